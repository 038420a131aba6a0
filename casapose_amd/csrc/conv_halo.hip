// 3x3 / stride 1 / pad 1 convolution for the shallow, high-resolution layers (cout <= 64):
// decoder blocks 3-5 and 8-10 and the stage-1 residual convs.
//
// Why a second kernel: with cout = 32 every A element feeds only 32 MACs, so the implicit-GEMM
// kernel needs ~10 TB/s of operand delivery from L2 (each input pixel is fetched again for each
// of the 9 taps) and stalls at 44-75 TFLOP/s.  Here a block owns a 2-D output tile
// (4*TMW rows x 32 columns), stages the (rows+2) x 34 input HALO of one 32-channel slice in LDS
// ONCE and runs all 9 taps from it: global operand traffic drops ~5.6x and the kernel becomes
// MFMA-bound.  Blocks are persistent over tiles so the producers fetch the next tile's halo while
// the consumers finish the current tile's epilogue.
//
// Same conventions as conv_f32.hip: fp32 MFMA 32x32x2, [row][k] LDS tiles with a 4-float pad
// (conflict-free ds_read_b128), wave specialisation (waves 0-3 MFMA, 4-7 gather), raw barriers,
// range-checked buffer loads, the same operand modes (bilinear x2, guided-nearest x2, partial-conv
// tap mask -- applied by the consumers because it depends on (output pixel, tap)) and epilogue.
#include "common.h"
#include "epilogue.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HW_COLS = 34;  // 32 output columns + 2 halo columns
enum : int { H_BILINEAR = 2, H_PARTIAL = 4, H_SEL = 8 };

struct HSrc {
    const float* data;
    const uint8_t* sel;
    int C, ld, mode, Hs, Ws;
    unsigned bytes;
};

struct HaloK {
    HSrc s[2];            // 32-multiple sources (s[1].data may be null)
    const float* img;     // optional 4-channel image source [B,H,W,4] (always direct)
    unsigned img_bytes;
    const float* W;       // [wide chunk][tap][cout_pad][32] then, if img, [cout_pad][40] (k = tap*4 + c)
    unsigned w_bytes;
    int B, H, Wd, Cout;
    int nwide0, nwide;    // wide (32-channel) chunks of source 0 / of both sources
    int tiles_y, tiles_x, ntiles;
    const uint8_t* label;  // partial conv + CLADE label map (same grid)
    unsigned lab_bytes;
    const float* residual;
    int res_ld;
    const float* scale;
    const float* shift;
    int clade;  // scale/shift indexed by label
    int act;
    float* out_raw;
    int raw_ld;
    float* out_act;
    int act_ld;
};

#define CP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Per tile the consumers run   nwide x 9 "tap steps" (16*TN MFMAs each, one 32-channel slice of one tap)
//                            + 1 "image step" (20*TN MFMAs: the 9 taps x 4 image channels as K = 40)
// with one barrier per step.  The producers run the same step sequence and keep three streams ahead
// of it -- weights (3-step register ring), halo fills of the NEXT 32-channel slice (2-step ring, 7
// elements per thread per slice), the image halo of the next tile -- using only incremental integer
// state (no divisions in the steady state; a step is only ~1000 MFMA cycles long).
template <int TMW, int TN, int MODE>
__global__ __launch_bounds__(512, (TMW * TN <= 2) ? 4 : 2) void conv_halo_kernel(const HaloK p) {
    constexpr int TH = 4 * TMW;             // output rows per tile
    constexpr int HR = TH + 2;              // halo rows
    constexpr int HP = HR * HW_COLS;        // halo pixels
    constexpr int BN = 32 * TN;
    constexpr int AS = 36;                  // wide halo pixel stride / weight row stride (floats)
    constexpr bool BILINEAR = (MODE & H_BILINEAR) != 0;
    constexpr bool PARTIAL = (MODE & H_PARTIAL) != 0;
    constexpr bool SEL = (MODE & H_SEL) != 0;
    constexpr int NV = BILINEAR ? 4 : 1;
    constexpr int NIT = (HP * 8 + 255) / 256;   // wide halo float4 elements per producer thread
    constexpr int NB = TN + 1;                  // float4 of weights per producer thread per step (image step: BN*40/4 = 320*TN)
    constexpr unsigned OOB = 0x80000000u;
    static_assert(NIT <= 7, "halo elements are issued at taps 0..6, the image element at tap 7");
    static_assert(HP <= 256, "one image-halo element per producer thread");

    const bool has_img = p.img != nullptr;
    const int BS = has_img ? 44 : 36;       // weight-stage row stride: the image step has 40-wide rows (+4 pad)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;                        // [2][HP][AS]
    float* bst = halo + 2 * HP * AS;           // [2][BN][BS]
    float* imgh = bst + 2 * BN * BS;           // [2][HP][4]

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;
    const int spt = 9 * p.nwide + (has_img ? 1 : 0);  // steps per tile

    const int bid = cp::xcd_remap(blockIdx.x, gridDim.x);  // neighbouring tiles (shared halo rows) on one XCD
    const int my_tiles = (p.ntiles - bid + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_steps = my_tiles * spt;
    auto tile_origin = [&](int k, int& n, int& y0, int& x0) {  // divisions: once per tile only
        int t = bid + k * (int)gridDim.x;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        x0 = tx * 32;
        y0 = (t % p.tiles_y) * TH;
        n = t / p.tiles_y;
    };

    if (producer) {
        // ------------------------------ producers --------------------------------------------
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.s[0].data, 0, p.s[0].bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.s[1].data ? p.s[1].data : p.s[0].data), 0,
                                                                              p.s[1].data ? p.s[1].bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)(has_img ? p.img : p.s[0].data), 0,
                                                                              has_img ? p.img_bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.w_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc((void*)(SEL ? (const void*)p.s[0].sel : (const void*)p.W), 0,
                                                                              SEL ? p.lab_bytes : 0u, 0x00020000);
        auto ldb4 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned off) -> float4 {
#ifdef HX_NOLOAD
            return make_float4(__builtin_bit_cast(float, off & 0x3fffffu), 1.f, 2.f, 3.f);  // timing experiment
#else
            return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
#endif
        };
        // per-thread constants: wide element `it` is halo pixel it*32 + tid/8, float4 slot tid%8
        const int f4 = tid & 7;
        const int hp0 = tid >> 3;
        int e_hy[NIT], e_hx[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int hp = it * 32 + hp0;
            e_hy[it] = hp < HP ? hp / HW_COLS : 0x4000;  // 0x4000: never in bounds
            e_hx[it] = hp % HW_COLS;
        }
        const int i_hy = tid < HP ? tid / HW_COLS : 0x4000, i_hx = tid % HW_COLS;

        // rings (slot = step % 3, compile-time because the step loop is unrolled by 3)
        float4 hreg[3][NV];
        int hdst[3];   // LDS float offset incl. stage, -1 = empty; bit 30 set = image element
        int hflag[3];
        float4 bring[3][NB];
        int selc[NIT], seln[NIT];

        // ---- stream state ---------------------------------------------------------------------
        int ft = 0, fc = 0;          // fill stream: tile, wide chunk in tile, next element
        int fn, fy0, fx0;                     // its tile origin
        int fgc = 0;                          // global wide-chunk counter of the fill (stage = fgc & 1)
        int bt = 0, bstep = 0;                // weight stream: tile, step in tile
        int ct = 0, cstep = 0, ctap = 0;      // consumer position: tile, step in tile, tap within the wide chunk
        int cgc = 0;                          // consumers' global wide-chunk counter
        int it_tile = 1;                      // image stream: next tile whose image halo is still to be fetched
        tile_origin(0, fn, fy0, fx0);

        auto load_sel_tile = [&](int k, int (&dst)[NIT]) {
            if constexpr (SEL) {
                int n, y0, x0;
                if (k >= my_tiles) return;
                tile_origin(k, n, y0, x0);
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int y = y0 - 1 + e_hy[it], x = x0 - 1 + e_hx[it];
                    const bool inb = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd;
                    dst[it] = __builtin_amdgcn_raw_buffer_load_b8(rss, inb ? ((n * p.H + y) * p.Wd + x) : (int)OOB, 0, 0);
                }
            }
        };
        // issue element `it` of wide chunk `c` of the tile at (n,y0,x0) into ring slot
        auto issue_wide = [&](int slot, int n, int y0, int x0, int c, int it, int stage, int selbyte) {
            const int si = c >= p.nwide0 ? 1 : 0;
            const __amdgpu_buffer_rsrc_t rs = si ? rs1 : rs0;
            const int sld = si ? p.s[1].ld : p.s[0].ld;
            const int cb = ((c - (si ? p.nwide0 : 0)) * 32 + f4 * 4) * 4;
            const int y = y0 - 1 + e_hy[it], x = x0 - 1 + e_hx[it];
            const bool inb = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd;
            hdst[slot] = e_hy[it] < 0x4000 ? (stage * HP * AS + (it * 32 + hp0) * AS + f4 * 4) : -1;
            hflag[slot] = ((y & 1) << 1) | ((x & 1) << 2) | (si << 3);
            if constexpr (BILINEAR) {
                unsigned o00, o01, o10, o11;
                if (si == 0) {
                    const int Hs = p.s[0].Hs, Ws = p.s[0].Ws;
                    int ys = (y >> 1) - ((y & 1) ? 0 : 1), xs = (x >> 1) - ((x & 1) ? 0 : 1);
                    int y1 = min(ys + 1, Hs - 1), x1 = min(xs + 1, Ws - 1);
                    ys = max(ys, 0);
                    xs = max(xs, 0);
                    const int nb = n * Hs * Ws;
                    o00 = inb ? (unsigned)(((nb + ys * Ws + xs) * sld) * 4 + cb) : OOB;
                    o01 = inb ? (unsigned)(((nb + ys * Ws + x1) * sld) * 4 + cb) : OOB;
                    o10 = inb ? (unsigned)(((nb + y1 * Ws + xs) * sld) * 4 + cb) : OOB;
                    o11 = inb ? (unsigned)(((nb + y1 * Ws + x1) * sld) * 4 + cb) : OOB;
                } else {
                    o00 = o01 = o10 = o11 = inb ? (unsigned)((((n * p.H + y) * p.Wd + x) * sld) * 4 + cb) : OOB;
                }
                hreg[slot][0] = ldb4(rs, o00);
                hreg[slot][1] = ldb4(rs, o01);
                hreg[slot][2] = ldb4(rs, o10);
                hreg[slot][3] = ldb4(rs, o11);
            } else if constexpr (SEL) {
                unsigned o;
                if (si == 0) {
                    const int sl = selbyte;
                    o = inb ? (unsigned)((((n * p.s[0].Hs + (y >> 1) + (sl >> 1)) * p.s[0].Ws + (x >> 1) + (sl & 1)) * sld) * 4 + cb) : OOB;
                } else {
                    o = inb ? (unsigned)((((n * p.H + y) * p.Wd + x) * sld) * 4 + cb) : OOB;
                }
                hreg[slot][0] = ldb4(rs, o);
            } else {
                hreg[slot][0] = ldb4(rs, inb ? (unsigned)((((n * p.H + y) * p.Wd + x) * sld) * 4 + cb) : OOB);
            }
        };
        auto issue_img = [&](int slot, int k) {  // image halo element of tile k -> image stage k&1
            int n, y0, x0;
            tile_origin(k, n, y0, x0);
            const int y = y0 - 1 + i_hy, x = x0 - 1 + i_hx;
            const bool inb = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd;
            hdst[slot] = i_hy < 0x4000 ? (0x40000000 | ((k & 1) * HP * 4 + tid * 4)) : -1;
            hflag[slot] = 8;  // no interpolation
            hreg[slot][0] = ldb4(rsi, inb ? (unsigned)(((n * p.H + y) * p.Wd + x) * 16) : OOB);
        };
        auto store_elem = [&](int slot) {
            const int d = hdst[slot];
            if (d < 0) return;
            float4 v = hreg[slot][0];
            if constexpr (BILINEAR) {
                if (!(hflag[slot] & 8)) {
                    const float fy = (hflag[slot] & 2) ? 0.25f : 0.75f, fx = (hflag[slot] & 4) ? 0.25f : 0.75f;
                    const float gy = 1.f - fy, gx = 1.f - fx;
                    const float4 v01 = hreg[slot][1], v10 = hreg[slot][2], v11 = hreg[slot][3];
                    v.x = (v.x * gx + v01.x * fx) * gy + (v10.x * gx + v11.x * fx) * fy;
                    v.y = (v.y * gx + v01.y * fx) * gy + (v10.y * gx + v11.y * fx) * fy;
                    v.z = (v.z * gx + v01.z * fx) * gy + (v10.z * gx + v11.z * fx) * fy;
                    v.w = (v.w * gx + v01.w * fx) * gy + (v10.w * gx + v11.w * fx) * fy;
                }
            }
            float* dst = (d & 0x40000000) ? (imgh + (d & 0x3fffffff)) : (halo + d);
            *reinterpret_cast<float4*>(dst) = v;
            hdst[slot] = -1;
        };
        // weights of step `st` of a tile: wide step st = c*9+tap -> block of BN*32 floats; image step -> BN*40 floats
        auto issue_b = [&](int slot, int st, bool valid) {
            const bool wide = st < 9 * p.nwide;
            const unsigned base = (unsigned)(wide ? st * BN * 32 : 9 * p.nwide * BN * 32) * 4u;
            const int nf4 = wide ? BN * 8 : BN * 10;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int e = j * 256 + tid;
                bring[slot][j] = ldb4(rsw, (valid && e < nf4) ? base + (unsigned)e * 16u : OOB);
            }
        };
        auto store_b = [&](int slot, int st, int stage) {
            const bool wide = st < 9 * p.nwide;
            float* b = bst + stage * BN * BS;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int e = j * 256 + tid;
                if (wide) {
                    if (e < BN * 8) *reinterpret_cast<float4*>(b + (e >> 3) * AS + (e & 7) * 4) = bring[slot][j];
                } else {
                    if (e < BN * 10) *reinterpret_cast<float4*>(b + (e / 10) * BS + (e % 10) * 4) = bring[slot][j];
                }
            }
        };

        // ---- prologue ---------------------------------------------------------------------------
        if constexpr (SEL) {
            load_sel_tile(0, selc);
            load_sel_tile(1, seln);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {  // fill 0: wide chunk 0 of tile 0 -> stage 0
            issue_wide(it % 3, fn, fy0, fx0, 0, it, 0, SEL ? selc[it] : 0);
            store_elem(it % 3);
        }
        if (has_img) {
            issue_img(0, 0);
            store_elem(0);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) hdst[q] = -1;
        // advance the fill stream to the next wide chunk
        fgc = 1;
        fc = 1;
        if (fc == p.nwide) {
            fc = 0;
            ft = 1;
            if (ft < my_tiles) tile_origin(ft, fn, fy0, fx0);
            if constexpr (SEL) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) selc[it] = seln[it];
                load_sel_tile(2, seln);
            }
        }
        // weights: step 0 straight to LDS, steps 1 and 2 into ring slots 1 and 2
        issue_b(0, 0, total_steps > 0);
        store_b(0, 0, 0);
        issue_b(1, 1 % spt, total_steps > 1);
        issue_b(2, 2 % spt, total_steps > 2);
        bstep = 3 % spt;  // next step (within a tile) whose weights are to be issued
        bt = 3 / spt;
        CP_BARRIER();

        // ---- steady state: one iteration per consumer step, ring slots static ---------------------
        auto do_step = [&](int cur, int gs) {
            // (1) weights of step gs+1 -> LDS stage (gs+1)&1 ; weights of step gs+3 -> ring slot cur
            {
                int st1 = cstep + 1;
                if (st1 == spt) st1 = 0;
                if (gs + 1 < total_steps) store_b((cur + 1) % 3, st1, (gs + 1) & 1);
                issue_b(cur, bstep, bt < my_tiles);
                if (++bstep == spt) { bstep = 0; ++bt; }
            }
            // (2) element issued two steps ago -> LDS
            store_elem((cur + 1) % 3);
            // (3) issue one element: halo fill of the next wide chunk at taps 0..NIT-1, image halo at tap 7
            const bool in_wide = cstep < 9 * p.nwide;
            if (in_wide && ctap < NIT) {
                if (fgc == cgc + 1 && ft < my_tiles) issue_wide(cur, fn, fy0, fx0, fc, ctap, fgc & 1, SEL ? selc[ctap] : 0);
            } else if (in_wide && ctap == 7 && has_img) {
                // the image stage of tile ct+1 is free once tile ct-1's image step is done: issue it in the LAST wide chunk
                if (cstep >= 9 * (p.nwide - 1) && it_tile == ct + 1 && it_tile < my_tiles) {
                    issue_img(cur, it_tile);
                    ++it_tile;
                }
            }
            CP_BARRIER();
            // (4) advance the consumer position; at a wide-chunk boundary the fill stream moves on
            ++cstep;
            if (in_wide) {
                if (++ctap == 9) {
                    ctap = 0;
                    ++cgc;
                    ++fgc;
                    if (++fc == p.nwide) {
                        fc = 0;
                        ++ft;
                        if (ft < my_tiles) tile_origin(ft, fn, fy0, fx0);
                        if constexpr (SEL) {
#pragma unroll
                            for (int it = 0; it < NIT; ++it) selc[it] = seln[it];
                            load_sel_tile(ft + 1, seln);
                        }
                    }
                }
            }
            if (cstep == spt) {
                cstep = 0;
                ++ct;
            }
        };
        for (int gs = 0; gs < total_steps; gs += 3) {
            do_step(0, gs);
            if (gs + 1 < total_steps) do_step(1, gs + 1);
            if (gs + 2 < total_steps) do_step(2, gs + 2);
        }
        return;
    }

    // ---------------------------------- consumers ------------------------------------------------
    const int wy = wave;  // consumer wave w owns tile rows [w*TMW, (w+1)*TMW)
    const int lrow = lane & 31;
    const int khalf = (lane >> 5) * 4;
    const int hi4 = (lane >> 5) * 4;
    const int half = lane >> 5;
    f32x16 acc[TMW][TN];
    float4 fa[2][TMW], fb[2][TN];
    int gs = 0, gwc = 0;  // global step / global wide-chunk counters
    cp::EpiArgs ea;
    ea.row_scale = nullptr; ea.label = p.clade ? p.label : nullptr; ea.residual = p.residual; ea.scale = p.scale; ea.shift = p.shift;
    ea.out_raw = p.out_raw; ea.out_act = p.out_act; ea.res_ld = p.res_ld; ea.raw_ld = p.raw_ld; ea.act_ld = p.act_ld;
    ea.cout = p.Cout; ea.act = p.act; ea.npix = (unsigned)(p.B * p.H * p.Wd);
    const cp::EpiRsrc er = cp::epi_make(ea, p.W);
    const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(PARTIAL ? (const void*)p.label : (const void*)p.W), 0,
                                                                          PARTIAL ? p.lab_bytes : 0u, 0x00020000);
    CP_BARRIER();  // prologue data is in LDS
    for (int k = 0; k < my_tiles; ++k) {
        int n, y0, x0;
        tile_origin(k, n, y0, x0);
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // partial conv: 9-bit tap mask of this lane's pixel in each of its rows.  The nine label
        // bytes are fetched as one batch of range-checked loads (outside the image -> never equal)
        int pmask[TMW];
        if constexpr (PARTIAL) {
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                const int y = y0 + wy * TMW + i, x = x0 + lrow;
                int lb[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    const bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.Wd;
                    lb[t] = __builtin_amdgcn_raw_buffer_load_b8(rsl, ok ? ((n * p.H + yy) * p.Wd + xx) : (int)OOB, 0, 0) | (ok ? 0 : 0xff00);
                }
                int m = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) m |= (lb[t] == lb[4]) ? (1 << t) : 0;
                pmask[i] = (lb[4] & 0xff00) ? 0 : m;
            }
        }
        auto mfma4 = [&](int slot, int tap_lo, int tap_hi) {
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                float4 av = fa[slot][i];
                if constexpr (PARTIAL) {
                    const int tp = half ? tap_hi : tap_lo;
                    if (!((pmask[i] >> tp) & 1)) av = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#ifdef HX_NOMFMA
                    acc[i][j][0] += av.x * fb[slot][j].x + av.y * fb[slot][j].y + av.z * fb[slot][j].z + av.w * fb[slot][j].w;  // timing experiment
#else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, fb[slot][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, fb[slot][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, fb[slot][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, fb[slot][j].w, acc[i][j], 0, 0, 0);
#endif
                }
            }
        };
        // ---- wide chunks: 9 taps x 4 k8-steps, one barrier per tap ----------------------------------
        for (int c = 0; c < p.nwide; ++c, ++gwc) {
            const float* hb = halo + (gwc & 1) * HP * AS + ((wy * TMW) * HW_COLS + lrow) * AS + khalf;
            auto read_wide = [&](int tap, int k8, int slot, int bstage) {
                const int ky = tap / 3, kx = tap - ky * 3;  // tap is a compile-time constant after unrolling
                const float* a = hb + (ky * HW_COLS + kx) * AS + k8 * 8;
                const float* b = bst + bstage * BN * BS + lrow * AS + k8 * 8 + khalf;
#pragma unroll
                for (int i = 0; i < TMW; ++i) fa[slot][i] = *reinterpret_cast<const float4*>(a + i * HW_COLS * AS);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const float4*>(b + j * 32 * AS);
            };
            read_wide(0, 0, 0, gs & 1);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap, ++gs) {
                read_wide(tap, 1, 1, gs & 1);
                mfma4(0, tap, tap);
                read_wide(tap, 2, 0, gs & 1);
                mfma4(1, tap, tap);
                read_wide(tap, 3, 1, gs & 1);
                mfma4(0, tap, tap);
                CP_BARRIER();  // all reads of weight stage gs&1 are complete; stage (gs+1)&1 is ready
                if (tap < 8) read_wide(tap + 1, 0, 0, (gs + 1) & 1);
                mfma4(1, tap, tap);
            }
        }
        // ---- image step: K = 9 taps x 4 channels (+4 zero) = 5 k8-steps; half-wave h handles tap 2s+h ----
        if (has_img) {
            const float* ib = imgh + (k & 1) * HP * 4 + ((wy * TMW) * HW_COLS + lrow) * 4;
            const float* b0 = bst + (gs & 1) * BN * BS + lrow * BS + khalf;
#pragma unroll
            for (int s5 = 0; s5 < 5; ++s5) {
                const int tl = 2 * s5, th = (2 * s5 + 1 < 9) ? 2 * s5 + 1 : 8;  // tap 9 does not exist: its weights are zero
                const int offl = ((tl / 3) * HW_COLS + tl % 3) * 4, offh = ((th / 3) * HW_COLS + th % 3) * 4;
                const float* a = ib + (half ? offh : offl);
#pragma unroll
                for (int i = 0; i < TMW; ++i) fa[s5 & 1][i] = *reinterpret_cast<const float4*>(a + i * HW_COLS * 4);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[s5 & 1][j] = *reinterpret_cast<const float4*>(b0 + j * 32 * BS + s5 * 8);
                if (s5 == 4) CP_BARRIER();  // reads of this weight stage / image stage are complete
                mfma4(s5 & 1, tl, th);
            }
            ++gs;
        }
        // ---- epilogue (epilogue.h) ------------------------------------------------------------
        int cos[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) cos[j] = j * 32 + lrow;
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            const int y = y0 + wy * TMW + i;
            float rsp[16];
            if constexpr (PARTIAL) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pm = __shfl(pmask[i], (r & 3) + 8 * (r >> 2) + hi4);  // the mask lives in the lane owning that column
                    rsp[r] = 9.0f / (float)max(__popc(pm), 1);
                }
            }
            const int rowbase = (n * p.H + y) * p.Wd + x0 + hi4;
#ifdef HX_NOEPI
            if (p.B < 0)  // timing experiment: never true, keeps the accumulators alive
#endif
            cp::epilogue_block<TN, (TN == 1) ? 8 : 4>(acc[i], cos, ea, er,
                                   [&](int r) { const int xr = (r & 3) + 8 * (r >> 2); return (y < p.H && x0 + hi4 + xr < p.Wd) ? rowbase + xr : -1; },
                                   PARTIAL ? rsp : nullptr);
        }
    }
}

template <int TMW, int TN, int MODE>
int launch_halo(HaloK k, hipStream_t st) {
    constexpr int TH = 4 * TMW, HP = (TH + 2) * HW_COLS, BN = 32 * TN;
    k.tiles_y = (k.H + TH - 1) / TH;
    k.tiles_x = (k.Wd + 31) / 32;
    k.ntiles = k.B * k.tiles_y * k.tiles_x;
    const bool img = k.img != nullptr;
    const size_t lds = (size_t)(2 * HP * 36 + 2 * BN * (img ? 44 : 36) + (img ? 2 * HP * 4 : 0)) * sizeof(float);
    const size_t lds_max = (size_t)(2 * HP * 36 + 2 * BN * 44 + 2 * HP * 4) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<TMW, TN, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        attr_set = true;
    }
    const int blocks_per_cu = (lds * 2 <= 160 * 1024) ? 2 : 1;
    int grid = 256 * blocks_per_cu;
    if (grid > k.ntiles) grid = k.ntiles;
    CP_LAUNCH((conv_halo_kernel<TMW, TN, MODE>), dim3(grid), dim3(512), lds, st, k);
    return cp::check_launch("cp_conv2d_fwd_f32(halo)");
}

}  // namespace

namespace cp {

// host packing of the halo kernel's weight layout: [chunk][tap][cout_pad][kc]
int halo_weight_floats(int cout, int num_sources, const int* channels) {
    const int bn = cout <= 32 ? 32 : 64;
    int n = 0;
    for (int s = 0; s < num_sources; ++s) n += (channels[s] == 4) ? bn * 40 : (channels[s] / 32) * 9 * bn * 32;
    return n;
}

// layout: for every 32-channel slice (sources in order): [tap][cout_pad][32]; then, for a trailing 4-channel image
// source, one block [cout_pad][40] with k = tap*4 + channel (36 real values + 4 zeros)
int halo_pack_weights(const float* w, int layout, int cout, int num_sources, const int* channels, const int* real_channels, float* dst) {
    const int bn = cout <= 32 ? 32 : 64;
    int cin = 0;
    for (int s = 0; s < num_sources; ++s) cin += real_channels[s];
    const int total = halo_weight_floats(cout, num_sources, channels);
    for (int i = 0; i < total; ++i) dst[i] = 0.f;
    size_t base = 0;
    int cbase = 0;
    auto src_index = [&](int ci, int t, int co) {
        const int ky = t / 3, kx = t % 3;
        return (layout == 0) ? ((((size_t)ky * 3 + kx) * cin + ci) * cout + co) : ((((size_t)ci * 3 + ky) * 3 + kx) * cout + co);
    };
    for (int s = 0; s < num_sources; ++s) {
        const int C = channels[s], Cr = real_channels[s];
        if (C == 4) {
            for (int co = 0; co < cout; ++co)
                for (int t = 0; t < 9; ++t)
                    for (int c = 0; c < Cr; ++c) dst[base + (size_t)co * 40 + t * 4 + c] = w[src_index(cbase + c, t, co)];
            base += (size_t)bn * 40;
        } else {
            for (int ch = 0; ch < C / 32; ++ch)
                for (int t = 0; t < 9; ++t)
                    for (int co = 0; co < cout; ++co)
                        for (int kk = 0; kk < 32; ++kk) {
                            const int c = ch * 32 + kk;
                            if (c < Cr) dst[base + ((size_t)(ch * 9 + t) * bn + co) * 32 + kk] = w[src_index(cbase + c, t, co)];
                        }
            base += (size_t)(C / 32) * 9 * bn * 32;
        }
        cbase += Cr;
    }
    return CP_OK;
}

bool halo_applicable(const cp_conv_desc* d) {
    if (!d->weights_halo) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->dilation != 1 || d->pad != 1) return false;
    if (d->cout > 64) return false;
    if (d->src[0].channels % 32 != 0 || d->src[0].pre_scale) return false;
    if (d->num_sources == 2) {
        if (d->src[1].mode != CP_SRC_DIRECT || d->src[1].pre_scale) return false;
        if (!(d->src[1].channels == 4 || d->src[1].channels % 32 == 0)) return false;
    }
    if (d->tap_label && d->epi_label && d->tap_label != d->epi_label) return false;
    if (d->tap_label && !d->row_scale) return false;  // the kernel always applies 9/count with the mask
    if (!d->tap_label && d->row_scale) return false;
    if (d->src[0].mode == CP_SRC_BILINEAR_X2 && d->tap_label) return false;
    if (d->src[0].mode == CP_SRC_NEAREST_SEL && !d->tap_label) return false;
    return true;
}

int launch_halo_conv(const cp_conv_desc* d, hipStream_t st) {
    HaloK k{};
    int wide = 0;
    for (int s = 0; s < d->num_sources; ++s) {
        const cp_conv_source& in = d->src[s];
        const int Hs = (in.mode == CP_SRC_DIRECT) ? d->in_h : d->in_h / 2, Ws = (in.mode == CP_SRC_DIRECT) ? d->in_w : d->in_w / 2;
        const long long nbytes = (long long)d->batch * Hs * Ws * in.ld * 4;
        CP_REQUIRE(nbytes < (1LL << 31), "cp_conv2d_fwd_f32: source %d spans %lld bytes; 32-bit range-checked addressing needs < 2 GiB", s, nbytes);
        if (in.channels == 4) {  // the image source (always the last one, always direct)
            k.img = in.data;
            k.img_bytes = (unsigned)nbytes;
            CP_REQUIRE(in.ld == 4, "cp_conv2d_fwd_f32: the 4-channel source must be dense (ld == 4)");
            continue;
        }
        HSrc& o = k.s[s];
        o.data = in.data;
        o.sel = in.sel;
        o.C = in.channels;
        o.ld = in.ld;
        o.mode = in.mode;
        o.Hs = Hs;
        o.Ws = Ws;
        o.bytes = (unsigned)nbytes;
        if (s == 0) k.nwide0 = in.channels / 32;
        wide += in.channels / 32;
    }
    k.nwide = wide;
    int chans[2] = {d->src[0].channels, d->num_sources > 1 ? d->src[1].channels : 0};
    k.W = d->weights_halo;
    k.w_bytes = (unsigned)(halo_weight_floats(d->cout, d->num_sources, chans) * sizeof(float));
    k.B = d->batch; k.H = d->in_h; k.Wd = d->in_w; k.Cout = d->cout;
    k.label = d->tap_label ? d->tap_label : d->epi_label;
    k.lab_bytes = (unsigned)((size_t)d->batch * d->in_h * d->in_w);
    k.residual = d->residual; k.res_ld = d->residual_ld;
    k.scale = d->scale; k.shift = d->shift; k.clade = d->epi_label != nullptr; k.act = d->act;
    k.out_raw = d->out_raw; k.raw_ld = d->out_raw_ld; k.out_act = d->out_act; k.act_ld = d->out_act_ld;
    const bool partial = d->tap_label != nullptr;
    const bool bil = d->src[0].mode == CP_SRC_BILINEAR_X2, sel = d->src[0].mode == CP_SRC_NEAREST_SEL;
    const int tn = d->cout <= 32 ? 1 : 2;
    const int mode = (bil ? H_BILINEAR : 0) | (partial ? H_PARTIAL : 0) | (sel ? H_SEL : 0);
#define CP_HALO_CASE(M)                                                     \
    case M:                                                                 \
        return tn == 1 ? launch_halo<1, 1, M>(k, st) : launch_halo<1, 2, M>(k, st);
    switch (mode) {
        CP_HALO_CASE(0)
        CP_HALO_CASE(H_BILINEAR)
        CP_HALO_CASE(H_PARTIAL)
        CP_HALO_CASE(H_PARTIAL | H_SEL)
        default: break;
    }
#undef CP_HALO_CASE
    cp::set_error("cp_conv2d_fwd_f32: halo kernel does not cover operand mode %d", mode);
    return CP_ERR_INVALID;
}

}  // namespace cp
