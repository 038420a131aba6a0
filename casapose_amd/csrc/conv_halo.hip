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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int HW_COLS = 34;  // 32 output columns + 2 halo columns
#ifndef CP_HALO_RING1
#define CP_HALO_RING1 6   // weight-prefetch ring of the 32-channel kernels (divides 36)
#endif
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
    int norm;   // partial conv: multiply by 9/count (forward); 0 for the data gradient, whose operand already carries that factor
    int act;
    float* out_raw;
    int raw_ld;
    float* out_act;
    int act_ld;
    const float* head_w;   // fused 1x1 head [4][2][32][4] (cout == 32 only)
    float* head_out;
    int head_cout, head_ld;
    uint8_t* head_lab;   // optional arg-max of the first head_lab_classes head channels
    int head_lab_classes;
};

#define CP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Structure (v3).  Per tile the consumers run, for every 32-channel slice ("wide chunk"), 9 taps x 4
// k8-steps of 4*TMW*TN MFMAs straight out of the LDS-resident halo, then (if the layer has the image
// source) 5 more k8-steps on the image halo (K = 9 taps x 4 channels, padded to 40).  The WEIGHT
// fragments do not go through LDS at all: they are tiny, shared by every tile and every wave, so each
// consumer lane fetches its 16 bytes per k8-step directly from L1/L2 with a 3-step register prefetch
// ring (the weight layout makes every such wave access one contiguous 1 KiB).  That leaves ONE
// barrier per wide chunk (halo stage flip); an earlier version staged weights per tap through LDS
// and its 10-18 barriers + producer bookkeeping per tile cost more than the MFMAs of a 32-channel
// layer.  The producers only refill the other halo stage once per chunk.
template <int TMW, int TN, int MODE>
__global__ __launch_bounds__(512, (TMW * TN <= 2) ? 4 : 2) void conv_halo_kernel(const HaloK p) {
    constexpr int TH = 4 * TMW;             // output rows per tile
    constexpr int HR = TH + 2;              // halo rows
    constexpr int HP = HR * HW_COLS;        // halo pixels
    constexpr int BN = 32 * TN;
    constexpr int AS = 36;                  // halo pixel stride (floats)
    constexpr bool BILINEAR = (MODE & H_BILINEAR) != 0;
    constexpr bool PARTIAL = (MODE & H_PARTIAL) != 0;
    constexpr bool SEL = (MODE & H_SEL) != 0;
    constexpr int NV = BILINEAR ? 4 : 1;
    constexpr int NIT = (HP * 8 + 255) / 256;   // wide halo float4 elements per producer thread
    constexpr unsigned OOB = 0x80000000u;
    static_assert(HP <= 256, "one image-halo element per producer thread");

    const bool has_img = p.img != nullptr;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;                        // [2][HP][AS]
    float* imgh = halo + 2 * HP * AS;          // [2][HP][4]

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;

    const int bid = cp::xcd_remap(blockIdx.x, gridDim.x);  // neighbouring tiles (shared halo rows) on one XCD
    const int my_tiles = (p.ntiles - bid + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_chunks = my_tiles * p.nwide;
    // Tile coordinates advance by the constant stride gridDim.x: mixed-radix increments, no divisions later
    struct TilePos { int tx, ty, n; };
    TilePos first;
    {
        int t = bid;
        first.tx = t % p.tiles_x;
        t /= p.tiles_x;
        first.ty = t % p.tiles_y;
        first.n = t / p.tiles_y;
    }
    const int g = (int)gridDim.x;
    const int d_tx = g % p.tiles_x, d_ty = (g / p.tiles_x) % p.tiles_y, d_n = g / (p.tiles_x * p.tiles_y);
    auto next_tile = [&](TilePos& t) {
        t.tx += d_tx;
        int cy = 0;
        if (t.tx >= p.tiles_x) { t.tx -= p.tiles_x; cy = 1; }
        t.ty += d_ty + cy;
        int cn = 0;
        if (t.ty >= p.tiles_y) { t.ty -= p.tiles_y; cn = 1; }
        t.n += d_n + cn;
    };

    if (producer) {
        // ------------------------------ producers --------------------------------------------
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.s[0].data, 0, p.s[0].bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.s[1].data ? p.s[1].data : p.s[0].data), 0,
                                                                              p.s[1].data ? p.s[1].bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)(has_img ? p.img : p.s[0].data), 0,
                                                                              has_img ? p.img_bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc((void*)(SEL ? (const void*)p.s[0].sel : (const void*)p.W), 0,
                                                                              SEL ? p.lab_bytes : 0u, 0x00020000);
        auto ldb4 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned off) -> float4 {
#ifdef HX_NOLOAD
            return make_float4(__builtin_bit_cast(float, off & 0x3fffffu), 1.f, 2.f, 3.f);  // timing experiment
#else
            return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
#endif
        };
        // wide element `it` of a thread is halo pixel it*32 + tid/8, float4 slot tid%8
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

        // fill the halo stage of wide chunk `c` of tile `tp` (elements [it0, it1)) -- issue all, then store all
        auto fill_wide = [&](const TilePos& tp, int c, int stage, int it0, int it1) {
            const int n = tp.n, y0 = tp.ty * TH, x0 = tp.tx * 32;
            const int si = c >= p.nwide0 ? 1 : 0;
            const __amdgpu_buffer_rsrc_t rs = si ? rs1 : rs0;
            const int sld = si ? p.s[1].ld : p.s[0].ld;
            const int cb = ((c - (si ? p.nwide0 : 0)) * 32 + f4 * 4) * 4;
            float4 v[NIT][NV];
            int par[NIT];
            int selb[NIT];
            if constexpr (SEL) {
                if (si == 0) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        if (it < it0 || it >= it1) continue;
                        const int y = y0 - 1 + e_hy[it], x = x0 - 1 + e_hx[it];
                        const bool inb = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd;
                        selb[it] = __builtin_amdgcn_raw_buffer_load_b8(rss, inb ? ((n * p.H + y) * p.Wd + x) : (int)OOB, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (it < it0 || it >= it1) continue;
                const int y = y0 - 1 + e_hy[it], x = x0 - 1 + e_hx[it];
                const bool inb = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd;
                par[it] = ((y & 1) << 1) | ((x & 1) << 2);
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
                    v[it][0] = ldb4(rs, o00);
                    v[it][1] = ldb4(rs, o01);
                    v[it][2] = ldb4(rs, o10);
                    v[it][3] = ldb4(rs, o11);
                } else if constexpr (SEL) {
                    unsigned o;
                    if (si == 0) {
                        const int sl = selb[it];
                        o = inb ? (unsigned)((((n * p.s[0].Hs + (y >> 1) + (sl >> 1)) * p.s[0].Ws + (x >> 1) + (sl & 1)) * sld) * 4 + cb) : OOB;
                    } else {
                        o = inb ? (unsigned)((((n * p.H + y) * p.Wd + x) * sld) * 4 + cb) : OOB;
                    }
                    v[it][0] = ldb4(rs, o);
                } else {
                    v[it][0] = ldb4(rs, inb ? (unsigned)((((n * p.H + y) * p.Wd + x) * sld) * 4 + cb) : OOB);
                }
            }
            float* h = halo + stage * HP * AS;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (it < it0 || it >= it1) continue;
                if (e_hy[it] >= 0x4000) continue;
                float4 r = v[it][0];
                if constexpr (BILINEAR) {
                    if (si == 0) {
                        const float fy = (par[it] & 2) ? 0.25f : 0.75f, fx = (par[it] & 4) ? 0.25f : 0.75f;
                        const float gy = 1.f - fy, gx = 1.f - fx;
                        const float4 v01 = v[it][1], v10 = v[it][2], v11 = v[it][3];
                        r.x = (r.x * gx + v01.x * fx) * gy + (v10.x * gx + v11.x * fx) * fy;
                        r.y = (r.y * gx + v01.y * fx) * gy + (v10.y * gx + v11.y * fx) * fy;
                        r.z = (r.z * gx + v01.z * fx) * gy + (v10.z * gx + v11.z * fx) * fy;
                        r.w = (r.w * gx + v01.w * fx) * gy + (v10.w * gx + v11.w * fx) * fy;
                    }
                }
                *reinterpret_cast<float4*>(h + (it * 32 + hp0) * AS + f4 * 4) = r;
            }
        };
        auto fill = [&](const TilePos& tp, int c, int stage) {
            if constexpr (BILINEAR) {  // 4 taps per element: two batches keep the register footprint down
                fill_wide(tp, c, stage, 0, (NIT + 1) / 2);
                fill_wide(tp, c, stage, (NIT + 1) / 2, NIT);
            } else {
                fill_wide(tp, c, stage, 0, NIT);
            }
        };
        auto fill_img = [&](const TilePos& tp, int stage) {
            const int n = tp.n, y = tp.ty * TH - 1 + i_hy, x = tp.tx * 32 - 1 + i_hx;
            const bool inb = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd;
            const float4 v = ldb4(rsi, inb ? (unsigned)(((n * p.H + y) * p.Wd + x) * 16) : OOB);
            if (i_hy < 0x4000) *reinterpret_cast<float4*>(imgh + stage * HP * 4 + tid * 4) = v;
        };

        TilePos ftile = first;  // tile of the NEXT chunk to fill
        int fk = 0, fc = 0;     // its tile counter / wide chunk index
        fill(ftile, 0, 0);
        if (has_img) fill_img(ftile, 0);
        CP_BARRIER();
        for (int gc = 0; gc < total_chunks; ++gc) {
            // consumers work on chunk gc (stage gc&1); refill the other stage with chunk gc+1
            if (++fc == p.nwide) {
                fc = 0;
                ++fk;
                next_tile(ftile);
                if (has_img && fk < my_tiles) fill_img(ftile, fk & 1);  // image stage fk&1 was last read in tile fk-2
            }
            if (gc + 1 < total_chunks) fill(ftile, fc, (gc + 1) & 1);
            CP_BARRIER();
        }
        return;
    }

    // ---------------------------------- consumers ------------------------------------------------
    const int wy = wave;  // consumer wave w owns tile rows [w*TMW, (w+1)*TMW)
    const int lrow = lane & 31;
    const int khalf = (lane >> 5) * 4;
    const int half = lane >> 5;
    // Accumulators are TRANSPOSED (MFMA A = weights, B = pixels): lane l holds pixel column l&31 of its tile row and, in register
    // r, output channel (r&3) + 8*(r>>2) + 4*(l>>5) of block j.  Everything that is per pixel (label, partial-conv norm, CLADE
    // table row, bounds) is therefore per LANE, four consecutive channels sit in four consecutive registers (16-byte loads and
    // stores), and the activated tile is already in the B-operand layout of the fused 1x1 head -- no LDS round trip.
    f32x16 acc[TMW][TN];
    const unsigned npix = (unsigned)(p.B * p.H * p.Wd);
    const bool has_lab = PARTIAL || p.clade;
    const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(has_lab ? (const void*)p.label : (const void*)p.W), 0,
                                                                          has_lab ? p.lab_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_tab_s = __builtin_amdgcn_make_buffer_rsrc((void*)(p.scale ? (const void*)p.scale : (const void*)p.W), 0,
                                                                              p.scale ? (unsigned)((p.clade ? 256 : 1) * p.Cout * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_tab_b = __builtin_amdgcn_make_buffer_rsrc((void*)(p.scale ? (const void*)p.shift : (const void*)p.W), 0,
                                                                              p.scale ? (unsigned)((p.clade ? 256 : 1) * p.Cout * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual ? (const void*)p.residual : (const void*)p.W), 0,
                                                                            p.residual ? npix * (unsigned)p.res_ld * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_raw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out_raw ? (void*)p.out_raw : (void*)p.W), 0,
                                                                            p.out_raw ? npix * (unsigned)p.raw_ld * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_act = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out_act ? (void*)p.out_act : (void*)p.W), 0,
                                                                            p.out_act ? npix * (unsigned)p.act_ld * 4u : 0u, 0x00020000);
    const bool head = (TN == 1) && p.head_out != nullptr;
    const __amdgpu_buffer_rsrc_t r_head = __builtin_amdgcn_make_buffer_rsrc((void*)(head ? (void*)p.head_out : (void*)p.W), 0,
                                                                             head ? npix * (unsigned)p.head_ld * 4u : 0u, 0x00020000);
    // fused-head weights: A fragments of the 16 MFMA steps (constant for the whole launch)
    float4 hw[4];
    if (head) {
        const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc((void*)p.head_w, 0, 4096u, 0x00020000);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            hw[g4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsh, (int)(((g4 * 2 + half) * 32 + lrow) * 16), 0, 0));
    }
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.w_bytes, 0x00020000);
    // weight fragment of k8-step u of wide chunk c: 16 B at ((c*36 + u)*2 + half)*BN + co  (x16 B); image block after the wide part
    const unsigned wlane = (unsigned)((half * BN + lrow) * 16);
    auto ldw = [&](unsigned step_bytes, int j) -> float4 {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)(step_bytes + wlane + (unsigned)j * 512u), 0, 0));
    };
    constexpr unsigned STEP_BYTES = 2u * BN * 16u;      // one k8-step of weights
    constexpr unsigned CHUNK_BYTES = 36u * STEP_BYTES;
    const unsigned img_w_off = (unsigned)p.nwide * CHUNK_BYTES;
    // ring: the fragment of step u lives in slot u%RING and is fetched RD = RING-1 steps ahead.  A step is 4*TMW*TN MFMAs (256 cycles
    // for a 32-channel layer), and the weight stream of all but the smallest layers comes from L2 (> 32 KiB), so the 32-channel
    // kernels need the deeper ring to cover that latency.  RING divides 36: the slot indices stay compile-time constants.
    constexpr int RING = (TMW * TN == 1) ? CP_HALO_RING1 : 4;
    constexpr int RD = RING - 1;
    static_assert(36 % RING == 0 && RD >= 3, "ring must divide the 36 steps of a wide chunk");
    float4 fb[RING][TN];
    float4 fa[2][TMW];
    // prime the ring with steps 0..RD-1 of chunk 0
#pragma unroll
    for (int u = 0; u < RD; ++u)
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[u][j] = ldw(u * STEP_BYTES, j);

    static_assert(TMW == 1, "one tile row per consumer wave");
    // ---- per-tile label state -------------------------------------------------------------------
    // partial conv: 9-bit tap mask of this lane's pixel; CLADE: label of this lane's pixel.  The nine label bytes are fetched as
    // one batch of range-checked loads (outside the image -> never equal to the centre).
    int pmask = 0, clab = 0;
    int lbn[9];
    auto issue_labels = [&](const TilePos& t) {
        const int y = t.ty * TH + wy, x = t.tx * 32 + lrow;
        if constexpr (PARTIAL) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                const bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.Wd;
                lbn[tp] = __builtin_amdgcn_raw_buffer_load_b8(rsl, ok ? ((t.n * p.H + yy) * p.Wd + xx) : (int)OOB, 0, 0) | (ok ? 0 : 0xff00);
            }
        } else {
            const bool ok = y < p.H && x < p.Wd;
            lbn[4] = __builtin_amdgcn_raw_buffer_load_b8(rsl, (ok && p.clade) ? ((t.n * p.H + y) * p.Wd + x) : (int)OOB, 0, 0);
        }
    };
    auto finish_labels = [&]() {
        if constexpr (PARTIAL) {
            int m = 0;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) m |= (lbn[tp] == lbn[4]) ? (1 << tp) : 0;
            pmask = (lbn[4] & 0xff00) ? 0 : m;
        }
        clab = lbn[4] & 0xff;
    };

    // ---- epilogue: lane = pixel, 4 consecutive channels per 16-byte access -----------------------
    // scale / shift of this lane's 16 channels per block (row `clab` of a CLADE table).  The 32-channel kernels request them
    // during the tile's last chunk; the 64-channel kernels have no registers to park them and fetch per block in the epilogue.
    constexpr bool EPI_PREFETCH = (TN == 1);
    float4 esc[TN][4], esh[TN][4];
    auto load_tables = [&](int j) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int ch = j * 32 + g4 * 8 + half * 4;
            const unsigned to = (ch < p.Cout) ? (unsigned)((clab * (p.clade ? p.Cout : 0) + ch) * 4) : OOB;
            esc[j][g4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_tab_s, (int)to, 0, 0));
            esh[j][g4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_tab_b, (int)to, 0, 0));
        }
    };
    auto epilogue = [&](int n, int y0, int x0) {
        const int y = y0 + wy, x = x0 + lrow;
        const bool pok = y < p.H && x < p.Wd;
        const unsigned pix = (unsigned)((n * p.H + y) * p.Wd + x);
        float f = 1.f;
        if constexpr (PARTIAL) f = p.norm ? 9.0f / (float)max(__popc(pmask), 1) : 1.0f;
        float4 keep[4];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float4 res[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int ch = j * 32 + g4 * 8 + half * 4;
                const unsigned o = (pok && ch < p.Cout) ? (pix * (unsigned)p.res_ld + (unsigned)ch) * 4u : OOB;
                res[g4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_res, (int)o, 0, 0));  // empty descriptor without a residual: zeros
            }
            if constexpr (!EPI_PREFETCH) load_tables(j);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int ch = j * 32 + g4 * 8 + half * 4;
                const bool ok = pok && ch < p.Cout;
                float4 v;
                v.x = acc[0][j][g4 * 4 + 0] * f + res[g4].x;
                v.y = acc[0][j][g4 * 4 + 1] * f + res[g4].y;
                v.z = acc[0][j][g4 * 4 + 2] * f + res[g4].z;
                v.w = acc[0][j][g4 * 4 + 3] * f + res[g4].w;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_raw, (int)(ok ? (pix * (unsigned)p.raw_ld + (unsigned)ch) * 4u : OOB), 0, 0);
                float4 t = v;
                if (p.scale) {
                    t.x = v.x * esc[j][g4].x + esh[j][g4].x;
                    t.y = v.y * esc[j][g4].y + esh[j][g4].y;
                    t.z = v.z * esc[j][g4].z + esh[j][g4].z;
                    t.w = v.w * esc[j][g4].w + esh[j][g4].w;
                }
                if (p.act == CP_ACT_RELU) {
                    t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
                } else if (p.act == CP_ACT_LEAKY01) {
                    t.x = fmaxf(t.x, 0.f) - fmaxf(-0.1f * t.x, 0.f);
                    t.y = fmaxf(t.y, 0.f) - fmaxf(-0.1f * t.y, 0.f);
                    t.z = fmaxf(t.z, 0.f) - fmaxf(-0.1f * t.z, 0.f);
                    t.w = fmaxf(t.w, 0.f) - fmaxf(-0.1f * t.w, 0.f);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), r_act, (int)(ok ? (pix * (unsigned)p.act_ld + (unsigned)ch) * 4u : OOB), 0, 0);
                if (j == 0) keep[g4] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if constexpr (TN == 1) {
            if (head) {
                // Fused 1x1 head: out[q][pixel] = sum_c Wh[c][q] * t[c][pixel].  Step (g4, e) multiplies channel 8*g4 + 4*half + e on
                // both operands: A = the head-weight fragment (row q = lane), B = the activated value already in this lane.
                f32x16 a2;
#pragma unroll
                for (int r = 0; r < 16; ++r) a2[r] = 0.f;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(hw[g4].x, keep[g4].x, a2, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(hw[g4].y, keep[g4].y, a2, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(hw[g4].z, keep[g4].z, a2, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(hw[g4].w, keep[g4].w, a2, 0, 0, 0);
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int q0 = g4 * 8 + half * 4;
                    const int nq = pok ? p.head_cout - q0 : 0;  // valid head channels of this group (<= 0: none)
                    const unsigned o = (pix * (unsigned)p.head_ld + (unsigned)q0) * 4u;
                    const unsigned v0 = __builtin_bit_cast(unsigned, (float)a2[g4 * 4 + 0]), v1 = __builtin_bit_cast(unsigned, (float)a2[g4 * 4 + 1]);
                    const unsigned v2 = __builtin_bit_cast(unsigned, (float)a2[g4 * 4 + 2]), v3 = __builtin_bit_cast(unsigned, (float)a2[g4 * 4 + 3]);
                    if (nq >= 4) __builtin_amdgcn_raw_buffer_store_b128(u32x4{v0, v1, v2, v3}, r_head, (int)o, 0, 0);
                    else if (nq == 3) __builtin_amdgcn_raw_buffer_store_b96(u32x3{v0, v1, v2}, r_head, (int)o, 0, 0);
                    else if (nq == 2) __builtin_amdgcn_raw_buffer_store_b64(u32x2{v0, v1}, r_head, (int)o, 0, 0);
                    else if (nq == 1) __builtin_amdgcn_raw_buffer_store_b32(v0, r_head, (int)o, 0, 0);
                }
                if (p.head_lab) {   // the hard label map straight from the head's registers (no second pass over the strided records)
                    const int lab = cp::head_argmax(a2, half, p.head_lab_classes);
                    if (half == 0 && pok) p.head_lab[pix] = (uint8_t)lab;
                }
            }
        }
    };

    int gwc = 0;  // global wide-chunk counter
    TilePos ctile = first;
    CP_BARRIER();  // prologue data is in LDS
    for (int k = 0; k < my_tiles; ++k) {
        const int n = ctile.n, y0 = ctile.ty * TH, x0 = ctile.tx * 32;
        if (has_lab) {
            issue_labels(ctile);
            finish_labels();
        }
        next_tile(ctile);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
        auto mfma4 = [&](int aslot, int bslot, int tap_lo, int tap_hi) {
            float4 av = fa[aslot][0];
            if constexpr (PARTIAL) {
                const int tp = half ? tap_hi : tap_lo;
                if (!((pmask >> tp) & 1)) av = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#ifdef HX_NOMFMA
                acc[0][j][0] += av.x * fb[bslot][j].x + av.y * fb[bslot][j].y + av.z * fb[bslot][j].z + av.w * fb[bslot][j].w;  // timing experiment
#else
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[bslot][j].x, av.x, acc[0][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[bslot][j].y, av.y, acc[0][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[bslot][j].z, av.z, acc[0][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[bslot][j].w, av.w, acc[0][j], 0, 0, 0);
#endif
            }
        };
        // ---- wide chunks: 36 k8-steps each, no barrier inside --------------------------------------
        for (int c = 0; c < p.nwide; ++c, ++gwc) {
            const float* hb = halo + (gwc & 1) * HP * AS + (wy * HW_COLS + lrow) * AS + khalf;
            const bool last = (c + 1 == p.nwide);
            // where the weight stream continues after this chunk: next chunk, the image block, or chunk 0 of the next tile
            const unsigned wcur = (unsigned)c * CHUNK_BYTES;
            const unsigned wnext = last ? (has_img ? img_w_off : 0u) : wcur + CHUNK_BYTES;
            auto read_a = [&](int u, int slot) {
                const int tap = u >> 2, k8 = u & 3;  // compile-time after unrolling
                const int ky = tap / 3, kx = tap - ky * 3;
                fa[slot][0] = *reinterpret_cast<const float4*>(hb + (ky * HW_COLS + kx) * AS + k8 * 8);
            };
            read_a(0, 0);
#pragma unroll
            for (int u = 0; u < 36; ++u) {
                // prefetch the weights of step u+RD (possibly in the next block of the stream) and the halo fragment of step u+1
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nu = u + RD - 36;  // >= 0: past this chunk
                    fb[(u + RD) % RING][j] = (nu < 0)                        ? ldw(wcur + (unsigned)(u + RD) * STEP_BYTES, j)
                                             : (last && has_img && nu >= 5) ? ldw((unsigned)(nu - 5) * STEP_BYTES, j)  // beyond the 5 image steps: next tile
                                                                            : ldw(wnext + (unsigned)nu * STEP_BYTES, j);
                }
                if (u + 1 < 36) read_a(u + 1, (u + 1) & 1);
                if (u == 16) {
                    if constexpr (EPI_PREFETCH) {
                        if (last) load_tables(0);
                    }
                }
                mfma4(u & 1, u % RING, u >> 2, u >> 2);
#ifndef HX_NO_STEP_FENCE
                // keep the step's weight prefetch IN the step: left alone the scheduler sinks the loads of step u + RD to just before their
                // use (two live fragments instead of RING), which turns the ring into a one-step prefetch and exposes the L2 latency
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            if (!last) CP_BARRIER();  // halo stage consumed; the other stage is ready (the tile's last barrier follows below)
        }
        // ---- image step: K = 9 taps x 4 channels (+4 zero) = 5 k8-steps; half-wave h handles tap 2s+h ----
        if (has_img) {
            const float* ib = imgh + (k & 1) * HP * 4 + (wy * HW_COLS + lrow) * 4;
#pragma unroll
            for (int s5 = 0; s5 < 5; ++s5) {
                const int tl = 2 * s5, th = (2 * s5 + 1 < 9) ? 2 * s5 + 1 : 8;  // tap 9 does not exist: its weights are zero
                const int offl = ((tl / 3) * HW_COLS + tl % 3) * 4, offh = ((th / 3) * HW_COLS + th % 3) * 4;
                fa[s5 & 1][0] = *reinterpret_cast<const float4*>(ib + (half ? offh : offl));
                // the ring continues from the wide part (36 % RING == 0): image step s5 sits in slot s5 % RING
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nu = s5 + RD;  // next fragments: the remaining image steps, then steps 0.. of the next tile's chunk 0
                    fb[nu % RING][j] = (nu < 5) ? ldw(img_w_off + (unsigned)nu * STEP_BYTES, j) : ldw((unsigned)(nu - 5) * STEP_BYTES, j);
                }
                mfma4(s5 & 1, s5 % RING, tl, th);
            }
            // the ring is now 5 steps out of phase: step v of the next tile sits in slot (5 + v) % RING -> realign to slot v
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float4 tmp[RD];
#pragma unroll
                for (int v = 0; v < RD; ++v) tmp[v] = fb[(5 + v) % RING][j];
#pragma unroll
                for (int v = 0; v < RD; ++v) fb[v][j] = tmp[v];
            }
        }
#ifdef HX_NOEPI
        if (p.B < 0)  // timing experiment: never true, keeps the accumulators alive
#endif
        epilogue(n, y0, x0);
        CP_BARRIER();  // tile done: the consumed halo stage may be refilled
    }
}

template <int TMW, int TN, int MODE>
int launch_halo(HaloK k, hipStream_t st) {
    constexpr int TH = 4 * TMW, HP = (TH + 2) * HW_COLS, BN = 32 * TN;
    k.tiles_y = (k.H + TH - 1) / TH;
    k.tiles_x = (k.Wd + 31) / 32;
    k.ntiles = k.B * k.tiles_y * k.tiles_x;
    const bool img = k.img != nullptr;
    const size_t lds = (size_t)(2 * HP * 36 + (img ? 2 * HP * 4 : 0)) * sizeof(float);
    const size_t lds_max = (size_t)(2 * HP * 36 + 2 * HP * 4) * sizeof(float);
    (void)BN;
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

// Fragment-major layout (what one consumer wave loads per k8-step is one contiguous 1 KiB per 32 output channels):
//   every 32-channel slice (sources in order): [tap (9)][k8 (4)][half (2)][cout_pad][4]  = W[co][slice*32 + k8*8 + half*4 + e] at `tap`
//   a trailing 4-channel image source:         [s5 (5)][half (2)][cout_pad][4]           = W[co][image channel e] at tap 2*s5+half
//   (tap 9 and the 4th image channel do not exist: zeros)
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
            for (int s5 = 0; s5 < 5; ++s5)
                for (int half = 0; half < 2; ++half) {
                    const int t = 2 * s5 + half;
                    if (t >= 9) continue;
                    for (int co = 0; co < cout; ++co)
                        for (int e = 0; e < Cr; ++e)
                            dst[base + (((size_t)s5 * 2 + half) * bn + co) * 4 + e] = w[src_index(cbase + e, t, co)];
                }
            base += (size_t)bn * 40;
        } else {
            for (int ch = 0; ch < C / 32; ++ch)
                for (int t = 0; t < 9; ++t)
                    for (int k8 = 0; k8 < 4; ++k8)
                        for (int half = 0; half < 2; ++half)
                            for (int co = 0; co < cout; ++co)
                                for (int e = 0; e < 4; ++e) {
                                    const int c = ch * 32 + k8 * 8 + half * 4 + e;
                                    if (c < Cr)
                                        dst[base + (((((size_t)ch * 9 + t) * 4 + k8) * 2 + half) * bn + co) * 4 + e] = w[src_index(cbase + c, t, co)];
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
    if (d->cout > 64 || d->cout % 4 != 0 || d->group_rows) return false;
    // the epilogue moves four channels per 16-byte access
    if ((d->out_raw && d->out_raw_ld % 4) || (d->out_act && d->out_act_ld % 4) || (d->residual && d->residual_ld % 4)) return false;
    if ((((uintptr_t)d->out_raw) | ((uintptr_t)d->out_act) | ((uintptr_t)d->residual)) & 15) return false;
    if (d->src[0].mode == CP_SRC_ZERO_INSERT_X2) return false;  // transposed-conv gather: generic kernel only
    if (d->src[0].channels % 32 != 0 || d->src[0].pre_scale) return false;
    if (d->num_sources == 2) {
        if (d->src[1].mode != CP_SRC_DIRECT || d->src[1].pre_scale) return false;
        if (!(d->src[1].channels == 4 || d->src[1].channels % 32 == 0)) return false;
    }
    if (d->tap_label && d->epi_label && d->tap_label != d->epi_label) return false;
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
    k.norm = d->row_scale != nullptr;  // tap mask without row_scale = un-normalised (the data gradient of a partial convolution)
    k.out_raw = d->out_raw; k.raw_ld = d->out_raw_ld; k.out_act = d->out_act; k.act_ld = d->out_act_ld;
    k.head_w = d->head_out ? d->head_weights : nullptr; k.head_out = d->head_out; k.head_cout = d->head_cout; k.head_ld = d->head_out_ld;
    k.head_lab = d->head_out ? d->head_label_out : nullptr; k.head_lab_classes = d->head_label_classes;
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
