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

constexpr int HW_COLS = 34;  // 32 output columns + 2 halo columns
enum : int { H_BILINEAR = 2, H_PARTIAL = 4, H_SEL = 8 };

struct HSrc {
    const float* data;
    const uint8_t* sel;
    int C, ld, mode, Hs, Ws;
    unsigned bytes;
    int nchunks;  // C/32, or 1 for the 4-channel image source
    int kc;       // 32, or 8 for the 4-channel source (4 data + 4 zero)
};

struct HaloK {
    HSrc s[2];
    const float* W;   // [chunk][tap][cout_pad][kc]
    unsigned w_bytes;
    int B, H, Wd, Cout, nchunks;
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

template <int TMW, int TN, int MODE>
__global__ __launch_bounds__(512, (TMW * TN <= 2) ? 4 : 2) void conv_halo_kernel(const HaloK p) {
    constexpr int TH = 4 * TMW;             // output rows per tile
    constexpr int HR = TH + 2;              // halo rows
    constexpr int HP = HR * HW_COLS;        // halo pixels
    constexpr int BN = 32 * TN;
    constexpr int AS = 36;                  // halo pixel stride (floats) for 32-channel chunks
    constexpr int AS4 = 12;                 // for the 8-wide (4+4) chunk
    constexpr bool BILINEAR = (MODE & H_BILINEAR) != 0;
    constexpr bool PARTIAL = (MODE & H_PARTIAL) != 0;
    constexpr bool SEL = (MODE & H_SEL) != 0;
    constexpr int NV = BILINEAR ? 4 : 1;
    constexpr int NIT = (HP * 8 + 255) / 256;   // halo float4 elements per producer thread (32-ch chunk)
    constexpr int IPS = (NIT + 7) / 8;          // iterations issued per tap step (8 issuing steps)
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;                        // [2][HP][AS]
    float* bst = smem + 2 * HP * AS;           // [2][BN][AS]

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.s[0].data, 0, p.s[0].bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.s[1].data ? p.s[1].data : p.s[0].data), 0,
                                                                          p.s[1].data ? p.s[1].bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc((void*)(SEL ? (const void*)p.s[0].sel : (const void*)p.W), 0,
                                                                          SEL ? p.lab_bytes : 0u, 0x00020000);
    auto ldb4 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned off) -> float4 {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    };

    // tiles of this block: blockIdx.x, blockIdx.x + gridDim.x, ...
    const int bid = cp::xcd_remap(blockIdx.x, gridDim.x);  // neighbouring tiles (shared halo rows) on one XCD
    const int my_tiles = (p.ntiles - bid + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_chunks = my_tiles * p.nchunks;
    auto tile_origin = [&](int k, int& n, int& y0, int& x0) {
        int t = bid + k * (int)gridDim.x;
        x0 = (t % p.tiles_x) * 32;
        t /= p.tiles_x;
        y0 = (t % p.tiles_y) * TH;
        n = t / p.tiles_y;
    };

    if (producer) {
        // ------------------------------ producers --------------------------------------------
        float4 hreg[IPS][NV];
        int hdst[IPS];      // LDS float offset (-1 = nothing)
        int hflag[IPS];     // bilinear parity bits + bit0 valid
        float4 breg[TN];
        int selb[IPS];      // SEL bytes for the NEXT issue step

        auto chunk_info = [&](int gc, int& n, int& y0, int& x0, int& si, int& c0, int& kc) {
            const int k = gc / p.nchunks, c = gc - k * p.nchunks;
            tile_origin(k, n, y0, x0);
            si = (c >= p.s[0].nchunks) ? 1 : 0;
            const int cl = c - (si ? p.s[0].nchunks : 0);
            kc = si ? p.s[1].kc : p.s[0].kc;
            c0 = cl * 32;
        };
        auto load_selbytes = [&](int gc, int it0) {
            if constexpr (SEL) {
                int n, y0, x0, si, c0, kc;
                if (gc >= total_chunks) return;
                chunk_info(gc, n, y0, x0, si, c0, kc);
#pragma unroll
                for (int u = 0; u < IPS; ++u) {
                    const int e = (it0 + u) * 256 + tid;
                    const int hp = kc == 32 ? (e >> 3) : e;
                    const int hy = hp / HW_COLS, hx = hp - hy * HW_COLS;
                    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
                    const bool inb = hp < HP && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd;
                    selb[u] = __builtin_amdgcn_raw_buffer_load_b8(rss, inb ? ((n * p.H + y) * p.Wd + x) : (int)OOB, 0, 0);
                }
            }
        };
        auto issue_halo = [&](int gc, int it0) {
            int n, y0, x0, si, c0, kc;
            if (gc >= total_chunks) {
#pragma unroll
                for (int u = 0; u < IPS; ++u) hdst[u] = -1;
                return;
            }
            chunk_info(gc, n, y0, x0, si, c0, kc);
            const __amdgpu_buffer_rsrc_t rs = si ? rs1 : rs0;
            const int sld = si ? p.s[1].ld : p.s[0].ld;
            const int as = kc == 32 ? AS : AS4;
            const int per = kc == 32 ? 8 : 1;
#pragma unroll
            for (int u = 0; u < IPS; ++u) {
                const int e = (it0 + u) * 256 + tid;
                const int hp = e / per, f = e - hp * per;
                const int hy = hp / HW_COLS, hx = hp - hy * HW_COLS;
                const int y = y0 - 1 + hy, x = x0 - 1 + hx;
                const bool in_tile = hp < HP;
                const bool inb = in_tile && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd;
                hdst[u] = in_tile ? (hp * as + f * 4) : -1;
                hflag[u] = (inb ? 1 : 0) | ((y & 1) << 1) | ((x & 1) << 2) | (si << 3) | (kc == 8 ? 16 : 0);
                const int cb = (c0 + f * 4) * 4;
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
                    hreg[u][0] = ldb4(rs, o00);
                    hreg[u][1] = ldb4(rs, o01);
                    hreg[u][2] = ldb4(rs, o10);
                    hreg[u][3] = ldb4(rs, o11);
                } else if constexpr (SEL) {
                    unsigned o;
                    if (si == 0) {
                        const int sl = selb[u];
                        o = inb ? (unsigned)((((n * p.s[0].Hs + (y >> 1) + (sl >> 1)) * p.s[0].Ws + (x >> 1) + (sl & 1)) * sld) * 4 + cb) : OOB;
                    } else {
                        o = inb ? (unsigned)((((n * p.H + y) * p.Wd + x) * sld) * 4 + cb) : OOB;
                    }
                    hreg[u][0] = ldb4(rs, o);
                } else {
                    hreg[u][0] = ldb4(rs, inb ? (unsigned)((((n * p.H + y) * p.Wd + x) * sld) * 4 + cb) : OOB);
                }
            }
        };
        auto store_halo = [&](int stage) {
            float* h = halo + stage * HP * AS;
#pragma unroll
            for (int u = 0; u < IPS; ++u) {
                if (hdst[u] < 0) continue;
                float4 v = hreg[u][0];
                if constexpr (BILINEAR) {
                    if (!(hflag[u] & 8)) {
                        const float fy = (hflag[u] & 2) ? 0.25f : 0.75f, fx = (hflag[u] & 4) ? 0.25f : 0.75f;
                        const float gy = 1.f - fy, gx = 1.f - fx;
                        const float4 v01 = hreg[u][1], v10 = hreg[u][2], v11 = hreg[u][3];
                        v.x = (v.x * gx + v01.x * fx) * gy + (v10.x * gx + v11.x * fx) * fy;
                        v.y = (v.y * gx + v01.y * fx) * gy + (v10.y * gx + v11.y * fx) * fy;
                        v.z = (v.z * gx + v01.z * fx) * gy + (v10.z * gx + v11.z * fx) * fy;
                        v.w = (v.w * gx + v01.w * fx) * gy + (v10.w * gx + v11.w * fx) * fy;
                    }
                }
                *reinterpret_cast<float4*>(h + hdst[u]) = v;
                if (hflag[u] & 16)  // 8-wide chunk: k 4..7 meet zero weights but must be finite
                    *reinterpret_cast<float4*>(h + hdst[u] + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        // weights of flattened step gs = gc*9 + tap: block [cout_pad][kc] at chunk-major offset
        auto issue_b = [&](int gs) {
            if (gs >= total_chunks * 9) return;
            const int gc = gs / 9, tap = gs - gc * 9;
            const int c = gc % p.nchunks;
            const int si = (c >= p.s[0].nchunks) ? 1 : 0;
            const int kc = si ? p.s[1].kc : p.s[0].kc;
            // chunk c starts at: (#32-chunks before it) * 9*BN*32 (+ nothing before the single 8-chunk, which is always last)
            const unsigned base = (unsigned)(c * 9 * BN * 32 + tap * BN * kc) * 4u;
            const int per_row = kc >> 2;  // float4 per weight row
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int e = j * 256 + tid;
                breg[j] = (e < BN * per_row) ? ldb4(rsw, base + (unsigned)e * 16u) : make_float4(0, 0, 0, 0);
            }
        };
        auto store_b = [&](int gs) {
            if (gs >= total_chunks * 9) return;
            const int gc = gs / 9;
            const int c = gc % p.nchunks;
            const int si = (c >= p.s[0].nchunks) ? 1 : 0;
            const int kc = si ? p.s[1].kc : p.s[0].kc;
            const int per_row = kc >> 2;
            const int as = kc == 32 ? AS : AS4;
            float* b = bst + (gs & 1) * BN * AS;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int e = j * 256 + tid;
                if (e < BN * per_row) {
                    const int row = e / per_row, f = e - row * per_row;
                    *reinterpret_cast<float4*>(b + row * as + f * 4) = breg[j];
                }
            }
        };

        // prologue: whole halo of chunk 0, weights of step 0; prefetch weights of step 1
        for (int it = 0; it < NIT; it += IPS) {
            load_selbytes(0, it);
            issue_halo(0, it);
            store_halo(0);
        }
        issue_b(0);
        store_b(0);
        issue_b(1);
        load_selbytes(1, 0);
        CP_BARRIER();
        int gs = 0;
        for (int gc = 0; gc < total_chunks; ++gc) {
            for (int tap = 0; tap < 9; ++tap, ++gs) {
                // consumers multiply step gs from stages (gc&1, gs&1)
                store_b(gs + 1);                           // issued one step ago
                issue_b(gs + 2);
                if (tap >= 1) store_halo((gc + 1) & 1);    // halo elements issued at tap-1
                if (tap < 8) {
                    issue_halo(gc + 1, tap * IPS);
                    load_selbytes(gc + 1, (tap + 1) * IPS);
                } else {
                    load_selbytes(gc + 2, 0);
                }
                CP_BARRIER();
            }
        }
        return;
    }

    // ---------------------------------- consumers ------------------------------------------------
    const int wy = wave;  // consumer wave w owns tile rows [w*TMW, (w+1)*TMW)
    const int lrow = lane & 31;
    const int khalf = (lane >> 5) * 4;
    const int hi4 = (lane >> 5) * 4;
    f32x16 acc[TMW][TN];
    float4 fa[2][TMW], fb[2][TN];
    int gs = 0;
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
        // partial conv: 9-bit tap mask of this lane's pixel in each of its rows
        int pmask[TMW];
        if constexpr (PARTIAL) {
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                const int y = y0 + wy * TMW + i, x = x0 + lrow;
                int m = 0;
                if (y < p.H && x < p.Wd) {
                    const uint8_t* lb = p.label + (size_t)n * p.H * p.Wd;
                    const int c = lb[(size_t)y * p.Wd + x];
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                        const bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.Wd && lb[(size_t)yy * p.Wd + xx] == c;
                        m |= ok ? (1 << t) : 0;
                    }
                }
                pmask[i] = m;
            }
        }
        for (int c = 0; c < p.nchunks; ++c) {
            const int gc = k * p.nchunks + c;
            const bool narrow = c >= p.s[0].nchunks && p.s[1].kc == 8;
            const float* hb = halo + (gc & 1) * HP * AS;
            const int as = narrow ? AS4 : AS;
            const int nk8 = narrow ? 1 : 4;
            auto read_frags = [&](int tap, int k8, int slot, int bstage) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const float* a = hb + ((wy * TMW + ky) * HW_COLS + lrow + kx) * as + k8 * 8 + khalf;
                const float* b = bst + bstage * BN * AS + lrow * as + k8 * 8 + khalf;
#pragma unroll
                for (int i = 0; i < TMW; ++i) fa[slot][i] = *reinterpret_cast<const float4*>(a + i * HW_COLS * as);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const float4*>(b + j * 32 * as);
            };
            auto mfma_step = [&](int tap, int slot) {
#pragma unroll
                for (int i = 0; i < TMW; ++i) {
                    float4 av = fa[slot][i];
                    if constexpr (PARTIAL) {
                        if (!((pmask[i] >> tap) & 1)) av = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, fb[slot][j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, fb[slot][j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, fb[slot][j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, fb[slot][j].w, acc[i][j], 0, 0, 0);
                    }
                }
            };
            if (nk8 == 4) {
                read_frags(0, 0, 0, gs & 1);
                for (int tap = 0; tap < 9; ++tap, ++gs) {
                    read_frags(tap, 1, 1, gs & 1);
                    mfma_step(tap, 0);
                    read_frags(tap, 2, 0, gs & 1);
                    mfma_step(tap, 1);
                    read_frags(tap, 3, 1, gs & 1);
                    mfma_step(tap, 0);
                    CP_BARRIER();  // all reads of weight stage gs&1 are complete; stage (gs+1)&1 is ready
                    if (tap < 8) read_frags(tap + 1, 0, 0, (gs + 1) & 1);
                    mfma_step(tap, 1);
                }
            } else {
                for (int tap = 0; tap < 9; ++tap, ++gs) {
                    read_frags(tap, 0, 0, gs & 1);
                    CP_BARRIER();
                    mfma_step(tap, 0);
                }
            }
        }
        // ---- epilogue -----------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            const int y = y0 + wy * TMW + i;
            if (y >= p.H) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xr = (r & 3) + 8 * (r >> 2) + hi4;
                const int x = x0 + xr;
                float rs = 1.f;
                if constexpr (PARTIAL) {
                    const int pm = __shfl(pmask[i], xr);  // the mask lives in the lane that owns column xr (all lanes active here)
                    rs = 9.0f / (float)max(__popc(pm), 1);
                }
                if (x >= p.Wd) continue;
                const size_t m = ((size_t)n * p.H + y) * p.Wd + x;
                const int lab = p.clade ? (int)p.label[m] : 0;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int co = j * 32 + lrow;
                    if (co >= p.Cout) continue;
                    float v = acc[i][j][r] * rs;
                    if (p.residual) v += p.residual[m * p.res_ld + co];
                    if (p.out_raw) p.out_raw[m * p.raw_ld + co] = v;
                    if (p.out_act) {
                        float t = v;
                        if (p.scale) t = t * p.scale[lab * p.Cout + co] + p.shift[lab * p.Cout + co];
                        if (p.act == CP_ACT_RELU) t = fmaxf(t, 0.f);
                        else if (p.act == CP_ACT_LEAKY01) t = fmaxf(t, 0.f) - fmaxf(-0.1f * t, 0.f);
                        p.out_act[m * p.act_ld + co] = t;
                    }
                }
            }
        }
    }
}

template <int TMW, int TN, int MODE>
int launch_halo(HaloK k, hipStream_t st) {
    constexpr int TH = 4 * TMW, HP = (TH + 2) * HW_COLS, BN = 32 * TN;
    k.tiles_y = (k.H + TH - 1) / TH;
    k.tiles_x = (k.Wd + 31) / 32;
    k.ntiles = k.B * k.tiles_y * k.tiles_x;
    const size_t lds = (size_t)(2 * HP * 36 + 2 * BN * 36) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<TMW, TN, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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
    for (int s = 0; s < num_sources; ++s) n += (channels[s] == 4) ? 9 * bn * 8 : (channels[s] / 32) * 9 * bn * 32;
    return n;
}

int halo_pack_weights(const float* w, int layout, int cout, int num_sources, const int* channels, const int* real_channels, float* dst) {
    const int bn = cout <= 32 ? 32 : 64;
    int cin = 0;
    for (int s = 0; s < num_sources; ++s) cin += real_channels[s];
    const int total = halo_weight_floats(cout, num_sources, channels);
    for (int i = 0; i < total; ++i) dst[i] = 0.f;
    size_t base = 0;
    int cbase = 0;
    for (int s = 0; s < num_sources; ++s) {
        const int C = channels[s], Cr = real_channels[s];
        const int kc = (C == 4) ? 8 : 32;
        const int nch = (C == 4) ? 1 : C / 32;
        for (int ch = 0; ch < nch; ++ch)
            for (int t = 0; t < 9; ++t)
                for (int co = 0; co < cout; ++co)
                    for (int kk = 0; kk < kc; ++kk) {
                        const int c = ch * 32 + kk;
                        if (c >= Cr || (C == 4 && kk >= 4)) continue;
                        const int ci = cbase + c, ky = t / 3, kx = t % 3;
                        const size_t src = (layout == 0) ? ((((size_t)ky * 3 + kx) * cin + ci) * cout + co)
                                                         : ((((size_t)ci * 3 + ky) * 3 + kx) * cout + co);
                        dst[base + ((size_t)(ch * 9 + t) * bn + co) * kc + kk] = w[src];
                    }
        base += (size_t)nch * 9 * bn * kc;
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
    int nchunks = 0;
    for (int s = 0; s < d->num_sources; ++s) {
        const cp_conv_source& in = d->src[s];
        HSrc& o = k.s[s];
        o.data = in.data;
        o.sel = in.sel;
        o.C = in.channels;
        o.ld = in.ld;
        o.mode = in.mode;
        o.Hs = (in.mode == CP_SRC_DIRECT) ? d->in_h : d->in_h / 2;
        o.Ws = (in.mode == CP_SRC_DIRECT) ? d->in_w : d->in_w / 2;
        const long long nbytes = (long long)d->batch * o.Hs * o.Ws * in.ld * 4;
        CP_REQUIRE(nbytes < (1LL << 31), "cp_conv2d_fwd_f32: source %d spans %lld bytes; 32-bit range-checked addressing needs < 2 GiB", s, nbytes);
        o.bytes = (unsigned)nbytes;
        o.kc = in.channels == 4 ? 8 : 32;
        o.nchunks = in.channels == 4 ? 1 : in.channels / 32;
        nchunks += o.nchunks;
    }
    int chans[2] = {d->src[0].channels, d->num_sources > 1 ? d->src[1].channels : 0};
    k.W = d->weights_halo;
    k.w_bytes = (unsigned)(halo_weight_floats(d->cout, d->num_sources, chans) * sizeof(float));
    k.B = d->batch; k.H = d->in_h; k.Wd = d->in_w; k.Cout = d->cout; k.nchunks = nchunks;
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
