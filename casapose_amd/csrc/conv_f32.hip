// Fused implicit-GEMM convolution forward, exact fp32 on v_mfma_f32_32x32x2_f32 (gfx950).
//
// GEMM view: M = batch*out_h*out_w output pixels, N = cout, K = sum over sources of
// taps*channels.  A (pixels x K) is gathered on the fly from the NHWC activations --
// with the source's spatial mode (direct / guided-nearest x2 / bilinear x2), the optional
// per-channel affine and the partial-convolution tap mask applied while the tile is in
// registers -- and staged through LDS; B (cout x K) comes from the pre-packed weights.
// A block of 256 threads (4 waves, one per SIMD) owns a BM x BN output tile; each wave
// owns TM x TN accumulators of 32x32.  K advances in chunks of 32 with register-staged
// double buffering: global loads for chunk q+1 are in flight while chunk q is multiplied.
//
// Operand trick: an MFMA 32x32x2 consumes ONE f32 of A and B per lane (lane l supplies
// k = l>>5).  The order of K inside a GEMM is free as long as A and B agree, so both
// tiles are stored [row][k] with k contiguous and read with one ds_read_b128 per 8 k:
// half-wave 0 takes k 0..3, half-wave 1 takes k 4..7, and MFMA step j multiplies the
// pair (j, 4+j).  Row stride 36 floats (9 x 16 B) makes those reads conflict-free.
//
// Reference call sites replaced: see include/casapose_hip.h (cp_conv2d_fwd_f32).
#include "common.h"
#include "epilogue.h"

#include <algorithm>

namespace cp {
int halo_weight_floats(int cout, int num_sources, const int* channels);
int halo_pack_weights(const float* w, int layout, int cout, int num_sources, const int* channels, const int* real_channels, float* dst);
bool halo_applicable(const cp_conv_desc* d);
int launch_halo_conv(const cp_conv_desc* d, hipStream_t st);
bool stem_applicable(const cp_conv_desc* d);
int launch_stem_conv(const cp_conv_desc* d, hipStream_t st);
}  // namespace cp

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_STRIDE = 36;  // floats per tile row (32 + 4 pad)
constexpr int MAX_TAPS = 64;
#ifndef CP_CONV_WAVES
#define CP_CONV_WAVES 2
#endif
#ifndef CP_CONV_NBUF
#define CP_CONV_NBUF 2
#endif
constexpr int NBUF = CP_CONV_NBUF;  // LDS stages: 2 = one barrier per K chunk, 1 = two barriers but half the LDS (more blocks per CU)

struct SrcK {
    const float* data;
    const uint8_t* sel;
    const float* pre_scale;
    const float* pre_shift;
    int C, ld, mode, Hs, Ws;
    unsigned bytes;  // extent of `data` for the buffer descriptor (range-checked loads)
    int c4;       // 1: channels == 4, eight taps per chunk
    int cpt;      // chunks per tap (C/32) when !c4
    int nchunks;  // K chunks contributed by this source
};

struct ConvK {
    SrcK s[2];
    const float* W;
    int ktot;
    unsigned w_bytes, lab_bytes;  // extents of W and of the tap_label / sel maps
    int B, Hin, Win, Ho, Wo, Cout, KH, KW, stride, dil, pad;
    int M;
    const uint8_t* tap_label;
    const float* row_scale;
    const float* residual;
    int res_ld;
    const float* scale;
    const float* shift;
    const uint8_t* epi_label;
    int act;
    float* out_raw;
    int raw_ld;
    float* out_act;
    int act_ld;
    int tiles_m, tiles_n, nchunks;
    int group_rows, group_wstride;  // grouped GEMM: rows [g*group_rows, ...) read weights at W + g*group_wstride
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Operand-loader variants (compile-time so that the prefetch path has no data-dependent
// control flow: every load is issued unconditionally from a clamped, always-valid address
// and zeroing / affine / interpolation happen when the registers are written to LDS,
// AFTER the MFMA block -- this is what keeps the loads in flight under the MFMAs).
enum : int { F_PRE = 1, F_BILINEAR = 2, F_PARTIAL = 4, F_SEL = 8 };

template <int WGM, int WGN, int TM, int TN, int MODE>
__global__ __launch_bounds__(512, (TM * TN >= 4) ? 2 : 4) void conv_f32_kernel(const ConvK p) {
    constexpr int BM = WGM * TM * 32;
    constexpr int BN = WGN * TN * 32;
    constexpr int RM = BM / 32;  // A rows staged per thread
    constexpr int RN = BN / 32;  // B rows staged per thread
    constexpr bool PRE = (MODE & F_PRE) != 0;
    constexpr bool BILINEAR = (MODE & F_BILINEAR) != 0;
    constexpr bool PARTIAL = (MODE & F_PARTIAL) != 0;
    constexpr bool SEL = (MODE & F_SEL) != 0;
    constexpr int NV = BILINEAR ? 4 : 1;
    static_assert(WGM * WGN == 4, "4 consumer waves per block");
    static_assert(NBUF == 2, "producer/consumer pipeline uses two LDS stages");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                            // [2][BM][LDS_STRIDE]
    float* Bs = smem + NBUF * BM * LDS_STRIDE;   // [NBUF][BN][LDS_STRIDE]
    int* tapoff = reinterpret_cast<int*>(Bs + NBUF * BN * LDS_STRIDE);  // [MAX_TAPS] (dy<<16)|(dx & 0xffff)

    // Wave specialisation: waves 0-3 (one per SIMD) only read fragments from LDS and issue
    // MFMAs; waves 4-7 (their SIMD partners) only gather operands (global -> registers ->
    // LDS).  The matrix pipe and the VALU/VMEM/LDS pipes are separate, so the hardware
    // overlaps the two instruction streams without depending on compiler scheduling.
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int tid = threadIdx.x & 255;  // index within the role
    const int lane = tid & 63;
    const int wm = (wave & 3) / WGN, wn = (wave & 3) % WGN;

    const int logical = cp::xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tile_n = logical % p.tiles_n;
    const int tile_m = logical / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const int ntaps = p.KH * p.KW;
    if (threadIdx.x < MAX_TAPS) {
        int ky = tid / p.KW, kx = tid - ky * p.KW;
        tapoff[tid] = ((ky * p.dil) << 16) | ((kx * p.dil) & 0xffff);
    }

    // ---- per-thread staging coordinates -------------------------------------------------
    // All operand fetches are range-checked raw-buffer loads with 32-bit byte offsets: an
    // offset of OOB (>= the descriptor's extent) returns zeros, which is exactly the zero
    // padding / tile-edge behaviour needed, with no branches and one VALU add per row.
    constexpr unsigned OOB = 0x80000000u;
    const int col4 = tid & 7;   // which float4 of the 32-wide K chunk
    const int rbase = tid >> 3; // 0..31
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.s[0].data, 0, p.s[0].bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.s[1].data ? p.s[1].data : p.s[0].data), 0,
                                                                          p.s[1].data ? p.s[1].bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(PARTIAL ? (const void*)p.tap_label : (const void*)p.W), 0,
                                                                          PARTIAL ? p.lab_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc((void*)(SEL ? (const void*)p.s[0].sel : (const void*)p.W), 0,
                                                                          SEL ? p.lab_bytes : 0u, 0x00020000);
    int r_pix[RM];   // (n*Hin + iy0)*Win + ix0 : pixel index of the tap-(0,0) position (may be "negative")
    int r_iy0[RM], r_ix0[RM], r_clab[RM];
    int r_nb[RM];    // SEL / BILINEAR: n * Hs * Ws
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        int m = m0 + rbase + 32 * i;
        if (m < p.M) {
            int n = m / (p.Ho * p.Wo);
            int rem = m - n * (p.Ho * p.Wo);
            int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            r_iy0[i] = oy * p.stride - p.pad;
            r_ix0[i] = ox * p.stride - p.pad;
            r_pix[i] = (n * p.Hin + r_iy0[i]) * p.Win + r_ix0[i];
            r_nb[i] = n * p.s[0].Hs * p.s[0].Ws;
            r_clab[i] = PARTIAL ? (int)p.tap_label[((size_t)n * p.Hin + oy) * p.Win + ox] : 0;
        } else {
            r_iy0[i] = -0x10000000;  // always out of bounds (also when added to the invalid-tap dy)
            r_ix0[i] = 0;
            r_pix[i] = 0;
            r_nb[i] = 0;
            r_clab[i] = -1;
        }
    }
    unsigned r_wof[RN];  // byte offset of this thread's float4 in weight row co (OOB for co >= Cout)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int co = n0 + rbase + 32 * j;
        const unsigned gofs = p.group_rows ? (unsigned)(m0 / p.group_rows) * (unsigned)p.group_wstride * 4u : 0u;
        r_wof[j] = (co < p.Cout) ? (unsigned)((co * p.ktot + col4 * 4) * 4) + gofs : OOB;
    }
    __syncthreads();  // tapoff visible

    // PARTIAL: one bit per tap and staged row, [label(tap position) == label(centre)], computed ONCE per tile (nine byte loads per
    // row) instead of one label load per row and K chunk inside the pipeline
    unsigned long long r_pm[RM];
    if constexpr (PARTIAL) {
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            unsigned long long m = 0;
            for (int t = 0; t < ntaps; ++t) {
                const int to = tapoff[t];
                const int iy = r_iy0[i] + (to >> 16), ix = r_ix0[i] + (int)(short)(to & 0xffff);
                const bool inb = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);
                const int lb = __builtin_amdgcn_raw_buffer_load_b8(rsl, inb ? (r_pix[i] + (to >> 16) * p.Win + (int)(short)(to & 0xffff)) : (int)OOB, 0, 0);
                m |= (inb && lb == r_clab[i]) ? (1ull << t) : 0ull;
            }
            r_pm[i] = m;
        }
    }

    // ---- staging registers (chunk q+1 while chunk q is multiplied) ------------------------
    float4 areg[RM][NV], breg[RN];
    float4 pre_s, pre_b;          // PRE
    int aflag[RM];                // bit0 in-bounds, bit1 y parity, bit2 x parity
    int st_tap = 0;               // PARTIAL: tap of the staged chunk (per thread for 4-channel sources)
    int selreg[RM];               // SEL: neighbour index for the NEXT chunk to be issued
    int st_si = 0;                // source of the staged chunk (uniform)
    bool st_pre = false;

    auto ldb4 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned off, int soff) -> float4 {
#ifdef CP_EXP_NOLOAD
        return make_float4(__builtin_bit_cast(float, off), 1.f, 2.f, (float)soff);  // timing experiment only
#else
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, soff, 0));
#endif
    };
    auto chunk_tap = [&](int q, int& si, int& tap, int& coff) {
        si = (q >= p.s[0].nchunks) ? 1 : 0;
        const int ql = q - (si ? p.s[0].nchunks : 0);
        const int c4 = si ? p.s[1].c4 : p.s[0].c4;
        const int cpt = si ? p.s[1].cpt : p.s[0].cpt;
        if (c4) {
            tap = ql * 8 + col4;
            coff = 0;
        } else {
            tap = ql / cpt;
            coff = (ql - tap * cpt) * 32 + col4 * 4;
        }
    };
    auto tap_delta = [&](int tap, int& dy, int& dx) {
        const int to = tapoff[tap < MAX_TAPS ? tap : MAX_TAPS - 1];
        dy = (tap < ntaps) ? (to >> 16) : 0x20000000;
        dx = (int)(short)(to & 0xffff);
    };

    // SEL: fetch the neighbour-selection bytes of chunk q (consumed one iteration later)
    auto load_sel = [&](int q) {
        if constexpr (SEL) {
            int si, tap, coff, dy, dx;
            chunk_tap(q < p.nchunks ? q : p.nchunks - 1, si, tap, coff);
            tap_delta(tap, dy, dx);
            const int dpix = dy * p.Win + dx;
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                const int iy = r_iy0[i] + dy, ix = r_ix0[i] + dx;
                const bool inb = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);
                selreg[i] = __builtin_amdgcn_raw_buffer_load_b8(rss, inb ? (r_pix[i] + dpix) : (int)OOB, 0, 0);
            }
        }
    };

    const bool zins = p.s[0].mode == CP_SRC_ZERO_INSERT_X2;
    auto issue_chunk = [&](int q) {
#pragma unroll
        for (int j = 0; j < RN; ++j) breg[j] = ldb4(rsw, r_wof[j], q * (BK * 4));
        int si, tap, coff, dy, dx;
        chunk_tap(q, si, tap, coff);
        tap_delta(tap, dy, dx);
        const __amdgpu_buffer_rsrc_t rs = si ? rs1 : rs0;
        const int sld = si ? p.s[1].ld : p.s[0].ld;
        const int dpix = dy * p.Win + dx;
        st_si = si;
        st_tap = tap < MAX_TAPS ? tap : MAX_TAPS - 1;
        if constexpr (PRE) {
            const float* ps = si ? p.s[1].pre_scale : p.s[0].pre_scale;
            const float* pb = si ? p.s[1].pre_shift : p.s[0].pre_shift;
            st_pre = ps != nullptr;
            pre_s = ld4((st_pre ? ps : p.W) + coff);  // clamped to valid memory when unused
            pre_b = ld4((st_pre ? pb : p.W) + coff);
        }
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            const int iy = r_iy0[i] + dy, ix = r_ix0[i] + dx;
            const bool inb = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);
            const int gpix = r_pix[i] + dpix;  // valid when inb
            aflag[i] = (inb ? 1 : 0) | ((iy & 1) << 1) | ((ix & 1) << 2);
            if constexpr (BILINEAR) {
                unsigned o00, o01, o10, o11;
                if (si == 0) {  // uniform: the x2 source
                    const int Hs = p.s[0].Hs, Ws = p.s[0].Ws;
                    int y0 = (iy >> 1) - ((iy & 1) ? 0 : 1), x0 = (ix >> 1) - ((ix & 1) ? 0 : 1);
                    int y1 = min(y0 + 1, Hs - 1), x1 = min(x0 + 1, Ws - 1);
                    y0 = max(y0, 0);
                    x0 = max(x0, 0);
                    const int c4b = coff * 4;
                    o00 = inb ? (unsigned)(((r_nb[i] + y0 * Ws + x0) * sld) * 4 + c4b) : OOB;
                    o01 = inb ? (unsigned)(((r_nb[i] + y0 * Ws + x1) * sld) * 4 + c4b) : OOB;
                    o10 = inb ? (unsigned)(((r_nb[i] + y1 * Ws + x0) * sld) * 4 + c4b) : OOB;
                    o11 = inb ? (unsigned)(((r_nb[i] + y1 * Ws + x1) * sld) * 4 + c4b) : OOB;
                } else {
                    o00 = o01 = o10 = o11 = inb ? (unsigned)((gpix * sld + coff) * 4) : OOB;
                }
                areg[i][0] = ldb4(rs, o00, 0);
                areg[i][1] = ldb4(rs, o01, 0);
                areg[i][2] = ldb4(rs, o10, 0);
                areg[i][3] = ldb4(rs, o11, 0);
            } else if constexpr (SEL) {
                unsigned o;
                if (si == 0) {
                    const int sl = selreg[i];
                    o = inb ? (unsigned)(((r_nb[i] + ((iy >> 1) + (sl >> 1)) * p.s[0].Ws + (ix >> 1) + (sl & 1)) * sld + coff) * 4) : OOB;
                } else {
                    o = inb ? (unsigned)((gpix * sld + coff) * 4) : OOB;
                }
                areg[i][0] = ldb4(rs, o, 0);
            } else {
                unsigned o = inb ? (unsigned)((gpix * sld + coff) * 4) : OOB;
                if (zins && si == 0)  // zero-insertion x2 (transposed convolution): only even positions carry data
                    o = (inb && !((iy | ix) & 1)) ? (unsigned)(((r_nb[i] + (iy >> 1) * p.s[0].Ws + (ix >> 1)) * sld + coff) * 4) : OOB;
                areg[i][0] = ldb4(rs, o, 0);
            }
        }
    };

    auto store_chunk = [&](int buf) {
#ifdef CP_EXP_NOSTORE
        if (p.M >= 0) { asm volatile("" :: "v"(areg[0][0].x), "v"(breg[0].x)); return; }  // timing experiment only
#endif
        float* a = As + buf * BM * LDS_STRIDE;
        float* b = Bs + buf * BN * LDS_STRIDE;
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            float4 v = areg[i][0];
            if constexpr (BILINEAR) {
                if (st_si == 0) {
                    const float fy = (aflag[i] & 2) ? 0.25f : 0.75f, fx = (aflag[i] & 4) ? 0.25f : 0.75f;
                    const float gy = 1.f - fy, gx = 1.f - fx;
                    const float4 v01 = areg[i][1], v10 = areg[i][2], v11 = areg[i][3];
                    v.x = (v.x * gx + v01.x * fx) * gy + (v10.x * gx + v11.x * fx) * fy;
                    v.y = (v.y * gx + v01.y * fx) * gy + (v10.y * gx + v11.y * fx) * fy;
                    v.z = (v.z * gx + v01.z * fx) * gy + (v10.z * gx + v11.z * fx) * fy;
                    v.w = (v.w * gx + v01.w * fx) * gy + (v10.w * gx + v11.w * fx) * fy;
                }
            }
            if constexpr (PRE) {
                // the affine applies to real pixels only: padding stays exactly zero
                if (st_pre && (aflag[i] & 1)) {
                    v.x = v.x * pre_s.x + pre_b.x;
                    v.y = v.y * pre_s.y + pre_b.y;
                    v.z = v.z * pre_s.z + pre_b.z;
                    v.w = v.w * pre_s.w + pre_b.w;
                }
            }
            if constexpr (PARTIAL) {
                if (!((r_pm[i] >> st_tap) & 1ull)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            *reinterpret_cast<float4*>(a + (rbase + 32 * i) * LDS_STRIDE + col4 * 4) = v;
        }
#pragma unroll
        for (int j = 0; j < RN; ++j)
            *reinterpret_cast<float4*>(b + (rbase + 32 * j) * LDS_STRIDE + col4 * 4) = breg[j];
    };

    // raw barrier: LDS traffic must be complete, but the producers' global loads stay in
    // flight across it (a __syncthreads() would drain vmcnt and serialise the prefetch)
#define CP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    if (producer) {
        load_sel(0);
        issue_chunk(0);
        load_sel(1);
        store_chunk(0);
        if (p.nchunks > 1) {
            issue_chunk(1);
            load_sel(2);
        }
        CP_BARRIER();
        for (int q = 0; q < p.nchunks; ++q) {
            // stage (q+1)&1 was last read by the consumers in iteration q-1: free since the barrier
            if (q + 1 < p.nchunks) {
                store_chunk((q + 1) & 1);
                if (q + 2 < p.nchunks) {
                    issue_chunk(q + 2);
                    load_sel(q + 3);
                }
            }
            CP_BARRIER();
        }
        return;
    }

    // ---------------------------------- consumers ------------------------------------------
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lrow = lane & 31;
    const int khalf = (lane >> 5) * 4;

    // MFMA fragments, double-buffered across the four 8-deep k steps of a chunk so that the
    // LDS reads of step k+1 are in flight while step k multiplies.
    float4 fa[2][TM], fb[2][TN];
    auto read_frags = [&](int buf, int k8, int slot) {
        const float* a = As + buf * BM * LDS_STRIDE + (wm * TM * 32 + lrow) * LDS_STRIDE + khalf + k8 * 8;
        const float* b = Bs + buf * BN * LDS_STRIDE + (wn * TN * 32 + lrow) * LDS_STRIDE + khalf + k8 * 8;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][i] = *reinterpret_cast<const float4*>(a + i * 32 * LDS_STRIDE);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const float4*>(b + j * 32 * LDS_STRIDE);
    };
    auto mfma_step = [&](int slot) {
        // element-major order: consecutive MFMAs hit different accumulators
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float av = e == 0 ? fa[slot][i].x : e == 1 ? fa[slot][i].y : e == 2 ? fa[slot][i].z : fa[slot][i].w;
                    const float bv = e == 0 ? fb[slot][j].x : e == 1 ? fb[slot][j].y : e == 2 ? fb[slot][j].z : fb[slot][j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                }
    };

#ifdef CP_EXP_SETPRIO
    __builtin_amdgcn_s_setprio(CP_EXP_SETPRIO);  // MFMA waves win issue arbitration against their producer partners
#endif
    CP_BARRIER();  // stage 0 is ready
    read_frags(0, 0, 0);
    for (int q = 0; q < p.nchunks; ++q) {
        const int buf = q & 1;
        read_frags(buf, 1, 1);
        mfma_step(0);
        read_frags(buf, 2, 0);
        mfma_step(1);
        read_frags(buf, 3, 1);
        mfma_step(0);
        mfma_step(1);
        CP_BARRIER();  // stage buf^1 now holds chunk q+1; stage buf may be overwritten
        if (q + 1 < p.nchunks) read_frags(buf ^ 1, 0, 0);
    }
#undef CP_BARRIER

    // ---- epilogue (epilogue.h: batched, branch-free) ---------------------------------------
    const int hi4 = (lane >> 5) * 4;
    cp::EpiArgs ea;
    ea.row_scale = p.row_scale; ea.label = p.epi_label; ea.residual = p.residual; ea.scale = p.scale; ea.shift = p.shift;
    ea.out_raw = p.out_raw; ea.out_act = p.out_act; ea.res_ld = p.res_ld; ea.raw_ld = p.raw_ld; ea.act_ld = p.act_ld;
    ea.cout = p.Cout; ea.act = p.act; ea.npix = (unsigned)p.M;
    const cp::EpiRsrc er = cp::epi_make(ea, p.W);
    int cos[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) cos[j] = n0 + (wn * TN + j) * 32 + lrow;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = m0 + (wm * TM + i) * 32 + hi4;
        float unused_keep[16][TN];
        cp::epilogue_block<TN, (TM * TN == 1) ? 4 : ((TN == 1) ? 8 : 4)>(acc[i], cos, ea, er, [&](int r) { const int m = mb + (r & 3) + 8 * (r >> 2); return m < p.M ? m : -1; }, nullptr, unused_keep);
    }
}

template <int WGM, int WGN, int TM, int TN, int MODE>
int launch(const ConvK& k, hipStream_t st) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    ConvK kk = k;
    kk.tiles_m = (k.M + BM - 1) / BM;
    kk.tiles_n = (k.Cout + BN - 1) / BN;
    size_t lds = (size_t)NBUF * (BM + BN) * LDS_STRIDE * sizeof(float) + MAX_TAPS * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f32_kernel<WGM, WGN, TM, TN, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid(kk.tiles_m * kk.tiles_n);
    CP_LAUNCH((conv_f32_kernel<WGM, WGN, TM, TN, MODE>), grid, dim3(512), lds, st, kk);
    return cp::check_launch("cp_conv2d_fwd_f32");
}

template <int MODE>
int launch_tile(int tile, const ConvK& k, hipStream_t st) {
    switch (tile) {
        case CP_TILE_128x128: return launch<2, 2, 2, 2, MODE>(k, st);
        case CP_TILE_64x128: return launch<2, 2, 1, 2, MODE>(k, st);
        case CP_TILE_128x64: return launch<2, 2, 2, 1, MODE>(k, st);
        case CP_TILE_64x64: return launch<2, 2, 1, 1, MODE>(k, st);
        case CP_TILE_128x32: return launch<4, 1, 1, 1, MODE>(k, st);
        case CP_TILE_256x32:
            if constexpr ((MODE & F_BILINEAR) != 0) return launch<4, 1, 1, 1, MODE>(k, st);  // 8 rows x 4 taps would not fit the VGPR budget
            else return launch<4, 1, 2, 1, MODE>(k, st);
        default: cp::set_error("cp_conv2d_fwd_f32: unknown tile_hint %d", tile); return CP_ERR_INVALID;
    }
}

int chunks_for(int taps, int C) { return (C == 4) ? (taps + 7) / 8 : taps * (C / 32); }

// Tile choice from the measured sweep on MI355X (profiles/): 64x64 blocks win or tie
// for every layer with cout >= 64 (3+ blocks per CU hide the staging of each other); the
// 32-channel full-resolution layers and the 9/27-channel heads want all four waves along M.
int pick_tile(long long M, int N) {
    (void)M;
    return N <= 32 ? CP_TILE_128x32 : CP_TILE_64x64;
}

}  // namespace

extern "C" int cp_conv_ktot(int kh, int kw, int num_sources, const int* channels) {
    int chunks = 0;
    for (int s = 0; s < num_sources; ++s) chunks += chunks_for(kh * kw, channels[s]);
    return chunks * BK;
}

extern "C" int cp_conv_pack_weights_host(const float* w, int layout, int kh, int kw, int cout, int num_sources,
                                         const int* channels, const int* real_channels, float* dst) {
    CP_REQUIRE(w && dst && channels && real_channels, "cp_conv_pack_weights_host: null pointer");
    CP_REQUIRE(num_sources == 1 || num_sources == 2, "cp_conv_pack_weights_host: num_sources must be 1 or 2");
    const int taps = kh * kw;
    int cin = 0;
    for (int s = 0; s < num_sources; ++s) {
        CP_REQUIRE(channels[s] == 4 || (channels[s] > 0 && channels[s] % 32 == 0),
                   "cp_conv_pack_weights_host: source channels must be 4 or a multiple of 32 (got %d)", channels[s]);
        CP_REQUIRE(real_channels[s] > 0 && real_channels[s] <= channels[s], "cp_conv_pack_weights_host: bad real_channels");
        cin += real_channels[s];
    }
    const int ktot = cp_conv_ktot(kh, kw, num_sources, channels);
    for (size_t i = 0; i < (size_t)cout * ktot; ++i) dst[i] = 0.f;
    int kbase = 0, cbase = 0;
    for (int s = 0; s < num_sources; ++s) {
        const int C = channels[s], Cr = real_channels[s];
        for (int t = 0; t < taps; ++t) {
            const int ky = t / kw, kx = t % kw;
            for (int c = 0; c < Cr; ++c) {
                const int k = kbase + t * C + c;  // both modes: tap-major, channel-minor
                const int ci = cbase + c;
                for (int co = 0; co < cout; ++co) {
                    size_t src = (layout == 0) ? ((((size_t)ky * kw + kx) * cin + ci) * cout + co)
                                               : ((((size_t)ci * kh + ky) * kw + kx) * cout + co);
                    dst[(size_t)co * ktot + k] = w[src];
                }
            }
        }
        kbase += chunks_for(taps, C) * BK;
        cbase += Cr;
    }
    return CP_OK;
}

extern "C" int cp_conv_selected_tile(const cp_conv_desc* d) {
    CP_REQUIRE_DESC(d, "cp_conv_selected_tile");
    if (d->tile_hint) return d->tile_hint;
    if (cp::stem_applicable(d)) return CP_TILE_STEM;
    if (cp::halo_applicable(d)) return CP_TILE_HALO;
    return pick_tile((long long)d->batch * d->out_h * d->out_w, d->cout);
}

extern "C" int cp_conv_pack_head_weights_host(const float* w, int head_cout, float* dst) {
    CP_REQUIRE(w && dst && head_cout >= 1 && head_cout <= 32, "cp_conv_pack_head_weights_host: bad arguments");
    // fragment-major like the halo weights: [k8 (4)][half (2)][q (32)][4] = Wh[c = k8*8 + half*4 + e][q]
    for (int i = 0; i < 1024; ++i) dst[i] = 0.f;
    for (int k8 = 0; k8 < 4; ++k8)
        for (int half = 0; half < 2; ++half)
            for (int q = 0; q < head_cout; ++q)
                for (int e = 0; e < 4; ++e) dst[((k8 * 2 + half) * 32 + q) * 4 + e] = w[(k8 * 8 + half * 4 + e) * head_cout + q];
    return CP_OK;
}

extern "C" int cp_conv_halo_weight_floats(int cout, int num_sources, const int* channels) {
    if (!channels || num_sources < 1 || num_sources > 2 || cout < 1 || cout > 64) return CP_ERR_INVALID;
    return cp::halo_weight_floats(cout, num_sources, channels);
}

extern "C" int cp_conv_pack_weights_halo_host(const float* w, int layout, int cout, int num_sources, const int* channels,
                                              const int* real_channels, float* dst) {
    CP_REQUIRE(w && dst && channels && real_channels, "cp_conv_pack_weights_halo_host: null pointer");
    CP_REQUIRE(cout >= 1 && cout <= 64 && (num_sources == 1 || num_sources == 2), "cp_conv_pack_weights_halo_host: cout must be <= 64, 1-2 sources");
    for (int s = 0; s < num_sources; ++s)
        CP_REQUIRE((channels[s] == 4 && s == num_sources - 1 && s > 0) || (channels[s] > 0 && channels[s] % 32 == 0),
                   "cp_conv_pack_weights_halo_host: source channels must be multiples of 32 (a 4-channel source only as the last of two)");
    return cp::halo_pack_weights(w, layout, cout, num_sources, channels, real_channels, dst);
}

extern "C" int cp_conv2d_fwd_f32(const cp_conv_desc* d, void* stream) {
    CP_REQUIRE_DESC(d, "cp_conv2d_fwd_f32");
    CP_REQUIRE(d->num_sources == 1 || d->num_sources == 2, "cp_conv2d_fwd_f32: num_sources must be 1 or 2");
    CP_REQUIRE(d->kh * d->kw <= MAX_TAPS && d->kh > 0 && d->kw > 0, "cp_conv2d_fwd_f32: unsupported kernel %dx%d", d->kh, d->kw);
    CP_REQUIRE(d->stride >= 1 && d->dilation >= 1 && d->pad >= 0, "cp_conv2d_fwd_f32: bad stride/dilation/pad");
    CP_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cout > 0, "cp_conv2d_fwd_f32: empty tensor");
    const int eh = (d->kh - 1) * d->dilation + 1, ew = (d->kw - 1) * d->dilation + 1;
    CP_REQUIRE(d->out_h == (d->in_h + 2 * d->pad - eh) / d->stride + 1 && d->out_w == (d->in_w + 2 * d->pad - ew) / d->stride + 1,
               "cp_conv2d_fwd_f32: out size %dx%d inconsistent with input %dx%d k%d s%d d%d p%d", d->out_h, d->out_w,
               d->in_h, d->in_w, d->kh, d->stride, d->dilation, d->pad);
    CP_REQUIRE(d->weights, "cp_conv2d_fwd_f32: null weights");
    CP_REQUIRE(d->out_raw || d->out_act || d->head_out, "cp_conv2d_fwd_f32: no output requested");
    if (d->head_out) {
        CP_REQUIRE(d->head_weights && d->head_cout >= 1 && d->head_cout <= 32 && d->head_out_ld >= d->head_cout && d->cout == 32,
                   "cp_conv2d_fwd_f32: fused head needs head_weights, 1 <= head_cout <= 32 <= ... and cout == 32");
        CP_REQUIRE(cp::halo_applicable(d) && (d->tile_hint == 0 || d->tile_hint == CP_TILE_HALO), "cp_conv2d_fwd_f32: the fused head is implemented by the halo-tile kernel only (3x3/s1/p1, weights_halo)");
        CP_REQUIRE(!d->head_label_out || (d->head_label_classes >= 1 && d->head_label_classes <= d->head_cout), "cp_conv2d_fwd_f32: head_label_classes must be in [1, head_cout]");
    } else {
        CP_REQUIRE(!d->head_label_out, "cp_conv2d_fwd_f32: head_label_out needs a fused head");
    }
    CP_REQUIRE(!d->tap_label || d->stride == 1, "cp_conv2d_fwd_f32: tap_label needs stride 1");
    CP_REQUIRE((d->scale == nullptr) == (d->shift == nullptr), "cp_conv2d_fwd_f32: scale and shift come together");
    CP_REQUIRE(!d->epi_label || d->scale, "cp_conv2d_fwd_f32: epi_label needs a scale/shift table");
    CP_REQUIRE((long long)d->batch * d->out_h * d->out_w < (1LL << 31), "cp_conv2d_fwd_f32: too many output pixels");

    ConvK k{};
    int chans[2] = {0, 0};
    for (int s = 0; s < d->num_sources; ++s) {
        const cp_conv_source& in = d->src[s];
        CP_REQUIRE(in.data, "cp_conv2d_fwd_f32: source %d has null data", s);
        CP_REQUIRE(in.channels == 4 || (in.channels > 0 && in.channels % 32 == 0),
                   "cp_conv2d_fwd_f32: source %d channels must be 4 or a multiple of 32 (got %d)", s, in.channels);
        CP_REQUIRE(in.ld >= in.channels && in.ld % 4 == 0, "cp_conv2d_fwd_f32: source %d ld must be >= channels and a multiple of 4", s);
        CP_REQUIRE(((uintptr_t)in.data & 15) == 0, "cp_conv2d_fwd_f32: source %d not 16-byte aligned", s);
        CP_REQUIRE(in.mode >= 0 && in.mode <= 3, "cp_conv2d_fwd_f32: source %d bad mode", s);
        CP_REQUIRE(in.mode != CP_SRC_ZERO_INSERT_X2 || (s == 0 && !in.pre_scale && !d->tap_label), "cp_conv2d_fwd_f32: zero-insertion applies to source 0 only, without pre-affine / tap mask");
        CP_REQUIRE(in.mode != CP_SRC_NEAREST_SEL || in.sel, "cp_conv2d_fwd_f32: source %d needs a sel map", s);
        CP_REQUIRE(in.mode == CP_SRC_DIRECT || (d->in_h % 2 == 0 && d->in_w % 2 == 0),
                   "cp_conv2d_fwd_f32: x2 source modes need even in_h/in_w");
        CP_REQUIRE((in.pre_scale == nullptr) == (in.pre_shift == nullptr), "cp_conv2d_fwd_f32: pre_scale/pre_shift come together");
        SrcK& o = k.s[s];
        o.data = in.data;
        o.sel = in.sel;
        o.pre_scale = in.pre_scale;
        o.pre_shift = in.pre_shift;
        o.C = in.channels;
        o.ld = in.ld;
        o.mode = in.mode;
        o.Hs = (in.mode == CP_SRC_DIRECT) ? d->in_h : d->in_h / 2;
        o.Ws = (in.mode == CP_SRC_DIRECT) ? d->in_w : d->in_w / 2;
        {
            const long long px = (long long)d->batch * o.Hs * o.Ws;
            const long long nbytes = px * in.ld * 4;
            CP_REQUIRE(nbytes < (1LL << 31), "cp_conv2d_fwd_f32: source %d spans %lld bytes; 32-bit range-checked addressing needs < 2 GiB", s, nbytes);
            o.bytes = (unsigned)nbytes;
        }
        o.c4 = in.channels == 4;
        o.cpt = o.c4 ? 1 : in.channels / 32;
        o.nchunks = chunks_for(d->kh * d->kw, in.channels);
        chans[s] = in.channels;
    }
    k.W = d->weights;
    k.ktot = cp_conv_ktot(d->kh, d->kw, d->num_sources, chans);
    k.nchunks = k.ktot / BK;
    k.w_bytes = (unsigned)((size_t)d->cout * k.ktot * 4);
    k.group_rows = d->group_rows;
    k.group_wstride = d->group_weight_stride;
    if (d->group_rows) {
        const long long rows = (long long)d->batch * d->out_h * d->out_w;
        CP_REQUIRE(d->group_rows > 0 && d->group_rows % 128 == 0 && rows % d->group_rows == 0 && d->group_weight_stride >= d->cout * k.ktot,
                   "cp_conv2d_fwd_f32: group_rows must be a multiple of 128 dividing the output pixels; group_weight_stride >= cout*ktot");
        CP_REQUIRE(d->tile_hint != CP_TILE_256x32, "cp_conv2d_fwd_f32: the 256-row tile does not support grouped weights");
        const long long wb = (rows / d->group_rows) * (long long)d->group_weight_stride * 4;
        CP_REQUIRE(wb < (1LL << 31), "cp_conv2d_fwd_f32: grouped weights span >= 2 GiB");
        k.w_bytes = (unsigned)wb;
    }
    k.lab_bytes = (unsigned)((size_t)d->batch * d->in_h * d->in_w);
    k.B = d->batch; k.Hin = d->in_h; k.Win = d->in_w; k.Ho = d->out_h; k.Wo = d->out_w; k.Cout = d->cout;
    k.KH = d->kh; k.KW = d->kw; k.stride = d->stride; k.dil = d->dilation; k.pad = d->pad;
    k.M = d->batch * d->out_h * d->out_w;
    k.tap_label = d->tap_label; k.row_scale = d->row_scale;
    k.residual = d->residual; k.res_ld = d->residual_ld;
    k.scale = d->scale; k.shift = d->shift; k.epi_label = d->epi_label; k.act = d->act;
    k.out_raw = d->out_raw; k.raw_ld = d->out_raw_ld; k.out_act = d->out_act; k.act_ld = d->out_act_ld;
    CP_REQUIRE(!d->out_raw || d->out_raw_ld >= d->cout, "cp_conv2d_fwd_f32: out_raw_ld < cout");
    CP_REQUIRE(!d->out_act || d->out_act_ld >= d->cout, "cp_conv2d_fwd_f32: out_act_ld < cout");
    CP_REQUIRE(!d->residual || d->residual_ld >= d->cout, "cp_conv2d_fwd_f32: residual_ld < cout");
    {
        const long long npx = (long long)d->batch * d->out_h * d->out_w;
        const long long ldmax = std::max(std::max(d->out_raw ? d->out_raw_ld : 0, d->out_act ? d->out_act_ld : 0), d->residual ? d->residual_ld : 0);
        CP_REQUIRE(npx * ldmax * 4 < (1LL << 31), "cp_conv2d_fwd_f32: output/residual tensor spans >= 2 GiB; split the batch (32-bit range-checked addressing)");
    }

    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->tile_hint == CP_TILE_STEM || (d->tile_hint == 0 && cp::stem_applicable(d))) {
        CP_REQUIRE(cp::stem_applicable(d), "cp_conv2d_fwd_f32: CP_TILE_STEM requested but the layer is not the 7x7/s2/p3 4->64 stem with stem-packed weights");
        return cp::launch_stem_conv(d, st);
    }
    if (d->tile_hint == CP_TILE_HALO || (d->tile_hint == 0 && cp::halo_applicable(d))) {
        CP_REQUIRE(cp::halo_applicable(d), "cp_conv2d_fwd_f32: CP_TILE_HALO requested but the layer does not qualify (3x3/s1/p1, cout<=64, weights_halo)");
        return cp::launch_halo_conv(d, st);
    }
    int tile = d->tile_hint ? d->tile_hint : pick_tile(k.M, k.Cout);
    // operand-loader variant
    const bool any_pre = k.s[0].pre_scale || (d->num_sources > 1 && k.s[1].pre_scale);
    const bool bil = k.s[0].mode == CP_SRC_BILINEAR_X2, sel = k.s[0].mode == CP_SRC_NEAREST_SEL;
    CP_REQUIRE(d->num_sources == 1 || d->src[1].mode == CP_SRC_DIRECT, "cp_conv2d_fwd_f32: only source 0 may use an x2 mode");
    const int mode = (any_pre ? F_PRE : 0) | (bil ? F_BILINEAR : 0) | (k.tap_label ? F_PARTIAL : 0) | (sel ? F_SEL : 0);
    switch (mode) {
        case 0: return launch_tile<0>(tile, k, st);
        case F_PRE: return launch_tile<F_PRE>(tile, k, st);
        case F_BILINEAR: return launch_tile<F_BILINEAR>(tile, k, st);
        case F_PARTIAL: return launch_tile<F_PARTIAL>(tile, k, st);
        case F_SEL: return launch_tile<F_SEL>(tile, k, st);
        case F_PARTIAL | F_SEL: return launch_tile<F_PARTIAL | F_SEL>(tile, k, st);
        default:
            cp::set_error("cp_conv2d_fwd_f32: unsupported combination of operand modes (pre=%d bilinear=%d partial=%d sel=%d)",
                          (int)any_pre, (int)bil, (int)(k.tap_label != nullptr), (int)sel);
            return CP_ERR_INVALID;
    }
}
