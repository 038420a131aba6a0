// Grouped GEMM of the Winograd path on the bf16 matrix pipe with fp32-equivalent accuracy (OPT-IN):
//     M[p][t][n] = sum_k V[p][t][k] * U[p][n][k]
// Every fp32 operand is split EXACTLY into three bf16 terms (8 + 8 + 8 significand bits):
//     hi = trunc_bf16(x),  mid = trunc_bf16(x - hi),  lo = x - hi - mid
// and the six products  hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi  are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Each
// product of two bf16 numbers is exact in fp32; the three dropped products (mid*lo, lo*mid, lo*lo) are <= 2^-24 of the full product,
// i.e. the rounding an fp32 multiply makes anyway.  Six bf16 MFMAs of K = 16 cost 6 x 32 cycles against 8 x 64 for the fp32 MFMA.
//
// Sizing (why this is not wino_gemm.hip with another instruction): at this MFMA rate a 32x32 per-wave tile would need ~3x the LDS
// bandwidth of a CU for its fragment reads, so a block is 128 x 128 with four consumer waves of 64 x 64.  The activations (A) stay fp32 in
// HBM and are split by the four producer waves while they stage them (three LDS stages of 30 KB: chunk c + 2 is stored while chunk c is multiplied, so the first
// fragments of chunk c + 1 are read before the barrier); the weights (B) are pre-split once into
// fragment-major bf16 planes (split_weights_kernel) and every consumer wave fetches its 12 fragments of a chunk straight from L2 one
// chunk ahead (issued unconditionally: a branch around the loads makes the compiler's wait counts conservative).  One persistent block per
// CU.  Measured 178-195 TFLOP/s-equivalent against 115-120 for the fp32 MFMA kernel = ~1.1 PFLOP/s of bf16 MFMA work.  Variants measured within 3 %: all six planes staged in LDS (182); fp32 in LDS, split in the consumers (149).
#include "common.h"
#include "split_f16.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;          // k per chunk
constexpr int BM = 128, BN = 128;
constexpr int ROWB = 80;        // bytes per staged row of one split: 32 bf16 (64 B) + 16 B pad -> conflict-free 16-byte fragment reads
constexpr int SPLIT_BYTES = BM * ROWB;          // one split plane of a 128-row tile
constexpr int TILE_BYTES = 3 * SPLIT_BYTES;     // hi, mid, lo
constexpr int STAGE_BYTES = TILE_BYTES;         // only A is staged; B fragments come pre-split straight from L2
constexpr int NSTAGE = 3;

// exact three-way split of four floats into packed bf16 pairs: hi = top 16 bits of x, mid = top 16 bits of x - hi, lo = top 16 bits of
// x - hi - mid (that last remainder has at most 8 significant bits, so taking its top half is exact).  v_perm_b32 packs the high halves.
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& mid, uint2& lo) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = __builtin_bit_cast(unsigned, x[e]);
        const float r1 = x[e] - __builtin_bit_cast(float, h[e] & 0xffff0000u);
        m[e] = __builtin_bit_cast(unsigned, r1);
        const float r2 = r1 - __builtin_bit_cast(float, m[e] & 0xffff0000u);
        l[e] = __builtin_bit_cast(unsigned, r2);
    }
    // perm(src0, src1, sel): byte k of the result = byte sel[k] of {src0 (bytes 4-7), src1 (bytes 0-3)}; 0x07060302 = [hi16(src1), hi16(src0)]
    hi = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
    mid = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
    lo = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
}

struct SplitK {
    const float* A;
    const unsigned char* B;   // pre-split weights: [group][n / 32][k / 16][hi, mid, lo][64 lanes][8 bf16] (cp_wino_split_weights_f32)
    float* C;
    int rows, N, K, group_rows, nchunks, tiles_m, tiles_n;
    int nb32;                 // 32-column blocks per group (N rounded up to 128, / 32)
    unsigned a_bytes, b_bytes;
    float c_scale;            // F16: 1 / (the power of two the weights were multiplied by before their split)
    uint32_t* mon;            // F16: f16x2 range monitor slot (common.h) or null: max |A| over every row the producers convert
};

// pre-split one weight row segment: thread = (group, 32-column block, k16 step, lane).  F16: planes 0 / 1 = hi / lo of the fp16 two-way split of
// U * scale (split_f16.h), plane 2 unused (the three-plane stride is kept so that both layouts address alike)
template <bool F16>
__global__ void split_weights_kernel(const float* __restrict__ U, int groups, int n, int k, int nb32, float scale, unsigned char* __restrict__ out) {
    const long long total = (long long)groups * nb32 * (k / 16) * 64;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        long long t = i >> 6;
        const int ks = (int)(t % (k / 16));
        t /= (k / 16);
        const int jb = (int)(t % nb32);
        const int g = (int)(t / nb32);
        const int col = jb * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (col < n) {
            const float* src = U + ((size_t)g * n + col) * k + k0;
            v0 = *reinterpret_cast<const float4*>(src);
            v1 = *reinterpret_cast<const float4*>(src + 4);
        }
        unsigned char* dst = out + ((((size_t)g * nb32 + jb) * (k / 16) + ks) * 3) * 1024 + lane * 16;
        if constexpr (F16) {
            uint2 h0, l0, h1, l1;
            cp::split4h(make_float4(v0.x * scale, v0.y * scale, v0.z * scale, v0.w * scale), h0, l0);
            cp::split4h(make_float4(v1.x * scale, v1.y * scale, v1.z * scale, v1.w * scale), h1, l1);
            *reinterpret_cast<uint4*>(dst) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
        } else {
            uint2 h0, m0, l0, h1, m1, l1;
            split4(v0, h0, m0, l0);
            split4(v1, h1, m1, l1);
            *reinterpret_cast<uint4*>(dst) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(m0.x, m0.y, m1.x, m1.y);
            *reinterpret_cast<uint4*>(dst + 2048) = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
    }
}

// NPL = 3: the exact three-way split (six products, fp32-equivalent).  NPL = 2: hi + mid planes only (16 significand bits per operand, products
// hi*hi, hi*mid, mid*hi): half the MFMAs, for the bf16 conv modes whose gates are 3e-2 -- NOT fp32-equivalent.  NPL = 2 with F16: the fp16
// two-way split of split_f16.h (operands reproduced to within one fp32 ulp, products hi*hi, hi*lo, lo*hi on v_mfma_f32_32x32x16_f16): fp32-level accuracy
// with half the MFMAs of the exact bf16 split; the weights come pre-multiplied by a power of two, the accumulators are multiplied by c_scale.
template <int NPL, bool F16, bool MON = false>
__global__ __launch_bounds__(512, 1) void wino_gemm_split_kernel(const SplitK p) {
    static_assert(!F16 || NPL == 2, "the fp16 split has two planes");
    static_assert(!MON || F16, "the range monitor goes with the fp16 split");   // (a compile-time variant: no branch in the producers' counted load / store phases)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NSTAGE stages][3 splits][128 rows][80 B]

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;

    // tile sequence: XCD x owns a contiguous run of the (m-major, n-minor) tile list (same scheme as wino_gemm.hip)
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nx = 8;
    const int xcd = blockIdx.x % nx, bidx = blockIdx.x / nx, nb = gridDim.x / nx;
    const int q_ = ntiles / nx, r_ = ntiles % nx;
    const int start = (xcd < r_) ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_;
    const int cnt = q_ + (xcd < r_ ? 1 : 0);
    const int my_items = (cnt > bidx) ? (cnt - bidx + nb - 1) / nb : 0;
    const int total_chunks = my_items * p.nchunks;
    if (total_chunks == 0) return;

#define CP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    if (producer) {
        if constexpr (F16) cp::f16_overflow_clamps();
        // 32 rows x 8 float4 per pass, four passes per operand.  A 16-lane group of a ds_write_b64 covers two rows: with consecutive rows (stride 80 B
        // = 20 banks) their 16-dword spans overlap on four banks (SQ_LDS_BANK_CONFLICT = one extra cycle per store, profiles/r04_pmc_gemm_isolated.txt);
        // rows r and r + 4 (80 dwords apart = 16 banks) do not: the eight row slots of a wave take rows 0, 4, 1, 5, 2, 6, 3, 7
        const int col4 = tid & 7;
#ifdef WS_ROWS_CONSECUTIVE
        const int rbase = tid >> 3;
#else
        const int rslot = tid >> 3, rbase = (rslot & ~7) | (((rslot & 7) >> 1) + 4 * (rslot & 1));
#endif
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
        // Two register sets: chunk c travels in set c & 1, is requested two phases before it is stored (round 4, with the fp16 split: a phase is 24
        // MFMAs per wave, and with ONE chunk in flight the operand latency was exposed -- the kernel without its A loads ran 25 % faster).  The loop
        // is unrolled by two phases and every phase issues its four loads UNCONDITIONALLY (out of range past the end: they fetch nothing), so the
        // compiler can count: the wait in front of a store leaves exactly the other set's four loads in flight.
        float4 areg[2][4];
        unsigned aoff[4];
        int it = -1, q = p.nchunks;
        auto advance = [&]() {
            if (++q >= p.nchunks) {
                q = 0;
                ++it;
                const int tile = start + bidx + it * nb;
                const int tm = tile / p.tiles_n;
                const int m0 = tm * BM;
#pragma unroll
                for (int i = 0; i < 4; ++i) aoff[i] = ((unsigned)(m0 + rbase + 32 * i) * (unsigned)p.K + col4 * 4) * 4u;
            }
        };
        auto issue = [&](auto setc, bool valid) __attribute__((always_inline)) {
            constexpr int S = decltype(setc)::value;
            if (valid) advance();
            const unsigned oob = valid ? 0u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < 4; ++i) areg[S][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(aoff[i] | oob), q * (BK * 4), 0));
        };
        float l_amax = 0.f;
        auto store = [&](auto setc, int buf) __attribute__((always_inline)) {
            constexpr int S = decltype(setc)::value;
            unsigned char* a = smem + buf * STAGE_BYTES + rbase * ROWB + col4 * 8;
            if constexpr (MON) {
#pragma unroll
                for (int i = 0; i < 4; ++i) l_amax = cp::amax4(l_amax, areg[S][i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint2 h, m, l;
                if constexpr (F16) cp::split4h(areg[S][i], h, m);
                else split4(areg[S][i], h, m, l);
                *reinterpret_cast<uint2*>(a + 32 * i * ROWB) = h;
                *reinterpret_cast<uint2*>(a + 32 * i * ROWB + SPLIT_BYTES) = m;
                if constexpr (NPL == 3) *reinterpret_cast<uint2*>(a + 32 * i * ROWB + 2 * SPLIT_BYTES) = l;
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        // three LDS stages: chunk c + 2 is stored while the consumers multiply chunk c, so chunk c + 1 is complete one barrier early and its
        // first fragments can be read during chunk c instead of behind the barrier
        issue(S0{}, true);                      // chunk 0
        issue(S1{}, total_chunks > 1);          // chunk 1
        store(S0{}, 0);
        issue(S0{}, total_chunks > 2);          // chunk 2
        if (total_chunks > 1) store(S1{}, 1);
        issue(S1{}, total_chunks > 3);          // chunk 3
        CP_BARRIER();
        int st = 2;   // stage of chunk c + 2
        for (int c = 0; c < total_chunks; c += 2) {
            if (c + 2 < total_chunks) store(S0{}, st);     // chunk c + 2 (requested two phases ago)
            issue(S0{}, c + 4 < total_chunks);             // chunk c + 4
            st = (st == 2) ? 0 : st + 1;
            CP_BARRIER();
            if (c + 1 >= total_chunks) break;
            if (c + 3 < total_chunks) store(S1{}, st);     // chunk c + 3
            issue(S1{}, c + 5 < total_chunks);             // chunk c + 5
            st = (st == 2) ? 0 : st + 1;
            CP_BARRIER();
        }
        if constexpr (MON) {
            cp::monitor_flush(p.mon, l_amax);
            cp::monitor_count_launch(p.mon, tid == 0);
        }
        return;
    }

    // ---------------------------------- consumers: 2 x 2 waves of 64 x 64 ------------------------------------
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, kh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][2][NPL];      // [slot][row block][split]: A fragments from LDS
    bf16x8 fb[2][2][2][NPL];   // [chunk parity][k16 step][column block][split]: B fragments of a whole chunk, fetched one chunk ahead from L2
    const __amdgpu_buffer_rsrc_t rbw = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
    const unsigned ks_total = (unsigned)(p.K / 16);
    auto read_a = [&](int buf, int ks, int slot) {
        const unsigned char* a = smem + buf * STAGE_BYTES + (wm * 64 + lrow) * ROWB + ks * 32 + kh * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < NPL; ++s) fa[slot][i][s] = *reinterpret_cast<const bf16x8*>(a + i * 32 * ROWB + s * SPLIT_BYTES);
    };
    // B fragments of flattened chunk `cc` (tile cc / nchunks of this block, chunk cc % nchunks) into register set `par`
    int f_it = -1, f_q = p.nchunks;
    unsigned f_base[2] = {0u, 0u};
    // `valid` = false issues the same twelve loads out of bounds (they fetch nothing): with a branch around the loads the compiler's
    // wait-count pass must assume they may not have been issued and then waits for the NEWEST outstanding loads before the first MFMA of
    // every chunk -- an L2 round trip per 48 MFMAs
    auto fetch_b = [&](int par, bool valid) {
        const unsigned oob = valid ? 0u : 0x80000000u;
        if (++f_q >= p.nchunks) {
            f_q = 0;
            ++f_it;
            const int tile = start + bidx + f_it * nb;
            const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
            const unsigned g = (unsigned)((tm * BM) / p.group_rows);
#pragma unroll
            for (int j = 0; j < 2; ++j) f_base[j] = ((g * (unsigned)p.nb32 + (unsigned)(tn * 4 + wn * 2 + j)) * ks_total * 3u) * 1024u + (unsigned)lane * 16u;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < NPL; ++s)
                    fb[par][ks][j][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                        rbw, (int)((f_base[j] + ((unsigned)(f_q * 2 + ks) * 3u + (unsigned)s) * 1024u) | oob), 0, 0));
    };
    auto mfma_step = [&](int slot, int par, int ks) {
        // smallest terms first; consecutive MFMAs hit different accumulators
        constexpr int NPR = (NPL == 3) ? 6 : 3;
#pragma unroll
        for (int t0 = 0; t0 < NPR; ++t0) {
            const int t = (NPL == 3) ? t0 : t0 + 3;   // two planes: the last three products (mid*hi, hi*mid, hi*hi)
            const int sa = (t == 0) ? 2 : (t == 1) ? 0 : (t == 2) ? 1 : (t == 3) ? 1 : 0;   // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
            const int sb = (t == 0) ? 0 : (t == 1) ? 2 : (t == 2) ? 1 : (t == 3) ? 0 : (t == 4) ? 1 : 0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if constexpr (F16)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cp::f16x8_t, fa[slot][i][sa]), __builtin_bit_cast(cp::f16x8_t, fb[par][ks][j][sb]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot][i][sa], fb[par][ks][j][sb], acc[i][j], 0, 0, 0);
        }
    };
    int it = 0, q = 0;
    int buf = 0;   // LDS stage of the current chunk (c % 3)
    auto chunk = [&](int c, int par) {
        const int nbuf = (buf == 2) ? 0 : buf + 1;
        fetch_b(par ^ 1, c + 1 < total_chunks);   // next chunk's weights: in flight during this chunk's 48 MFMAs
        read_a(buf, 1, 1);
        mfma_step(0, par, 0);
        if (c + 1 < total_chunks) read_a(nbuf, 0, 0);   // the next chunk's first fragments: its stage has been complete since the last barrier
        mfma_step(1, par, 1);
        // the six reads above go out behind the first MFMAs of the second step, not in front of the barrier
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NPL, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * ((NPL == 3) ? 6 : 3) - 4, 0);
        CP_BARRIER();
        buf = nbuf;
        if (++q == p.nchunks) {
            const int tile = start + bidx + it * nb;
            const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = tn * BN + wn * 64 + j * 32 + lrow;
                    float* dst = p.C + (size_t)(tm * BM + wm * 64 + i * 32 + kh * 4) * p.N + col;
                    if (col < p.N) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2)) * p.N] = F16 ? acc[i][j][r] * p.c_scale : acc[i][j][r];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
            q = 0;
            ++it;
        }
    };
    fetch_b(0, true);
    CP_BARRIER();  // stage 0 ready
    read_a(0, 0, 0);
    int c = 0;
    for (; c + 1 < total_chunks; c += 2) {
        chunk(c, 0);
        chunk(c + 1, 1);
    }
    if (c < total_chunks) chunk(c, 0);
#undef CP_BARRIER
}

}  // namespace

namespace cp {
bool wino_gemm_wide_applicable(int rows, int group_rows, int k, int n);   // wino_gemm_wide.hip
int wino_gemm_wide_launch(const float* V, const void* Usplit, float* M, int rows, int group_rows, int k, int n, float c_scale, unsigned a_bytes, unsigned b_bytes,
                          hipStream_t stream);
}

extern "C" size_t cp_wino_split_weights_bytes(int groups, int n, int k) {
    if (groups <= 0 || n <= 0 || k <= 0 || k % 16) return 0;
    const size_t nb32 = (size_t)((n + 127) / 128) * 4;
    return (size_t)groups * nb32 * (k / 16) * 3 * 1024;
}

extern "C" float cp_f16x2_weight_scale(float max_abs) {
    // the power of two that brings max |w| into [2^11, 2^12): low parts of all but vanishing weights are normal fp16 numbers, 16x headroom to 65504
    if (!(max_abs > 0.f) || !std::isfinite(max_abs)) return 1.f;
    int e = 0;
    (void)std::frexp(max_abs, &e);   // max_abs = m * 2^e, m in [0.5, 1)
    return std::ldexp(1.f, std::min(std::max(12 - e, -100), 100));
}

extern "C" int cp_wino_split_weights_scaled_f32(const float* U, int groups, int n, int k, int planes, float scale, void* out, void* stream) {
    CP_REQUIRE(U && out && groups > 0 && n > 0 && k > 0 && k % 32 == 0, "cp_wino_split_weights_f32: bad arguments (K must be a multiple of 32)");
    CP_REQUIRE(((uintptr_t)U & 15) == 0 && ((uintptr_t)out & 15) == 0, "cp_wino_split_weights_f32: pointers must be 16-byte aligned");
    CP_REQUIRE(planes == 3 || planes == 2 || planes == CP_PLANES_F16X2, "cp_wino_split_weights_scaled_f32: planes must be 3, 2 or CP_PLANES_F16X2");
    CP_REQUIRE(planes == CP_PLANES_F16X2 ? (scale > 0.f && std::isfinite(scale)) : scale == 1.f, "cp_wino_split_weights_scaled_f32: a scale other than 1 goes with CP_PLANES_F16X2 only");
    const int nb32 = ((n + 127) / 128) * 4;
    const long long total = (long long)groups * nb32 * (k / 16) * 64;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    if (planes == CP_PLANES_F16X2)
        CP_LAUNCH(split_weights_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, U, groups, n, k, nb32, scale, reinterpret_cast<unsigned char*>(out));
    else
        CP_LAUNCH(split_weights_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, U, groups, n, k, nb32, 1.f, reinterpret_cast<unsigned char*>(out));
    return cp::check_launch("cp_wino_split_weights_f32");
}

extern "C" int cp_wino_split_weights_f32(const float* U, int groups, int n, int k, void* out, void* stream) {
    return cp_wino_split_weights_scaled_f32(U, groups, n, k, 3, 1.f, out, stream);
}

extern "C" int cp_wino_gemm_split_scaled_f32(const float* V, const void* Usplit, float* M, int rows, int group_rows, int k, int n, int planes, float c_scale, void* stream) {
    CP_REQUIRE(planes == 3 || planes == 2 || planes == CP_PLANES_F16X2, "cp_wino_gemm_split_planes_f32: planes must be 3 (exact split), 2 (hi + mid) or CP_PLANES_F16X2");
    CP_REQUIRE(planes == CP_PLANES_F16X2 ? (c_scale > 0.f && std::isfinite(c_scale)) : c_scale == 1.f, "cp_wino_gemm_split_scaled_f32: a scale other than 1 goes with CP_PLANES_F16X2 only");
    CP_REQUIRE(V && Usplit && M, "cp_wino_gemm_split_f32: null pointer");
    CP_REQUIRE(rows > 0 && group_rows > 0 && rows % group_rows == 0 && group_rows % 128 == 0, "cp_wino_gemm_split_f32: rows must be whole groups of a multiple of 128 rows");
    CP_REQUIRE(k > 0 && k % 32 == 0 && n > 0, "cp_wino_gemm_split_f32: K must be a multiple of 32");
    const long long ab = (long long)rows * k * 4, cb = (long long)rows * n * 4;
    const long long bb = (long long)cp_wino_split_weights_bytes(rows / group_rows, n, k);
    CP_REQUIRE(ab < (1LL << 31) && bb < (1LL << 31) && cb < (1LL << 33), "cp_wino_gemm_split_f32: operand spans >= 2 GiB");
    CP_REQUIRE(((uintptr_t)V & 15) == 0 && ((uintptr_t)Usplit & 15) == 0, "cp_wino_gemm_split_f32: operands must be 16-byte aligned");
    SplitK g{};
    g.A = V; g.B = reinterpret_cast<const unsigned char*>(Usplit); g.C = M;
    g.rows = rows; g.N = n; g.K = k; g.group_rows = group_rows; g.nchunks = k / BK;
    g.tiles_m = rows / BM; g.tiles_n = (n + BN - 1) / BN;
    g.nb32 = g.tiles_n * 4;
    g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;
    g.c_scale = c_scale;
    g.mon = planes == CP_PLANES_F16X2 ? cp::f16x2_monitor() : nullptr;
    if (planes == CP_PLANES_F16X2 && cp::wino_gemm_wide_applicable(rows, group_rows, k, n))   // 128 x 256 tiles, both operands through LDS (wino_gemm_wide.hip)
        return cp::wino_gemm_wide_launch(V, Usplit, M, rows, group_rows, k, n, c_scale, g.a_bytes, g.b_bytes, (hipStream_t)stream);
    const size_t lds = (size_t)NSTAGE * STAGE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_gemm_split_kernel<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_gemm_split_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_gemm_split_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_gemm_split_kernel<2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (planes == 3) CP_LAUNCH((wino_gemm_split_kernel<3, false>), dim3(cp::persistent_blocks()), dim3(512), lds, (hipStream_t)stream, g);
    else if (planes == 2) CP_LAUNCH((wino_gemm_split_kernel<2, false>), dim3(cp::persistent_blocks()), dim3(512), lds, (hipStream_t)stream, g);
    else if (g.mon) CP_LAUNCH((wino_gemm_split_kernel<2, true, true>), dim3(cp::persistent_blocks()), dim3(512), lds, (hipStream_t)stream, g);
    else CP_LAUNCH((wino_gemm_split_kernel<2, true>), dim3(cp::persistent_blocks()), dim3(512), lds, (hipStream_t)stream, g);
    return cp::check_launch("cp_wino_gemm_split_f32");
}

extern "C" int cp_wino_gemm_split_planes_f32(const float* V, const void* Usplit, float* M, int rows, int group_rows, int k, int n, int planes, void* stream) {
    CP_REQUIRE(planes == 3 || planes == 2, "cp_wino_gemm_split_planes_f32: planes must be 3 (exact split) or 2 (hi + mid)");
    return cp_wino_gemm_split_scaled_f32(V, Usplit, M, rows, group_rows, k, n, planes, 1.f, stream);
}

extern "C" int cp_wino_gemm_split_f32(const float* V, const void* Usplit, float* M, int rows, int group_rows, int k, int n, void* stream) {
    return cp_wino_gemm_split_scaled_f32(V, Usplit, M, rows, group_rows, k, n, 3, 1.f, stream);
}
