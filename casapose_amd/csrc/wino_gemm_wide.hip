// The Winograd planes' grouped GEMM in the fp16 two-way split (split_f16.h) with a 128 x 256 block tile and BOTH operands staged in LDS
// (round 4).  Why a second kernel beside wino_gemm_split.hip: with three products per fp32 product that kernel is bound by operand delivery, not
// by the matrix pipe -- compiled without its A loads it runs 25 % faster, without its B fetches 16 %, without both 43 %
// (round-4 probe, CHANGELOG 8.1).  Per 32-wide chunk a 128 x 128 block pulls 16 KB of V through L2 and its four consumer waves fetch 32 KB of
// weight fragments from L2 (the two waves of a column pair fetch the same 8 KB).  Here
//   * a block owns 128 rows x 256 columns: V is read half as often (N = 256: once; N = 512: twice), a chunk feeds 48 MFMAs per wave instead of 24;
//   * the weight fragments of a chunk (32 KB, pre-split and fragment-major: 1 KB pieces) go global -> LDS by DMA (buffer_load ... lds) from the
//     producer waves, once per block, and the consumers read them with ds_read_b128 -- no consumer VMEM traffic, half the L2 -> CU bytes;
//   * three A stages (61 KB) + three B stages (96 KB) = 159.7 KB of LDS: chunk c + 2 of both operands lands while chunk c is multiplied.
// Only the two-plane fp16 split fits (a third plane would need 220 KB); the exact bf16 split and every N that is not a multiple of 256 stay on
// wino_gemm_split.hip.  Same arithmetic, same results (tests/test_gpu_f16x2.py compares the two kernels bit for bit).
#include "common.h"
#include "split_f16.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

// Hand-counted waits: `s_waitcnt vmcnt(N)` lets the N youngest loads stay in flight, which is only right while the compiler emits at least N loads
// between the LDS-DMA pieces a wait must cover and the wait itself.  tests/test_asm_invariants.py counts them in the generated ISA of every build;
// -DCP_SAFE_WAITS (tools/build_variant.sh) replaces every such wait by vmcnt(0) -- slower, and independent of instruction selection.
#ifdef CP_SAFE_WAITS
#define CP_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define CP_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));   // (container type of a 16-byte fragment; the MFMA below reads it as 8 x fp16)

constexpr int BK = 32;
constexpr int BM = 128, BN = 256;
constexpr int ROWB = 80;                         // bytes per staged A row of one plane: 32 x 2 B + 16 B pad (conflict-free 16-byte fragment reads)
constexpr int SPLIT_BYTES = BM * ROWB;
constexpr int A_STAGE = 2 * SPLIT_BYTES;         // hi, lo
constexpr int NA = 3;
constexpr int B_PIECES = 2 * 8 * 2;              // (k16 step, 32-column block, plane) pieces of 1 KB per chunk
constexpr int B_STAGE = B_PIECES * 1024;
constexpr int NB = 3;
constexpr int LDS_BYTES = NA * A_STAGE + NB * B_STAGE;   // 159744

struct WideK {
    const float* A;
    const unsigned char* B;   // [group][n / 32][k / 16][3 plane slots][64 lanes][8 x 2 B] (cp_wino_split_weights_scaled_f32: slots 0 / 1 = hi / lo)
    float* C;
    int rows, N, K, group_rows, nchunks, tiles_m, tiles_n, nb32;
    unsigned a_bytes, b_bytes;
    float c_scale;
    uint32_t* mon;   // f16x2 range monitor slot (common.h) or null: max |A| over every row the producers convert
};

#define CP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// MON: the producers also fold max |A| into the f16x2 range monitor slot p.mon (a compile-time variant: a run-time branch inside store() would put
// basic-block boundaries between the DMA pieces and the hand-counted waits that cover them, tests/test_asm_invariants.py)
template <bool MON>
__global__ __launch_bounds__(512, 1) void wino_gemm_wide_kernel(const WideK p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ast = smem;                   // [NA][2 planes][128 rows][80 B]
    unsigned char* bst = smem + NA * A_STAGE;    // [NB][32 pieces][1 KB]

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;

    // tile sequence: XCD x owns a contiguous run of the (m-major, n-minor) tile list (as wino_gemm_split.hip)
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nx = 8;
    const int xcd = blockIdx.x % nx, bidx = blockIdx.x / nx, nb = gridDim.x / nx;
    const int q_ = ntiles / nx, r_ = ntiles % nx;
    const int start = (xcd < r_) ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_;
    const int cnt = q_ + (xcd < r_ ? 1 : 0);
    const int my_items = (cnt > bidx) ? (cnt - bidx + nb - 1) / nb : 0;
    const int total_chunks = my_items * p.nchunks;
    if (total_chunks == 0) return;
    const unsigned ks_total = (unsigned)(p.K / 16);

    if (producer) {
        cp::f16_overflow_clamps();
        const int pw = wave - 4;
        const int col4 = tid & 7;
        const int rslot = tid >> 3, rbase = (rslot & ~7) | (((rslot & 7) >> 1) + 4 * (rslot & 1));   // conflict-free ds_write_b64 row order
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
        float4 areg[2][4];   // chunk c in set c & 1, two chunks in flight (compile-time set numbers only)
        unsigned aoff[4];
        int it = -1, q = p.nchunks;
        auto advance = [&]() __attribute__((always_inline)) {
            if (++q >= p.nchunks) {
                q = 0;
                ++it;
                const int tile = start + bidx + it * nb;
                const int m0 = (tile / p.tiles_n) * BM;
#pragma unroll
                for (int i = 0; i < 4; ++i) aoff[i] = ((unsigned)(m0 + rbase + 32 * i) * (unsigned)p.K + col4 * 4) * 4u;
            }
        };
        auto issue = [&](auto setc, bool valid) __attribute__((always_inline)) {   // always four loads (out of range when !valid): countable
            constexpr int S = decltype(setc)::value;
            if (valid) advance();
            const unsigned oob = valid ? 0u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < 4; ++i) areg[S][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(aoff[i] | oob), q * (BK * 4), 0));
        };
        float l_amax = 0.f;
        auto store = [&](auto setc, int buf) __attribute__((always_inline)) {
            constexpr int S = decltype(setc)::value;
            unsigned char* a = ast + buf * A_STAGE + rbase * ROWB + col4 * 8;
            if constexpr (MON) {
#pragma unroll
                for (int i = 0; i < 4; ++i) l_amax = cp::amax4(l_amax, areg[S][i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint2 h, l;
                cp::split4h(areg[S][i], h, l);
                *reinterpret_cast<uint2*>(a + 32 * i * ROWB) = h;
                *reinterpret_cast<uint2*>(a + 32 * i * ROWB + SPLIT_BYTES) = l;
            }
        };
        // weight fragments of the next chunk in the block's chunk list: this wave's 8 of the 32 pieces, global -> LDS stage `buf` by DMA
        int d_it = -1, d_q = p.nchunks;
        unsigned d_base = 0u;
        auto dma_b = [&](int buf) __attribute__((always_inline)) {
            if (++d_q >= p.nchunks) {
                d_q = 0;
                ++d_it;
                const int tile = start + bidx + d_it * nb;
                const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
                const unsigned g = (unsigned)((tm * BM) / p.group_rows);
                d_base = (g * (unsigned)p.nb32 + (unsigned)(tn * 8)) * ks_total * 3u * 1024u;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pi = pw * 8 + i, ks = pi >> 4, jb = (pi >> 1) & 7, s = pi & 1;
                const unsigned src = d_base + (((unsigned)jb * ks_total + (unsigned)(d_q * 2 + ks)) * 3u + (unsigned)s) * 1024u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(bst + buf * B_STAGE + pi * 1024), 16, (int)(lane * 16), (int)src, 0, 0);
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        issue(S0{}, true);                      // chunk 0
        issue(S1{}, total_chunks > 1);          // chunk 1
        dma_b(0);                               // weights of chunk 0
        if (total_chunks > 1) dma_b(1);
        store(S0{}, 0);
        if (total_chunks > 1) store(S1{}, 1);
        issue(S0{}, total_chunks > 2);          // chunk 2
        issue(S1{}, total_chunks > 3);          // chunk 3
        CP_WAIT_VM(8);   // everything but the eight loads just issued: the DMA pieces have landed
        CP_BARRIER();
        int st = 2;   // stage (of both operands) of chunk c + 2
        for (int c = 0; c < total_chunks; c += 2) {
            if (c + 2 < total_chunks) {
                store(S0{}, st);                // chunk c + 2 (requested two phases ago)
                dma_b(st);
            }
            issue(S0{}, c + 4 < total_chunks);  // chunk c + 4
            CP_WAIT_VM(4);   // the DMA pieces (older than the four loads above) have landed
            st = (st == 2) ? 0 : st + 1;
            CP_BARRIER();
            if (c + 1 >= total_chunks) break;
            if (c + 3 < total_chunks) {
                store(S1{}, st);                // chunk c + 3
                dma_b(st);
            }
            issue(S1{}, c + 5 < total_chunks);  // chunk c + 5
            CP_WAIT_VM(4);
            st = (st == 2) ? 0 : st + 1;
            CP_BARRIER();
        }
        if constexpr (MON) {
            cp::monitor_flush(p.mon, l_amax);
            cp::monitor_count_launch(p.mon, tid == 0);
        }
        return;
    }

    // ---------------------------------- consumers: 2 x 2 waves of 64 rows x 128 columns ------------------------------------
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, kh = lane >> 5;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][2][2];   // [slot][row block][plane]
    bf16x8 fb[2][4][2];   // [slot][column block][plane]
    auto read_ab = [&](int buf, int ks, int slot) __attribute__((always_inline)) {
        const unsigned char* a = ast + buf * A_STAGE + (wm * 64 + lrow) * ROWB + ks * 32 + kh * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) fa[slot][i][s] = *reinterpret_cast<const bf16x8*>(a + i * 32 * ROWB + s * SPLIT_BYTES);
        const unsigned char* b = bst + buf * B_STAGE + (unsigned)(((ks * 8 + wn * 4) * 2) * 1024) + (unsigned)lane * 16u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) fb[slot][j][s] = *reinterpret_cast<const bf16x8*>(b + (j * 2 + s) * 1024);
    };
    auto mfma_step = [&](int slot) __attribute__((always_inline)) {
        // lo*hi, hi*lo, hi*hi: smallest terms first; consecutive MFMAs hit different accumulators
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int sa = (t == 0) ? 1 : 0, sb = (t == 1) ? 1 : 0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cp::f16x8_t, fa[slot][i][sa]), __builtin_bit_cast(cp::f16x8_t, fb[slot][j][sb]),
                                                                       acc[i][j], 0, 0, 0);
        }
    };
    int it = 0, q = 0;
    int buf = 0;   // LDS stage of the current chunk (c % 3)
    CP_BARRIER();  // chunks 0 and 1 are in LDS
    read_ab(0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // (nothing pending at the loop's top on the entry path either)
    for (int c = 0; c < total_chunks; ++c) {
        const int nbuf = (buf == 2) ? 0 : buf + 1;
        // Fences pin the order: twelve fragment reads, then the 24 MFMAs that do NOT depend on them (768 cycles: the LDS latency is hidden), twice per
        // chunk.  Left alone the scheduler sinks every read to just in front of its first use (register pressure: 224 VGPRs) and the second half of
        // the chunk waits for its twelve reads one by one.
        read_ab(buf, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(0);
        __builtin_amdgcn_sched_barrier(0);
        // the next chunk's first fragments: both its stages have been complete since the last barrier.  UNCONDITIONAL (after the last chunk it reads a
        // stale stage and nobody uses it): a branch here splits the loop body into blocks, and the sinking pass then moves the reads of the first
        // group down into the block of their first use, past the fences
        read_ab(nbuf, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(1);
        __builtin_amdgcn_sched_barrier(0);
        // lgkmcnt(0) through the BUILTIN (simm16 0xc07f: vmcnt / expcnt at their maxima): the wait-count pass reads it and knows that nothing is
        // pending at the loop's top -- with the wait inside an asm string it assumed the twelve reads above still in flight there and put an
        // lgkmcnt(0) between the next iteration's reads and the MFMAs that do not need them
        __builtin_amdgcn_s_waitcnt(0xc07f);
        asm volatile("s_barrier" ::: "memory");
        buf = nbuf;
        if (++q == p.nchunks) {
            const int tile = start + bidx + it * nb;
            const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = tn * BN + wn * 128 + j * 32 + lrow;
                    float* dst = p.C + (size_t)(tm * BM + wm * 64 + i * 32 + kh * 4) * p.N + col;
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2)) * p.N] = acc[i][j][r] * p.c_scale;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
            q = 0;
            ++it;
        }
    }
#undef CP_BARRIER
}

}  // namespace

namespace cp {

// 1 when the wide kernel covers the shape (N a multiple of 256; CASAPOSE_GEMM_WIDE=0 switches it off for A/B runs)
bool wino_gemm_wide_applicable(int rows, int group_rows, int k, int n) {
    static const bool enabled = [] { const char* e = std::getenv("CASAPOSE_GEMM_WIDE"); return !(e && e[0] == '0'); }();
    return enabled && n % BN == 0 && k % BK == 0 && rows % group_rows == 0 && group_rows % BM == 0;
}

int wino_gemm_wide_launch(const float* V, const void* Usplit, float* M, int rows, int group_rows, int k, int n, float c_scale, unsigned a_bytes, unsigned b_bytes,
                          hipStream_t stream) {
    WideK g{};
    g.mon = cp::f16x2_monitor();
    g.A = V; g.B = reinterpret_cast<const unsigned char*>(Usplit); g.C = M;
    g.rows = rows; g.N = n; g.K = k; g.group_rows = group_rows; g.nchunks = k / BK;
    g.tiles_m = rows / BM; g.tiles_n = n / BN;
    g.nb32 = ((n + 127) / 128) * 4;
    g.a_bytes = a_bytes; g.b_bytes = b_bytes;
    g.c_scale = c_scale;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_gemm_wide_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_gemm_wide_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    if (g.mon) CP_LAUNCH(wino_gemm_wide_kernel<true>, dim3(cp::persistent_blocks()), dim3(512), LDS_BYTES, stream, g);
    else CP_LAUNCH(wino_gemm_wide_kernel<false>, dim3(cp::persistent_blocks()), dim3(512), LDS_BYTES, stream, g);
    return cp::check_launch("cp_wino_gemm_split_f32 (wide)");
}

}  // namespace cp
