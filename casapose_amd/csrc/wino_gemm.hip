// Grouped GEMM of the Winograd path:  M[p][t][n] = sum_k V[p][t][k] * U[p][n][k]   (36 planes, K = input channels).
//
// Same MFMA core as conv_f32.hip (exact fp32 v_mfma_f32_32x32x2_f32, [row][k] LDS tiles with 36-float rows, one
// ds_read_b128 per 8 k, four consumer + four producer waves), but written for what this GEMM is: K is only 256-512, so
// a 64x64 tile lives for 8-16 chunks and the per-tile start-up (first operand rows arrive from HBM, address set-up,
// epilogue) is a large fraction of its life.  Therefore
//   * blocks are PERSISTENT: 1024 blocks (4 per CU) walk the tile list, and the operand stream is flattened across tiles -- while
//     the consumers finish tile i (last chunks + stores) the producers already fetch the first chunks of tile i+1;
//   * the tile order keeps the blocks that run concurrently on one XCD on the same few row-panels of V (8 n-tiles of
//     a row-tile side by side), so V is fetched from memory once per XCD and re-read from that XCD's L2;
//   * no per-row coordinate arithmetic at all: a row is a contiguous K-vector.
#include "common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LS = 36;   // LDS row stride (floats)
constexpr int BM = 64, BN = 64;

struct GemmK {
    const float* A;
    const float* B;
    float* C;
    int rows, N, K, group_rows, nchunks, tiles_m, tiles_n;
    unsigned a_bytes, b_bytes;
    unsigned b_group_stride_bytes;
};

__global__ __launch_bounds__(512, 4) void wino_gemm_kernel(const GemmK p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                    // [2][BM][LS]
    float* Bs = smem + 2 * BM * LS;      // [2][BN][LS]
    constexpr unsigned OOB = 0x80000000u;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;

    // ---- this block's tile sequence: XCD x owns a contiguous run of the (m-major, n-minor) tile list; the blocks of
    // that XCD walk it in lock-step strides so that at any time they cover adjacent n-tiles of the same row-tiles
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
        const int col4 = tid & 7, rbase = tid >> 3;  // 32 rows x 8 float4 per pass, two passes per operand
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
        float4 areg[2], breg[2];
        unsigned aoff[2], boff[2];
        int it = -1, q = p.nchunks;  // chunk cursor of the NEXT issue
        auto advance = [&]() {
            if (++q >= p.nchunks) {
                q = 0;
                ++it;
                const int tile = start + bidx + it * nb;
                const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
                const int m0 = tm * BM, n0 = tn * BN;
                const unsigned gofs = (unsigned)(m0 / p.group_rows) * p.b_group_stride_bytes;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = rbase + 32 * i;
                    aoff[i] = ((unsigned)(m0 + r) * (unsigned)p.K + col4 * 4) * 4u;
                    boff[i] = (n0 + r < p.N) ? gofs + ((unsigned)(n0 + r) * (unsigned)p.K + col4 * 4) * 4u : OOB;
                }
            }
        };
        auto issue = [&]() {
            advance();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                areg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)aoff[i], q * (BK * 4), 0));
                breg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)boff[i], q * (BK * 4), 0));
            }
        };
        auto store = [&](int buf) {
            float* a = As + buf * BM * LS + rbase * LS + col4 * 4;
            float* b = Bs + buf * BN * LS + rbase * LS + col4 * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<float4*>(a + 32 * i * LS) = areg[i];
                *reinterpret_cast<float4*>(b + 32 * i * LS) = breg[i];
            }
        };
        issue();
        store(0);
        if (total_chunks > 1) issue();
        CP_BARRIER();
        for (int c = 0; c < total_chunks; ++c) {
            if (c + 1 < total_chunks) {
                store((c + 1) & 1);
                if (c + 2 < total_chunks) issue();
            }
            CP_BARRIER();
        }
        return;
    }

    // ---------------------------------- consumers ---------------------------------------------
    const int wm = (wave & 3) >> 1, wn = (wave & 3) & 1;
    const int lrow = lane & 31, khalf = (lane >> 5) * 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 fa[2], fb[2];
    auto read_frags = [&](int buf, int k8, int slot) {
        fa[slot] = *reinterpret_cast<const float4*>(As + buf * BM * LS + (wm * 32 + lrow) * LS + khalf + k8 * 8);
        fb[slot] = *reinterpret_cast<const float4*>(Bs + buf * BN * LS + (wn * 32 + lrow) * LS + khalf + k8 * 8);
    };
    auto mfma4 = [&](int slot) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot].x, fb[slot].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot].y, fb[slot].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot].z, fb[slot].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot].w, fb[slot].w, acc, 0, 0, 0);
    };
    CP_BARRIER();  // stage 0 ready
    read_frags(0, 0, 0);
    int it = 0, q = 0;
    for (int c = 0; c < total_chunks; ++c) {
        const int buf = c & 1;
        read_frags(buf, 1, 1);
        mfma4(0);
        read_frags(buf, 2, 0);
        mfma4(1);
        read_frags(buf, 3, 1);
        mfma4(0);
        mfma4(1);
        CP_BARRIER();
        if (c + 1 < total_chunks) read_frags(buf ^ 1, 0, 0);
        if (++q == p.nchunks) {  // tile finished: store and restart the accumulator (the producers are already a tile ahead)
            const int tile = start + bidx + it * nb;
            const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
            const int col = tn * BN + wn * 32 + lrow;
            float* dst = p.C + (size_t)(tm * BM + wm * 32 + (lane >> 5) * 4) * p.N + col;
            if (col < p.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2)) * p.N] = acc[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            q = 0;
            ++it;
        }
    }
#undef CP_BARRIER
}

}  // namespace

extern "C" int cp_wino_gemm_f32(const float* V, const float* U, float* M, int rows, int group_rows, int k, int n, void* stream) {
    CP_REQUIRE(V && U && M, "cp_wino_gemm_f32: null pointer");
    CP_REQUIRE(rows > 0 && group_rows > 0 && rows % group_rows == 0 && group_rows % 64 == 0, "cp_wino_gemm_f32: rows must be whole groups of a multiple of 64 rows");
    CP_REQUIRE(k > 0 && k % 32 == 0 && n > 0, "cp_wino_gemm_f32: K must be a multiple of 32");
    const long long ab = (long long)rows * k * 4, bb = (long long)(rows / group_rows) * n * k * 4, cb = (long long)rows * n * 4;
    CP_REQUIRE(ab < (1LL << 31) && bb < (1LL << 31) && cb < (1LL << 33), "cp_wino_gemm_f32: operand spans >= 2 GiB");
    CP_REQUIRE(((uintptr_t)V & 15) == 0 && ((uintptr_t)U & 15) == 0, "cp_wino_gemm_f32: operands must be 16-byte aligned");
    GemmK g{};
    g.A = V; g.B = U; g.C = M;
    g.rows = rows; g.N = n; g.K = k; g.group_rows = group_rows; g.nchunks = k / BK;
    g.tiles_m = rows / BM; g.tiles_n = (n + BN - 1) / BN;
    g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;
    g.b_group_stride_bytes = (unsigned)((long long)n * k * 4);
    const size_t lds = (size_t)2 * (BM + BN) * LS * sizeof(float);
    static int grid = 0;
    if (!grid) { const char* e = getenv("CP_WINO_GRID"); grid = e ? atoi(e) : 1024; }
    CP_LAUNCH(wino_gemm_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, g);
    return cp::check_launch("cp_wino_gemm_f32");
}
