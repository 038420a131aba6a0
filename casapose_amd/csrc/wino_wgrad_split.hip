// Weight gradient of the Winograd layers on the bf16 matrix pipe (gfx950): the grouped "TN" GEMM
//
//     dU[g][n][k] = sum over tiles t of  dM[g][t][n] * V[g][t][k]            g = 36 Winograd planes, n = cout, k = cin
//
// (dM = A dY A^T from cp_wino_dy_transform_f32, V = the forward's transformed input, kept per layer) that cp_conv2d_wgrad_f32 computes in its
// grouped mode with fp32 MFMAs (wgrad_gemm128_kernel, 6.3 ms of a 62 ms training step).  Here every fp32 operand is split EXACTLY into three bf16
// terms and six bf16 x bf16 products (each exact in fp32) are accumulated in fp32 -- the arithmetic of wino_gemm_split.hip / conv_wgrad_split.hip,
// fp32-equivalent -- or, with planes = 1, operands are rounded to bf16 (CASAPOSE_CONV_MODE=bf16).  The reference obtains this product from
// tf.GradientTape (train_casapose.py:594-611 -> Conv2DBackpropFilter of the layers.Conv2D call sites of resnet.py:97-103, casapose.py:71-74).
// Round 6, planes = CP_PLANES_F16X2: both operands as fp16 pairs (split_f16.h: hi = rn_f16(x s), lo = rn_f16(x s - hi), three exact products per
// fp32 product instead of six), each multiplied by a power of two s the CALLER picks so that its maximum sits inside fp16's band (a gradient has
// no natural magnitude: cp_wino_dy_transform_f32 reports max |dM| into the armed monitor slot); the accumulators take 1 / (s_a s_b) on the way out.
//
// The reduction runs over the ROWS of both operands, so the MFMA fragments (8 consecutive t of one column per lane) are transposes of the
// [t][channel] layout in HBM: ds_read_b64_tr_b16 from an LDS image kept in the native layout delivers them without shuffles (conv_wgrad_split.hip).
// A block owns a 128 (n) x 128 (k) tile of one plane: 4 loader waves fetch 32-row slabs of both operands two or three steps ahead into
// registers, split them and store the planes ([row][32 channels] = 64-byte rows, conflict-free for the transpose reads); 4 consumer waves hold
// 64 x 64 of the tile each (four 32x32 accumulators) and issue 24 MFMAs per (16 rows, 6 products).  One barrier per slab, two LDS slots.
// Work = (plane, chunk of the row range, tile) items of up to `lc` slabs, dealt ROUND-ROBIN to the persistent blocks: at any moment the 256 blocks
// hold 256 consecutive items, i.e. the 16 tiles of a (plane, chunk) run side by side on consecutive blocks -- which the XCD-aware block order
// puts on one L2 -- and walk the same rows in step, so every operand strip (read by 4 tiles) comes from HBM once.  (A first version gave each
// block a contiguous share of the (tile, slab) stream: concurrent blocks then sat in different tiles, nothing was shared and the kernel ran at
// the 3.7 GB of a 4x re-read: 125 TF/s-equivalent, 177 with bf16 operands.)  An item ends with fp32 atomics into the zeroed dU.
#include "common.h"
#include "split_f16.h"

#include <algorithm>
#include <cmath>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int TS = 32;              // rows (tiles of the Winograd grid) per slab
constexpr int BLK = 4;              // 32-channel blocks per operand and block tile: 128 columns
constexpr int ROWB = TS * 64;       // bytes of one (32-channel block, plane) of a slab: [32 rows][32 bf16]
constexpr int NSLOT = 2;

struct TnK {
    const float* a;   // dM [G][T][N]
    const float* b;   // V  [G][T][K]
    float* c;         // dU [G][N][K]
    unsigned a_bytes, b_bytes;
    int G, T, N, K;
    int tiles_n, tiles_k, steps;   // steps = slabs of the whole row range
    int lc, chunks;                // slabs per chunk, chunks per plane
    int items;                     // G * chunks * tiles_n * tiles_k
    float sa, sb, descale;         // f16x2: powers of two on the operands, 1 / (sa sb) on the accumulators
};

#define TN_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ unsigned pack_hi16(unsigned a_lo, unsigned b_hi) { return __builtin_amdgcn_perm(b_hi, a_lo, 0x07060302u); }

// exact three-way split of 8 floats into packed bf16 planes: hi = top 16 bits, mid = top 16 bits of x - hi, lo = top 16 bits of the rest
__device__ __forceinline__ void split8(const float4 v0, const float4 v1, uint4& hi, uint4& mid, uint4& lo) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h[e] = __builtin_bit_cast(unsigned, x[e]);
        const float r1 = x[e] - __builtin_bit_cast(float, h[e] & 0xffff0000u);
        m[e] = __builtin_bit_cast(unsigned, r1);
        const float r2 = r1 - __builtin_bit_cast(float, m[e] & 0xffff0000u);
        l[e] = __builtin_bit_cast(unsigned, r2);
    }
    hi = make_uint4(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]), pack_hi16(h[4], h[5]), pack_hi16(h[6], h[7]));
    mid = make_uint4(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]), pack_hi16(m[4], m[5]), pack_hi16(m[6], m[7]));
    lo = make_uint4(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]), pack_hi16(l[4], l[5]), pack_hi16(l[6], l[7]));
}

__device__ __forceinline__ uint4 round8(const float4 v0, const float4 v1) {   // round to nearest even
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned u = __builtin_bit_cast(unsigned, x[e]);
        r[e] = u + 0x7fffu + ((u >> 16) & 1u);
    }
    return make_uint4(pack_hi16(r[0], r[1]), pack_hi16(r[2], r[3]), pack_hi16(r[4], r[5]), pack_hi16(r[6], r[7]));
}

template <int NP, bool F16>
__device__ __forceinline__ void store_planes(unsigned char* dst, int plane_stride, const float4 v0, const float4 v1, float s) {
    if constexpr (F16) {
        uint2 h0, l0, h1, l1;
        cp::split4h(make_float4(v0.x * s, v0.y * s, v0.z * s, v0.w * s), h0, l0);
        cp::split4h(make_float4(v1.x * s, v1.y * s, v1.z * s, v1.w * s), h1, l1);
        *reinterpret_cast<uint4*>(dst) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4*>(dst + plane_stride) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    } else if constexpr (NP == 3) {
        uint4 h, m, l;
        split8(v0, v1, h, m, l);
        *reinterpret_cast<uint4*>(dst) = h;
        *reinterpret_cast<uint4*>(dst + plane_stride) = m;
        *reinterpret_cast<uint4*>(dst + 2 * plane_stride) = l;
    } else {
        *reinterpret_cast<uint4*>(dst) = round8(v0, v1);
    }
}

// 8 consecutive rows (the MFMA's k) of this lane's channel: two transpose reads of 4 rows each
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* a) {
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * 64));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

struct Item {
    int g, tn, tk, s0, len;   // plane, tile, first slab and number of slabs
};

__device__ __forceinline__ Item decode(const TnK& p, int q) {
    Item r;
    r.tk = q % p.tiles_k;
    q /= p.tiles_k;
    r.tn = q % p.tiles_n;
    q /= p.tiles_n;
    const int c = q % p.chunks;
    r.g = q / p.chunks;
    r.s0 = c * p.lc;
    r.len = min(p.lc, p.steps - r.s0);
    return r;
}

template <int NP, bool F16>
__global__ __launch_bounds__(512, 1) void wino_wgrad_split_kernel(const TnK p) {
    static_assert(!F16 || NP == 2, "the fp16 two-way split has two planes");
    constexpr unsigned OOB = 0x80000000u;
    constexpr int PLANE = BLK * ROWB, SLOT = NP * PLANE;   // one operand of one slab: [plane][block][row][32 ch]
    constexpr int D = 3;                                   // register sets of the loaders = slabs in flight
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Bs = smem + NSLOT * SLOT;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int bid = cp::xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int G = (int)gridDim.x;
    if (bid >= p.items) return;
    int NT = 0;   // slabs of this block's items bid, bid + G, ...
    for (int q = bid; q < p.items; q += G) NT += decode(p, q).len;
    const int NTP = (NT + D - 1) / D * D;

    if (wave >= 4) {
        // ------------------------------------------------ loaders ---------------------------------------------------------------
        if constexpr (F16) cp::f16_overflow_clamps();
        const int L = (wave - 4) * 64 + lane;   // 0..255
        const int oct = L & 3;                  // 8 channels of a 32-channel block
        const int blk = (L >> 2) & 3;           // 16 consecutive lanes cover 512 contiguous bytes of one row (4 blocks x 32 channels)
        const int row0 = L >> 4;                // 0..15; second item: + 16
        const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, p.b_bytes, 0x00020000);
        float4 ar[D][2][2], br[D][2][2];
        int iq = bid, is = 0;   // issue cursor: item, slab within it
        Item un = decode(p, bid);
        // the same loads every call, in one basic block (out-of-range work gets out-of-bounds offsets, OR-ed in so that no branch surrounds a
        // load): the wait-count pass then knows how many younger loads are in flight when a set is consumed (conv_wgrad_split.hip)
        auto issue = [&](int d) {
            const bool live = iq < p.items;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int t = (un.s0 + is) * TS + row0 + 16 * i;
                const bool ok = live && t < p.T;
                const long long rowi = (long long)un.g * p.T + t;
                const unsigned offa = (unsigned)((rowi * p.N + un.tn * 128 + blk * 32 + oct * 8) * 4) | (ok ? 0u : OOB);
                const unsigned offb = (unsigned)((rowi * p.K + un.tk * 128 + blk * 32 + oct * 8) * 4) | (ok ? 0u : OOB);
                ar[d][i][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsa, (int)offa, 0, 0));
                ar[d][i][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsa, (int)(offa + 16u), 0, 0));
                br[d][i][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsb, (int)offb, 0, 0));
                br[d][i][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsb, (int)(offb + 16u), 0, 0));
            }
            if (live && ++is >= un.len) {
                iq += G;
                is = 0;
                if (iq < p.items) un = decode(p, iq);
            }
        };
        auto write = [&](int d, int T) {
            unsigned char* ab = As + (T & 1) * SLOT + blk * ROWB + oct * 16;
            unsigned char* bb = Bs + (T & 1) * SLOT + blk * ROWB + oct * 16;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                store_planes<NP, F16>(ab + (row0 + 16 * i) * 64, PLANE, ar[d][i][0], ar[d][i][1], p.sa);
                store_planes<NP, F16>(bb + (row0 + 16 * i) * 64, PLANE, br[d][i][0], br[d][i][1], p.sb);
            }
        };
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d);
        for (int T = 0; T < NTP; T += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                write(d, T + d);
                issue(d);
                TN_BARRIER();
            }
        }
        return;
    }

    // ---------------------------------------------------- consumers -------------------------------------------------------------
    const int wn = wave & 1, wk = wave >> 1;   // this wave's 64 x 64 quarter of the tile: n blocks 2wn, 2wn+1; k blocks 2wk, 2wk+1
    const int kg = lane >> 5, half = (lane >> 4) & 1, li = lane & 15;
    const int lane_off = (8 * kg + (li >> 2)) * 64 + (16 * half + 4 * (li & 3)) * 2;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto flush = [&](const Item& un) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n0 = un.tn * 128 + (2 * wn + i) * 32 + 4 * (lane >> 5);
                const int k = un.tk * 128 + (2 * wk + j) * 32 + (lane & 31);
                float* dst = p.c + ((size_t)un.g * p.N) * p.K + k;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + (r & 3) + 8 * (r >> 2);
                    atomicAdd(dst + (size_t)n * p.K, F16 ? acc[i][j][r] * p.descale : acc[i][j][r]);
                    acc[i][j][r] = 0.f;
                }
            }
    };

    int T = 0;
    for (int q = bid; q < p.items; q += G) {
      const Item un = decode(p, q);
      for (int s = 0; s < un.len; ++s, ++T) {
        TN_BARRIER();
        const unsigned char* abase = As + (T & 1) * SLOT + (2 * wn) * ROWB + lane_off;
        const unsigned char* bbase = Bs + (T & 1) * SLOT + (2 * wk) * ROWB + lane_off;
        bf16x8 a[2][2][NP], b[2][2][NP];   // [sub-step parity][block][plane]
        auto load = [&](int j) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    a[j & 1][i][pl] = frag_tr(abase + pl * PLANE + i * ROWB + j * 1024);
                    b[j & 1][i][pl] = frag_tr(bbase + pl * PLANE + i * ROWB + j * 1024);
                }
        };
        load(0);
#pragma unroll
        for (int j = 0; j < TS / 16; ++j) {
            if (j + 1 < TS / 16) load(j + 1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x16& c = acc[i][q];
                    bf16x8(&aa)[NP] = a[j & 1][i];
                    bf16x8(&bb)[NP] = b[j & 1][q];
                    if constexpr (F16) {   // lo * hi, hi * lo, hi * hi
                        const cp::f16x8_t a0 = __builtin_bit_cast(cp::f16x8_t, aa[0]), a1 = __builtin_bit_cast(cp::f16x8_t, aa[1]);
                        const cp::f16x8_t b0 = __builtin_bit_cast(cp::f16x8_t, bb[0]), b1 = __builtin_bit_cast(cp::f16x8_t, bb[1]);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
                    } else if constexpr (NP == 3) {   // smallest products first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[2], bb[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[0], bb[2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[1], bb[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[1], bb[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[0], bb[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[0], bb[0], c, 0, 0, 0);
                    } else {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[0], bb[0], c, 0, 0, 0);
                    }
                }
        }
      }
      flush(un);
    }
    for (; T < NTP; ++T) TN_BARRIER();
}

template <int NP, bool F16>
int launch(TnK k, hipStream_t st) {
    k.tiles_n = k.N / 128;
    k.tiles_k = k.K / 128;
    k.steps = (k.T + TS - 1) / TS;
    // chunk length: about 28 slabs (896 rows): long enough that the 64 KB of atomics closing an item are ~7 % of the 0.9 MB it loads, short
    // enough that the items of all planes deal out evenly over 256 blocks
    k.chunks = std::max(1, (k.steps + 27) / 28);
    k.lc = (k.steps + k.chunks - 1) / k.chunks;
    k.chunks = (k.steps + k.lc - 1) / k.lc;
    const long long items = (long long)k.G * k.chunks * k.tiles_n * k.tiles_k;
    if (items >= (1LL << 30)) {
        cp::set_error("cp_wino_wgrad_split_f32: too many work items");
        return CP_ERR_INVALID;
    }
    k.items = (int)items;
    const int grid = (int)std::min<long long>(256, items);
    const size_t lds = (size_t)2 * NSLOT * NP * BLK * ROWB;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_wgrad_split_kernel<NP, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    CP_LAUNCH((wino_wgrad_split_kernel<NP, F16>), dim3((unsigned)grid), dim3(512), lds, st, k);
    return cp::check_launch("cp_wino_wgrad_split_f32");
}

}  // namespace

extern "C" int cp_wino_wgrad_split_applicable(int groups, int rows, int n, int k) {
    return groups > 0 && rows > 0 && n > 0 && k > 0 && n % 128 == 0 && k % 128 == 0 && (long long)groups * rows * std::max(n, k) * 4 < (1LL << 31) ? 1 : 0;
}

extern "C" int cp_wino_wgrad_split_f32(const float* dm, const float* v, float* du, int groups, int rows, int n, int k, int planes, void* stream) {
    return cp_wino_wgrad_split_scaled_f32(dm, v, du, groups, rows, n, k, planes, 1.f, 1.f, stream);
}

extern "C" int cp_wino_wgrad_split_scaled_f32(const float* dm, const float* v, float* du, int groups, int rows, int n, int k, int planes, float dm_scale,
                                              float v_scale, void* stream) {
    CP_REQUIRE(dm && v && du, "cp_wino_wgrad_split_f32: null pointer");
    CP_REQUIRE(planes == 1 || planes == 3 || planes == CP_PLANES_F16X2, "cp_wino_wgrad_split_f32: planes must be 3 (exact split), 1 (bf16) or CP_PLANES_F16X2");
    CP_REQUIRE(planes == CP_PLANES_F16X2 ? (dm_scale > 0.f && v_scale > 0.f && std::isfinite(dm_scale) && std::isfinite(v_scale)) : (dm_scale == 1.f && v_scale == 1.f),
               "cp_wino_wgrad_split_scaled_f32: operand factors are positive powers of two with CP_PLANES_F16X2 and 1 otherwise");
    CP_REQUIRE(cp_wino_wgrad_split_applicable(groups, rows, n, k),
               "cp_wino_wgrad_split_f32: n and k must be multiples of 128 and each operand smaller than 2 GiB (groups %d, rows %d, n %d, k %d)", groups, rows, n, k);
    CP_REQUIRE((((uintptr_t)dm) | ((uintptr_t)v) | ((uintptr_t)du)) % 16 == 0, "cp_wino_wgrad_split_f32: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(du, 0, sizeof(float) * (size_t)groups * n * k, st) != hipSuccess) return cp::check_launch("cp_wino_wgrad_split_f32 memset");
    TnK p{};
    p.a = dm;
    p.b = v;
    p.c = du;
    p.G = groups;
    p.T = rows;
    p.N = n;
    p.K = k;
    p.a_bytes = (unsigned)((size_t)groups * rows * n * 4);
    p.b_bytes = (unsigned)((size_t)groups * rows * k * 4);
    p.sa = dm_scale;
    p.sb = v_scale;
    p.descale = 1.f / (dm_scale * v_scale);
    if (planes == CP_PLANES_F16X2) return launch<2, true>(p, st);
    return planes == 3 ? launch<3, false>(p, st) : launch<1, false>(p, st);
}
