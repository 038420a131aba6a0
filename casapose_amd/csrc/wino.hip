// Winograd F(4x4, 3x3) for the deep 3x3 / stride-1 convolutions (Lavin & Gray's minimal filtering): the 144
// multiplications of a 4x4 output tile become 36, i.e. the MFMA work of the layer drops 4x, at the price of two
// streaming transform passes.  Same convolution, different summation order -- measured error vs fp64 is ~8e-6 of the
// output range in fp32 (direct: ~2e-6), far inside the parity gate.  The reference reaches the same layers through
// cuDNN, which makes this choice for fp32 3x3 convolutions on its own (layers.Conv2D, resnet.py:97-103; casapose.py:71-74).
//
//   V[p][t][c] = (B^T d B)[p]        input transform : 6x6 input patch d of tile t, channel c      (this file)
//   M[p][t][o] = sum_c V[p][t][c] * U[p][o][c]        36 independent GEMMs = ONE grouped 1x1 launch  (conv_f32.hip)
//   Y          = A^T M A, epilogue   output transform : residual / affine table / activation / stores (this file)
//
// Dilation d (2 and 4 in stages 3/4) is handled exactly by sub-grid decomposition: the pixels with equal (y mod d,
// x mod d) form d*d independent dilation-1 problems with zero padding at their own borders.
// Tile id: t = ((n*d*d + sy*d + sx)*Tu + tu)*Tv + tv; planes are padded to Tp tiles (a multiple of 64) so that a GEMM
// block never straddles two planes.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int THREADS = 256;

struct WinoGeom {
    int B, H, W, d, Tu, Tv, T, Tp;
};

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }

// B^T applied to six values
__device__ __forceinline__ void bt6(const float4 (&d)[6], float4 (&t)[6]) {
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = d[3] + d[4] - 4.f * (d[1] + d[2]);
    t[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    t[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
    t[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

// A^T applied to six values
__device__ __forceinline__ void at6(const float4 (&m)[6], float4 (&y)[4]) {
    const float4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}

__device__ __forceinline__ void tile_coords(int t, const WinoGeom& g, int& n, int& sy, int& sx, int& tu, int& tv) {
    tv = t % g.Tv;
    int r = t / g.Tv;
    tu = r % g.Tu;
    r /= g.Tu;
    const int s = r % (g.d * g.d);
    n = r / (g.d * g.d);
    sy = s / g.d;
    sx = s - sy * g.d;
}

// pre_scale / pre_shift (per channel, may be null) + pre_act: the producer's normalisation + activation applied to every REAL pixel while it
// is loaded -- y = act(fma(x, scale, shift)), the expression of affine_act_kernel (train_kernels.hip), so that the training step need not
// store y for a layer whose only consumer is this transform; the zero padding stays zero.
__device__ __forceinline__ float pre_act_f(float t, int act) {
    if (act == CP_ACT_RELU) return fmaxf(t, 0.f);
    if (act == CP_ACT_LEAKY01) return fmaxf(t, 0.f) - fmaxf(-0.1f * t, 0.f);
    return t;
}

// PRE: 0 = plain, 4 = a per-channel factor only, 1 + act = normalise + activation `act` (CP_ACT_NONE / RELU / LEAKY01) -- compile-time: with a run-time activation switch the
// 144 elements of a patch cost two compares and selects each and the transform turned from HBM-bound (102 us) into VALU-bound (173 us)
template <int PRE>
__global__ __launch_bounds__(THREADS) void wino_in_kernel(const float* __restrict__ src, int ld, int C, WinoGeom g, float* __restrict__ V, int ldv,
                                                          int c_off, const float* __restrict__ pre_scale, const float* __restrict__ pre_shift, int pre_act,
                                                          uint32_t* mon) {
    const int c4n = C >> 2;
    const long long total = (long long)g.T * c4n;
    float amax = 0.f;   // f16x2 range monitor (common.h): max |V| over what this launch writes
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int t = (int)(i / c4n);
        int n, sy, sx, tu, tv;
        tile_coords(t, g, n, sy, sx, tu, tv);
        float4 ps = f4(1.f), pb = f4(0.f);
        if constexpr (PRE) {
            ps = *reinterpret_cast<const float4*>(pre_scale + c4 * 4);
            if constexpr (PRE != 4) pb = *reinterpret_cast<const float4*>(pre_shift + c4 * 4);
        }
        float4 tt[6][6];  // (B^T d): column by column
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int x = sx + g.d * (4 * tv + j - 1);
            float4 col[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int y = sy + g.d * (4 * tu + r - 1);
                const bool ok = (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;  // outside the image (or before the sub-grid's first row/column): zero padding
                col[r] = ok ? *reinterpret_cast<const float4*>(src + (((size_t)n * g.H + y) * g.W + x) * ld + c4 * 4) : f4(0.f);
                if constexpr (PRE == 4) {   // a per-channel factor only (no shift, no activation): zero padding stays zero, so no condition
                    const float4 v = col[r];
                    col[r] = make_float4(v.x * ps.x, v.y * ps.y, v.z * ps.z, v.w * ps.w);
                } else if (PRE && ok) {
                    constexpr int ACT = PRE - 1;
                    const float4 v = col[r];
                    col[r] = make_float4(pre_act_f(__builtin_fmaf(v.x, ps.x, pb.x), ACT), pre_act_f(__builtin_fmaf(v.y, ps.y, pb.y), ACT),
                                         pre_act_f(__builtin_fmaf(v.z, ps.z, pb.z), ACT), pre_act_f(__builtin_fmaf(v.w, ps.w, pb.w), ACT));
                }
            }
            float4 o[6];
            bt6(col, o);
#pragma unroll
            for (int r = 0; r < 6; ++r) tt[r][j] = o[r];
        }
        float* dst = V + (size_t)t * ldv + c_off + c4 * 4;
        const size_t plane = (size_t)g.Tp * ldv;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float4 o[6];
            bt6(tt[r], o);
            if (mon) {   // uniform
#pragma unroll
                for (int j = 0; j < 6; ++j) amax = cp::amax4(amax, o[j]);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<float4*>(dst + (size_t)(r * 6 + j) * plane) = o[j];
        }
    }
    if (mon) {
        cp::monitor_flush(mon, amax);
        cp::monitor_count_launch(mon, threadIdx.x == 0);
    }
}

struct WinoEpi {
    const float* residual;
    int res_ld;
    const float* scale;
    const float* shift;
    const uint8_t* label;
    int act;
    float* out_raw;
    int raw_ld;
    float* out_act;
    int act_ld;
    double* stats;   // [2][cout] or null: sum and sum of squares of the RAW output (after the residual) over the real pixels -- the batch statistics
                     // of the normalisation layer that follows, so that the training step needs no separate pass over the tensor for them
};

template <bool STATS>
__global__ __launch_bounds__(THREADS) void wino_out_kernel(const float* __restrict__ M, int cout, WinoGeom g, WinoEpi e) {
    extern __shared__ double sstat[];   // [2][cout] when STATS
    const int c4n = cout >> 2;
    const long long total = (long long)g.T * c4n;
    if constexpr (STATS) {
        for (int i = threadIdx.x; i < 2 * cout; i += THREADS) sstat[i] = 0.0;
        __syncthreads();
    }
    int cur_c4 = -1;
    double ss[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    auto flush = [&]() {
        if (cur_c4 < 0) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            atomicAdd(&sstat[cur_c4 * 4 + k], ss[k]);
            atomicAdd(&sstat[cout + cur_c4 * 4 + k], sq[k]);
            ss[k] = sq[k] = 0.0;
        }
    };
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int t = (int)(i / c4n);
        if constexpr (STATS) {
            if (c4 != cur_c4) {
                flush();
                cur_c4 = c4;
            }
        }
        int n, sy, sx, tu, tv;
        tile_coords(t, g, n, sy, sx, tu, tv);
        const float* src = M + (size_t)t * cout + c4 * 4;
        const size_t plane = (size_t)g.Tp * cout;
        float4 tt[4][6];  // A^T m, column by column
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float4 col[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) col[r] = *reinterpret_cast<const float4*>(src + (size_t)(r * 6 + j) * plane);
            float4 o[4];
            at6(col, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) tt[r][j] = o[r];
        }
        float4 sc = f4(1.f), sh = f4(0.f);
        const bool aff = e.scale != nullptr;
        if (aff && !e.label) {
            sc = *reinterpret_cast<const float4*>(e.scale + c4 * 4);
            sh = *reinterpret_cast<const float4*>(e.shift + c4 * 4);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 o[4];
            at6(tt[r], o);
            const int y = sy + g.d * (4 * tu + r);
            if (y >= g.H) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = sx + g.d * (4 * tv + j);
                if (x >= g.W) continue;
                const size_t pix = ((size_t)n * g.H + y) * g.W + x;
                float4 v = o[j];
                if (e.residual) v = v + *reinterpret_cast<const float4*>(e.residual + pix * e.res_ld + c4 * 4);
                if constexpr (STATS) {   // fp64 from the first addition, like bn_stats_kernel: the two routes agree to ~1e-16, so a layer's activation
                                         // branches do not depend on which of them produced its statistics
                    ss[0] += v.x; ss[1] += v.y; ss[2] += v.z; ss[3] += v.w;
                    sq[0] += (double)v.x * v.x; sq[1] += (double)v.y * v.y; sq[2] += (double)v.z * v.z; sq[3] += (double)v.w * v.w;
                }
                if (e.out_raw) *reinterpret_cast<float4*>(e.out_raw + pix * e.raw_ld + c4 * 4) = v;
                if (e.out_act) {
                    if (aff && e.label) {
                        const int l = e.label[pix];
                        sc = *reinterpret_cast<const float4*>(e.scale + (size_t)l * cout + c4 * 4);
                        sh = *reinterpret_cast<const float4*>(e.shift + (size_t)l * cout + c4 * 4);
                    }
                    float4 w = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
                    if (e.act == CP_ACT_RELU) {
                        w = make_float4(fmaxf(w.x, 0.f), fmaxf(w.y, 0.f), fmaxf(w.z, 0.f), fmaxf(w.w, 0.f));
                    } else if (e.act == CP_ACT_LEAKY01) {
                        w = make_float4(fmaxf(w.x, 0.f) - fmaxf(-0.1f * w.x, 0.f), fmaxf(w.y, 0.f) - fmaxf(-0.1f * w.y, 0.f),
                                        fmaxf(w.z, 0.f) - fmaxf(-0.1f * w.z, 0.f), fmaxf(w.w, 0.f) - fmaxf(-0.1f * w.w, 0.f));
                    }
                    *reinterpret_cast<float4*>(e.out_act + pix * e.act_ld + c4 * 4) = w;
                }
            }
        }
    }
    if constexpr (STATS) {
        flush();
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * cout; i += THREADS)
            if (sstat[i] != 0.0) atomicAdd(&e.stats[i], sstat[i]);
    }
}

// Output transform of layer A fused with the input transform of layer B (round 4): where a Winograd convolution's activated output feeds
// nothing but the next Winograd convolution of the SAME geometry (the conv1 -> conv2 / conv2 -> next unit's conv1 chains of stages 3 and 4,
// resnet.py:57-113), the activated map makes no round trip through HBM: a block owns ONE sub-grid of one image (dilation d: the pixels with
// equal (y mod d, x mod d) -- an independent dilation-1 problem, so the 6x6 patches of its tiles never leave it) and CS channels; it computes
// Y = A^T M A + epilogue for its Tu x Tv tiles, keeps act(Y) in LDS with the zero ring of the NEXT convolution's padding, and writes
// V' = B^T d B of layer B from there.  The raw output (a residual for later) and the activated map (when something else reads it too) are
// still stored on request.  Thread = (tile, channel quad).
template <int CS4, int MAXT>
__global__ __launch_bounds__(MAXT) void wino_out_in_kernel(const float* __restrict__ M, int cout, WinoGeom g, WinoEpi e, float* __restrict__ V, int ldv, int c_off,
                                                           uint32_t* mon) {
    extern __shared__ __attribute__((aligned(16))) float4 sub[];   // [(4 Tu + 2)][(4 Tv + 2)][CS4]
    const int RW = 4 * g.Tv + 2, RH = 4 * g.Tu + 2;
    const int slices = (cout / 4 + CS4 - 1) / CS4;
    const int slice = blockIdx.x % slices, sg = blockIdx.x / slices;     // sg = n * d * d + sy * d + sx
    const int s = sg % (g.d * g.d), n = sg / (g.d * g.d);
    const int sy = s / g.d, sx = s - sy * g.d;
    const int c4l = threadIdx.x % CS4, tl = threadIdx.x / CS4;
    const int c4 = slice * CS4 + c4l;
    const bool live = tl < g.Tu * g.Tv && c4 < cout / 4;
    const int tu = tl / g.Tv, tv = tl - tu * g.Tv;
    for (int i = threadIdx.x; i < RH * RW * CS4; i += blockDim.x) sub[i] = f4(0.f);   // the ring and the rows / columns beyond the image stay zero
    __syncthreads();
    const int t = (sg * g.Tu + tu) * g.Tv + tv;
    if (live) {
        const float* src = M + (size_t)t * cout + c4 * 4;
        const size_t plane = (size_t)g.Tp * cout;
        float4 tt[4][6];  // A^T m, column by column
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float4 col[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) col[r] = *reinterpret_cast<const float4*>(src + (size_t)(r * 6 + j) * plane);
            float4 o[4];
            at6(col, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) tt[r][j] = o[r];
        }
        float4 sc = f4(1.f), sh = f4(0.f);
        if (e.scale) {
            sc = *reinterpret_cast<const float4*>(e.scale + c4 * 4);
            sh = *reinterpret_cast<const float4*>(e.shift + c4 * 4);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 o[4];
            at6(tt[r], o);
            const int y = sy + g.d * (4 * tu + r);
            if (y >= g.H) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = sx + g.d * (4 * tv + j);
                if (x >= g.W) continue;
                const size_t pix = ((size_t)n * g.H + y) * g.W + x;
                float4 v = o[j];
                if (e.residual) v = v + *reinterpret_cast<const float4*>(e.residual + pix * e.res_ld + c4 * 4);
                if (e.out_raw) *reinterpret_cast<float4*>(e.out_raw + pix * e.raw_ld + c4 * 4) = v;
                float4 w = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);   // wino_out_kernel's expression
                if (e.act == CP_ACT_RELU) {
                    w = make_float4(fmaxf(w.x, 0.f), fmaxf(w.y, 0.f), fmaxf(w.z, 0.f), fmaxf(w.w, 0.f));
                } else if (e.act == CP_ACT_LEAKY01) {
                    w = make_float4(fmaxf(w.x, 0.f) - fmaxf(-0.1f * w.x, 0.f), fmaxf(w.y, 0.f) - fmaxf(-0.1f * w.y, 0.f),
                                    fmaxf(w.z, 0.f) - fmaxf(-0.1f * w.z, 0.f), fmaxf(w.w, 0.f) - fmaxf(-0.1f * w.w, 0.f));
                }
                if (e.out_act) *reinterpret_cast<float4*>(e.out_act + pix * e.act_ld + c4 * 4) = w;
                sub[((4 * tu + r + 1) * RW + 4 * tv + j + 1) * CS4 + c4l] = w;
            }
        }
    }
    __syncthreads();
    float amax = 0.f;   // f16x2 range monitor (common.h): max |V'| over what this launch writes
    if (live) {
        float4 tt[6][6];  // (B^T d): column by column
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float4 col[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) col[r] = sub[((4 * tu + r) * RW + 4 * tv + j) * CS4 + c4l];
            float4 o[6];
            bt6(col, o);
#pragma unroll
            for (int r = 0; r < 6; ++r) tt[r][j] = o[r];
        }
        float* dst = V + (size_t)t * ldv + c_off + c4 * 4;
        const size_t plane = (size_t)g.Tp * ldv;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float4 o[6];
            bt6(tt[r], o);
            if (mon) {   // uniform
#pragma unroll
                for (int j = 0; j < 6; ++j) amax = cp::amax4(amax, o[j]);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<float4*>(dst + (size_t)(r * 6 + j) * plane) = o[j];
        }
    }
    if (mon) {   // (every thread of the block reaches this point: dead threads report 0)
        cp::monitor_flush(mon, amax);
        cp::monitor_count_launch(mon, threadIdx.x == 0);
    }
}

// U[p][o][k_off + c] = (G g G^T)[p] on the device, g read from the MASTER weights through strides so that one kernel serves
// the forward layout (HWIO or IHWO) and the flipped / transposed data-gradient layout; run after every optimizer step.
__global__ void wino_weight_kernel(const float* __restrict__ w, long long s_ky, long long s_kx, long long s_in, long long s_out, int flip, int channels,
                                   int cout, int ldk, int k_off, float* __restrict__ U) {
    const long long total = (long long)channels * cout;
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % channels);   // consecutive lanes -> consecutive k: coalesced U stores
        const int o = (int)(i / channels);
        float g[3][3], t[6][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                g[ky][kx] = w[(flip ? 2 - ky : ky) * s_ky + (flip ? 2 - kx : kx) * s_kx + c * s_in + o * s_out];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) t[a][kx] = G[a][0] * g[0][kx] + G[a][1] * g[1][kx] + G[a][2] * g[2][kx];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b)
                U[((size_t)(a * 6 + b) * cout + o) * ldk + k_off + c] = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
    }
}

// A applied to four values (the transpose of the output transform): dM = A dY A^T turns the gradient of a 4x4 output tile into
// the gradient of the 36 Winograd products (weight gradient: dU[p] = sum_t dM[p][t]^T V[p][t])
__device__ __forceinline__ void a6(const float4 (&y)[4], float4 (&m)[6]) {
    m[0] = y[0];
    m[1] = y[0] + y[1] + y[2] + y[3];
    m[2] = y[0] - y[1] + y[2] - y[3];
    m[3] = y[0] + 2.f * y[1] + 4.f * y[2] + 8.f * y[3];
    m[4] = y[0] - 2.f * y[1] + 4.f * y[2] - 8.f * y[3];
    m[5] = y[3];
}

__global__ __launch_bounds__(THREADS) void wino_dy_kernel(const float* __restrict__ dy, int ld, int C, WinoGeom g, float* __restrict__ dM, uint32_t* mon) {
    float amax = 0.f;   // f16x2 range monitor (common.h): max |dM| over what this launch writes (the weight-gradient GEMM may convert it to fp16 pairs)
    const int c4n = C >> 2;
    const long long total = (long long)g.Tp * c4n;  // the padding tiles are ZEROED: the weight-gradient GEMM reduces over all Tp rows
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int t = (int)(i / c4n);
        if (t >= g.T) {
            float* z = dM + (size_t)t * C + c4 * 4;
            for (int p = 0; p < 36; ++p) *reinterpret_cast<float4*>(z + (size_t)p * g.Tp * C) = f4(0.f);
            continue;
        }
        int n, sy, sx, tu, tv;
        tile_coords(t, g, n, sy, sx, tu, tv);
        float4 tt[6][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = sx + g.d * (4 * tv + j);
            float4 col[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = sy + g.d * (4 * tu + r);
                col[r] = (y < g.H && x < g.W) ? *reinterpret_cast<const float4*>(dy + (((size_t)n * g.H + y) * g.W + x) * ld + c4 * 4) : f4(0.f);
            }
            float4 o[6];
            a6(col, o);
#pragma unroll
            for (int r = 0; r < 6; ++r) tt[r][j] = o[r];
        }
        float* dst = dM + (size_t)t * C + c4 * 4;
        const size_t plane = (size_t)g.Tp * C;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float4 o[6];
            a6(tt[r], o);
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<float4*>(dst + (size_t)(r * 6 + j) * plane) = o[j];
            if (mon) {   // uniform
#pragma unroll
                for (int j = 0; j < 6; ++j) amax = cp::amax4(amax, o[j]);
            }
        }
    }
    if (mon) {
        cp::monitor_flush(mon, amax);
        cp::monitor_count_launch(mon, threadIdx.x == 0);
    }
}

// dW(ky,kx,c,o) = sum_{a,b} G[a][ky] G[b][kx] dU[a*6+b][o][k_off + c]   (the adjoint of wino_weight_kernel), written through strides
__global__ void wino_weight_grad_kernel(const float* __restrict__ dU, int channels, int cout, int ldk, int k_off, long long s_ky, long long s_kx,
                                        long long s_in, long long s_out, float* __restrict__ dw, int accumulate) {
    const long long total = (long long)channels * cout;
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % channels);
        const int o = (int)(i / channels);
        float t[3][6];  // G^T dU
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int b = 0; b < 6; ++b) t[ky][b] = 0.f;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const float u = dU[((size_t)(a * 6 + b) * cout + o) * ldk + k_off + c];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) t[ky][b] += G[a][ky] * u;
            }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float v = 0.f;
#pragma unroll
                for (int b = 0; b < 6; ++b) v += t[ky][b] * G[b][kx];
                float* dst = dw + ky * s_ky + kx * s_kx + c * s_in + o * s_out;
                *dst = accumulate ? *dst + v : v;
            }
    }
}

int make_geom(int batch, int h, int w, int dil, WinoGeom& g) {
    if (batch <= 0 || h <= 0 || w <= 0 || dil <= 0) return CP_ERR_INVALID;
    g.B = batch; g.H = h; g.W = w; g.d = dil;
    const int hs = (h + dil - 1) / dil, ws = (w + dil - 1) / dil;
    g.Tu = (hs + 3) / 4;
    g.Tv = (ws + 3) / 4;
    const long long T = (long long)batch * dil * dil * g.Tu * g.Tv;
    if (T * 36 >= (1LL << 31)) return CP_ERR_INVALID;
    g.T = (int)T;
    g.Tp = (int)((T + 127) / 128 * 128);
    return CP_OK;
}

inline int grid_for(long long n) {
    long long b = (n + THREADS - 1) / THREADS;
    return (int)(b < 1 ? 1 : (b > 256 * 16 ? 256 * 16 : b));
}

}  // namespace

extern "C" int cp_wino_tiles(int batch, int h, int w, int dilation, int* tiles, int* tiles_padded) {
    WinoGeom g;
    if (make_geom(batch, h, w, dilation, g) != CP_OK) {
        cp::set_error("cp_wino_tiles: bad geometry %dx%dx%d dilation %d", batch, h, w, dilation);
        return CP_ERR_INVALID;
    }
    if (tiles) *tiles = g.T;
    if (tiles_padded) *tiles_padded = g.Tp;
    return CP_OK;
}

extern "C" int cp_wino_pack_weights_host(const float* w_hwio, int cin_total, int cout, int c_begin, int channels, int real_channels, int ldk,
                                         int k_off, float* dst) {
    // dst[p][co][k_off + c] = (G g G^T)[p] for input channel c_begin + c; dst is [36][cout][ldk], zero-filled by the caller once
    CP_REQUIRE(w_hwio && dst && cout > 0 && real_channels > 0 && real_channels <= channels && k_off >= 0 && k_off + channels <= ldk,
               "cp_wino_pack_weights_host: bad arguments");
    static const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    for (int c = 0; c < real_channels; ++c)
        for (int co = 0; co < cout; ++co) {
            double g[3][3], t[6][3];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w_hwio[(((size_t)ky * 3 + kx) * cin_total + c_begin + c) * cout + co];
            for (int a = 0; a < 6; ++a)
                for (int kx = 0; kx < 3; ++kx) t[a][kx] = G[a][0] * g[0][kx] + G[a][1] * g[1][kx] + G[a][2] * g[2][kx];
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b)
                    dst[((size_t)(a * 6 + b) * cout + co) * ldk + k_off + c] = (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]);
        }
    return CP_OK;
}

extern "C" int cp_wino_input_transform_pre_f32(const float* src, int ld, int channels, int batch, int h, int w, int dilation, float* V, int ldv,
                                               int c_off, const float* pre_scale, const float* pre_shift, int pre_act, void* stream) {
    CP_REQUIRE(src && V && channels > 0 && channels % 4 == 0 && ld >= channels && ld % 4 == 0 && c_off >= 0 && c_off % 4 == 0 && c_off + channels <= ldv,
               "cp_wino_input_transform_f32: bad arguments");
    CP_REQUIRE((pre_scale || !pre_shift) && (pre_shift || !pre_scale || pre_act == CP_ACT_NONE) && (((uintptr_t)pre_scale | (uintptr_t)pre_shift) & 15) == 0,
               "cp_wino_input_transform_pre_f32: pre_scale and pre_shift come together (pre_scale alone = a per-channel factor, no activation), 16-byte aligned");
    WinoGeom g;
    CP_REQUIRE(make_geom(batch, h, w, dilation, g) == CP_OK, "cp_wino_input_transform_f32: bad geometry");
    CP_REQUIRE(!pre_scale || pre_act == CP_ACT_NONE || pre_act == CP_ACT_RELU || pre_act == CP_ACT_LEAKY01, "cp_wino_input_transform_pre_f32: unknown activation %d", pre_act);
    const dim3 grid(grid_for((long long)g.T * (channels / 4)));
#define CP_WIN(P_) CP_LAUNCH(wino_in_kernel<P_>, grid, dim3(THREADS), 0, (hipStream_t)stream, src, ld, channels, g, V, ldv, c_off, pre_scale, pre_shift, pre_act, cp::f16x2_monitor())
    if (!pre_scale) CP_WIN(0);
    else if (!pre_shift) CP_WIN(4);
    else if (pre_act == CP_ACT_RELU) CP_WIN(1 + CP_ACT_RELU);
    else if (pre_act == CP_ACT_LEAKY01) CP_WIN(1 + CP_ACT_LEAKY01);
    else CP_WIN(1 + CP_ACT_NONE);
#undef CP_WIN
    return cp::check_launch("cp_wino_input_transform_f32");
}

extern "C" int cp_wino_input_transform_f32(const float* src, int ld, int channels, int batch, int h, int w, int dilation, float* V, int ldv,
                                           int c_off, void* stream) {
    return cp_wino_input_transform_pre_f32(src, ld, channels, batch, h, w, dilation, V, ldv, c_off, nullptr, nullptr, 0, stream);
}

extern "C" int cp_wino_output_transform_f32(const float* M, int cout, int batch, int h, int w, int dilation, const float* residual, int residual_ld,
                                            const float* scale, const float* shift, const uint8_t* epi_label, int act, float* out_raw,
                                            int out_raw_ld, float* out_act, int out_act_ld, void* stream) {
    return cp_wino_output_transform_stats_f32(M, cout, batch, h, w, dilation, residual, residual_ld, scale, shift, epi_label, act, out_raw, out_raw_ld,
                                              out_act, out_act_ld, nullptr, stream);
}

extern "C" int cp_wino_output_transform_stats_f32(const float* M, int cout, int batch, int h, int w, int dilation, const float* residual, int residual_ld,
                                                  const float* scale, const float* shift, const uint8_t* epi_label, int act, float* out_raw,
                                                  int out_raw_ld, float* out_act, int out_act_ld, double* stats, void* stream) {
    CP_REQUIRE(M && cout > 0 && cout % 4 == 0 && (out_raw || out_act), "cp_wino_output_transform_f32: bad arguments");
    CP_REQUIRE((scale == nullptr) == (shift == nullptr) && (!epi_label || scale), "cp_wino_output_transform_f32: scale/shift/label combination");
    CP_REQUIRE((!out_raw || out_raw_ld >= cout) && (!out_act || out_act_ld >= cout) && (!residual || residual_ld >= cout), "cp_wino_output_transform_f32: ld < cout");
    WinoGeom g;
    CP_REQUIRE(make_geom(batch, h, w, dilation, g) == CP_OK, "cp_wino_output_transform_f32: bad geometry");
    WinoEpi e{residual, residual_ld, scale, shift, epi_label, act, out_raw, out_raw_ld, out_act, out_act_ld, stats};
    hipStream_t st = (hipStream_t)stream;
    int blocks = grid_for((long long)g.T * (cout / 4));
    size_t lds = 0;
    if (stats) {   // fewer, longer-lived blocks: every block ends with 2 * cout fp64 atomics
        CP_REQUIRE(cout <= 2048, "cp_wino_output_transform_stats_f32: cout <= 2048");
        if (hipMemsetAsync(stats, 0, sizeof(double) * 2 * cout, st) != hipSuccess) return cp::check_launch("cp_wino_output_transform_stats_f32 memset");
        static const int cap = getenv("CP_WINO_STATS_BLOCKS") ? atoi(getenv("CP_WINO_STATS_BLOCKS")) : 1024;   // tuning aid
        if (blocks > cap) blocks = cap;
        lds = sizeof(double) * 2 * cout;
    }
    if (stats)
        CP_LAUNCH(wino_out_kernel<true>, dim3(blocks), dim3(THREADS), lds, st, M, cout, g, e);
    else
        CP_LAUNCH(wino_out_kernel<false>, dim3(blocks), dim3(THREADS), 0, st, M, cout, g, e);
    return cp::check_launch("cp_wino_output_transform_f32");
}


// block shape of the fused transform: 32 channels per block (8 quads, 128-byte segments of every M / V row) when a sub-grid's tiles x 8 fit 256
// threads and its padded map 64 KB of LDS (60x80 at dilation 4: 4 x 5 tiles); else 16 channels (4 quads) with up to 384 threads and 96 KB
// (the 30x40 sub-grids of dilation 2: 8 x 10 tiles); 0 = not applicable (dilation 1 at this size: one sub-grid is the whole image)
static int out_in_quads(const WinoGeom& g) {
    const int tiles = g.Tu * g.Tv;
    const size_t px = (size_t)(4 * g.Tu + 2) * (4 * g.Tv + 2);
    if (tiles * 8 <= 256 && px * 8 * 16 <= 64 * 1024) return 8;
    if (tiles * 4 <= 384 && px * 4 * 16 <= 96 * 1024) return 4;
    return 0;
}

extern "C" int cp_wino_output_input_applicable(int batch, int h, int w, int dilation, int cout) {
    WinoGeom g;
    if (make_geom(batch, h, w, dilation, g) != CP_OK || cout <= 0 || cout % 4) return 0;
    // measured (bs 16, 60x80): the 32-channel form saves 0.13 ms over the three stage-4 pairs; the 16-channel form (64-byte segments of every M / V
    // row) LOSES 0.1 ms on the stage-3 pairs against the two separate passes -- so only the former is used unless CP_WINO_OUT_IN_MIN_QUADS=4
    static const int max_quads = getenv("CP_WINO_OUT_IN_MIN_QUADS") ? atoi(getenv("CP_WINO_OUT_IN_MIN_QUADS")) : 8;
    const int q = out_in_quads(g);
    return q >= max_quads ? 1 : 0;
}

extern "C" int cp_wino_output_input_transform_f32(const float* M, int cout, int batch, int h, int w, int dilation, const float* residual, int residual_ld,
                                                  const float* scale, const float* shift, int act, float* out_raw, int out_raw_ld, float* out_act,
                                                  int out_act_ld, float* V, int ldv, int c_off, void* stream) {
    CP_REQUIRE(M && V && cout > 0 && cout % 4 == 0 && c_off >= 0 && c_off % 4 == 0 && c_off + cout <= ldv, "cp_wino_output_input_transform_f32: bad arguments");
    CP_REQUIRE((scale == nullptr) == (shift == nullptr), "cp_wino_output_input_transform_f32: scale and shift come together");
    CP_REQUIRE((!out_raw || out_raw_ld >= cout) && (!out_act || out_act_ld >= cout) && (!residual || residual_ld >= cout), "cp_wino_output_input_transform_f32: ld < cout");
    WinoGeom g;
    CP_REQUIRE(make_geom(batch, h, w, dilation, g) == CP_OK, "cp_wino_output_input_transform_f32: bad geometry");
    const int quads = out_in_quads(g);
    CP_REQUIRE(quads > 0, "cp_wino_output_input_transform_f32: a sub-grid of %dx%d at dilation %d does not fit one block (cp_wino_output_input_applicable)", h, w, dilation);
    WinoEpi e{residual, residual_ld, scale, shift, nullptr, act, out_raw, out_raw_ld, out_act, out_act_ld, nullptr};
    const int slices = (cout / 4 + quads - 1) / quads;
    const int threads = (g.Tu * g.Tv * quads + 63) / 64 * 64;
    const size_t lds = (size_t)(4 * g.Tu + 2) * (4 * g.Tv + 2) * quads * 16;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_out_in_kernel<8, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_out_in_kernel<4, 384>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    const dim3 grid(batch * dilation * dilation * slices);
    if (quads == 8)
        CP_LAUNCH((wino_out_in_kernel<8, 256>), grid, dim3(threads), lds, (hipStream_t)stream, M, cout, g, e, V, ldv, c_off, cp::f16x2_monitor());
    else
        CP_LAUNCH((wino_out_in_kernel<4, 384>), grid, dim3(threads), lds, (hipStream_t)stream, M, cout, g, e, V, ldv, c_off, cp::f16x2_monitor());
    return cp::check_launch("cp_wino_output_input_transform_f32");
}

extern "C" int cp_wino_transform_weights_f32(const float* w, long long stride_ky, long long stride_kx, long long stride_in, long long stride_out,
                                             int flip, int channels, int cout, int ldk, int k_off, float* U, void* stream) {
    CP_REQUIRE(w && U && channels > 0 && cout > 0 && k_off >= 0 && k_off + channels <= ldk, "cp_wino_transform_weights_f32: bad arguments");
    CP_LAUNCH(wino_weight_kernel, dim3(grid_for((long long)channels * cout)), dim3(THREADS), 0, (hipStream_t)stream, w, stride_ky, stride_kx, stride_in,
              stride_out, flip, channels, cout, ldk, k_off, U);
    return cp::check_launch("cp_wino_transform_weights_f32");
}


extern "C" int cp_wino_dy_transform_f32(const float* dy, int ld, int channels, int batch, int h, int w, int dilation, float* dM, void* stream) {
    CP_REQUIRE(dy && dM && channels > 0 && channels % 4 == 0 && ld >= channels && ld % 4 == 0, "cp_wino_dy_transform_f32: bad arguments");
    WinoGeom g;
    CP_REQUIRE(make_geom(batch, h, w, dilation, g) == CP_OK, "cp_wino_dy_transform_f32: bad geometry");
    CP_LAUNCH(wino_dy_kernel, dim3(grid_for((long long)g.Tp * (channels / 4))), dim3(THREADS), 0, (hipStream_t)stream, dy, ld, channels, g, dM, cp::f16x2_monitor());
    return cp::check_launch("cp_wino_dy_transform_f32");
}

extern "C" int cp_wino_weight_grad_f32(const float* dU, int channels, int cout, int ldk, int k_off, long long stride_ky, long long stride_kx,
                                       long long stride_in, long long stride_out, float* dw, int accumulate, void* stream) {
    CP_REQUIRE(dU && dw && channels > 0 && cout > 0 && k_off >= 0 && k_off + channels <= ldk, "cp_wino_weight_grad_f32: bad arguments");
    CP_LAUNCH(wino_weight_grad_kernel, dim3(grid_for((long long)channels * cout)), dim3(THREADS), 0, (hipStream_t)stream, dU, channels, cout, ldk, k_off,
              stride_ky, stride_kx, stride_in, stride_out, dw, accumulate);
    return cp::check_launch("cp_wino_weight_grad_f32");
}
