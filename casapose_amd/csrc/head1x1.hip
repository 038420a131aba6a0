// The two 1x1 output heads of the TRAINING step (pv_final_conv_segmentation: 32 -> K, pv_final_conv_vertex: 32 -> ver_dim; pose_models.py:546,616)
// as streaming kernels: 32 input channels, <= 32 output channels, full resolution -- 6.4 M pixels at bs 32 / 448 x 448, 128 bytes in per pixel and a
// few GFLOP, i.e. HBM-bound.  The general implicit-GEMM kernel (one K chunk per tile, pipeline fill and drain per 128 pixels) ran these at
// 1.4-1.6 TB/s; here a wave keeps the whole 32 x 32 weight tile in registers and streams pixels through exact fp32 MFMAs
// (v_mfma_f32_32x32x2_f32, the arithmetic of cp_conv2d_fwd_f32 / cp_conv2d_wgrad_f32):
//   forward        out[p][q]  = sum_c x[p][c] W[c][q]          A = pixels x channels (each lane loads ITS 16 channels: 64 contiguous bytes)
//   data gradient  dx[p][c] (+)= sum_q dy[p][q] W[c][q]        same shape with the weight tile transposed
//   weight grad    dW[c][q] (+)= sum_p x[p][c] dy[p][q]        k = pixels: both operands are coalesced 128-byte rows
// Weights and their gradient are addressed in the Keras layout [cin][cout] directly (no packing, no scatter).  (In inference the heads are
// fused into the epilogues of decoder blocks 5 / 10; in training batch normalisation sits between, so they are separate passes.)
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CIN = 32;

// k-pair of MFMA m: channels (m, 16 + m) -- lane half h supplies channel 16 h + m, i.e. each lane needs 16 CONSECUTIVE channels
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, int ld_x, long long pixels, const float* __restrict__ w, int cout,
                                                       float* __restrict__ out, int ld_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, h = lane >> 5;
    float wb[16];   // B[k = 16h + m][j = row]
#pragma unroll
    for (int m = 0; m < 16; ++m) wb[m] = row < cout ? w[(16 * h + m) * cout + row] : 0.f;
    const long long groups = (pixels + 31) >> 5;
    const long long gstep = (long long)gridDim.x * 4;
    auto load = [&](long long g, float4 (&a)[4]) {
        const long long p = g * 32 + row;
        if (g < groups && p < pixels) {
            const float4* src = reinterpret_cast<const float4*>(x + p * ld_x + 16 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = src[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float4 a[4], an[4];
    long long g = (long long)blockIdx.x * 4 + wave;
    load(g, a);
    for (; g < groups; g += gstep) {
        load(g + gstep, an);   // the next group's rows are in flight while this one is multiplied and stored
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, wb[4 * i + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, wb[4 * i + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, wb[4 * i + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, wb[4 * i + 3], acc, 0, 0, 0);
        }
        // D[i = pixel][j = q]: this lane holds column q = row of pixels (r & 3) + 8 (r >> 2) + 4 h
        if (row < cout) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long pp = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (pp < pixels) out[pp * ld_out + row] = acc[r];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = an[i];
    }
}

// WIDE: every dy row has 32 readable floats from `dy` on (16-byte aligned): four 16-byte loads per lane, columns >= cout masked after the load;
// otherwise only the cout real columns are touched, one 4-byte load each
template <bool WIDE>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ dy, int ld_dy, long long pixels, const float* __restrict__ w, int cout,
                                                         float* __restrict__ dx, int ld_dx, int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, h = lane >> 5;
    float wb[16];   // B[k = q = 16h + m][j = c = row] = W[c][q]
#pragma unroll
    for (int m = 0; m < 16; ++m) wb[m] = (16 * h + m) < cout ? w[row * cout + 16 * h + m] : 0.f;
    const long long groups = (pixels + 31) >> 5;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const long long p = g * 32 + row;
        float av[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) av[m] = 0.f;
        if (p < pixels) {
            if constexpr (WIDE) {
                const float4* src = reinterpret_cast<const float4*>(dy + p * ld_dy + 16 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 v = src[i];
                    av[4 * i + 0] = v.x; av[4 * i + 1] = v.y; av[4 * i + 2] = v.z; av[4 * i + 3] = v.w;
                }
#pragma unroll
                for (int m = 0; m < 16; ++m)
                    if (16 * h + m >= cout) av[m] = 0.f;   // whatever lives there (padding, another head's gradient) must not meet even a zero weight
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m)
                    if (16 * h + m < cout) av[m] = dy[p * ld_dy + 16 * h + m];
            }
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], wb[m], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long pp = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (pp < pixels) {
                float* d = dx + pp * ld_dx + row;
                *d = accumulate ? *d + acc[r] : acc[r];
            }
        }
    }
}

// k = pixels: MFMA m of a 32-pixel group multiplies pixels (2m, 2m + 1); A[i = c][k] = x[pixel][c], B[k][j = q] = dy[pixel][q]
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ x, int ld_x, const float* __restrict__ dy, int ld_dy, long long pixels,
                                                         int cout, float* __restrict__ dw) {
    __shared__ float red[4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const long long groups = (pixels + 31) >> 5;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        float av[16], bv[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const long long p = g * 32 + 2 * m + h;
            const bool ok = p < pixels;
            av[m] = ok ? x[p * ld_x + row] : 0.f;
            bv[m] = (ok && row < cout) ? dy[p * ld_dy + row] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[m], acc, 0, 0, 0);
    }
    // block reduction of the four waves, then one atomic per (c, q) and block
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
        const int r = i >> 6, l = i & 63;
        const float s = red[0][r][l] + red[1][r][l] + red[2][r][l] + red[3][r][l];
        const int c = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), q = l & 31;   // D[i = c][j = q]
        if (q < cout && s != 0.f) atomicAdd(&dw[c * cout + q], s);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Fused normalisation of decoder blocks 5 / 10 (casapose.py:76-105 is ONE Keras block: convolution, (class-adaptive) normalisation,
// activation; the heads follow, pose_models.py:546,616).  With batch statistics the raw convolution output x [pixels][32] has to be in
// memory once (the statistics are a global reduction), but the ACTIVATED tensor y = act(x * scale[l] + shift[l]) and its gradient need
// not: the head's forward and weight gradient recompute y from x (table rows from LDS), and the two passes of the normalisation backward
// recompute the head's data gradient g_y = dOut W^T on the matrix pipe instead of reading a stored one.  Per block that removes one
// write + two reads of y and one write + two reads of g_y (822 MB each at bs 32 / 448 x 448).
// tables: scale / shift / gamma [classes][32]; labels = nullptr means class 0 everywhere.
constexpr int MAX_CLASSES = 64;

__device__ __forceinline__ float head_act(float t, int act) {
    if (act == CP_ACT_RELU) return fmaxf(t, 0.f);
    if (act == CP_ACT_LEAKY01) return fmaxf(t, 0.f) - fmaxf(-0.1f * t, 0.f);
    return t;
}
__device__ __forceinline__ float head_act_grad(float t, int act) {
    if (act == CP_ACT_RELU) return t > 0.f ? 1.f : 0.f;
    if (act == CP_ACT_LEAKY01) return t > 0.f ? 1.f : (t < 0.f ? 0.1f : 0.f);
    return 1.f;
}

// out[p][q] = sum_c act(fma(x[p][c], scale[l_p][c], shift[l_p][c])) W[c][q]
// RECORD: the head writes COMPLETE output records out[p][0 .. pre_n + cout) = [pre[p][0 .. pre_n) | its own cout columns].  A head that writes
// its 9 or 27 columns into 36-float records leaves every 128-byte line partly written, and a partly dirty line costs the memory system a
// read-modify-write: 556 us for the 9-column head into the records against 209 us into dense rows of 9 floats at bs 32 / 448 x 448
// (tools/debug/head_probe.py).  So the FIRST head writes dense rows and the LAST one copies them in front of its own columns: whole lines
// only.  The copy is dealt over the wave by ELEMENT (32 pixels x pre_n floats, 64 per round: with dense prefix rows a round is one
// contiguous 256-byte load), requested before the group's MFMAs and stored behind them, next to the head columns of the same pixels.
constexpr int PRE_ROUNDS = 8;   // 32 * pre_n <= 64 * PRE_ROUNDS: pre_n <= 16
template <bool RECORD>
__global__ __launch_bounds__(256) void head_fwd_affine_kernel(const float* __restrict__ x, int ld_x, long long pixels, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const uint8_t* __restrict__ labels, int classes, int act,
                                                              const float* __restrict__ w, int cout, float* __restrict__ out, int ld_out,
                                                              const float* __restrict__ pre, int ld_pre, int pre_n) {
    extern __shared__ float4 ftab[];   // [2][classes * 8]
    float4* tab[2] = {ftab, ftab + classes * 8};
    for (int i = threadIdx.x; i < classes * 8; i += 256) {
        tab[0][i] = reinterpret_cast<const float4*>(scale)[i];
        tab[1][i] = reinterpret_cast<const float4*>(shift)[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, h = lane >> 5;
    float wb[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) wb[m] = row < cout ? w[(16 * h + m) * cout + row] : 0.f;
    // element e = 64 t + lane of a group's prefix block: pixel e / pre_n, float e % pre_n -- the same for every group, kept as 32-bit offsets
    const int pre_e = 32 * pre_n, pre_rounds = (pre_e + 63) >> 6;
    int pre_pix[PRE_ROUNDS], pre_in[PRE_ROUNDS], pre_out[PRE_ROUNDS];
    if constexpr (RECORD) {
        const uint32_t magic = 65536u / (uint32_t)pre_n + 1u;   // e / pre_n == (e * magic) >> 16 for e < 1024, pre_n <= 32
#pragma unroll
        for (int t = 0; t < PRE_ROUNDS; ++t) {
            const uint32_t e = 64u * t + lane, pix = (e * magic) >> 16, f = e - pix * pre_n;
            pre_pix[t] = (int)e < pre_e ? (int)pix : 32;   // (32 = never inside a group)
            pre_in[t] = (int)(pix * ld_pre + f);
            pre_out[t] = (int)(pix * ld_out + f);
        }
    }
    const long long groups = (pixels + 31) >> 5;
    const long long gstep = (long long)gridDim.x * 4;
    auto load = [&](long long g, float4 (&a)[4], int& l) {
        const long long p = g * 32 + row;
        l = 0;
        if (g < groups && p < pixels) {
            const float4* src = reinterpret_cast<const float4*>(x + p * ld_x + 16 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = src[i];
            if (labels) l = labels[p];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float4 a[4], an[4];
    int l, ln;
    long long g = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave);
    load(g, a, l);
    for (; g < groups; g += gstep) {
        load(g + gstep, an, ln);
        float pv[PRE_ROUNDS];
        const int rem = pixels - g * 32 < 32 ? (int)(pixels - g * 32) : 32;   // pixels of this group
        if constexpr (RECORD) {
#pragma unroll
            for (int t = 0; t < PRE_ROUNDS; ++t)
                if (t < pre_rounds) pv[t] = pre_pix[t] < rem ? (pre + g * 32 * ld_pre)[pre_in[t]] : 0.f;
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const bool live = g * 32 + row < pixels;   // rows past the end stay zero (act(shift) would not be)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 s = tab[0][l * 8 + 4 * h + i], b = tab[1][l * 8 + 4 * h + i];
            float4 y;
            y.x = live ? head_act(__builtin_fmaf(a[i].x, s.x, b.x), act) : 0.f;
            y.y = live ? head_act(__builtin_fmaf(a[i].y, s.y, b.y), act) : 0.f;
            y.z = live ? head_act(__builtin_fmaf(a[i].z, s.z, b.z), act) : 0.f;
            y.w = live ? head_act(__builtin_fmaf(a[i].w, s.w, b.w), act) : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.x, wb[4 * i + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.y, wb[4 * i + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.z, wb[4 * i + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.w, wb[4 * i + 3], acc, 0, 0, 0);
        }
        float* const o = RECORD ? out + pre_n : out;
        if (row < cout) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long pp = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (pp < pixels) o[pp * ld_out + row] = acc[r];
            }
        }
        if constexpr (RECORD) {
#pragma unroll
            for (int t = 0; t < PRE_ROUNDS; ++t)
                if (t < pre_rounds && pre_pix[t] < rem) (out + g * 32 * ld_out)[pre_out[t]] = pv[t];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = an[i];
        l = ln;
    }
}

// dW[c][q] (+)= sum_p act(fma(x[p][c], scale[l_p][c], shift[l_p][c])) dy[p][q]
// LABELS = false: one class, the lane's two table entries live in registers.  LABELS = true: the 32 labels of a pixel group arrive as two 16-byte
// loads (the same address for every lane), the table rows come from LDS ([classes][32] floats: consecutive lanes, consecutive banks).
template <bool LABELS>
__global__ __launch_bounds__(256) void head_wgrad_affine_kernel(const float* __restrict__ x, int ld_x, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const uint8_t* __restrict__ labels, int classes, int act,
                                                                const float* __restrict__ dy, int ld_dy, long long pixels, int cout, float* __restrict__ dw) {
    extern __shared__ float wsm[];   // [4][16][64] block reduction, then [2][classes * 32] tables
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(wsm);
    float* tab0 = wsm + 4 * 16 * 64;
    float* tab1 = tab0 + classes * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, h = lane >> 5;
    if constexpr (LABELS) {
        for (int i = threadIdx.x; i < classes * 32; i += 256) {
            tab0[i] = scale[i];
            tab1[i] = shift[i];
        }
        __syncthreads();
    }
    const float s0 = scale[row], b0 = shift[row];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const long long groups = pixels >> 5;   // pixels % 32 == 0
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        float av[16], bv[16];
        uint32_t lw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (LABELS) {
            const uint4 l0 = *reinterpret_cast<const uint4*>(labels + g * 32), l1 = *reinterpret_cast<const uint4*>(labels + g * 32 + 16);
            lw[0] = l0.x; lw[1] = l0.y; lw[2] = l0.z; lw[3] = l0.w; lw[4] = l1.x; lw[5] = l1.y; lw[6] = l1.z; lw[7] = l1.w;
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const long long p = g * 32 + 2 * m + h;
            av[m] = x[p * ld_x + row];
            bv[m] = row < cout ? dy[p * ld_dy + row] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            float s = s0, b = b0;
            if constexpr (LABELS) {
                const int l = (int)((lw[m >> 1] >> (16 * (m & 1) + 8 * h)) & 255u);   // pixel 2m + h of the group
                s = tab0[l * 32 + row];
                b = tab1[l * 32 + row];
            }
            av[m] = head_act(__builtin_fmaf(av[m], s, b), act);
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[m], acc, 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
        const int r = i >> 6, l = i & 63;
        const float s = red[0][r][l] + red[1][r][l] + red[2][r][l] + red[3][r][l];
        const int c = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), q = l & 31;
        if (q < cout && s != 0.f) atomicAdd(&dw[c * cout + q], s);
    }
}

struct HeadBn {
    const float* x;        // raw convolution output [pixels][ld_x], 32 channels
    int ld_x;
    const float* dout;     // gradient of the head's output, rows of ld_dout floats, >= 32 readable floats per row from here on (16-byte aligned)
    int ld_dout;
    long long pixels;      // multiple of 32
    const float* w;        // head weights [32][cout] (Keras)
    int cout;
    const float* mean;     // [32]
    const float* rstd;     // [32]
    const float* gamma;    // [classes][32]
    const float* fscale;   // [classes][32] the forward's folded tables: the activation branch is decided by fma(x, fscale, fshift)
    const float* fshift;
    const uint8_t* labels; // or nullptr
    int classes, act;
};

// g_y tile of pixel group g: acc[r] = (dOut W^T)[pixel g*32 + (r & 3) + 8 (r >> 2) + 4 h][channel row]
__device__ __forceinline__ f32x16 head_gy_tile(const HeadBn& k, long long g, int row, int h, const float (&wb)[16]) {
    const long long p = g * 32 + row;
    float av[16];
    const float4* src = reinterpret_cast<const float4*>(k.dout + p * k.ld_dout + 16 * h);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = src[i];
        av[4 * i + 0] = v.x; av[4 * i + 1] = v.y; av[4 * i + 2] = v.z; av[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int m = 0; m < 16; ++m)
        if (16 * h + m >= k.cout) av[m] = 0.f;   // another head's gradient / padding lives there
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], wb[m], acc, 0, 0, 0);
    return acc;
}

// reduce pass of the normalisation backward with g_y recomputed: red[(l*32 + c)*2 + {0,1}] += {g, g*xhat}, chan[c*2 + {0,1}] += gamma[l][c] * {g, g*xhat}
__global__ __launch_bounds__(256) void head_bn_bwd_reduce_kernel(HeadBn k, double* __restrict__ red, double* __restrict__ chan) {
    extern __shared__ double sred[];   // [classes*32*2] + [32*2], then three float tables [classes*32]
    const int nred = k.classes * 64;
    float* tab = reinterpret_cast<float*>(sred + nred + 64);
    float* t_fs = tab;
    float* t_fb = tab + k.classes * 32;
    float* t_gm = tab + 2 * k.classes * 32;
    for (int i = threadIdx.x; i < nred + 64; i += 256) sred[i] = 0.0;
    for (int i = threadIdx.x; i < k.classes * 32; i += 256) {
        t_fs[i] = k.fscale[i];
        t_fb[i] = k.fshift[i];
        t_gm[i] = k.gamma ? k.gamma[i] : 1.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, h = lane >> 5;
    float wb[16];   // B[k = q = 16h + m][j = c = row] = W[c][q]
#pragma unroll
    for (int m = 0; m < 16; ++m) wb[m] = (16 * h + m) < k.cout ? k.w[row * k.cout + 16 * h + m] : 0.f;
    const float mu = k.mean[row], rs = k.rstd[row];
    int cur = -1;
    double a0 = 0.0, a1 = 0.0;
    float fs = 0.f, fb = 0.f;
    auto flush = [&]() {
        if (cur < 0) return;
        const double gm = (double)t_gm[cur * 32 + row];
        atomicAdd(&sred[(cur * 32 + row) * 2 + 0], a0);
        atomicAdd(&sred[(cur * 32 + row) * 2 + 1], a1);
        atomicAdd(&sred[nred + row * 2 + 0], a0 * gm);
        atomicAdd(&sred[nred + row * 2 + 1], a1 * gm);
    };
    const long long groups = k.pixels >> 5;
    // a wave takes runs of 8 consecutive groups (256 consecutive pixels of a row: long label runs), the runs grid-strided
    const long long nrun = (groups + 7) >> 3;
    for (long long run = (long long)blockIdx.x * 4 + wave; run < nrun; run += (long long)gridDim.x * 4)
        for (long long g = run * 8; g < run * 8 + 8 && g < groups; ++g) {
            const f32x16 gy = head_gy_tile(k, g, row, h, wb);
            uint32_t lw[4] = {0, 0, 0, 0};
            if (k.labels) {
#pragma unroll
                for (int j = 0; j < 4; ++j) lw[j] = *reinterpret_cast<const uint32_t*>(k.labels + g * 32 + 8 * j + 4 * h);
            }
            float xv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = k.x[(g * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * k.ld_x + row];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int l = (int)((lw[r >> 2] >> (8 * (r & 3))) & 255u);
                if (l != cur) {
                    flush();
                    cur = l;
                    a0 = a1 = 0.0;
                    fs = t_fs[l * 32 + row];
                    fb = t_fb[l * 32 + row];
                }
                const float t = __builtin_fmaf(xv[r], fs, fb);
                const float gg = gy[r] * head_act_grad(t, k.act);
                const float xh = (xv[r] - mu) * rs;
                a0 += gg;
                a1 += (double)gg * xh;
            }
        }
    flush();
    __syncthreads();
    for (int i = threadIdx.x; i < nred; i += 256)
        if (sred[i] != 0.0) atomicAdd(&red[i], sred[i]);
    for (int i = threadIdx.x; i < 64; i += 256)
        if (sred[nred + i] != 0.0) atomicAdd(&chan[i], sred[nred + i]);
}

// apply pass: dx[p][c] = rstd[c] * (g*gamma[l][c] - m1[c] - xhat*m2[c]) * row_scale[p], g = g_y * act'(t) recomputed as in the reduce pass
__global__ __launch_bounds__(256) void head_bn_bwd_apply_kernel(HeadBn k, const double* __restrict__ chan, double inv_n, const float* __restrict__ row_scale,
                                                                float* __restrict__ dx, int ld_dx) {
    extern __shared__ float atab[];   // three float tables [classes*32]
    float* t_fs = atab;
    float* t_fb = atab + k.classes * 32;
    float* t_gm = atab + 2 * k.classes * 32;
    for (int i = threadIdx.x; i < k.classes * 32; i += 256) {
        t_fs[i] = k.fscale[i];
        t_fb[i] = k.fshift[i];
        t_gm[i] = k.gamma ? k.gamma[i] : 1.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, h = lane >> 5;
    float wb[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) wb[m] = (16 * h + m) < k.cout ? k.w[row * k.cout + 16 * h + m] : 0.f;
    const float mu = k.mean[row], rs = k.rstd[row];
    const float m1 = (float)(chan[row * 2 + 0] * inv_n), m2 = (float)(chan[row * 2 + 1] * inv_n);
    const long long groups = k.pixels >> 5;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const f32x16 gy = head_gy_tile(k, g, row, h, wb);
        uint32_t lw[4] = {0, 0, 0, 0};
        float4 rsc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k.labels) lw[j] = *reinterpret_cast<const uint32_t*>(k.labels + g * 32 + 8 * j + 4 * h);
            rsc[j] = row_scale ? *reinterpret_cast<const float4*>(row_scale + g * 32 + 8 * j + 4 * h) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
        float xv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) xv[r] = k.x[(g * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * k.ld_x + row];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int l = (int)((lw[r >> 2] >> (8 * (r & 3))) & 255u);
            const float t = __builtin_fmaf(xv[r], t_fs[l * 32 + row], t_fb[l * 32 + row]);
            const float gg = gy[r] * head_act_grad(t, k.act) * t_gm[l * 32 + row];
            const float xh = (xv[r] - mu) * rs;
            const float4 rv = rsc[r >> 2];
            const float sc = (r & 3) == 0 ? rv.x : ((r & 3) == 1 ? rv.y : ((r & 3) == 2 ? rv.z : rv.w));
            dx[(g * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * ld_dx + row] = rs * (gg - m1 - xh * m2) * sc;
        }
    }
}

// Grid-stride kernels, one wave per 32-pixel group at a time: as many blocks as are RESIDENT at once (occupancy x 256 CUs) and no more -- a grid
// of 1.3 or 2.7 resident sets leaves the chip a third empty for its last round (the kernels run for 0.2-0.5 ms, a round is most of that:
// cp_head1x1_bn_bwd_reduce_f32 476 -> 388 us at bs 32 / 448 x 448).
template <typename K>
int grid_for_groups(K kernel, size_t lds, long long pixels, long long groups_per_wave = 1) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess || per_cu < 1) per_cu = 4;
    const long long resident = 256LL * per_cu;
    const long long blocks = ((pixels + 31) / 32 + 4 * groups_per_wave - 1) / (4 * groups_per_wave);
    return (int)(blocks < 1 ? 1 : (blocks > resident ? resident : blocks));
}

}  // namespace

extern "C" int cp_head1x1_fwd_f32(const float* x, int ld_x, long long pixels, const float* w, int cout, float* out, int ld_out, void* stream) {
    CP_REQUIRE(x && w && out && pixels > 0 && cout > 0 && cout <= 32, "cp_head1x1_fwd_f32: bad arguments (32 input channels, 1 <= cout <= 32)");
    CP_REQUIRE(ld_x >= CIN && ld_x % 4 == 0 && ((uintptr_t)x & 15) == 0 && ld_out >= cout, "cp_head1x1_fwd_f32: x rows must be 16-byte aligned float4 rows of >= 32 channels");
    CP_LAUNCH(head_fwd_kernel, dim3(grid_for_groups(head_fwd_kernel, 0, pixels)), dim3(256), 0, (hipStream_t)stream, x, ld_x, pixels, w, cout, out, ld_out);
    return cp::check_launch("cp_head1x1_fwd_f32");
}

extern "C" int cp_head1x1_dgrad_f32(const float* dy, int ld_dy, int dy_row_floats, long long pixels, const float* w, int cout, float* dx, int ld_dx,
                                    int accumulate, void* stream) {
    CP_REQUIRE(dy && w && dx && pixels > 0 && cout > 0 && cout <= 32 && ld_dy >= cout && dy_row_floats >= cout && ld_dx >= CIN, "cp_head1x1_dgrad_f32: bad arguments");
    if (dy_row_floats >= 32 && ld_dy % 4 == 0 && ((uintptr_t)dy & 15) == 0)
        CP_LAUNCH(head_dgrad_kernel<true>, dim3(grid_for_groups(head_dgrad_kernel<true>, 0, pixels)), dim3(256), 0, (hipStream_t)stream, dy, ld_dy, pixels, w, cout, dx, ld_dx, accumulate);
    else
        CP_LAUNCH(head_dgrad_kernel<false>, dim3(grid_for_groups(head_dgrad_kernel<false>, 0, pixels)), dim3(256), 0, (hipStream_t)stream, dy, ld_dy, pixels, w, cout, dx, ld_dx, accumulate);
    return cp::check_launch("cp_head1x1_dgrad_f32");
}

extern "C" int cp_head1x1_wgrad_f32(const float* x, int ld_x, const float* dy, int ld_dy, long long pixels, int cout, float* dw, int accumulate,
                                    void* stream) {
    CP_REQUIRE(x && dy && dw && pixels > 0 && cout > 0 && cout <= 32 && ld_x >= CIN && ld_dy >= cout, "cp_head1x1_wgrad_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate)
        if (hipMemsetAsync(dw, 0, sizeof(float) * CIN * cout, st) != hipSuccess) return cp::check_launch("cp_head1x1_wgrad_f32 memset");
    int blocks = grid_for_groups(head_wgrad_kernel, 0, pixels);
    if (blocks > 1024) blocks = 1024;
    CP_LAUNCH(head_wgrad_kernel, dim3(blocks), dim3(256), 0, st, x, ld_x, dy, ld_dy, pixels, cout, dw);
    return cp::check_launch("cp_head1x1_wgrad_f32");
}

extern "C" int cp_head1x1_fwd_affine_f32(const float* x, int ld_x, long long pixels, const float* scale, const float* shift, const uint8_t* labels,
                                         int classes, int act, const float* w, int cout, float* out, int ld_out, void* stream) {
    CP_REQUIRE(x && w && out && scale && shift && pixels > 0 && cout > 0 && cout <= 32, "cp_head1x1_fwd_affine_f32: bad arguments (32 input channels, 1 <= cout <= 32)");
    CP_REQUIRE(ld_x >= CIN && ld_x % 4 == 0 && ((uintptr_t)x & 15) == 0 && ld_out >= cout, "cp_head1x1_fwd_affine_f32: x rows must be 16-byte aligned float4 rows of >= 32 channels");
    CP_REQUIRE(classes >= 1 && classes <= MAX_CLASSES && (classes == 1 || labels) && ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0,
               "cp_head1x1_fwd_affine_f32: 1 <= classes <= 64, labels for the class-adaptive form, 16-byte aligned tables");
    const size_t lds = 2 * (size_t)classes * 32 * sizeof(float);
    CP_LAUNCH(head_fwd_affine_kernel<false>, dim3(grid_for_groups(head_fwd_affine_kernel<false>, lds, pixels)), dim3(256), lds, (hipStream_t)stream, x, ld_x, pixels,
              scale, shift, labels, classes, act, w, cout, out, ld_out, (const float*)nullptr, 0, 0);
    return cp::check_launch("cp_head1x1_fwd_affine_f32");
}

extern "C" int cp_head1x1_fwd_affine_record_f32(const float* x, int ld_x, long long pixels, const float* scale, const float* shift, const uint8_t* labels,
                                                int classes, int act, const float* w, int cout, const float* prefix, int ld_prefix, int prefix_n,
                                                float* out, int ld_out, void* stream) {
    CP_REQUIRE(x && w && out && scale && shift && prefix && pixels > 0 && cout > 0 && cout <= 32, "cp_head1x1_fwd_affine_record_f32: bad arguments (32 input channels, 1 <= cout <= 32)");
    CP_REQUIRE(ld_x >= CIN && ld_x % 4 == 0 && ((uintptr_t)x & 15) == 0, "cp_head1x1_fwd_affine_record_f32: x rows must be 16-byte aligned float4 rows of >= 32 channels");
    CP_REQUIRE(prefix_n >= 1 && prefix_n <= 2 * PRE_ROUNDS && ld_prefix >= prefix_n && ld_out >= prefix_n + cout,
               "cp_head1x1_fwd_affine_record_f32: 1 <= prefix_n <= 16, rows of >= prefix_n floats in, records of >= prefix_n + cout floats out");
    CP_REQUIRE(classes >= 1 && classes <= MAX_CLASSES && (classes == 1 || labels) && ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0,
               "cp_head1x1_fwd_affine_record_f32: 1 <= classes <= 64, labels for the class-adaptive form, 16-byte aligned tables");
    const size_t lds = 2 * (size_t)classes * 32 * sizeof(float);
    CP_LAUNCH(head_fwd_affine_kernel<true>, dim3(grid_for_groups(head_fwd_affine_kernel<true>, lds, pixels)), dim3(256), lds, (hipStream_t)stream, x, ld_x, pixels,
              scale, shift, labels, classes, act, w, cout, out, ld_out, prefix, ld_prefix, prefix_n);
    return cp::check_launch("cp_head1x1_fwd_affine_record_f32");
}

extern "C" int cp_head1x1_wgrad_affine_f32(const float* x, int ld_x, const float* scale, const float* shift, const uint8_t* labels, int classes, int act,
                                           const float* dy, int ld_dy, long long pixels, int cout, float* dw, int accumulate, void* stream) {
    CP_REQUIRE(x && dy && dw && scale && shift && pixels > 0 && cout > 0 && cout <= 32 && ld_x >= CIN && ld_dy >= cout, "cp_head1x1_wgrad_affine_f32: bad arguments");
    CP_REQUIRE(classes >= 1 && classes <= MAX_CLASSES && (classes == 1 || labels) && (!labels || ((uintptr_t)labels & 15) == 0) && pixels % 32 == 0,
               "cp_head1x1_wgrad_affine_f32: 1 <= classes <= 64, 16-byte aligned labels for the class-adaptive form, pixels %% 32 == 0");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate)
        if (hipMemsetAsync(dw, 0, sizeof(float) * CIN * cout, st) != hipSuccess) return cp::check_launch("cp_head1x1_wgrad_affine_f32 memset");
    const size_t lds = sizeof(float) * (4 * 16 * 64 + 2 * (size_t)classes * 32);
    const bool lab = labels && classes > 1;
    int blocks = lab ? grid_for_groups(head_wgrad_affine_kernel<true>, lds, pixels) : grid_for_groups(head_wgrad_affine_kernel<false>, lds, pixels);
    if (blocks > 1024) blocks = 1024;
    if (lab)
        CP_LAUNCH(head_wgrad_affine_kernel<true>, dim3(blocks), dim3(256), lds, st, x, ld_x, scale, shift, labels, classes, act, dy, ld_dy, pixels, cout, dw);
    else
        CP_LAUNCH(head_wgrad_affine_kernel<false>, dim3(blocks), dim3(256), lds, st, x, ld_x, scale, shift, labels, classes, act, dy, ld_dy, pixels, cout, dw);
    return cp::check_launch("cp_head1x1_wgrad_affine_f32");
}

static int head_bn_args(const char* fn, HeadBn& k, const float* x, int ld_x, const float* dout, int ld_dout, int dout_row_floats, long long pixels,
                        const float* w, int cout, const float* mean, const float* rstd, const float* gamma, const float* fwd_scale, const float* fwd_shift,
                        const uint8_t* labels, int classes, int act) {
    CP_REQUIRE(x && dout && w && mean && rstd && fwd_scale && fwd_shift, "%s: null pointer", fn);
    CP_REQUIRE(pixels > 0 && pixels % 32 == 0 && cout > 0 && cout <= 32 && ld_x >= CIN, "%s: pixels must be a positive multiple of 32, 1 <= cout <= 32", fn);
    CP_REQUIRE(dout_row_floats >= 32 && ld_dout % 4 == 0 && ((uintptr_t)dout & 15) == 0, "%s: every dout row needs 32 readable floats, 16-byte aligned", fn);
    CP_REQUIRE(classes >= 1 && classes <= MAX_CLASSES && (classes == 1 || labels) && (!labels || ((uintptr_t)labels & 3) == 0),
               "%s: 1 <= classes <= 64, 4-byte aligned labels for the class-adaptive form", fn);
    k = HeadBn{x, ld_x, dout, ld_dout, pixels, w, cout, mean, rstd, gamma, fwd_scale, fwd_shift, labels, classes, act};
    return CP_OK;
}

extern "C" int cp_head1x1_bn_bwd_reduce_f32(const float* x, int ld_x, const float* dout, int ld_dout, int dout_row_floats, long long pixels, const float* w,
                                            int cout, const float* mean, const float* rstd, const float* gamma, const float* fwd_scale,
                                            const float* fwd_shift, const uint8_t* labels, int classes, int act, double* red, double* chan, void* stream) {
    HeadBn k;
    if (int rc = head_bn_args("cp_head1x1_bn_bwd_reduce_f32", k, x, ld_x, dout, ld_dout, dout_row_floats, pixels, w, cout, mean, rstd, gamma, fwd_scale,
                              fwd_shift, labels, classes, act))
        return rc;
    CP_REQUIRE(red && chan, "cp_head1x1_bn_bwd_reduce_f32: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const size_t nred = (size_t)classes * 64;
    if (chan == red + nred) {
        if (hipMemsetAsync(red, 0, (nred + 64) * sizeof(double), st) != hipSuccess) return cp::check_launch("cp_head1x1_bn_bwd_reduce_f32 memset");
    } else {
        if (hipMemsetAsync(red, 0, nred * sizeof(double), st) != hipSuccess) return cp::check_launch("cp_head1x1_bn_bwd_reduce_f32 memset");
        if (hipMemsetAsync(chan, 0, 64 * sizeof(double), st) != hipSuccess) return cp::check_launch("cp_head1x1_bn_bwd_reduce_f32 memset");
    }
    const size_t lds = (nred + 64) * sizeof(double) + 3 * (size_t)classes * 32 * sizeof(float);
    const int blocks = grid_for_groups(head_bn_bwd_reduce_kernel, lds, pixels, 8);   // a wave takes runs of 8 groups: >= 1 run per wave
    CP_LAUNCH(head_bn_bwd_reduce_kernel, dim3((unsigned)blocks), dim3(256), lds, st, k, red, chan);
    return cp::check_launch("cp_head1x1_bn_bwd_reduce_f32");
}

extern "C" int cp_head1x1_bn_bwd_apply_f32(const float* x, int ld_x, const float* dout, int ld_dout, int dout_row_floats, long long pixels, const float* w,
                                           int cout, const float* mean, const float* rstd, const float* gamma, const float* fwd_scale,
                                           const float* fwd_shift, const uint8_t* labels, int classes, int act, const double* chan, double global_pixels,
                                           const float* row_scale, float* dx, int ld_dx, void* stream) {
    HeadBn k;
    if (int rc = head_bn_args("cp_head1x1_bn_bwd_apply_f32", k, x, ld_x, dout, ld_dout, dout_row_floats, pixels, w, cout, mean, rstd, gamma, fwd_scale,
                              fwd_shift, labels, classes, act))
        return rc;
    CP_REQUIRE(chan && dx && ld_dx >= CIN && global_pixels > 0 && (!row_scale || ((uintptr_t)row_scale & 15) == 0), "cp_head1x1_bn_bwd_apply_f32: bad arguments");
    const size_t lds = 3 * (size_t)classes * 32 * sizeof(float);
    CP_LAUNCH(head_bn_bwd_apply_kernel, dim3(grid_for_groups(head_bn_bwd_apply_kernel, lds, pixels)), dim3(256), lds, (hipStream_t)stream, k, chan, 1.0 / global_pixels, row_scale, dx, ld_dx);
    return cp::check_launch("cp_head1x1_bn_bwd_apply_f32");
}
