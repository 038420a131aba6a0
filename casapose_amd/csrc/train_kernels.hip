// Streaming kernels of the TRAINING path: batch-norm statistics, normalise + activate, their
// backward, the adjoint of the resampling layers, weight re-packing and Adam.
//
// Reference call sites: (Sync)BatchNormalization with training=True (resnet.py:78,100,247,250,303;
// casapose.py:77; _normalization_layers.py:108 -- batch statistics over (N,H,W), biased variance,
// SURVEY B5), ClassAdaptiveWeightedNormalization.calc (_normalization_layers.py:119-139), ReLU and the
// leaky pair (casapose.py:98-107), MaxPooling2D / UpSampling2D(bilinear) / GuidedUpsampling gradients
// (the reference gets them from tf.GradientTape, train_casapose.py:594), Keras Adam
// (train_casapose.py:334-347, eps 1e-7).  All of them are HBM-bound: 16 B per lane per access,
// grid-stride over a capped grid, fp64 accumulation for the statistics.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int THREADS = 256;

inline int grid_for(long long n) {
    long long b = (n + THREADS - 1) / THREADS;
    return (int)(b < 1 ? 1 : (b > 256 * 8 ? 256 * 8 : b));
}

// ---------------------------------------------------------------------------------------------
// statistics: sums[c] += sum x, sums[C + c] += sum x^2
// A block walks rows (pixels) with stride gridDim; thread t owns channel group (t % C4) and row lane
// (t / C4): fp64 partials per thread, LDS reduce, one fp64 atomic per value and block.
// 1024-thread blocks, at most 512 of them: every block ends with 2C atomics on the SAME 2C addresses, and same-address fp64 atomics retire one
// after the other (~9 ns each, measured with tools/debug/stats_probe.py: 2048 blocks of 256 threads spent 18 us of a 49 us launch waiting for
// that chain; 100352 x 512: 70 -> 47 us, 401408 x 64: 49 -> 31 us at equal thread counts).
constexpr int STAT_THREADS = 1024;
__global__ __launch_bounds__(STAT_THREADS) void bn_stats_kernel(const float* __restrict__ x, long long pixels, int C, int ld,
                                                                double* __restrict__ sums) {
    extern __shared__ double sred[];  // [2][C]
    const int c4n = C >> 2;
    const int lanes_per_row = c4n;              // threads covering one pixel
    const int rows_per_pass = STAT_THREADS / lanes_per_row;  // C <= 4096
    const int c4 = threadIdx.x % lanes_per_row;
    const int rl = threadIdx.x / lanes_per_row;
    for (int i = threadIdx.x; i < 2 * C; i += STAT_THREADS) sred[i] = 0.0;
    __syncthreads();
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (rl < rows_per_pass) {
        for (long long r0 = (long long)blockIdx.x * rows_per_pass; r0 < pixels; r0 += (long long)gridDim.x * rows_per_pass) {
            const long long r = r0 + rl;
            if (r < pixels) {
                const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c4 * 4);
                s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
                q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(&sred[c4 * 4 + e], s[e]);
            atomicAdd(&sred[C + c4 * 4 + e], q[e]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += STAT_THREADS) atomicAdd(&sums[i], sred[i]);
}

__device__ __forceinline__ float act_fwd(float t, int act) {
    if (act == CP_ACT_RELU) return fmaxf(t, 0.f);
    if (act == CP_ACT_LEAKY01) return fmaxf(t, 0.f) - fmaxf(-0.1f * t, 0.f);
    return t;
}
__device__ __forceinline__ float act_grad(float t, int act) {  // derivative wrt the pre-activation t
    if (act == CP_ACT_RELU) return t > 0.f ? 1.f : 0.f;
    if (act == CP_ACT_LEAKY01) return t > 0.f ? 1.f : (t < 0.f ? 0.1f : 0.f);
    return 1.f;
}

// y = act(x*scale[l][c] + shift[l][c]) (+ add), l = labels ? labels[pixel] : 0
__global__ void affine_act_kernel(const float* __restrict__ x, long long pixels, int C, int ld_x, const float* __restrict__ scale,
                                  const float* __restrict__ shift, const uint8_t* __restrict__ labels, int act,
                                  float* __restrict__ y, int ld_y) {
    const int c4n = C >> 2;
    const long long total = pixels * c4n;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    if (!labels && nthreads % c4n == 0) {   // a thread keeps its channel group: the two table rows are loaded once (see bn_act_bwd_apply_kernel)
        const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
        const int c4 = (int)(i0 % c4n);
        const float4 s = *reinterpret_cast<const float4*>(scale + c4 * 4);
        const float4 b = *reinterpret_cast<const float4*>(shift + c4 * 4);
        for (long long i = i0; i < total; i += nthreads) {
            const long long r = i / c4n;
            const float4 v = *reinterpret_cast<const float4*>(x + r * ld_x + c4 * 4);
            float4 o;
            o.x = act_fwd(__builtin_fmaf(v.x, s.x, b.x), act); o.y = act_fwd(__builtin_fmaf(v.y, s.y, b.y), act);
            o.z = act_fwd(__builtin_fmaf(v.z, s.z, b.z), act); o.w = act_fwd(__builtin_fmaf(v.w, s.w, b.w), act);
            *reinterpret_cast<float4*>(y + r * ld_y + c4 * 4) = o;
        }
        return;
    }
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const long long r = i / c4n;
        const int l = labels ? labels[r] : 0;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld_x + c4 * 4);
        const float4 s = *reinterpret_cast<const float4*>(scale + (size_t)l * C + c4 * 4);
        const float4 b = *reinterpret_cast<const float4*>(shift + (size_t)l * C + c4 * 4);
        float4 o;
        // one explicit fused multiply-add per element: the backward kernels repeat exactly this expression to decide the activation's branch
        o.x = act_fwd(__builtin_fmaf(v.x, s.x, b.x), act); o.y = act_fwd(__builtin_fmaf(v.y, s.y, b.y), act);
        o.z = act_fwd(__builtin_fmaf(v.z, s.z, b.z), act); o.w = act_fwd(__builtin_fmaf(v.w, s.w, b.w), act);
        *reinterpret_cast<float4*>(y + r * ld_y + c4 * 4) = o;
    }
}

// backward of y = act(t), t = gamma[l][c]*xhat + beta[l][c], xhat = (x - mean[c]) * rstd[c]
// reduce pass: red[(l*C + c)*2 + {0,1}] += {g, g*xhat} (g = dy*act'(t))   -> dbeta[l][c], dgamma[l][c]
//              chan[c*2 + {0,1}]       += {g*gamma, g*gamma*xhat}           -> the two means of the BN backward
// (1024-thread blocks as in bn_stats_kernel were measured and lose here: 7.06 against 6.65 ms of normalisation backward per step)
constexpr int RED_THREADS = 256;
__global__ __launch_bounds__(RED_THREADS) void bn_act_bwd_reduce_kernel(const float* __restrict__ x, int ld_x, const float* __restrict__ dy, int ld_dy,
                                                                    long long pixels, int C, int classes, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, const uint8_t* __restrict__ labels,
                                                                    int act, const float* __restrict__ fscale, const float* __restrict__ fshift,
                                                                    double* __restrict__ red, double* __restrict__ chan) {
    extern __shared__ double sred[];  // [classes*C*2] + [C*2]
    const int nred = classes * C * 2, nch = C * 2;
    for (int i = threadIdx.x; i < nred + nch; i += RED_THREADS) sred[i] = 0.0;
    __syncthreads();
    const int c4n = C >> 2;
    const int rows_per_pass = RED_THREADS / c4n;
    const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n;
    if (rl < rows_per_pass) {
        const float4 mu = *reinterpret_cast<const float4*>(mean + c4 * 4);
        const float4 rs = *reinterpret_cast<const float4*>(rstd + c4 * 4);
        int cur = -1;
        double a[4][4];  // per channel: g, g*xhat, g*gamma, g*gamma*xhat for the current label
        float4 gm = make_float4(1, 1, 1, 1), bt = make_float4(0, 0, 0, 0);
        float4 fs = make_float4(0, 0, 0, 0), fb = make_float4(0, 0, 0, 0);   // the forward's folded table row (branch decision)
        auto flush = [&]() {
            if (cur < 0) return;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = c4 * 4 + e;
                atomicAdd(&sred[((size_t)cur * C + c) * 2 + 0], a[e][0]);
                atomicAdd(&sred[((size_t)cur * C + c) * 2 + 1], a[e][1]);
                atomicAdd(&sred[nred + c * 2 + 0], a[e][2]);
                atomicAdd(&sred[nred + c * 2 + 1], a[e][3]);
            }
        };
        // a block takes runs of SUB consecutive passes (a thread's successive pixels inside a run are rows_per_pass apart, i.e. neighbours in
        // the label map), the runs grid-strided: with a plain grid-strided walk a thread's next pixel lay ~32 k pixels away and nearly every
        // step of a CLADE layer changed the label -> 16 LDS fp64 atomics per step (block 10: 1.37 ms against 0.79 ms without labels)
        constexpr int SUB = 8;
        const long long run = (long long)SUB * rows_per_pass;
        for (long long rr = (long long)blockIdx.x * run; rr < pixels; rr += (long long)gridDim.x * run) {
            // all SUB row pairs of the run are requested before the first is used (16 loads in flight per thread instead of 2)
            float4 xq[SUB], dq[SUB];
            int lq[SUB];
#pragma unroll
            for (int k = 0; k < SUB; ++k) {
                const long long r = rr + (long long)k * rows_per_pass + rl;
                const bool ok = r < pixels;
                const long long rc = ok ? r : 0;
                xq[k] = *reinterpret_cast<const float4*>(x + rc * ld_x + c4 * 4);
                dq[k] = *reinterpret_cast<const float4*>(dy + rc * ld_dy + c4 * 4);
                lq[k] = ok ? (labels ? (int)labels[rc] : 0) : -1;
            }
#pragma unroll
            for (int k = 0; k < SUB; ++k) {
                const int l = lq[k];
                if (l < 0) continue;
                if (l != cur) {
                    flush();
                    cur = l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e][0] = a[e][1] = a[e][2] = a[e][3] = 0.0;
                    gm = gamma ? *reinterpret_cast<const float4*>(gamma + (size_t)l * C + c4 * 4) : make_float4(1, 1, 1, 1);
                    bt = beta ? *reinterpret_cast<const float4*>(beta + (size_t)l * C + c4 * 4) : make_float4(0, 0, 0, 0);
                    if (fscale) {
                        fs = *reinterpret_cast<const float4*>(fscale + (size_t)l * C + c4 * 4);
                        fb = *reinterpret_cast<const float4*>(fshift + (size_t)l * C + c4 * 4);
                    }
                }
                const float4 xv = xq[k], dv = dq[k];
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
                const float mus[4] = {mu.x, mu.y, mu.z, mu.w}, rss[4] = {rs.x, rs.y, rs.z, rs.w};
                const float gms[4] = {gm.x, gm.y, gm.z, gm.w}, bts[4] = {bt.x, bt.y, bt.z, bt.w};
                const float fss[4] = {fs.x, fs.y, fs.z, fs.w}, fbs[4] = {fb.x, fb.y, fb.z, fb.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (xs[e] - mus[e]) * rss[e];
                    // the branch the FORWARD took (affine_act_kernel's own expression) when its tables are given, else recomputed from gamma / beta
                    const float t = fscale ? __builtin_fmaf(xs[e], fss[e], fbs[e]) : gms[e] * xh + bts[e];
                    const float g = ds[e] * act_grad(t, act);
                    a[e][0] += g;
                    a[e][1] += (double)g * xh;
                    a[e][2] += (double)g * gms[e];
                    a[e][3] += (double)g * gms[e] * xh;
                }
            }
        }
        flush();
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nred; i += RED_THREADS)
        if (sred[i] != 0.0) atomicAdd(&red[i], sred[i]);
    for (int i = threadIdx.x; i < nch; i += RED_THREADS)
        if (sred[nred + i] != 0.0) atomicAdd(&chan[i], sred[nred + i]);
}

// apply pass: dx = rstd * (g*gamma - m1[c] - xhat*m2[c]), m1 = chan[c][0]/N, m2 = chan[c][1]/N  (N = GLOBAL pixel count)
__global__ void bn_act_bwd_apply_kernel(const float* __restrict__ x, int ld_x, const float* __restrict__ dy, int ld_dy, long long pixels, int C,
                                        const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, const uint8_t* __restrict__ labels, int act,
                                        const float* __restrict__ fscale, const float* __restrict__ fshift,
                                        const double* __restrict__ chan, double inv_n, const float* __restrict__ row_scale,
                                        float* __restrict__ dx, int ld_dx, int accumulate) {
    const int c4n = C >> 2;
    const long long total = pixels * c4n;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    if (!labels && nthreads % c4n == 0) {
        // one class and a thread count that is a multiple of the channel groups: a thread keeps ITS channel group for the whole walk, so the
        // ten table rows are loaded once instead of once per element (they shared the L1 with the x / dy stream); same expressions, same bits
        const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
        const int c4 = (int)(i0 % c4n);
        const float4 mu = *reinterpret_cast<const float4*>(mean + c4 * 4);
        const float4 rs = *reinterpret_cast<const float4*>(rstd + c4 * 4);
        const float4 gm = gamma ? *reinterpret_cast<const float4*>(gamma + c4 * 4) : make_float4(1, 1, 1, 1);
        const float4 bt = beta ? *reinterpret_cast<const float4*>(beta + c4 * 4) : make_float4(0, 0, 0, 0);
        const float4 fs = fscale ? *reinterpret_cast<const float4*>(fscale + c4 * 4) : make_float4(0, 0, 0, 0);
        const float4 fb = fscale ? *reinterpret_cast<const float4*>(fshift + c4 * 4) : make_float4(0, 0, 0, 0);
        const float mus[4] = {mu.x, mu.y, mu.z, mu.w}, rss[4] = {rs.x, rs.y, rs.z, rs.w};
        const float gms[4] = {gm.x, gm.y, gm.z, gm.w}, bts[4] = {bt.x, bt.y, bt.z, bt.w};
        const float fss[4] = {fs.x, fs.y, fs.z, fs.w}, fbs[4] = {fb.x, fb.y, fb.z, fb.w};
        float m1[4], m2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            m1[e] = (float)(chan[(c4 * 4 + e) * 2 + 0] * inv_n);
            m2[e] = (float)(chan[(c4 * 4 + e) * 2 + 1] * inv_n);
        }
        for (long long i = i0; i < total; i += nthreads) {
            const long long r = i / c4n;
            const float4 xv = *reinterpret_cast<const float4*>(x + r * ld_x + c4 * 4);
            const float4 dv = *reinterpret_cast<const float4*>(dy + r * ld_dy + c4 * 4);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
            const float rsc = row_scale ? row_scale[r] : 1.f;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xs[e] - mus[e]) * rss[e];
                const float t = fscale ? __builtin_fmaf(xs[e], fss[e], fbs[e]) : gms[e] * xh + bts[e];
                const float g = ds[e] * act_grad(t, act) * gms[e];
                o[e] = rss[e] * (g - m1[e] - xh * m2[e]) * rsc;
            }
            float4* dst = reinterpret_cast<float4*>(dx + r * ld_dx + c4 * 4);
            if (accumulate) { const float4 old = *dst; o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w; }
            *dst = make_float4(o[0], o[1], o[2], o[3]);
        }
        return;
    }
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const long long r = i / c4n;
        const int l = labels ? labels[r] : 0;
        const float4 xv = *reinterpret_cast<const float4*>(x + r * ld_x + c4 * 4);
        const float4 dv = *reinterpret_cast<const float4*>(dy + r * ld_dy + c4 * 4);
        const float4 mu = *reinterpret_cast<const float4*>(mean + c4 * 4);
        const float4 rs = *reinterpret_cast<const float4*>(rstd + c4 * 4);
        const float4 gm = gamma ? *reinterpret_cast<const float4*>(gamma + (size_t)l * C + c4 * 4) : make_float4(1, 1, 1, 1);
        const float4 bt = beta ? *reinterpret_cast<const float4*>(beta + (size_t)l * C + c4 * 4) : make_float4(0, 0, 0, 0);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
        const float mus[4] = {mu.x, mu.y, mu.z, mu.w}, rss[4] = {rs.x, rs.y, rs.z, rs.w};
        const float gms[4] = {gm.x, gm.y, gm.z, gm.w}, bts[4] = {bt.x, bt.y, bt.z, bt.w};
        const float4 fs = fscale ? *reinterpret_cast<const float4*>(fscale + (size_t)l * C + c4 * 4) : make_float4(0, 0, 0, 0);
        const float4 fb = fscale ? *reinterpret_cast<const float4*>(fshift + (size_t)l * C + c4 * 4) : make_float4(0, 0, 0, 0);
        const float fss[4] = {fs.x, fs.y, fs.z, fs.w}, fbs[4] = {fb.x, fb.y, fb.z, fb.w};
        float o[4];
        const float rsc = row_scale ? row_scale[r] : 1.f;  // partial convolution: the normalised tensor was rowscale * conv
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xs[e] - mus[e]) * rss[e];
            const float t = fscale ? __builtin_fmaf(xs[e], fss[e], fbs[e]) : gms[e] * xh + bts[e];   // same branch as the forward and the reduce pass
            const float g = ds[e] * act_grad(t, act) * gms[e];
            const float m1 = (float)(chan[(c4 * 4 + e) * 2 + 0] * inv_n), m2 = (float)(chan[(c4 * 4 + e) * 2 + 1] * inv_n);
            o[e] = rss[e] * (g - m1 - xh * m2) * rsc;
        }
        float4* dst = reinterpret_cast<float4*>(dx + r * ld_dx + c4 * 4);
        if (accumulate) { const float4 old = *dst; o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w; }
        *dst = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// adjoints of the resampling layers (gather form: no atomics, deterministic)
// max-pool 3x3/s2 zero-pad-1: dx(iy,ix) = sum over the <=4 windows containing it where x(iy,ix) is the
// window's FIRST maximum in raster order (the tie rule of the forward arg-max; post-ReLU inputs: the zero
// padding can tie with zeros, in which case the padded tap wins only if it comes first)
// A thread owns the 2x2 input block (2oy + {0,1}, 2ox + {0,1}) of four channels: the four windows that can contain those inputs,
// (oy + {0,1}, ox + {0,1}), cover a 5x5 input patch, which is loaded ONCE (25 x 16 bytes; the first version re-derived the arg-max of up to
// four windows per input element, 36 loads each: 0.74 ms at bs 32 against ~0.1 ms of traffic).
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int H, int W, int C, int Ho, int Wo,
                                   float* __restrict__ dx, int accumulate) {
    const int c4n = C >> 2;
    const int Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;   // 2x2 input blocks
    const long long total = (long long)B * Hb * Wb * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        long long t = i / c4n;
        const int bx = (int)(t % Wb);
        t /= Wb;
        const int by = (int)(t % Hb);
        const int n = (int)(t / Hb);
        // patch rows 2by-1 .. 2by+3, columns 2bx-1 .. 2bx+3 (zero outside the image: the forward pads with zeros)
        float4 pt[5][5];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int yy = 2 * by - 1 + r, xx = 2 * bx - 1 + q;
                pt[r][q] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                               ? *reinterpret_cast<const float4*>(x + (((size_t)n * H + yy) * W + xx) * C + c4 * 4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        float o[2][2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) o[a][b][e] = 0.f;
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int oy = by + wy, ox = bx + wx;
                if (oy >= Ho || ox >= Wo) continue;
                // first maximum of the window in raster order, per channel (its taps are patch rows 2wy.., columns 2wx..)
                float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int bidx[4] = {-1, -1, -1, -1};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float4 v = pt[2 * wy + ky][2 * wx + kx];
                        const float vs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (vs[e] > best[e]) { best[e] = vs[e]; bidx[e] = ky * 3 + kx; }
                    }
                const float4 g = *reinterpret_cast<const float4*>(dy + (((size_t)n * Ho + oy) * Wo + ox) * C + c4 * 4);
                const float gs[4] = {g.x, g.y, g.z, g.w};
                // the block's inputs inside this window: input (a, b) sits at patch (1 + a, 1 + b) = window tap (1 + a - 2wy, 1 + b - 2wx)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int ky = 1 + a - 2 * wy, kx = 1 + b - 2 * wx;
                        if (ky < 0 || ky > 2 || kx < 0 || kx > 2) continue;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (bidx[e] == ky * 3 + kx) o[a][b][e] += gs[e];
                    }
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int iy = 2 * by + a, ix = 2 * bx + b;
                if (iy >= H || ix >= W) continue;
                float4* dst = reinterpret_cast<float4*>(dx + (((size_t)n * H + iy) * W + ix) * C + c4 * 4);
                float4 r = make_float4(o[a][b][0], o[a][b][1], o[a][b][2], o[a][b][3]);
                if (accumulate) { const float4 old = *dst; r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w; }
                *dst = r;
            }
    }
}

// The same adjoint from the arg-max TAP INDEX the forward recorded (cp_maxpool3x3s2_idx_f32: one byte per output element, tap 0..8 in raster order,
// first maximum wins, the zero padding takes part): no x reads at all -- per 2x2 input block the four windows' index bytes and dy values instead of
// a 5x5 patch of x (0.60 -> ~0.25 ms at bs 32: the 411 MB input is no longer touched).
__global__ void maxpool_bwd_idx_kernel(const uint8_t* __restrict__ idx, const float* __restrict__ dy, int B, int H, int W, int C, int Ho, int Wo,
                                       float* __restrict__ dx, int accumulate) {
    const int c4n = C >> 2;
    const int Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
    const long long total = (long long)B * Hb * Wb * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        long long t = i / c4n;
        const int bx = (int)(t % Wb);
        t /= Wb;
        const int by = (int)(t % Hb);
        const int n = (int)(t / Hb);
        float o[2][2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) o[a][b][e] = 0.f;
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int oy = by + wy, ox = bx + wx;
                if (oy >= Ho || ox >= Wo) continue;
                const size_t op = (((size_t)n * Ho + oy) * Wo + ox) * C + c4 * 4;
                const uint32_t iw = *reinterpret_cast<const uint32_t*>(idx + op);
                const float4 g = *reinterpret_cast<const float4*>(dy + op);
                const float gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int ky = 1 + a - 2 * wy, kx = 1 + b - 2 * wx;   // input (2by + a, 2bx + b) as a tap of window (oy, ox)
                        if (ky < 0 || ky > 2 || kx < 0 || kx > 2) continue;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if ((int)((iw >> (8 * e)) & 255u) == ky * 3 + kx) o[a][b][e] += gs[e];
                    }
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int iy = 2 * by + a, ix = 2 * bx + b;
                if (iy >= H || ix >= W) continue;
                float4* dst = reinterpret_cast<float4*>(dx + (((size_t)n * H + iy) * W + ix) * C + c4 * 4);
                float4 r = make_float4(o[a][b][0], o[a][b][1], o[a][b][2], o[a][b][3]);
                if (accumulate) { const float4 old = *dst; r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w; }
                *dst = r;
            }
    }
}

// forward that records the arg-max tap (training): same maximum as maxpool_kernel (aux_kernels.hip), first maximum in raster order
__global__ void maxpool_idx_kernel(const float* __restrict__ src, int B, int H, int W, int C, int Ho, int Wo, float* __restrict__ dst,
                                   uint8_t* __restrict__ idx) {
    const int c4n = C >> 2;
    const long long total = (long long)B * Ho * Wo * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const long long pix = i / c4n;
        const int ox = (int)(pix % Wo);
        const long long t = pix / Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                    v = *reinterpret_cast<const float4*>(src + (((size_t)n * H + iy) * W + ix) * C + c4 * 4);
                const float vs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (vs[e] > best[e]) { best[e] = vs[e]; bi[e] = ky * 3 + kx; }
            }
        *reinterpret_cast<float4*>(dst + (size_t)pix * C + c4 * 4) = make_float4(best[0], best[1], best[2], best[3]);
        *reinterpret_cast<uint32_t*>(idx + (size_t)pix * C + c4 * 4) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    }
}

// bilinear x2 (half-pixel centres): dx(y,x) = sum of dy over the <=16 hi-res pixels that sample (y,x)
__global__ void bilinear_x2_bwd_kernel(const float* __restrict__ dy, int ld_dy, int B, int H, int W, int C, float* __restrict__ dx) {
    const int c4n = C >> 2, Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)B * H * W * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        long long pix = i / c4n;
        const int x = (int)(pix % W);
        long long t = pix / W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        float o[4] = {0, 0, 0, 0};
        for (int oy = 2 * y - 2; oy <= 2 * y + 3; ++oy) {
            if (oy < 0 || oy >= Ho) continue;
            int y0 = (oy >> 1) - ((oy & 1) ? 0 : 1);
            const float fy = (oy & 1) ? 0.25f : 0.75f;
            const int y1 = min(y0 + 1, H - 1);
            y0 = max(y0, 0);
            const float wy = (y0 == y ? 1.f - fy : 0.f) + (y1 == y ? fy : 0.f);
            if (wy == 0.f) continue;
            for (int ox = 2 * x - 2; ox <= 2 * x + 3; ++ox) {
                if (ox < 0 || ox >= Wo) continue;
                int x0 = (ox >> 1) - ((ox & 1) ? 0 : 1);
                const float fx = (ox & 1) ? 0.25f : 0.75f;
                const int x1 = min(x0 + 1, W - 1);
                x0 = max(x0, 0);
                const float wx = (x0 == x ? 1.f - fx : 0.f) + (x1 == x ? fx : 0.f);
                if (wx == 0.f) continue;
                const float4 g = *reinterpret_cast<const float4*>(dy + (((size_t)n * Ho + oy) * Wo + ox) * ld_dy + c4 * 4);
                const float ww = wy * wx;
                o[0] += ww * g.x; o[1] += ww * g.y; o[2] += ww * g.z; o[3] += ww * g.w;
            }
        }
        *reinterpret_cast<float4*>(dx + (size_t)pix * C + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// guided nearest x2: dx(y,x) = sum of dy over the hi-res pixels whose selection points at (y,x):
// candidates live in the low-res cells (y,x) [sel 0], (y,x-1) [sel 1], (y-1,x) [sel 2], (y-1,x-1) [sel 3]
__global__ void guided_x2_bwd_kernel(const float* __restrict__ dy, int ld_dy, const uint8_t* __restrict__ sel, int B, int H, int W, int C,
                                     float* __restrict__ dx) {
    const int c4n = C >> 2, Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)B * H * W * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        long long pix = i / c4n;
        const int x = (int)(pix % W);
        long long t = pix / W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        float o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cy = y - (s >> 1), cx = x - (s & 1);  // low-res cell whose sub-pixels may select (y,x) with index s
            if (cy < 0 || cx < 0) continue;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const int oy = 2 * cy + (sub >> 1), ox = 2 * cx + (sub & 1);
                const size_t hp = ((size_t)n * Ho + oy) * Wo + ox;
                if (sel[hp] == s) {
                    const float4 g = *reinterpret_cast<const float4*>(dy + hp * ld_dy + c4 * 4);
                    o[0] += g.x; o[1] += g.y; o[2] += g.z; o[3] += g.w;
                }
            }
        }
        *reinterpret_cast<float4*>(dx + (size_t)pix * C + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// dst[i] = idx[i] >= 0 ? src[idx[i]] : 0            (re-pack master weights into a kernel layout)
__global__ void gather_kernel(const float* __restrict__ src, const int* __restrict__ idx, long long n, float* __restrict__ dst) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = idx[i];
        dst[i] = j >= 0 ? src[j] : 0.f;
    }
}
// dst[idx[i]] (+)= src[i] for idx[i] >= 0              (packed weight gradient -> master layout; idx is injective)
__global__ void scatter_kernel(const float* __restrict__ src, const int* __restrict__ idx, long long n, float* __restrict__ dst, int accumulate) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = idx[i];
        if (j >= 0) dst[j] = accumulate ? dst[j] + src[i] : src[i];
    }
}

__global__ void axpby_kernel(const float* __restrict__ a, float alpha, const float* __restrict__ b, float beta, long long n, float* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}

// Keras Adam (tf.keras.optimizers.Adam, eps 1e-7): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t * m / (sqrt(v) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                            float lr_t, float b1, float b2, float eps, float grad_scale) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// batch statistics -> normalisation constants, folded affine tables and moving-average update, one launch
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double inv_n, int C, int real_c, int classes, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, int pad_one, float momentum, float* __restrict__ moving_mean,
                                   float* __restrict__ moving_var, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                   float* __restrict__ gamma_full, float* __restrict__ beta_full, float* __restrict__ scale, float* __restrict__ shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= classes * C) return;
    const int l = i / C, c = i - l * C;
    const double mean = sums[c] * inv_n;
    double var = sums[C + c] * inv_n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const bool real = c < real_c;
    const double g = (gamma && real) ? (double)gamma[l * real_c + c] : 1.0;
    const double b = (beta && real) ? (double)beta[l * real_c + c] : 0.0;
    double sc = rstd * g, sh = b - mean * sc;
    if (pad_one && !real) { sc = 0.0; sh = 1.0; }
    scale[i] = (float)sc;
    shift[i] = (float)sh;
    gamma_full[i] = (float)g;
    beta_full[i] = (float)b;
    if (l == 0) {
        mean_out[c] = (float)mean;
        rstd_out[c] = (float)rstd;
        if (moving_mean && real) {
            moving_mean[c] = moving_mean[c] * momentum + (float)mean * (1.f - momentum);
            moving_var[c] = moving_var[c] * momentum + (float)var * (1.f - momentum);
        }
    }
}

// red[(l*C+c)*2 + {0,1}] (fp64) -> d beta[l][c], d gamma[l][c] (fp32, real channels only)
__global__ void bn_param_grads_kernel(const double* __restrict__ red, int C, int real_c, int classes, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= classes * real_c) return;
    const int l = i / real_c, c = i - l * real_c;
    const double* r = red + ((size_t)l * C + c) * 2;
    if (dbeta) dbeta[i] = (float)r[0];
    if (dgamma) dgamma[i] = (float)r[1];
}

}  // namespace

extern "C" int cp_bn_stats_f32(const float* x, long long pixels, int channels, int ld, double* sums, void* stream) {
    CP_REQUIRE(x && sums && pixels > 0, "cp_bn_stats_f32: bad arguments");
    CP_REQUIRE(channels % 4 == 0 && channels >= 4 && channels <= 1024 && ld >= channels && ld % 4 == 0, "cp_bn_stats_f32: channels must be a multiple of 4 (<= 1024), ld >= channels");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * channels, st) != hipSuccess) return cp::check_launch("cp_bn_stats_f32 memset");
    const int rows_per_pass = STAT_THREADS / (channels / 4);
    long long blocks = (pixels + rows_per_pass * 16 - 1) / (rows_per_pass * 16);
    static const int cap = getenv("CP_BN_STATS_BLOCKS") ? atoi(getenv("CP_BN_STATS_BLOCKS")) : 512;   // tuning aid
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    CP_LAUNCH(bn_stats_kernel, dim3((unsigned)blocks), dim3(STAT_THREADS), sizeof(double) * 2 * channels, st, x, pixels, channels, ld, sums);
    return cp::check_launch("cp_bn_stats_f32");
}

extern "C" int cp_bn_finalize_f32(const double* sums, double pixels, int channels, int real_channels, int classes, const float* gamma,
                                  const float* beta, float eps, int pad_one, float momentum, float* moving_mean, float* moving_var, float* mean,
                                  float* rstd, float* gamma_full, float* beta_full, float* scale, float* shift, void* stream) {
    CP_REQUIRE(sums && mean && rstd && gamma_full && beta_full && scale && shift && pixels > 0, "cp_bn_finalize_f32: null pointer");
    CP_REQUIRE(channels > 0 && real_channels > 0 && real_channels <= channels && classes >= 1, "cp_bn_finalize_f32: bad sizes");
    CP_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "cp_bn_finalize_f32: moving_mean/moving_var come together");
    const int n = classes * channels;
    CP_LAUNCH(bn_finalize_kernel, dim3((n + THREADS - 1) / THREADS), dim3(THREADS), 0, (hipStream_t)stream, sums, 1.0 / pixels, channels, real_channels,
              classes, gamma, beta, eps, pad_one, momentum, moving_mean, moving_var, mean, rstd, gamma_full, beta_full, scale, shift);
    return cp::check_launch("cp_bn_finalize_f32");
}

extern "C" int cp_bn_param_grads_f32(const double* red, int channels, int real_channels, int classes, float* dgamma, float* dbeta, void* stream) {
    CP_REQUIRE(red && (dgamma || dbeta) && channels > 0 && real_channels > 0 && real_channels <= channels && classes >= 1, "cp_bn_param_grads_f32: bad arguments");
    const int n = classes * real_channels;
    CP_LAUNCH(bn_param_grads_kernel, dim3((n + THREADS - 1) / THREADS), dim3(THREADS), 0, (hipStream_t)stream, red, channels, real_channels, classes, dgamma, dbeta);
    return cp::check_launch("cp_bn_param_grads_f32");
}

extern "C" int cp_affine_act_f32(const float* x, long long pixels, int channels, int ld_x, const float* scale, const float* shift,
                                 const uint8_t* labels, int act, float* y, int ld_y, void* stream) {
    CP_REQUIRE(x && y && scale && shift && pixels > 0 && channels % 4 == 0 && ld_x >= channels && ld_y >= channels, "cp_affine_act_f32: bad arguments");
    CP_LAUNCH(affine_act_kernel, dim3(grid_for(pixels * (channels / 4))), dim3(THREADS), 0, (hipStream_t)stream, x, pixels, channels, ld_x, scale,
              shift, labels, act, y, ld_y);
    return cp::check_launch("cp_affine_act_f32");
}

extern "C" int cp_bn_act_bwd_reduce_f32(const float* x, int ld_x, const float* dy, int ld_dy, long long pixels, int channels, int classes,
                                        const float* mean, const float* rstd, const float* gamma, const float* beta, const uint8_t* labels,
                                        int act, const float* fwd_scale, const float* fwd_shift, double* red, double* chan, void* stream) {
    CP_REQUIRE(x && dy && mean && rstd && red && chan && pixels > 0, "cp_bn_act_bwd_reduce_f32: null pointer");
    CP_REQUIRE((fwd_scale == nullptr) == (fwd_shift == nullptr), "cp_bn_act_bwd_reduce_f32: fwd_scale and fwd_shift come together");
    CP_REQUIRE(channels % 4 == 0 && channels <= 1024 && classes >= 1 && classes <= 64, "cp_bn_act_bwd_reduce_f32: channels %% 4, classes <= 64");
    CP_REQUIRE(classes == 1 || labels, "cp_bn_act_bwd_reduce_f32: class-adaptive form needs labels");
    hipStream_t st = (hipStream_t)stream;
    const size_t nred = (size_t)classes * channels * 2, nch = (size_t)channels * 2;
    const size_t lds = (nred + nch) * sizeof(double);
    CP_REQUIRE(lds <= 150 * 1024, "cp_bn_act_bwd_reduce_f32: classes*channels too large for the LDS reduction (%zu bytes)", lds);
    if (chan == red + nred) {   // one launch for both tables when the caller keeps them adjacent (the training plan does: ~30 fewer fills per step)
        if (hipMemsetAsync(red, 0, (nred + nch) * sizeof(double), st) != hipSuccess) return cp::check_launch("cp_bn_act_bwd_reduce_f32 memset");
    } else {
        if (hipMemsetAsync(red, 0, nred * sizeof(double), st) != hipSuccess) return cp::check_launch("cp_bn_act_bwd_reduce_f32 memset");
        if (hipMemsetAsync(chan, 0, nch * sizeof(double), st) != hipSuccess) return cp::check_launch("cp_bn_act_bwd_reduce_f32 memset");
    }
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_act_bwd_reduce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr = true;
    }
    const int rows_per_pass = RED_THREADS / (channels / 4);
    long long blocks = (pixels + rows_per_pass * 32 - 1) / (rows_per_pass * 32);
    static const int cap = getenv("CP_BN_REDUCE_BLOCKS") ? atoi(getenv("CP_BN_REDUCE_BLOCKS")) : 1024;   // tuning aid
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    CP_LAUNCH(bn_act_bwd_reduce_kernel, dim3((unsigned)blocks), dim3(RED_THREADS), lds, st, x, ld_x, dy, ld_dy, pixels, channels, classes, mean, rstd,
              gamma, beta, labels, act, fwd_scale, fwd_shift, red, chan);
    return cp::check_launch("cp_bn_act_bwd_reduce_f32");
}

extern "C" int cp_bn_act_bwd_apply_f32(const float* x, int ld_x, const float* dy, int ld_dy, long long pixels, int channels, const float* mean,
                                       const float* rstd, const float* gamma, const float* beta, const uint8_t* labels, int act,
                                       const float* fwd_scale, const float* fwd_shift,
                                       const double* chan, double global_pixels, const float* row_scale, float* dx, int ld_dx, int accumulate, void* stream) {
    CP_REQUIRE(x && dy && mean && rstd && chan && dx && pixels > 0 && channels % 4 == 0 && global_pixels > 0, "cp_bn_act_bwd_apply_f32: bad arguments");
    CP_REQUIRE((fwd_scale == nullptr) == (fwd_shift == nullptr), "cp_bn_act_bwd_apply_f32: fwd_scale and fwd_shift come together");
    CP_LAUNCH(bn_act_bwd_apply_kernel, dim3(grid_for(pixels * (channels / 4))), dim3(THREADS), 0, (hipStream_t)stream, x, ld_x, dy, ld_dy, pixels,
              channels, mean, rstd, gamma, beta, labels, act, fwd_scale, fwd_shift, chan, 1.0 / global_pixels, row_scale, dx, ld_dx, accumulate);
    return cp::check_launch("cp_bn_act_bwd_apply_f32");
}

extern "C" int cp_maxpool3x3s2_bwd_f32(const float* x, const float* dy, int batch, int h, int w, int channels, float* dx, int accumulate, void* stream) {
    CP_REQUIRE(x && dy && dx && batch > 0 && h > 0 && w > 0 && channels % 4 == 0, "cp_maxpool3x3s2_bwd_f32: bad arguments");
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    CP_LAUNCH(maxpool_bwd_kernel, dim3(grid_for((long long)batch * ((h + 1) / 2) * ((w + 1) / 2) * (channels / 4))), dim3(THREADS), 0, (hipStream_t)stream, x, dy,
              batch, h, w, channels, ho, wo, dx, accumulate);
    return cp::check_launch("cp_maxpool3x3s2_bwd_f32");
}

extern "C" int cp_maxpool3x3s2_idx_f32(const float* src, int batch, int h, int w, int channels, float* dst, uint8_t* idx, void* stream) {
    CP_REQUIRE(src && dst && idx && batch > 0 && h > 0 && w > 0 && channels % 4 == 0 && ((uintptr_t)idx & 3) == 0, "cp_maxpool3x3s2_idx_f32: bad arguments");
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    CP_LAUNCH(maxpool_idx_kernel, dim3(grid_for((long long)batch * ho * wo * (channels / 4))), dim3(THREADS), 0, (hipStream_t)stream, src, batch, h, w, channels,
              ho, wo, dst, idx);
    return cp::check_launch("cp_maxpool3x3s2_idx_f32");
}

extern "C" int cp_maxpool3x3s2_bwd_idx_f32(const uint8_t* idx, const float* dy, int batch, int h, int w, int channels, float* dx, int accumulate,
                                           void* stream) {
    CP_REQUIRE(idx && dy && dx && batch > 0 && h > 0 && w > 0 && channels % 4 == 0 && ((uintptr_t)idx & 3) == 0, "cp_maxpool3x3s2_bwd_idx_f32: bad arguments");
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    CP_LAUNCH(maxpool_bwd_idx_kernel, dim3(grid_for((long long)batch * ((h + 1) / 2) * ((w + 1) / 2) * (channels / 4))), dim3(THREADS), 0, (hipStream_t)stream, idx,
              dy, batch, h, w, channels, ho, wo, dx, accumulate);
    return cp::check_launch("cp_maxpool3x3s2_bwd_idx_f32");
}

extern "C" int cp_upsample_bilinear_x2_bwd_f32(const float* dy, int ld_dy, int batch, int h, int w, int channels, float* dx, void* stream) {
    CP_REQUIRE(dy && dx && batch > 0 && h > 0 && w > 0 && channels % 4 == 0 && ld_dy >= channels, "cp_upsample_bilinear_x2_bwd_f32: bad arguments");
    CP_LAUNCH(bilinear_x2_bwd_kernel, dim3(grid_for((long long)batch * h * w * (channels / 4))), dim3(THREADS), 0, (hipStream_t)stream, dy, ld_dy, batch,
              h, w, channels, dx);
    return cp::check_launch("cp_upsample_bilinear_x2_bwd_f32");
}

extern "C" int cp_guided_upsample_x2_bwd_f32(const float* dy, int ld_dy, const uint8_t* sel, int batch, int h, int w, int channels, float* dx,
                                             void* stream) {
    CP_REQUIRE(dy && sel && dx && batch > 0 && h > 0 && w > 0 && channels % 4 == 0 && ld_dy >= channels, "cp_guided_upsample_x2_bwd_f32: bad arguments");
    CP_LAUNCH(guided_x2_bwd_kernel, dim3(grid_for((long long)batch * h * w * (channels / 4))), dim3(THREADS), 0, (hipStream_t)stream, dy, ld_dy, sel,
              batch, h, w, channels, dx);
    return cp::check_launch("cp_guided_upsample_x2_bwd_f32");
}

extern "C" int cp_gather_f32(const float* src, const int32_t* idx, long long n, float* dst, void* stream) {
    CP_REQUIRE(src && idx && dst && n > 0, "cp_gather_f32: bad arguments");
    CP_LAUNCH(gather_kernel, dim3(grid_for(n)), dim3(THREADS), 0, (hipStream_t)stream, src, idx, n, dst);
    return cp::check_launch("cp_gather_f32");
}

extern "C" int cp_scatter_f32(const float* src, const int32_t* idx, long long n, float* dst, int accumulate, void* stream) {
    CP_REQUIRE(src && idx && dst && n > 0, "cp_scatter_f32: bad arguments");
    CP_LAUNCH(scatter_kernel, dim3(grid_for(n)), dim3(THREADS), 0, (hipStream_t)stream, src, idx, n, dst, accumulate);
    return cp::check_launch("cp_scatter_f32");
}

extern "C" int cp_axpby_f32(const float* a, float alpha, const float* b, float beta, long long n, float* out, void* stream) {
    CP_REQUIRE(a && out && n > 0, "cp_axpby_f32: bad arguments");
    CP_LAUNCH(axpby_kernel, dim3(grid_for(n)), dim3(THREADS), 0, (hipStream_t)stream, a, alpha, b, beta, n, out);
    return cp::check_launch("cp_axpby_f32");
}

extern "C" int cp_adam_step_f32(float* params, const float* grads, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                                int step, float grad_scale, void* stream) {
    CP_REQUIRE(params && grads && m && v && n > 0 && step >= 1, "cp_adam_step_f32: bad arguments");
    const double c1 = 1.0 - pow((double)beta1, (double)step), c2 = 1.0 - pow((double)beta2, (double)step);
    const float lr_t = (float)(lr * sqrt(c2) / c1);
    CP_LAUNCH(adam_kernel, dim3(grid_for(n)), dim3(THREADS), 0, (hipStream_t)stream, params, grads, m, v, n, lr_t, beta1, beta2, eps, grad_scale);
    return cp::check_launch("cp_adam_step_f32");
}
