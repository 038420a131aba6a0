// Confidence-weighted least-squares keypoint voting (CoordLSVotingWeighted.calc,
// casapose/pose_estimation/voting_layers_2d.py:83-122) as one streaming reduction.
//
// The reference broadcasts R = w(I - n n^T) to [B,H,W,objects,kp,2,2] and reduces in fp64;
// here every pixel is read exactly once (algorithmic bytes = H*W*ld*4 per image) and only
// the five distinct sums per (image, object, keypoint) are kept:
//     S00 = sum w(1-ny^2)   S01 = sum -w ny nx   S11 = sum w(1-nx^2)
//     T0  = sum (R c)_y     T1  = sum (R c)_x          c = ((y+.5)/H, (x+.5)/H)
// Per-pixel terms are formed in fp32 exactly as the reference does (:89-105) and
// accumulated in fp64 (:113-114).
//
// Work decomposition: a wave owns a strip of 64 columns x ROWS rows; lane l walks DOWN
// column x0+l, so the 64 lanes of each step read 64 consecutive pixels (one contiguous
// 64*ld*4-byte span, staged through LDS with 16-byte accesses) while each lane's object
// label stays constant for long runs.  A lane accumulates privately in fp64 registers and
// only flushes (LDS fp64 atomics) when its label changes or the strip ends; the block then
// adds its LDS table to the global fp64 sums with one atomic per entry.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int ROWS = 16;        // rows per strip
constexpr int MAXKP = 9;        // compile-time bound for the private accumulators
constexpr int WAVES = 4;

// softplus = max(x,0) + log1p(exp(-|x|)) on the hardware exp2 / log2 units: with t = exp(-|x|) in (0,1] and u = fl(1 + t),
// log1p(t) = log(u) + (t - (u - 1)) / u restores the bits the addition drops (error ~1e-7 of the result)
__device__ __forceinline__ float softplus_fast(float x) {
    const float t = __expf(-fabsf(x));
    const float u = 1.0f + t;
    return fmaxf(x, 0.f) + (__logf(u) + (t - (u - 1.0f)) * __frcp_rn(u));
}

template <int KP>
__global__ __launch_bounds__(256) void ls_accumulate_kernel(const float* __restrict__ field, int ld, int seg_off,
                                                            int dir_off, int conf_off, const uint8_t* __restrict__ labels,
                                                            int B, int H, int W, int objects, double* __restrict__ sums,
                                                            int strips_x, int strips_y, int sigmoid) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* acc_lds = reinterpret_cast<double*>(smem_raw);                         // [objects][KP][5]
    float* stage = reinterpret_cast<float*>(smem_raw + sizeof(double) * objects * KP * 5);  // [WAVES][64*ld]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nacc = objects * KP * 5;
    for (int i = tid; i < nacc; i += blockDim.x) acc_lds[i] = 0.0;
    __syncthreads();

    // strip id -> (image, strip row, strip column); all 4 waves of a block work on ONE image
    const int strips_per_img = strips_x * strips_y;
    const int blocks_per_img = (strips_per_img + WAVES - 1) / WAVES;
    const int img = blockIdx.x / blocks_per_img;
    const int sidx = (blockIdx.x % blocks_per_img) * WAVES + wave;
    float* wstage = stage + (size_t)wave * 64 * ld;

    if (sidx < strips_per_img) {
        const int sy = sidx / strips_x, sx = sidx % strips_x;
        const int x0 = sx * 64, y0 = sy * ROWS;
        const int x = x0 + lane;
        const int ncols = min(64, W - x0);
        const float invH = 1.0f;  // divide like the reference: (v + 0.5) / H in fp32
        (void)invH;
        const float cx = ((float)x + 0.5f) / (float)H;
        const int classes = objects + 1;

        double a[KP][5];
#pragma unroll
        for (int j = 0; j < KP; ++j)
#pragma unroll
            for (int c = 0; c < 5; ++c) a[j][c] = 0.0;
        int cur = 0;

        auto flush = [&]() {
            if (cur > 0) {
                double* dst = acc_lds + (size_t)(cur - 1) * KP * 5;
#pragma unroll
                for (int j = 0; j < KP; ++j)
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        atomicAdd(dst + j * 5 + c, a[j][c]);
                        a[j][c] = 0.0;
                    }
            }
        };

        const int y_end = min(y0 + ROWS, H);
        for (int y = y0; y < y_end; ++y) {
            // ---- stage this row segment: ncols*ld contiguous floats, 16 B per lane per step ----
            const float* g = field + (((size_t)img * H + y) * W + x0) * ld;
            const int nvec = (ncols * ld) >> 2;  // ld % 4 == 0
            for (int v = lane; v < nvec; v += 64)
                reinterpret_cast<float4*>(wstage)[v] = reinterpret_cast<const float4*>(g)[v];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes landed before the reads below
            int lab = 0;
            const float* px = wstage + lane * ld;
            if (lane < ncols) {
                if (labels) {
                    lab = labels[((size_t)img * H + y) * W + x];
                } else {
                    float best = px[seg_off];
                    for (int k = 1; k < classes; ++k) {
                        float v = px[seg_off + k];
                        if (v > best) { best = v; lab = k; }
                    }
                }
            }
            if (lab != cur) {
                flush();
                cur = lab;
            }
#ifdef LS_NOCOMPUTE
            if (lab > 250) {
#else
            if (lab > 0) {
#endif
                const float cy = ((float)y + 0.5f) / (float)H;
#pragma unroll
                for (int j = 0; j < KP; ++j) {
                    float dy = px[dir_off + 2 * j], dx = px[dir_off + 2 * j + 1];
                    float cf = px[conf_off + j];
#ifdef LS_NOSOFTPLUS
                    float w = fmaxf(cf, 0.f);
#else
                    float w = sigmoid ? 1.0f / (1.0f + expf(-cf))                    // sigmoid_weights=True (:32-33, sigmoid_scale = 1)
                                      : fmaxf(cf, 0.f) + log1pf(expf(-fabsf(cf)));   // softplus (:35)
#endif
                    float nrm = sqrtf(dy * dy + dx * dx);
                    float ny = (nrm > 0.f) ? dy / nrm : 0.f;  // divide_no_nan (:90)
                    float nx = (nrm > 0.f) ? dx / nrm : 0.f;
                    float r00 = (1.0f - ny * ny) * w;
                    float r01 = (0.0f - ny * nx) * w;
                    float r11 = (1.0f - nx * nx) * w;
                    float q0 = r00 * cy + r01 * cx;  // (:103-105)
                    float q1 = r01 * cy + r11 * cx;
                    a[j][0] += (double)r00;
                    a[j][1] += (double)r01;
                    a[j][2] += (double)r11;
                    a[j][3] += (double)q0;
                    a[j][4] += (double)q1;
                }
            }
            __builtin_amdgcn_wave_barrier();  // all lanes done reading before the next row overwrites
        }
        flush();
    }
    __syncthreads();
    double* gdst = sums + (size_t)img * nacc;
    for (int i = tid; i < nacc; i += blockDim.x) {
        double v = acc_lds[i];
        if (v != 0.0) atomicAdd(gdst + i, v);
    }
}

// Fast path for the production record [9 logits | 18 directions | 9 confidences] (ld = 36, 8 objects, 9 keypoints): same strip
// decomposition and arithmetic as above, but
//   * the next row segment (64 pixels = 9216 contiguous bytes = nine 16-byte loads per lane) is fetched into registers while the
//     current one is processed -- the generic kernel waits for every row's memory latency (measured 3.1 TB/s with the arithmetic
//     removed); and
//   * a lane reads its pixel back from the staging buffer as nine 16-byte LDS reads (144-byte pixel stride: conflict-free) with
//     compile-time channel positions, instead of 27 scalar reads with 4-way bank conflicts.
__global__ __launch_bounds__(256) void ls_accumulate36_kernel(const float* __restrict__ field, const uint8_t* __restrict__ labels, int B, int H,
                                                              int W, double* __restrict__ sums, int strips_x, int strips_y) {
    constexpr int KP = 9, LD = 36, OBJ = 8, NV = LD / 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* acc_lds = reinterpret_cast<double*>(smem_raw);                                // [OBJ][KP][5]
    float4* stage = reinterpret_cast<float4*>(smem_raw + sizeof(double) * OBJ * KP * 5);   // [WAVES][64*NV]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nacc = OBJ * KP * 5;
    for (int i = tid; i < nacc; i += blockDim.x) acc_lds[i] = 0.0;
    __syncthreads();

    const int strips_per_img = strips_x * strips_y;
    const int blocks_per_img = (strips_per_img + WAVES - 1) / WAVES;
    const int img = blockIdx.x / blocks_per_img;
    const int sidx = (blockIdx.x % blocks_per_img) * WAVES + wave;
    float4* wstage = stage + (size_t)wave * 64 * NV;

    if (sidx < strips_per_img) {
        const int sy = sidx / strips_x, sx = sidx % strips_x;
        const int x0 = sx * 64, y0 = sy * ROWS;
        const int x = x0 + lane;
        const int ncols = min(64, W - x0);
        const int nvec = ncols * NV;
        const float cx = ((float)x + 0.5f) / (float)H;

        double a[KP][5];
#pragma unroll
        for (int j = 0; j < KP; ++j)
#pragma unroll
            for (int c = 0; c < 5; ++c) a[j][c] = 0.0;
        int cur = 0;
        auto flush = [&]() {
            if (cur > 0) {
                double* dst = acc_lds + (size_t)(cur - 1) * KP * 5;
#pragma unroll
                for (int j = 0; j < KP; ++j)
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        atomicAdd(dst + j * 5 + c, a[j][c]);
                        a[j][c] = 0.0;
                    }
            }
        };

        float4 pre[NV];
        int lab_pre = 0;
        auto issue = [&](int y) {
            const float4* g = reinterpret_cast<const float4*>(field + (((size_t)img * H + y) * W + x0) * LD);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int v = lane + 64 * i;
                pre[i] = (v < nvec) ? g[v] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (labels) lab_pre = (lane < ncols) ? (int)labels[((size_t)img * H + y) * W + x] : 0;
        };
        const int y_end = min(y0 + ROWS, H);
        issue(y0);
        for (int y = y0; y < y_end; ++y) {
#pragma unroll
            for (int i = 0; i < NV; ++i) wstage[lane + 64 * i] = pre[i];
            int lab = lab_pre;
            if (y + 1 < y_end) issue(y + 1);  // in flight while this row is processed
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes have landed (the prefetch stays in flight)
            float r[LD];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float4 q = wstage[lane * NV + i];
                r[4 * i] = q.x; r[4 * i + 1] = q.y; r[4 * i + 2] = q.z; r[4 * i + 3] = q.w;
            }
            if (!labels) {
                lab = 0;
                float best = r[0];
#pragma unroll
                for (int k = 1; k <= OBJ; ++k)
                    if (r[k] > best) { best = r[k]; lab = k; }
                if (lane >= ncols) lab = 0;
            }
            if (lab != cur) {
                flush();
                cur = lab;
            }
#ifdef LS_NOCOMPUTE
            if (lab > 250) {
#else
            if (lab > 0) {
#endif
                const float cy = ((float)y + 0.5f) / (float)H;
#pragma unroll
                for (int j = 0; j < KP; ++j) {
                    const float dy = r[9 + 2 * j], dx = r[9 + 2 * j + 1];
                    const float cf = r[27 + j];
                    // softplus (:35) = max(x,0) + log1p(exp(-|x|)) on the hardware exp2 / log2 units: t = exp(-|x|) in (0,1], and
                    // log1p(t) = log(u) + (t - (u - 1)) / u with u = fl(1 + t) restores the bits the addition drops (error ~1e-7 of w;
                    // the libm calls of the generic kernel cost as much as the whole memory stream)
                    const float w = softplus_fast(cf);
                    const float n2 = dy * dy + dx * dx;
                    const float inv = (n2 > 0.f) ? __frsqrt_rn(n2) : 0.f;  // divide_no_nan (:90); one reciprocal square root for both components
                    const float ny = dy * inv, nx = dx * inv;
                    const float r00 = (1.0f - ny * ny) * w;
                    const float r01 = (0.0f - ny * nx) * w;
                    const float r11 = (1.0f - nx * nx) * w;
                    const float q0 = r00 * cy + r01 * cx;  // (:103-105)
                    const float q1 = r01 * cy + r11 * cx;
                    a[j][0] += (double)r00;
                    a[j][1] += (double)r01;
                    a[j][2] += (double)r11;
                    a[j][3] += (double)q0;
                    a[j][4] += (double)q1;
                }
            }
            __builtin_amdgcn_wave_barrier();  // all lanes done reading before the next row overwrites
        }
        flush();
    }
    __syncthreads();
    double* gdst = sums + (size_t)img * nacc;
    for (int i = tid; i < nacc; i += blockDim.x) {
        double v = acc_lds[i];
        if (v != 0.0) atomicAdd(gdst + i, v);
    }
}

// p = pinv([[S00,S01],[S01,S11]]) [T0,T1]^T * H   (voting_layers_2d.py:116-122);
// tf.linalg.pinv default rcond = 10 * max(m,n) * eps(fp64).
__global__ void ls_solve_kernel(const double* __restrict__ sums, int total, int H, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double* s = sums + (size_t)i * 5;
    double a = s[0], b = s[1], c = s[2], t0 = s[3], t1 = s[4];
    double half_tr = 0.5 * (a + c), half_df = 0.5 * (a - c);
    double rad = sqrt(half_df * half_df + b * b);
    double l1 = half_tr + rad, l2 = half_tr - rad;  // l1 >= l2 (PSD up to rounding)
    const double rcond = 10.0 * 2.0 * 2.220446049250313e-16;
    double p0 = 0.0, p1 = 0.0;
    double smax = fmax(fabs(l1), fabs(l2));
    if (smax > 0.0) {
        if (fabs(l2) > rcond * smax && fabs(l1) > rcond * smax) {
            double det = a * c - b * b;
            p0 = (c * t0 - b * t1) / det;
            p1 = (a * t1 - b * t0) / det;
        } else {
            // rank one: keep only the dominant eigen-pair
            double lam = (fabs(l1) >= fabs(l2)) ? l1 : l2;
            double vx = b, vy = lam - a;          // (A - a I) v: eigenvector candidates
            double wx = lam - c, wy = b;
            if (wx * wx + wy * wy > vx * vx + vy * vy) { vx = wx; vy = wy; }
            double n2 = vx * vx + vy * vy;
            if (n2 > 0.0) {
                double proj = (vx * t0 + vy * t1) / (n2 * lam);
                p0 = vx * proj;
                p1 = vy * proj;
            }
        }
    }
    out[2 * i] = (float)p0 * (float)H;
    out[2 * i + 1] = (float)p1 * (float)H;
}

// ---------------------------------------------------------------------------------------------
// backward of the voter (the reference differentiates CoordLSVotingWeighted with tf.GradientTape,
// train_casapose.py:555-579,594): with A = sum R, b = sum R c, p = A^-1 b and g = dL/dp,
//   u = A^-1 g,  dL/dR_i = u (c_i - p)^T  for every pixel i of the object, R_i = w_i (I - n_i n_i^T):
//   dL/dw_i = u.e - (u.n)(n.e)                 e = c_i - p
//   dL/dn_i = -w_i (u (e.n) + e (u.n))
//   dL/dd_i = (dL/dn - n (n.dL/dn)) / |d|      n = d/|d|
//   dL/dconf_i = dL/dw_i * sigmoid(conf_i)      w = softplus(conf)
// No reduction: one pass over the pixels given the per-keypoint pair (p, u).
__global__ void ls_bwd_prepare_kernel(const double* __restrict__ sums, const float* __restrict__ dkp, int total, int H, float* __restrict__ pu) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double* s = sums + (size_t)i * 5;
    const double a = s[0], b = s[1], c = s[2], t0 = s[3], t1 = s[4];
    const double g0 = (double)dkp[2 * i] * H, g1 = (double)dkp[2 * i + 1] * H;  // keypoints = p * H
    const double det = a * c - b * b;
    const double tr = a + c;
    double p0 = 0, p1 = 0, u0 = 0, u1 = 0;
    if (tr > 0.0 && fabs(det) > 1e-12 * tr * tr) {  // regular system; a degenerate one passes no gradient
        p0 = (c * t0 - b * t1) / det;
        p1 = (a * t1 - b * t0) / det;
        u0 = (c * g0 - b * g1) / det;
        u1 = (a * g1 - b * g0) / det;
    }
    pu[4 * i] = (float)p0; pu[4 * i + 1] = (float)p1; pu[4 * i + 2] = (float)u0; pu[4 * i + 3] = (float)u1;
}

template <int KP>
__global__ void ls_bwd_kernel(const float* __restrict__ field, int ld, int dir_off, int conf_off, const uint8_t* __restrict__ labels, int B, int H,
                              int W, int objects, const float* __restrict__ pu, const uint8_t* __restrict__ reg_labels,
                              const float* __restrict__ conf_coef, float* __restrict__ dfield, int dld, int ddir_off, int dconf_off,
                              int accumulate) {
    const long long total = (long long)B * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int lab = labels[i];
        const int rlab = reg_labels ? reg_labels[i] : 0;
        float* g = dfield + i * dld;
        if (lab == 0 && rlab == 0) {
            if (!accumulate) {
                for (int j = 0; j < 2 * KP; ++j) g[ddir_off + j] = 0.f;
                for (int j = 0; j < KP; ++j) g[dconf_off + j] = 0.f;
            }
            continue;
        }
        const int x = (int)(i % W);
        const long long t = i / W;
        const int y = (int)(t % H), b = (int)(t / H);
        const float cy = ((float)y + 0.5f) / (float)H, cx = ((float)x + 0.5f) / (float)H;
        const float* px = field + i * ld;
        const float* tab = pu + ((size_t)b * objects + (lab > 0 ? lab - 1 : 0)) * KP * 4;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const float cf = px[conf_off + j];
            const float sg = 1.f / (1.f + __expf(-cf));
            float gdy = 0.f, gdx = 0.f, gcf = 0.f;
            if (lab > 0) {
                const float dy = px[dir_off + 2 * j], dx = px[dir_off + 2 * j + 1];
                const float w = softplus_fast(cf);
                const float nrm = sqrtf(dy * dy + dx * dx);
                const float p0 = tab[4 * j], p1 = tab[4 * j + 1], u0 = tab[4 * j + 2], u1 = tab[4 * j + 3];
                const float e0 = cy - p0, e1 = cx - p1;
                if (nrm > 0.f) {
                    const float inr = 1.f / nrm;
                    const float ny = dy * inr, nx = dx * inr;
                    const float un = u0 * ny + u1 * nx, en = e0 * ny + e1 * nx, ue = u0 * e0 + u1 * e1;
                    gcf = (ue - un * en) * sg;
                    const float ln0 = -w * (u0 * en + e0 * un), ln1 = -w * (u1 * en + e1 * un);
                    const float nl = ny * ln0 + nx * ln1;
                    gdy = (ln0 - ny * nl) * inr;
                    gdx = (ln1 - nx * nl) * inr;
                } else {
                    gcf = (u0 * e0 + u1 * e1) * sg;  // n = 0: R = w I
                }
            }
            if (rlab > 0 && conf_coef) gcf += conf_coef[b * KP + j] * sg;
            if (accumulate) {
                g[ddir_off + 2 * j] += gdy;
                g[ddir_off + 2 * j + 1] += gdx;
                g[dconf_off + j] += gcf;
            } else {
                g[ddir_off + 2 * j] = gdy;
                g[ddir_off + 2 * j + 1] = gdx;
                g[dconf_off + j] = gcf;
            }
        }
    }
}

// The production record of ls_bwd_kernel (ld = 36: [9 logits | 18 directions | 9 confidences], gradient rows whose 27 values start on a 16-byte
// boundary with one writable float of padding behind them): a pixel's record and gradient row move as nine / seven 16-byte accesses with
// compile-time channel positions.  The generic kernel's 4-byte accesses at a 144 / 256-byte lane stride took 0.98 ms per training step for a
// quarter of the pixels (the foreground); same arithmetic, same results.
__global__ void ls_bwd36_kernel(const float* __restrict__ field, const uint8_t* __restrict__ labels, int B, int H, int W, int objects,
                                const float* __restrict__ pu, const uint8_t* __restrict__ reg_labels, const float* __restrict__ conf_coef,
                                float* __restrict__ dfield, int dld, int ddir_off, int accumulate) {
    constexpr int KP = 9;
    const long long total = (long long)B * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int lab = labels[i];
        const int rlab = reg_labels ? reg_labels[i] : 0;
        float4* g4 = reinterpret_cast<float4*>(dfield + i * dld + ddir_off);
        if (lab == 0 && rlab == 0) {
            if (!accumulate) {
#pragma unroll
                for (int q = 0; q < 7; ++q) g4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            continue;
        }
        const int x = (int)(i % W);
        const long long t = i / W;
        const int y = (int)(t % H), b = (int)(t / H);
        const float cy = ((float)y + 0.5f) / (float)H, cx = ((float)x + 0.5f) / (float)H;
        float rec[36];
        const float4* r4 = reinterpret_cast<const float4*>(field + i * 36);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const float4 v = r4[q];
            rec[4 * q] = v.x; rec[4 * q + 1] = v.y; rec[4 * q + 2] = v.z; rec[4 * q + 3] = v.w;
        }
        float out[28];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const float4 v = accumulate ? g4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
        }
        const float* tab = pu + ((size_t)b * objects + (lab > 0 ? lab - 1 : 0)) * KP * 4;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const float cf = rec[27 + j];
            const float sg = 1.f / (1.f + __expf(-cf));
            float gdy = 0.f, gdx = 0.f, gcf = 0.f;
            if (lab > 0) {
                const float dy = rec[9 + 2 * j], dx = rec[9 + 2 * j + 1];
                const float w = softplus_fast(cf);
                const float nrm = sqrtf(dy * dy + dx * dx);
                const float4 pv = *reinterpret_cast<const float4*>(tab + 4 * j);
                const float p0 = pv.x, p1 = pv.y, u0 = pv.z, u1 = pv.w;
                const float e0 = cy - p0, e1 = cx - p1;
                if (nrm > 0.f) {
                    const float inr = 1.f / nrm;
                    const float ny = dy * inr, nx = dx * inr;
                    const float un = u0 * ny + u1 * nx, en = e0 * ny + e1 * nx, ue = u0 * e0 + u1 * e1;
                    gcf = (ue - un * en) * sg;
                    const float ln0 = -w * (u0 * en + e0 * un), ln1 = -w * (u1 * en + e1 * un);
                    const float nl = ny * ln0 + nx * ln1;
                    gdy = (ln0 - ny * nl) * inr;
                    gdx = (ln1 - nx * nl) * inr;
                } else {
                    gcf = (u0 * e0 + u1 * e1) * sg;  // n = 0: R = w I
                }
            }
            if (rlab > 0 && conf_coef) gcf += conf_coef[b * KP + j] * sg;
            out[2 * j] += gdy;
            out[2 * j + 1] += gdx;
            out[18 + j] += gcf;
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) g4[q] = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
    }
}

// per image: counts of every class in two label maps and, over the foreground of `labels`, the sums of softplus(conf_j)
// (objects_available and the confidence regulariser of keypoint_reprojection_loss, loss_functions.py:236-262)
template <int KP>
__global__ __launch_bounds__(256) void kp_stats_kernel(const float* __restrict__ field, int ld, int conf_off, const uint8_t* __restrict__ labels,
                                                       const uint8_t* __restrict__ labels_est, int pix_per_img, int classes,
                                                       int* __restrict__ counts, double* __restrict__ conf_sums) {
    extern __shared__ int scount[];  // [2][classes]
    __shared__ double ssum[KP];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * classes; i += blockDim.x) scount[i] = 0;
    if (threadIdx.x < KP) ssum[threadIdx.x] = 0.0;
    __syncthreads();
    double cs[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) cs[j] = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pix_per_img; i += gridDim.x * blockDim.x) {
        const size_t p = (size_t)b * pix_per_img + i;
        const int l = labels[p];
        atomicAdd(&scount[l < classes ? l : 0], 1);
        if (labels_est) {
            const int le = labels_est[p];
            atomicAdd(&scount[classes + (le < classes ? le : 0)], 1);
        }
        if (l > 0) {
            const float* px = field + p * ld + conf_off;
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                const float cf = px[j];
                cs[j] += (double)softplus_fast(cf);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        double v = cs[j];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0 && v != 0.0) atomicAdd(&ssum[j], v);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * classes; i += blockDim.x)
        if (scount[i]) atomicAdd(&counts[((size_t)(i / classes) * gridDim.y + b) * classes + (i % classes)], scount[i]);
    if (threadIdx.x < KP && ssum[threadIdx.x] != 0.0) atomicAdd(&conf_sums[b * KP + threadIdx.x], ssum[threadIdx.x]);
}

// keypoint_reprojection_loss without BPnP (loss_functions.py:207-344): crop pixels -> original image through the per-image
// affine (transform_points_back_tf_batch, ransac_voting.py:124-158), distance to the projected ground-truth keypoints,
// smooth L1, soft cap at max_err, mean over keypoints, sum over available objects / their number.  One block.
__global__ void kp_reproj_loss_kernel(const float* __restrict__ coords_yx, const float* __restrict__ gt_xy, const float* __restrict__ affine,
                                      const float* __restrict__ avail, int batch, int objects, int kp, float max_err, float weight,
                                      float* __restrict__ g_yx, double* __restrict__ loss_out) {
    __shared__ double red[256];
    __shared__ double navail;
    const int N = batch * objects;
    if (threadIdx.x == 0) {
        double s = 0;
        for (int n = 0; n < N; ++n) s += avail[n];
        navail = s;
    }
    __syncthreads();
    const double na = navail;
    double local = 0.0;
    for (int i = threadIdx.x; i < N * kp; i += blockDim.x) {
        const int n = i / kp;
        const int b = n / objects;
        const float av = avail[n];
        const float* A = affine + b * 6;
        const float y = coords_yx[2 * i], x = coords_yx[2 * i + 1];
        const float X = A[0] * x + A[1] * y + A[2], Y = A[3] * x + A[4] * y + A[5];
        const float dx = (gt_xy[2 * i] - X) * av, dy = (gt_xy[2 * i + 1] - Y) * av;
        const float e = sqrtf(dx * dx + dy * dy);
        float l = e < 1.f ? 0.5f * e * e : e - 0.5f;
        float slope = e < 1.f ? e : 1.f;
        if (l > max_err) { l = max_err + (l - max_err) * 0.01f; slope *= 0.01f; }
        local += (double)(l * av);
        float gx = 0.f, gy = 0.f;
        if (e > 0.f && na > 0.0) {
            const float c = weight * slope * av * av / (e * (float)kp * (float)na);  // d loss / d (X,Y) = -c*(dx,dy)
            const float gX = -c * dx, gY = -c * dy;
            gx = gX * A[0] + gY * A[3];
            gy = gX * A[1] + gY * A[4];
        }
        g_yx[2 * i] = gy;
        g_yx[2 * i + 1] = gx;
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss_out = na > 0.0 ? red[0] / ((double)kp * na) : 0.0;
}

}  // namespace

extern "C" size_t cp_ls_vote_workspace_bytes(int batch, int objects, int kp) {
    return (size_t)batch * objects * kp * 5 * sizeof(double);
}

extern "C" int cp_ls_vote_w_f32(const float* field, int ld, int seg_off, int dir_off, int conf_off, const uint8_t* labels, int batch, int h, int w,
                                int objects, int kp, int sigmoid_weights, double* sums_ws, float* keypoints, void* stream);

extern "C" int cp_ls_vote_f32(const float* field, int ld, int seg_off, int dir_off, int conf_off, const uint8_t* labels,
                              int batch, int h, int w, int objects, int kp, double* sums_ws, float* keypoints,
                              void* stream) {
    return cp_ls_vote_w_f32(field, ld, seg_off, dir_off, conf_off, labels, batch, h, w, objects, kp, 0, sums_ws, keypoints, stream);
}

// the same voter with the pixel weight selectable: sigmoid_weights = 0 softplus(conf) (the configs' choice), 1 sigmoid(conf)
// (CoordLSVotingWeighted(sigmoid_weights=True), voting_layers_2d.py:32-33)
extern "C" int cp_ls_vote_w_f32(const float* field, int ld, int seg_off, int dir_off, int conf_off, const uint8_t* labels, int batch, int h, int w,
                                int objects, int kp, int sigmoid_weights, double* sums_ws, float* keypoints, void* stream) {
    CP_REQUIRE(field && sums_ws && keypoints, "cp_ls_vote_f32: null pointer");
    CP_REQUIRE(batch > 0 && h > 0 && w > 0 && objects > 0 && objects < 255, "cp_ls_vote_f32: bad sizes");
    CP_REQUIRE(kp == MAXKP, "cp_ls_vote_f32: built for %d keypoints (got %d)", MAXKP, kp);
    CP_REQUIRE(ld % 4 == 0 && ld <= 64 && ((uintptr_t)field & 15) == 0, "cp_ls_vote_f32: ld must be a multiple of 4 (<= 64) and field 16-byte aligned");
    CP_REQUIRE(seg_off >= 0 && seg_off + objects + 1 <= ld && dir_off >= 0 && dir_off + 2 * kp <= ld && conf_off >= 0 && conf_off + kp <= ld,
               "cp_ls_vote_f32: channel offsets outside the pixel record");
    hipStream_t st = (hipStream_t)stream;
    size_t nbytes = cp_ls_vote_workspace_bytes(batch, objects, kp);
    if (hipMemsetAsync(sums_ws, 0, nbytes, st) != hipSuccess) return cp::check_launch("cp_ls_vote_f32 memset");
    int strips_x = (w + 63) / 64, strips_y = (h + ROWS - 1) / ROWS;
    int blocks_per_img = (strips_x * strips_y + WAVES - 1) / WAVES;
    size_t lds = sizeof(double) * objects * kp * 5 + sizeof(float) * WAVES * 64 * ld;
    if (!sigmoid_weights && ld == 36 && seg_off == 0 && dir_off == 9 && conf_off == 27 && objects == 8 && !getenv("CP_LS_GENERIC")) {  // the production record
        CP_LAUNCH(ls_accumulate36_kernel, dim3(batch * blocks_per_img), dim3(256), lds, st, field, labels, batch, h, w, sums_ws, strips_x, strips_y);
    } else {
        CP_LAUNCH((ls_accumulate_kernel<MAXKP>), dim3(batch * blocks_per_img), dim3(256), lds, st, field, ld, seg_off,
                  dir_off, conf_off, labels, batch, h, w, objects, sums_ws, strips_x, strips_y, sigmoid_weights ? 1 : 0);
    }
    int total = batch * objects * kp;
    CP_LAUNCH(ls_solve_kernel, dim3((total + 255) / 256), dim3(256), 0, st, sums_ws, total, h, keypoints);
    return cp::check_launch("cp_ls_vote_f32");
}


extern "C" int cp_ls_vote_bwd_f32(const float* field, int ld, int dir_off, int conf_off, const uint8_t* labels, int batch, int h, int w,
                                  int objects, int kp, const double* sums_ws, const float* dkeypoints, float* pu_ws,
                                  const uint8_t* reg_labels, const float* conf_coef, float* dfield, int dld, int ddir_off, int dconf_off,
                                  int accumulate, void* stream) {
    CP_REQUIRE(field && labels && sums_ws && dkeypoints && pu_ws && dfield, "cp_ls_vote_bwd_f32: null pointer");
    CP_REQUIRE(kp == MAXKP, "cp_ls_vote_bwd_f32: built for %d keypoints (got %d)", MAXKP, kp);
    CP_REQUIRE(batch > 0 && h > 0 && w > 0 && objects > 0 && objects < 255, "cp_ls_vote_bwd_f32: bad sizes");
    CP_REQUIRE(dir_off >= 0 && dir_off + 2 * kp <= ld && conf_off >= 0 && conf_off + kp <= ld, "cp_ls_vote_bwd_f32: channel offsets outside the pixel record");
    CP_REQUIRE(ddir_off >= 0 && ddir_off + 2 * kp <= dld && dconf_off >= 0 && dconf_off + kp <= dld, "cp_ls_vote_bwd_f32: gradient offsets outside the row");
    CP_REQUIRE((reg_labels == nullptr) == (conf_coef == nullptr), "cp_ls_vote_bwd_f32: reg_labels and conf_coef come together");
    hipStream_t st = (hipStream_t)stream;
    const int total = batch * objects * kp;
    CP_LAUNCH(ls_bwd_prepare_kernel, dim3((total + 255) / 256), dim3(256), 0, st, sums_ws, dkeypoints, total, h, pu_ws);
    if (cp::check_launch("cp_ls_vote_bwd_f32 prepare") != CP_OK) return CP_ERR_LAUNCH;
    const long long px = (long long)batch * h * w;
    long long blocks = (px + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const bool rec36 = ld == 36 && dir_off == 9 && conf_off == 27 && dld % 4 == 0 && ddir_off % 4 == 0 && dconf_off == ddir_off + 18 && ddir_off + 28 <= dld &&
                       (((uintptr_t)field | (uintptr_t)dfield | (uintptr_t)pu_ws) & 15) == 0 && !getenv("CP_LS_GENERIC");
    if (rec36)   // the production record; in overwrite mode the float behind the 27 gradient values (padding of the row) is zeroed with them
        CP_LAUNCH(ls_bwd36_kernel, dim3((unsigned)blocks), dim3(256), 0, st, field, labels, batch, h, w, objects, pu_ws, reg_labels, conf_coef, dfield, dld,
                  ddir_off, accumulate);
    else
        CP_LAUNCH((ls_bwd_kernel<MAXKP>), dim3((unsigned)blocks), dim3(256), 0, st, field, ld, dir_off, conf_off, labels, batch, h, w, objects, pu_ws, reg_labels,
                  conf_coef, dfield, dld, ddir_off, dconf_off, accumulate);
    return cp::check_launch("cp_ls_vote_bwd_f32");
}

extern "C" int cp_kp_stats_f32(const float* field, int ld, int conf_off, const uint8_t* labels, const uint8_t* labels_est, int batch, int h, int w,
                               int classes, int kp, int32_t* counts, double* conf_sums, void* stream) {
    CP_REQUIRE(field && labels && counts && conf_sums, "cp_kp_stats_f32: null pointer");
    CP_REQUIRE(kp == MAXKP && classes >= 2 && classes <= 256 && conf_off >= 0 && conf_off + kp <= ld, "cp_kp_stats_f32: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * 2 * batch * classes, st) != hipSuccess) return cp::check_launch("cp_kp_stats_f32 memset");
    if (hipMemsetAsync(conf_sums, 0, sizeof(double) * batch * kp, st) != hipSuccess) return cp::check_launch("cp_kp_stats_f32 memset");
    const int ppi = h * w;
    int gx = (ppi + 255) / 256;
    if (gx > 256) gx = 256;
    CP_LAUNCH((kp_stats_kernel<MAXKP>), dim3(gx, batch), dim3(256), sizeof(int) * 2 * classes, st, field, ld, conf_off, labels, labels_est, ppi, classes, counts,
              conf_sums);
    return cp::check_launch("cp_kp_stats_f32");
}

extern "C" int cp_kp_reproj_loss_f32(const float* coords_yx, const float* gt_xy, const float* affine, const float* avail, int batch, int objects,
                                     int kp, float max_pixel_error, float weight, float* g_yx, double* loss_out, void* stream) {
    CP_REQUIRE(coords_yx && gt_xy && affine && avail && g_yx && loss_out && batch > 0 && objects > 0 && kp > 0, "cp_kp_reproj_loss_f32: bad arguments");
    CP_LAUNCH(kp_reproj_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, coords_yx, gt_xy, affine, avail, batch, objects, kp, max_pixel_error, weight,
              g_yx, loss_out);
    return cp::check_launch("cp_kp_reproj_loss_f32");
}
