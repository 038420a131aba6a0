// Training losses of compute_loss (train_casapose.py:40-145) for the merged-output models, forward and
// gradient in one pass over the network output:
//   mask   : mean softmax cross-entropy                                  (train_casapose.py:59-60)
//   vertex : smooth_l1_loss(dirs, target field, fg weights)              (utils/loss_functions.py:14-44)
//   proxy  : proxy_voting_loss_v2(loss_per_object=False)                  (utils/loss_functions.py:132-203)
// The target vector field (unit vectors pixel centre -> keypoint, image_utils.py:17-63) is evaluated on the
// fly from the keypoints instead of being materialised.  filter_vertex_with_segmentation
// (train_casapose.py:64-69) restricts the foreground to pixels whose arg-max prediction equals the label;
// the filtered map is a constant of the gradient (stop_gradient, :95).
// HBM-bound: one read of the output row, one write of the gradient row per pixel.
#include "common.h"

namespace {

constexpr int THREADS = 256;

// pass 1: filtered foreground label per pixel + per-image foreground count
__global__ void loss_prepare_kernel(const float* __restrict__ out, int ld, int K, const uint8_t* __restrict__ labels_fg, int pix_per_img, int batch,
                                    int filter, uint8_t* __restrict__ fg, int* __restrict__ count) {
    const int b = blockIdx.y;
    int local = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pix_per_img; i += gridDim.x * blockDim.x) {
        const size_t p = (size_t)b * pix_per_img + i;
        int l = labels_fg[p];
        if (filter && l != 0) {
            const float* z = out + p * ld;
            float best = z[0];
            int arg = 0;
            for (int k = 1; k < K; ++k)
                if (z[k] > best) { best = z[k]; arg = k; }  // first maximum, like tf.argmax
            if (arg != l) l = 0;
        }
        fg[p] = (uint8_t)l;
        local += l != 0;
    }
    // wave reduce, one atomic per wave
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&count[b], local);
}

__device__ __forceinline__ float sl1(float a) { return a < 1.f ? 0.5f * a * a : a - 0.5f; }

// pass 2: one pixel per lane; the output record and the gradient row are moved with 16-byte accesses when their strides allow it
// (ld % 4 == 0 -- true for the padded training record -- and dld % 4 == 0), the arithmetic runs on register copies
constexpr int MAXREC = 64;   // floats of the output record / gradient row held in registers

template <bool VEC>
__global__ __launch_bounds__(THREADS) void loss_main_kernel(const float* __restrict__ out, int ld, int K, int kp, const uint8_t* __restrict__ labels_ce,
                                                            const uint8_t* __restrict__ fg, const int* __restrict__ count,
                                                            const float* __restrict__ keypoints, int objects, int batch, int H, int W,
                                                            float mask_w, float vertex_w, float proxy_w, float* __restrict__ dout, int dld,
                                                            int vert_off, double* __restrict__ sums) {
    const int pix_per_img = H * W;
    const int b = blockIdx.y;
    const float inv_ce = 1.f / ((float)batch * (float)pix_per_img);
    const float nrm_b = 1.f / ((2.f * kp * (float)count[b] + 1e-3f) * (float)batch);
    const int nrec = K + 2 * kp;  // floats of the record that are read
    double s_mask = 0.0, s_vert = 0.0, s_proxy = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pix_per_img; i += gridDim.x * blockDim.x) {
        const size_t p = (size_t)b * pix_per_img + i;
        float z[MAXREC], g[MAXREC];
        if constexpr (VEC) {
            const float4* src = reinterpret_cast<const float4*>(out + p * ld);
#pragma unroll
            for (int q = 0; q < 8; ++q)  // the logits live in the first 32 floats
                if (4 * q < K) {
                    const float4 t = src[q];
                    z[4 * q] = t.x; z[4 * q + 1] = t.y; z[4 * q + 2] = t.z; z[4 * q + 3] = t.w;
                }
        } else {
#pragma unroll
            for (int q = 0; q < MAXREC; ++q)
                if (q < nrec) z[q] = out[p * ld + q];
        }
#pragma unroll
        for (int q = 0; q < MAXREC; ++q) g[q] = 0.f;
        // ---- cross-entropy ------------------------------------------------------------------
        float mx = z[0];
#pragma unroll
        for (int k = 1; k < 32; ++k)
            if (k < K) mx = fmaxf(mx, z[k]);
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (k < K) se += __expf(z[k] - mx);
        const int lc = labels_ce[p];
        const float lse = mx + __logf(se);
        const float inv_se = 1.f / se;
        float zl = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (k < K) {
                g[k] = mask_w * inv_ce * (__expf(z[k] - mx) * inv_se - (k == lc ? 1.f : 0.f));
                zl = (k == lc) ? z[k] : zl;
            }
        s_mask += (double)(lse - zl);
        // ---- vertex + proxy ------------------------------------------------------------------
        const int l = fg[p];
        if (l != 0) {
            const int y = i / W, x = i - y * W;
            const float cy = y + 0.5f, cx = x + 0.5f;
            const float* kpt = keypoints + ((size_t)b * objects + (l - 1)) * kp * 2;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j >= kp) continue;
                const float ky = kpt[2 * j], kx = kpt[2 * j + 1];
                const float ay = ky - cy, ax = kx - cx;  // pixel centre -> keypoint
                // the directions start at the run-time offset K: scalar loads (cache hits, the record was just read) instead of
                // dynamic register indexing
                const float vy = out[p * ld + K + 2 * j], vx = out[p * ld + K + 2 * j + 1];
                const float tn = sqrtf(ay * ay + ax * ax);
                const float it = 1.f / fmaxf(tn, 1e-12f);
                const float ey = vy - ay * it, ex = vx - ax * it;
                const float aey = fabsf(ey), aex = fabsf(ex);
                s_vert += (double)(sl1(aey) + sl1(aex));
                float gy = vertex_w * nrm_b * (aey < 1.f ? ey : copysignf(1.f, ey));
                float gx = vertex_w * nrm_b * (aex < 1.f ? ex : copysignf(1.f, ex));
                const float num = vy * ax - vx * ay;
                const float n2 = vy * vy + vx * vx;
                if (n2 > 0.f) {
                    const float nr = sqrtf(n2), inr = 1.f / nr;
                    const float dist = fabsf(num) * inr;
                    s_proxy += (double)sl1(dist);
                    const float dl = (dist < 1.f ? dist : 1.f) * proxy_w * nrm_b;
                    const float sg = num > 0.f ? 1.f : (num < 0.f ? -1.f : 0.f);
                    const float c2 = dist * inr * inr;  // |num| / nr^3
                    gy += dl * (sg * ax * inr - c2 * vy);
                    gx += dl * (-sg * ay * inr - c2 * vx);
                }
                if constexpr (VEC) {  // vert_off == 32 (checked by the launcher): static register indices
                    g[32 + 2 * j] = gy;
                    g[32 + 2 * j + 1] = gx;
                } else {
#pragma unroll
                    for (int q = 0; q < MAXREC; ++q) {
                        g[q] = (q == vert_off + 2 * j) ? gy : g[q];
                        g[q] = (q == vert_off + 2 * j + 1) ? gx : g[q];
                    }
                }
            }
        }
        if constexpr (VEC) {
            float4* dst = reinterpret_cast<float4*>(dout + p * dld);
#pragma unroll
            for (int q = 0; q < MAXREC / 4; ++q)
                if (4 * q < dld) dst[q] = make_float4(g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]);
        } else {
#pragma unroll
            for (int q = 0; q < MAXREC; ++q)
                if (q < dld) dout[p * dld + q] = g[q];
        }
    }
    // block reduce (fp64) -> 3 atomics per block
    __shared__ double red[3][THREADS / 64];
    for (int o = 32; o > 0; o >>= 1) {
        s_mask += __shfl_down(s_mask, o);
        s_vert += __shfl_down(s_vert, o);
        s_proxy += __shfl_down(s_proxy, o);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = s_mask; red[1][wv] = s_vert; red[2][wv] = s_proxy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, c = 0, d = 0;
        for (int q = 0; q < THREADS / 64; ++q) { a += red[0][q]; c += red[1][q]; d += red[2][q]; }
        atomicAdd(&sums[0], a * (double)inv_ce);
        atomicAdd(&sums[1], c * (double)nrm_b);
        atomicAdd(&sums[2], d * (double)nrm_b);
    }
}

// filter_high_proxy_errors (train_casapose.py:71-93 with proxy_voting_dist, loss_functions.py:47-129): per image and object the mean
// smooth-L1 proxy distance over its (already segmentation-filtered) pixels; objects whose value is >= 5 are removed from the
// foreground of the vertex / proxy losses (a constant of the gradient).
__global__ __launch_bounds__(THREADS) void proxy_object_sums_kernel(const float* __restrict__ out, int ld, int K, int kp, const uint8_t* __restrict__ fg,
                                                                    const float* __restrict__ keypoints, int objects, int H, int W,
                                                                    double* __restrict__ objsum, int* __restrict__ objcnt) {
    extern __shared__ double ssum[];                 // [objects] doubles, then [objects] ints
    int* scnt = reinterpret_cast<int*>(ssum + objects);
    const int b = blockIdx.y, ppi = H * W;
    for (int i = threadIdx.x; i < objects; i += blockDim.x) { ssum[i] = 0.0; scnt[i] = 0; }
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ppi; i += gridDim.x * blockDim.x) {
        const size_t p = (size_t)b * ppi + i;
        const int l = fg[p];
        if (l == 0) continue;
        const int y = i / W, x = i - y * W;
        const float cy = y + 0.5f, cx = x + 0.5f;
        const float* kpt = keypoints + ((size_t)b * objects + (l - 1)) * kp * 2;
        const float* v = out + p * ld + K;
        float s = 0.f;
        for (int j = 0; j < kp; ++j) {
            const float ay = kpt[2 * j] - cy, ax = kpt[2 * j + 1] - cx;
            const float vy = v[2 * j], vx = v[2 * j + 1];
            const float n2 = vy * vy + vx * vx;
            const float dist = n2 > 0.f ? fabsf(vy * ax - vx * ay) / sqrtf(n2) : 0.f;
            s += sl1(dist);
        }
        atomicAdd(&ssum[l - 1], (double)s);
        atomicAdd(&scnt[l - 1], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < objects; i += blockDim.x) {
        if (scnt[i]) {
            atomicAdd(&objsum[b * objects + i], ssum[i]);
            atomicAdd(&objcnt[b * objects + i], scnt[i]);
        }
    }
}

__global__ void proxy_filter_finalize_kernel(const double* __restrict__ objsum, const int* __restrict__ objcnt, int batch, int objects, int kp,
                                             int min_object_pixel, uint8_t* __restrict__ bad, int* __restrict__ count, float* __restrict__ values) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    int good = 0;
    for (int o = 0; o < objects; ++o) {
        const int c = objcnt[b * objects + o];
        const float v = c >= min_object_pixel ? (float)(objsum[b * objects + o] / ((double)kp * c + 1e-3)) : 0.f;
        const bool is_bad = !(v < 5.f);
        bad[b * objects + o] = is_bad ? 1 : 0;  // keep iff value < 5 (train_casapose.py:82)
        if (values) values[b * objects + o] = v;
        good += is_bad ? 0 : c;
    }
    if (count) count[b] = good;
}

__global__ void proxy_filter_apply_kernel(uint8_t* __restrict__ fg, const uint8_t* __restrict__ bad, int ppi, int objects) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ppi; i += gridDim.x * blockDim.x) {
        const size_t p = (size_t)b * ppi + i;
        const int l = fg[p];
        if (l && bad[b * objects + l - 1]) fg[p] = 0;
    }
}

}  // namespace

extern "C" size_t cp_pose_loss_workspace_bytes(int batch, int h, int w) {
    // filtered label map | per-image counts | per-object proxy sums (fp64), counts, flags (up to 64 objects)
    return (((size_t)batch * h * w + 255) & ~(size_t)255) + (((sizeof(int) * (size_t)batch) + 255) & ~(size_t)255) + (size_t)batch * 64 * (8 + 4 + 1) + 256;
}

extern "C" int cp_pose_loss_f32(const float* out, int ld, int seg_dim, int kp, const uint8_t* labels_ce, const uint8_t* labels_fg,
                                const float* keypoints_yx, int objects, int batch, int h, int w, int filter_with_segmentation,
                                int filter_high_proxy_errors, float mask_w, float vertex_w, float proxy_w, void* ws, float* dout, int dld,
                                int vert_off, double* loss_sums, float* object_loss_values, void* stream) {
    CP_REQUIRE(out && labels_ce && labels_fg && keypoints_yx && ws && dout && loss_sums, "cp_pose_loss_f32: null pointer");
    CP_REQUIRE(seg_dim >= 2 && seg_dim <= 64 && objects == seg_dim - 1 && kp >= 1, "cp_pose_loss_f32: seg_dim = objects + 1 (2..64), kp >= 1");
    CP_REQUIRE(ld >= seg_dim + 2 * kp, "cp_pose_loss_f32: ld < seg_dim + 2*kp");
    CP_REQUIRE(vert_off >= seg_dim && dld >= vert_off + 2 * kp, "cp_pose_loss_f32: gradient row layout [0,seg_dim) | [vert_off, vert_off+2kp) does not fit dld");
    CP_REQUIRE(seg_dim <= 32 && kp <= 16 && seg_dim + 2 * kp <= MAXREC && dld <= MAXREC, "cp_pose_loss_f32: at most 32 classes, 16 keypoints, 64-float rows");
    CP_REQUIRE(batch > 0 && h > 0 && w > 0 && (long long)batch * h * w < (1LL << 31), "cp_pose_loss_f32: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int ppi = h * w;
    uint8_t* fg = (uint8_t*)ws;
    int* count = (int*)((char*)ws + (((size_t)batch * ppi + 255) & ~(size_t)255));
    if (hipMemsetAsync(count, 0, sizeof(int) * batch, st) != hipSuccess) return cp::check_launch("cp_pose_loss_f32 memset");
    if (hipMemsetAsync(loss_sums, 0, sizeof(double) * 3, st) != hipSuccess) return cp::check_launch("cp_pose_loss_f32 memset");
    int gx = (ppi + THREADS - 1) / THREADS;
    if (gx > 512) gx = 512;
    CP_LAUNCH(loss_prepare_kernel, dim3(gx, batch), dim3(THREADS), 0, st, out, ld, seg_dim, labels_fg, ppi, batch, filter_with_segmentation, fg, count);
    if (cp::check_launch("cp_pose_loss_f32 prepare") != CP_OK) return CP_ERR_LAUNCH;
    if (filter_high_proxy_errors || object_loss_values) {
        char* base = (char*)count + (((sizeof(int) * (size_t)batch) + 255) & ~(size_t)255);
        double* objsum = (double*)base;
        int* objcnt = (int*)(base + (size_t)batch * 64 * 8);
        uint8_t* bad = (uint8_t*)(base + (size_t)batch * 64 * 12);
        if (hipMemsetAsync(base, 0, (size_t)batch * 64 * 13, st) != hipSuccess) return cp::check_launch("cp_pose_loss_f32 memset");
        CP_LAUNCH(proxy_object_sums_kernel, dim3(gx, batch), dim3(THREADS), objects * (sizeof(double) + sizeof(int)), st, out, ld, seg_dim, kp, fg, keypoints_yx,
                  objects, h, w, objsum, objcnt);
        // values are always reported against the segmentation-filtered mask; the foreground is only edited when the filter is on
        CP_LAUNCH(proxy_filter_finalize_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, objsum, objcnt, batch, objects, kp, 20, bad,
                  filter_high_proxy_errors ? count : (int*)nullptr, object_loss_values);
        if (filter_high_proxy_errors)
            CP_LAUNCH(proxy_filter_apply_kernel, dim3(gx, batch), dim3(THREADS), 0, st, fg, bad, ppi, objects);
        if (cp::check_launch("cp_pose_loss_f32 proxy filter") != CP_OK) return CP_ERR_LAUNCH;
    }
    const bool vec = ld % 4 == 0 && dld == 64 && vert_off == 32 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)dout & 15) == 0 && ((seg_dim + 3) & ~3) <= ld;
    if (vec)
        CP_LAUNCH(loss_main_kernel<true>, dim3(gx, batch), dim3(THREADS), 0, st, out, ld, seg_dim, kp, labels_ce, fg, count, keypoints_yx, objects, batch, h, w,
                  mask_w, vertex_w, proxy_w, dout, dld, vert_off, loss_sums);
    else
        CP_LAUNCH(loss_main_kernel<false>, dim3(gx, batch), dim3(THREADS), 0, st, out, ld, seg_dim, kp, labels_ce, fg, count, keypoints_yx, objects, batch, h, w,
                  mask_w, vertex_w, proxy_w, dout, dld, vert_off, loss_sums);
    return cp::check_launch("cp_pose_loss_f32");
}
