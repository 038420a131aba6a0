// The reference's loss / target-field FUNCTIONS as stand-alone device passes (the fused training step uses loss_kernels.hip; these back
// the importable casapose.utils.loss_functions / casapose.utils.image_utils surface and the separated-vector-field losses of `pvnet`):
//   cp_vector_field_f32    compute_vertex_hcoords_batch_v3 / get_all_vectorfields          (utils/image_utils.py:17-79)
//   cp_smooth_l1_f32       smooth_l1_loss                                                  (utils/loss_functions.py:14-44)
//   cp_proxy_voting_f32    proxy_voting_dist / proxy_voting_loss_v2                        (utils/loss_functions.py:47-203)
//   cp_pose_loss_sep_f32   compute_loss with separated vector fields (oc * 2kp channels)   (train_casapose.py:57,97-125), value + gradient
// All HBM-bound streaming passes, one pixel per lane, fp64 reductions (wave shuffle -> one atomic per wave).
#include "common.h"

namespace {

constexpr int THREADS = 256;

inline int grid_x(long long n) {
    long long b = (n + THREADS - 1) / THREADS;
    return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

__device__ __forceinline__ float sl1(float a) { return a < 1.f ? 0.5f * a * a : a - 0.5f; }

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// keypoints [b][objects][instances][kp][2] (y,x).  Instance of object `obj` whose centre (keypoint 0) is nearest to the pixel centre; first
// minimum wins (tf.argmin).
__device__ __forceinline__ int nearest_instance(const float* __restrict__ kobj, int instances, int kp, float cy, float cx) {
    int best = 0;
    float bd = 3.4e38f;
    for (int i = 0; i < instances; ++i) {
        const float dy = cy - kobj[(size_t)i * kp * 2], dx = cx - kobj[(size_t)i * kp * 2 + 1];
        const float d = sqrtf(dy * dy + dx * dx);
        if (d < bd) { bd = d; best = i; }
    }
    return best;
}

__global__ __launch_bounds__(THREADS) void vector_field_kernel(const uint8_t* __restrict__ labels, const float* __restrict__ keypoints, int H, int W,
                                                               int objects, int instances, int kp, int separated, int normalize,
                                                               float* __restrict__ out, int ld) {
    const int b = blockIdx.y, ppi = H * W;
    const int width = (separated ? objects : 1) * 2 * kp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ppi; i += gridDim.x * blockDim.x) {
        const size_t p = (size_t)b * ppi + i;
        float* o = out + p * ld;
        for (int q = 0; q < width; ++q) o[q] = 0.f;
        const int l = labels[p];
        if (l == 0 || l > objects) continue;
        const int y = i / W, x = i - y * W;
        const float cy = y + 0.5f, cx = x + 0.5f;
        const float* kobj = keypoints + ((size_t)b * objects + (l - 1)) * instances * kp * 2;
        const int inst = instances > 1 ? nearest_instance(kobj, instances, kp, cy, cx) : 0;
        const float* k = kobj + (size_t)inst * kp * 2;
        float* dst = o + (separated ? (l - 1) * 2 * kp : 0);
        for (int j = 0; j < kp; ++j) {
            float dy = k[2 * j] - cy, dx = k[2 * j + 1] - cx;
            if (normalize) {  // tf.math.l2_normalize: x * rsqrt(max(sum x^2, 1e-12))
                const float inv = 1.f / sqrtf(fmaxf(dy * dy + dx * dx, 1e-12f));
                dy *= inv;
                dx *= inv;
            }
            dst[2 * j] = dy;
            dst[2 * j + 1] = dx;
        }
    }
}

__device__ __forceinline__ float weight_of(const float* __restrict__ w, int wld, size_t p, int wmode) {
    if (wmode == 2 || !w) return 1.f;
    const float v = w[p * wld];
    return wmode == 1 ? fabsf(1.f - v) : v;
}

__global__ __launch_bounds__(THREADS) void smooth_l1_kernel(const float* __restrict__ pred, int pld, const float* __restrict__ target, int tld,
                                                            const float* __restrict__ weights, int wld, int wmode, int C, int ppi,
                                                            double* __restrict__ sums, float* __restrict__ elem) {
    const int b = blockIdx.y;
    double s_loss = 0.0, s_w = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ppi; i += gridDim.x * blockDim.x) {
        const size_t p = (size_t)b * ppi + i;
        const float w = weight_of(weights, wld, p, wmode);
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
            const float v = sl1(fabsf(w * (pred[p * pld + c] - target[p * tld + c])));
            acc += v;
            if (elem) elem[p * C + c] = v;
        }
        s_loss += (double)acc;
        s_w += (double)w;
    }
    s_loss = wave_sum(s_loss);
    s_w = wave_sum(s_w);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[2 * b], s_loss);
        atomicAdd(&sums[2 * b + 1], s_w);
    }
}

// perpendicular distance between the keypoint and the line through the pixel centre along the predicted direction, minimum over the
// object's instances: |v_y (k_x - c_x) - v_x (k_y - c_y)| / |v|, 0 where |v| = 0 (divide_no_nan)
__global__ __launch_bounds__(THREADS) void proxy_voting_kernel(const float* __restrict__ pred, int pld, int kp, const uint8_t* __restrict__ labels,
                                                               const float* __restrict__ weights, int wld, int wmode,
                                                               const float* __restrict__ keypoints, int objects, int instances, int H, int W,
                                                               double* __restrict__ img_sums, double* __restrict__ obj_sums, int* __restrict__ obj_counts,
                                                               float* __restrict__ dist_out, float* __restrict__ elem) {
    const int b = blockIdx.y, ppi = H * W;
    double s_loss = 0.0, s_w = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ppi; i += gridDim.x * blockDim.x) {
        const size_t p = (size_t)b * ppi + i;
        const int l = labels ? labels[p] : 1;
        const int obj = l > 0 ? (l <= objects ? l - 1 : objects - 1) : 0;   // arg-max of an all-zero one-hot row is 0
        const float w = weight_of(weights, wld, p, wmode);
        const int y = i / W, x = i - y * W;
        const float cy = y + 0.5f, cx = x + 0.5f;
        const float* kobj = keypoints + ((size_t)b * objects + obj) * instances * kp * 2;
        float acc = 0.f;
        for (int j = 0; j < kp; ++j) {
            const float vy = pred[p * pld + 2 * j], vx = pred[p * pld + 2 * j + 1];
            const float n2 = vy * vy + vx * vx;
            float best = 3.4e38f;
            for (int t = 0; t < instances; ++t) {
                const float ky = kobj[((size_t)t * kp + j) * 2], kx = kobj[((size_t)t * kp + j) * 2 + 1];
                const float num = fabsf(vy * (kx - cx) - vx * (ky - cy));
                const float d = n2 > 0.f ? num / sqrtf(n2) : 0.f;
                best = fminf(best, d);
            }
            const float dist = fabsf(w * best);
            if (dist_out) dist_out[p * kp + j] = dist;
            const float v = sl1(dist);
            if (elem) elem[p * kp + j] = v;
            acc += v;
        }
        s_loss += (double)acc;
        s_w += (double)w;
        if (obj_sums) {
            atomicAdd(&obj_sums[(size_t)b * objects + obj], (double)acc);
            if (l > 0) atomicAdd(&obj_counts[(size_t)b * objects + obj], 1);
        }
    }
    s_loss = wave_sum(s_loss);
    s_w = wave_sum(s_w);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&img_sums[2 * b], s_loss);
        atomicAdd(&img_sums[2 * b + 1], s_w);
    }
}

// ---- separated vector fields (pvnet): per-object slices of the field, losses normalised per (image, object) ---------------------------
__global__ __launch_bounds__(THREADS) void sep_prepare_kernel(const float* __restrict__ out, int ld, int K, const uint8_t* __restrict__ labels_fg, int ppi,
                                                              int filter, uint8_t* __restrict__ fg, int* __restrict__ counts) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ppi; i += gridDim.x * blockDim.x) {
        const size_t p = (size_t)b * ppi + i;
        int l = labels_fg[p];
        if (filter && l != 0) {
            const float* z = out + p * ld;
            float best = z[0];
            int arg = 0;
            for (int k = 1; k < K; ++k)
                if (z[k] > best) { best = z[k]; arg = k; }
            if (arg != l) l = 0;
        }
        fg[p] = (uint8_t)l;
        if (l != 0 && l < K) atomicAdd(&counts[b * K + l], 1);
    }
}

__global__ __launch_bounds__(THREADS) void sep_main_kernel(const float* __restrict__ out, int ld, int K, int kp, const uint8_t* __restrict__ labels_ce,
                                                           const uint8_t* __restrict__ fg, const int* __restrict__ counts,
                                                           const float* __restrict__ keypoints, int objects, int batch, int H, int W, float mask_w,
                                                           float vertex_w, float proxy_w, float* __restrict__ dout, int dld, int vert_off,
                                                           double* __restrict__ sums) {
    const int ppi = H * W, b = blockIdx.y;
    const float inv_ce = 1.f / ((float)batch * (float)ppi);
    double s_mask = 0.0, s_vert = 0.0, s_proxy = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ppi; i += gridDim.x * blockDim.x) {
        const size_t p = (size_t)b * ppi + i;
        const float* z = out + p * ld;
        float* g = dout + p * dld;
        for (int q = 0; q < dld; ++q) g[q] = 0.f;
        // cross-entropy over the K logits
        float mx = z[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, z[k]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += __expf(z[k] - mx);
        const int lc = labels_ce[p];
        const float inv_se = 1.f / se;
        for (int k = 0; k < K; ++k) g[k] = mask_w * inv_ce * (__expf(z[k] - mx) * inv_se - (k == lc ? 1.f : 0.f));
        s_mask += (double)(mx + __logf(se) - z[lc]);
        const int l = fg[p];
        if (l == 0 || l > objects) continue;
        // the slice of the pixel's own object: smooth-L1 to the unit vectors + proxy voting, both normalised by 2kp * (pixels of that object in
        // this image) + 1e-3 and averaged over the batch (smooth_l1_loss / proxy_voting_loss_v2 called once per object, train_casapose.py:97-125)
        const float nrm = 1.f / ((2.f * kp * (float)counts[b * K + l] + 1e-3f) * (float)batch);
        const int y = i / W, x = i - y * W;
        const float cy = y + 0.5f, cx = x + 0.5f;
        const float* kpt = keypoints + ((size_t)b * objects + (l - 1)) * kp * 2;
        const float* v = z + K + (size_t)(l - 1) * 2 * kp;
        float* gv = g + vert_off + (size_t)(l - 1) * 2 * kp;
        for (int j = 0; j < kp; ++j) {
            const float ay = kpt[2 * j] - cy, ax = kpt[2 * j + 1] - cx;
            const float vy = v[2 * j], vx = v[2 * j + 1];
            const float it = 1.f / sqrtf(fmaxf(ay * ay + ax * ax, 1e-24f));
            const float ey = vy - ay * it, ex = vx - ax * it;
            const float aey = fabsf(ey), aex = fabsf(ex);
            s_vert += (double)((sl1(aey) + sl1(aex)) * nrm);
            float gy = vertex_w * nrm * (aey < 1.f ? ey : copysignf(1.f, ey));
            float gx = vertex_w * nrm * (aex < 1.f ? ex : copysignf(1.f, ex));
            const float num = vy * ax - vx * ay, n2 = vy * vy + vx * vx;
            if (n2 > 0.f) {
                const float nr = sqrtf(n2), inr = 1.f / nr;
                const float dist = fabsf(num) * inr;
                s_proxy += (double)(sl1(dist) * nrm);
                const float dl = (dist < 1.f ? dist : 1.f) * proxy_w * nrm;
                const float sg = num > 0.f ? 1.f : (num < 0.f ? -1.f : 0.f);
                const float c2 = dist * inr * inr;
                gy += dl * (sg * ax * inr - c2 * vy);
                gx += dl * (-sg * ay * inr - c2 * vx);
            }
            gv[2 * j] = gy;
            gv[2 * j + 1] = gx;
        }
    }
    s_mask = wave_sum(s_mask);
    s_vert = wave_sum(s_vert);
    s_proxy = wave_sum(s_proxy);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[0], s_mask * (double)inv_ce);
        atomicAdd(&sums[1], s_vert);
        atomicAdd(&sums[2], s_proxy);
    }
}

}  // namespace

extern "C" int cp_vector_field_f32(const uint8_t* labels, const float* keypoints_yx, int batch, int h, int w, int objects, int instances, int kp,
                                   int separated, int normalize, float* out, int ld, void* stream) {
    CP_REQUIRE(labels && keypoints_yx && out, "cp_vector_field_f32: null pointer");
    CP_REQUIRE(batch > 0 && h > 0 && w > 0 && objects >= 1 && objects <= 255 && instances >= 1 && kp >= 1, "cp_vector_field_f32: bad shape");
    CP_REQUIRE(ld >= (separated ? objects : 1) * 2 * kp, "cp_vector_field_f32: ld smaller than the field width");
    CP_LAUNCH(vector_field_kernel, dim3(grid_x((long long)h * w), batch), dim3(THREADS), 0, (hipStream_t)stream, labels, keypoints_yx, h, w, objects, instances,
              kp, separated, normalize, out, ld);
    return cp::check_launch("cp_vector_field_f32");
}

extern "C" int cp_smooth_l1_f32(const float* pred, int pld, const float* target, int tld, const float* weights, int wld, int wmode, int channels,
                                int batch, long long pixels_per_image, double* sums, float* elem, void* stream) {
    CP_REQUIRE(pred && target && sums, "cp_smooth_l1_f32: null pointer");
    CP_REQUIRE(channels >= 1 && pld >= channels && tld >= channels && batch > 0 && pixels_per_image > 0 && pixels_per_image < (1LL << 31), "cp_smooth_l1_f32: bad shape");
    CP_REQUIRE(wmode >= 0 && wmode <= 2 && (wmode == 2 || weights) && (!weights || wld >= 1), "cp_smooth_l1_f32: wmode 0 (as given) / 1 (|1 - w|) need weights, 2 = ones");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * batch, st) != hipSuccess) return cp::check_launch("cp_smooth_l1_f32 memset");
    CP_LAUNCH(smooth_l1_kernel, dim3(grid_x(pixels_per_image), batch), dim3(THREADS), 0, st, pred, pld, target, tld, weights, wld, wmode, channels,
              (int)pixels_per_image, sums, elem);
    return cp::check_launch("cp_smooth_l1_f32");
}

extern "C" int cp_proxy_voting_f32(const float* pred, int pld, int kp, const uint8_t* labels, const float* weights, int wld, int wmode,
                                   const float* keypoints_yx, int objects, int instances, int batch, int h, int w, double* img_sums, double* obj_sums,
                                   int32_t* obj_counts, float* dist_out, float* elem, void* stream) {
    CP_REQUIRE(pred && keypoints_yx && img_sums, "cp_proxy_voting_f32: null pointer");
    CP_REQUIRE(kp >= 1 && pld >= 2 * kp && objects >= 1 && objects <= 255 && instances >= 1 && batch > 0 && h > 0 && w > 0, "cp_proxy_voting_f32: bad shape");
    CP_REQUIRE(labels || objects == 1, "cp_proxy_voting_f32: several objects need a label map");
    CP_REQUIRE(wmode >= 0 && wmode <= 2 && (wmode == 2 || weights), "cp_proxy_voting_f32: wmode 0 / 1 need weights");
    CP_REQUIRE((obj_sums == nullptr) == (obj_counts == nullptr), "cp_proxy_voting_f32: obj_sums and obj_counts come together");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(img_sums, 0, sizeof(double) * 2 * batch, st) != hipSuccess) return cp::check_launch("cp_proxy_voting_f32 memset");
    if (obj_sums) {
        if (hipMemsetAsync(obj_sums, 0, sizeof(double) * batch * objects, st) != hipSuccess) return cp::check_launch("cp_proxy_voting_f32 memset");
        if (hipMemsetAsync(obj_counts, 0, sizeof(int) * batch * objects, st) != hipSuccess) return cp::check_launch("cp_proxy_voting_f32 memset");
    }
    CP_LAUNCH(proxy_voting_kernel, dim3(grid_x((long long)h * w), batch), dim3(THREADS), 0, st, pred, pld, kp, labels, weights, wld, wmode, keypoints_yx, objects,
              instances, h, w, img_sums, obj_sums, obj_counts, dist_out, elem);
    return cp::check_launch("cp_proxy_voting_f32");
}

extern "C" size_t cp_pose_loss_sep_workspace_bytes(int batch, int h, int w, int seg_dim) {
    return (((size_t)batch * h * w + 255) & ~(size_t)255) + (size_t)batch * seg_dim * sizeof(int) + 256;
}

extern "C" int cp_pose_loss_sep_f32(const float* out, int ld, int seg_dim, int kp, const uint8_t* labels_ce, const uint8_t* labels_fg,
                                    const float* keypoints_yx, int objects, int batch, int h, int w, int filter_with_segmentation, float mask_w,
                                    float vertex_w, float proxy_w, void* ws, float* dout, int dld, int vert_off, double* loss_sums, void* stream) {
    CP_REQUIRE(out && labels_ce && labels_fg && keypoints_yx && ws && dout && loss_sums, "cp_pose_loss_sep_f32: null pointer");
    CP_REQUIRE(seg_dim >= 2 && seg_dim <= 64 && objects == seg_dim - 1 && kp >= 1, "cp_pose_loss_sep_f32: seg_dim = objects + 1 (2..64), kp >= 1");
    CP_REQUIRE(ld >= seg_dim + objects * 2 * kp, "cp_pose_loss_sep_f32: ld < seg_dim + objects*2*kp");
    CP_REQUIRE(vert_off >= seg_dim && dld >= vert_off + objects * 2 * kp, "cp_pose_loss_sep_f32: gradient row [0,seg_dim) | [vert_off, vert_off + objects*2kp) does not fit dld");
    CP_REQUIRE(batch > 0 && h > 0 && w > 0 && (long long)batch * h * w < (1LL << 31), "cp_pose_loss_sep_f32: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int ppi = h * w;
    uint8_t* fg = (uint8_t*)ws;
    int* counts = (int*)((char*)ws + (((size_t)batch * ppi + 255) & ~(size_t)255));
    if (hipMemsetAsync(counts, 0, sizeof(int) * batch * seg_dim, st) != hipSuccess) return cp::check_launch("cp_pose_loss_sep_f32 memset");
    if (hipMemsetAsync(loss_sums, 0, sizeof(double) * 3, st) != hipSuccess) return cp::check_launch("cp_pose_loss_sep_f32 memset");
    const int gx = grid_x(ppi);
    CP_LAUNCH(sep_prepare_kernel, dim3(gx, batch), dim3(THREADS), 0, st, out, ld, seg_dim, labels_fg, ppi, filter_with_segmentation, fg, counts);
    CP_LAUNCH(sep_main_kernel, dim3(gx, batch), dim3(THREADS), 0, st, out, ld, seg_dim, kp, labels_ce, fg, counts, keypoints_yx, objects, batch, h, w, mask_w,
              vertex_w, proxy_w, dout, dld, vert_off, loss_sums);
    return cp::check_launch("cp_pose_loss_sep_f32");
}
