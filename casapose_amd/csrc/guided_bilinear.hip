// GuidedBilinearUpsampling (casapose/pose_models/models/_normalization_layers.py:569-664), used by casapose_c_gcu4_bilat:
// a 2x upsampling whose four taps {(y,x),(y,x+1),(y+1,x),(y+1,x+1)} (zero padded bottom/right) are blended with the fixed
// per-sub-pixel weights [[1,0,0,0],[.5,.5,0,0],[.5,0,.5,0],[.25,.25,.25,.25]] (:596-604), but a tap whose LOW-resolution label differs
// from the HIGH-resolution label of the output pixel is replaced by the mean of the matching taps (:643-660; zero if none matches).
// That is linear in the taps: out = sum_j coef_j * tap_j with coef_j = match_j * (w_j + (sum of the weights of the non-matching
// taps) / #matching), so the whole layer is a 4-tap gather driven by a 4-bit match mask per output pixel; the adjoint gathers
// the same coefficients.  HBM-bound streaming kernels.
#include "common.h"

namespace {

constexpr int THREADS = 256;

inline int grid_for(long long n) {
    long long b = (n + THREADS - 1) / THREADS;
    return (int)(b < 1 ? 1 : (b > 256 * 8 ? 256 * 8 : b));
}

__global__ void match_mask_kernel(const uint8_t* __restrict__ hi, const uint8_t* __restrict__ lo, int B, int H, int W, uint8_t* __restrict__ mask) {
    const int Hl = H / 2, Wl = W / 2;
    const long long total = (long long)B * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long long t = i / W;
        const int y = (int)(t % H), n = (int)(t / H);
        const int c = hi[i], ly = y >> 1, lx = x >> 1;
        const uint8_t* lb = lo + (size_t)n * Hl * Wl;
        const bool xr = (lx + 1) < Wl, yb = (ly + 1) < Hl;
        int m = lb[(size_t)ly * Wl + lx] == c ? 1 : 0;
        if (xr && lb[(size_t)ly * Wl + lx + 1] == c) m |= 2;
        if (yb && lb[(size_t)(ly + 1) * Wl + lx] == c) m |= 4;
        if (xr && yb && lb[(size_t)(ly + 1) * Wl + lx + 1] == c) m |= 8;
        mask[i] = (uint8_t)m;
    }
}

__device__ __forceinline__ void coefficients(int m, int sub, float (&coef)[4]) {
    const float w[4] = {sub == 0 ? 1.f : (sub == 3 ? 0.25f : 0.5f), sub == 1 ? 0.5f : (sub == 3 ? 0.25f : 0.f), sub == 2 ? 0.5f : (sub == 3 ? 0.25f : 0.f),
                        sub == 3 ? 0.25f : 0.f};
    const int n = __popc(m & 15);
    float miss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) miss += ((m >> j) & 1) ? 0.f : w[j];
    const float share = n ? miss / (float)n : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) coef[j] = ((m >> j) & 1) ? w[j] + share : 0.f;
}

__global__ void guided_bilinear_kernel(const float* __restrict__ src, const uint8_t* __restrict__ mask, int B, int H, int W, int C, float* __restrict__ dst) {
    const int c4n = C >> 2, Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)B * Ho * Wo * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const long long pix = i / c4n;
        const int ox = (int)(pix % Wo);
        const long long t = pix / Wo;
        const int oy = (int)(t % Ho), n = (int)(t / Ho);
        float coef[4];
        coefficients(mask[pix], (oy & 1) * 2 + (ox & 1), coef);
        const int y = oy >> 1, x = ox >> 1;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (coef[j] == 0.f) continue;  // non-matching taps (including the zero padding) never contribute
            const float4 v = *reinterpret_cast<const float4*>(src + (((size_t)n * H + y + (j >> 1)) * W + x + (j & 1)) * C + c4 * 4);
            o.x += coef[j] * v.x; o.y += coef[j] * v.y; o.z += coef[j] * v.z; o.w += coef[j] * v.w;
        }
        *reinterpret_cast<float4*>(dst + (size_t)pix * C + c4 * 4) = o;
    }
}

// adjoint in gather form: low-res pixel (y,x) is tap j of the cells (y - (j>>1), x - (j&1))
__global__ void guided_bilinear_bwd_kernel(const float* __restrict__ dy, int ld_dy, const uint8_t* __restrict__ mask, int B, int H, int W, int C,
                                           float* __restrict__ dx) {
    const int c4n = C >> 2, Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)B * H * W * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const long long pix = i / c4n;
        const int x = (int)(pix % W);
        const long long t = pix / W;
        const int y = (int)(t % H), n = (int)(t / H);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cy = y - (j >> 1), cx = x - (j & 1);
            if (cy < 0 || cx < 0) continue;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const size_t hp = ((size_t)n * Ho + 2 * cy + (sub >> 1)) * Wo + 2 * cx + (sub & 1);
                float coef[4];
                coefficients(mask[hp], sub, coef);
                if (coef[j] == 0.f) continue;
                const float4 g = *reinterpret_cast<const float4*>(dy + hp * ld_dy + c4 * 4);
                o.x += coef[j] * g.x; o.y += coef[j] * g.y; o.z += coef[j] * g.z; o.w += coef[j] * g.w;
            }
        }
        *reinterpret_cast<float4*>(dx + (size_t)pix * C + c4 * 4) = o;
    }
}

}  // namespace

extern "C" int cp_guided_match_mask(const uint8_t* labels_hi, const uint8_t* labels_lo, int batch, int h_hi, int w_hi, uint8_t* mask, void* stream) {
    CP_REQUIRE(labels_hi && labels_lo && mask && batch > 0 && h_hi > 0 && w_hi > 0 && h_hi % 2 == 0 && w_hi % 2 == 0, "cp_guided_match_mask: bad arguments (even high-resolution size)");
    CP_LAUNCH(match_mask_kernel, dim3(grid_for((long long)batch * h_hi * w_hi)), dim3(THREADS), 0, (hipStream_t)stream, labels_hi, labels_lo, batch, h_hi, w_hi, mask);
    return cp::check_launch("cp_guided_match_mask");
}

extern "C" int cp_guided_bilinear_upsample_x2_f32(const float* src, const uint8_t* mask, int batch, int h, int w, int channels, float* dst, void* stream) {
    CP_REQUIRE(src && mask && dst && batch > 0 && h > 0 && w > 0 && channels % 4 == 0, "cp_guided_bilinear_upsample_x2_f32: bad arguments");
    CP_LAUNCH(guided_bilinear_kernel, dim3(grid_for((long long)batch * 4 * h * w * (channels / 4))), dim3(THREADS), 0, (hipStream_t)stream, src, mask, batch, h, w,
              channels, dst);
    return cp::check_launch("cp_guided_bilinear_upsample_x2_f32");
}

extern "C" int cp_guided_bilinear_upsample_x2_bwd_f32(const float* dy, int ld_dy, const uint8_t* mask, int batch, int h, int w, int channels, float* dx,
                                                      void* stream) {
    CP_REQUIRE(dy && mask && dx && batch > 0 && h > 0 && w > 0 && channels % 4 == 0 && ld_dy >= channels, "cp_guided_bilinear_upsample_x2_bwd_f32: bad arguments");
    CP_LAUNCH(guided_bilinear_bwd_kernel, dim3(grid_for((long long)batch * h * w * (channels / 4))), dim3(THREADS), 0, (hipStream_t)stream, dy, ld_dy, mask, batch,
              h, w, channels, dx);
    return cp::check_launch("cp_guided_bilinear_upsample_x2_bwd_f32");
}
