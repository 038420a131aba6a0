// Streaming (HBM-bound) kernels around the convolutions: channel padding, max-pool,
// stand-alone x2 upsamplers, arg-max labels and the label pyramid of decoder 2.
// All of them move 16 B per lane per access and grid-stride over a capped grid.
#include "common.h"

namespace {

constexpr int THREADS = 256;

inline int grid_for(long long work_items) {
    long long b = (work_items + THREADS - 1) / THREADS;
    if (b < 1) b = 1;
    if (b > 256 * 8) b = 256 * 8;
    return (int)b;
}

__global__ void pad3to4_kernel(const float* __restrict__ src, float* __restrict__ dst, long long pixels) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < pixels; i += (long long)gridDim.x * blockDim.x) {
        const float* s = src + i * 3;
        reinterpret_cast<float4*>(dst)[i] = make_float4(s[0], s[1], s[2], 0.f);
    }
}

// one thread = one output pixel x 4 channels
__global__ void maxpool_kernel(const float* __restrict__ src, int B, int H, int W, int C, int Ho, int Wo,
                               const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                               float* __restrict__ dst) {
    const int c4n = C >> 2;
    const long long total = (long long)B * Ho * Wo * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c4 = (int)(i % c4n);
        long long pix = i / c4n;
        int ox = (int)(pix % Wo);
        long long t = pix / Wo;
        int oy = (int)(t % Ho);
        int n = (int)(t / Ho);
        // zero padding takes part in the max (resnet.py:253): start from 0 only if a tap is padded
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                    v = *reinterpret_cast<const float4*>(src + (((size_t)n * H + iy) * W + ix) * C + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        if (scale) {
            float4 s = *reinterpret_cast<const float4*>(scale + c4 * 4);
            float4 b = *reinterpret_cast<const float4*>(shift + c4 * 4);
            m.x = m.x * s.x + b.x; m.y = m.y * s.y + b.y; m.z = m.z * s.z + b.z; m.w = m.w * s.w + b.w;
        }
        if (relu) { m.x = fmaxf(m.x, 0.f); m.y = fmaxf(m.y, 0.f); m.z = fmaxf(m.z, 0.f); m.w = fmaxf(m.w, 0.f); }
        *reinterpret_cast<float4*>(dst + (size_t)pix * C + c4 * 4) = m;
    }
}

__global__ void bilinear_x2_kernel(const float* __restrict__ src, int B, int H, int W, int C, float* __restrict__ dst) {
    const int c4n = C >> 2, Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)B * Ho * Wo * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c4 = (int)(i % c4n);
        long long pix = i / c4n;
        int ox = (int)(pix % Wo);
        long long t = pix / Wo;
        int oy = (int)(t % Ho);
        int n = (int)(t / Ho);
        int y0 = (oy >> 1) - ((oy & 1) ? 0 : 1), x0 = (ox >> 1) - ((ox & 1) ? 0 : 1);
        float fy = (oy & 1) ? 0.25f : 0.75f, fx = (ox & 1) ? 0.25f : 0.75f;
        int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        y0 = max(y0, 0); x0 = max(x0, 0);
        const float* base = src + (size_t)n * H * W * C + c4 * 4;
        float4 v00 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * W + x0) * C);
        float4 v01 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * W + x1) * C);
        float4 v10 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * W + x0) * C);
        float4 v11 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * W + x1) * C);
        float gx = 1.f - fx, gy = 1.f - fy;
        float4 o;
        o.x = (v00.x * gx + v01.x * fx) * gy + (v10.x * gx + v11.x * fx) * fy;
        o.y = (v00.y * gx + v01.y * fx) * gy + (v10.y * gx + v11.y * fx) * fy;
        o.z = (v00.z * gx + v01.z * fx) * gy + (v10.z * gx + v11.z * fx) * fy;
        o.w = (v00.w * gx + v01.w * fx) * gy + (v10.w * gx + v11.w * fx) * fy;
        *reinterpret_cast<float4*>(dst + (size_t)pix * C + c4 * 4) = o;
    }
}

__global__ void guided_x2_kernel(const float* __restrict__ src, const uint8_t* __restrict__ sel, int B, int H, int W,
                                 int C, float* __restrict__ dst) {
    const int c4n = C >> 2, Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)B * Ho * Wo * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c4 = (int)(i % c4n);
        long long pix = i / c4n;
        int ox = (int)(pix % Wo);
        long long t = pix / Wo;
        int oy = (int)(t % Ho);
        int n = (int)(t / Ho);
        int s = sel[pix];
        int sy = (oy >> 1) + (s >> 1), sx = (ox >> 1) + (s & 1);
        float4 v = *reinterpret_cast<const float4*>(src + (((size_t)n * H + sy) * W + sx) * C + c4 * 4);
        *reinterpret_cast<float4*>(dst + (size_t)pix * C + c4 * 4) = v;
    }
}

__global__ void argmax_kernel(const float* __restrict__ x, int ld, int K, long long pixels, uint8_t* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < pixels; i += (long long)gridDim.x * blockDim.x) {
        const float* p = x + i * ld;
        float best = p[0];
        int bi = 0;
        for (int k = 1; k < K; ++k) {
            float v = p[k];
            if (v > best) { best = v; bi = k; }
        }
        out[i] = (uint8_t)bi;
    }
}

// same result with 16-byte loads: the K (<= 12) logits of a pixel are the first floats of its 16-byte aligned record, so three
// float4 loads replace nine strided scalar loads (the record is touched once either way; this only cuts the request count)
template <int NV>
__global__ void argmax_vec_kernel(const float* __restrict__ x, int ld, int K, long long pixels, uint8_t* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < pixels; i += (long long)gridDim.x * blockDim.x) {
        const float4* p = reinterpret_cast<const float4*>(x + i * ld);
        float v[NV * 4];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const float4 t = p[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
        float best = v[0];
        int bi = 0;
#pragma unroll
        for (int k = 1; k < NV * 4; ++k)
            if (k < K && v[k] > best) { best = v[k]; bi = k; }
        out[i] = (uint8_t)bi;
    }
}

// level l+1 labels = level l labels [::2, ::2]  (HalfSize, _normalization_layers.py:294-299)
__global__ void half_labels_kernel(const uint8_t* __restrict__ in, int B, int H, int W, int Ho, int Wo, uint8_t* __restrict__ out) {
    const long long total = (long long)B * Ho * Wo;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int ox = (int)(i % Wo);
        long long t = i / Wo;
        int oy = (int)(t % Ho);
        int n = (int)(t / Ho);
        out[i] = in[((size_t)n * H + 2 * oy) * W + 2 * ox];
    }
}

// pnorm = 9 / #{in-bounds 3x3 neighbours with the centre label} (_normalization_layers.py:344-352)
__global__ void pnorm_kernel(const uint8_t* __restrict__ lab, int B, int H, int W, float* __restrict__ out) {
    const long long total = (long long)B * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int x = (int)(i % W);
        long long t = i / W;
        int y = (int)(t % H);
        int n = (int)(t / H);
        const uint8_t* base = lab + (size_t)n * H * W;
        int c = base[(size_t)y * W + x], cnt = 0;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                int yy = y + dy, xx = x + dx;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) cnt += (base[(size_t)yy * W + xx] == c);
            }
        out[i] = 9.0f / (float)cnt;  // the centre always matches, cnt >= 1
    }
}

// guided-upsampling neighbour selection (_normalization_layers.py:534-551): first of
// {(y,x),(y,x+1),(y+1,x),(y+1,x+1)} (zero padded bottom/right) in the LOW map whose label equals
// the HIGH label, else 0.
__global__ void guided_sel_kernel(const uint8_t* __restrict__ hi, const uint8_t* __restrict__ lo, int B, int H, int W,
                                  uint8_t* __restrict__ sel) {
    const int Hl = H / 2, Wl = W / 2;
    const long long total = (long long)B * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int x = (int)(i % W);
        long long t = i / W;
        int y = (int)(t % H);
        int n = (int)(t / H);
        int c = hi[i];
        int ly = y >> 1, lx = x >> 1;
        const uint8_t* lb = lo + (size_t)n * Hl * Wl;
        int s = 0;
        bool xr = (lx + 1) < Wl, yb = (ly + 1) < Hl;
        if (lb[(size_t)ly * Wl + lx] == c) s = 0;
        else if (xr && lb[(size_t)ly * Wl + lx + 1] == c) s = 1;
        else if (yb && lb[(size_t)(ly + 1) * Wl + lx] == c) s = 2;
        else if (xr && yb && lb[(size_t)(ly + 1) * Wl + lx + 1] == c) s = 3;
        sel[i] = (uint8_t)s;
    }
}

// ransac_voting_layer_all_masks front end (ransac_voting.py:276-301): object masks [b,h,w,oc] (float, > 0.5 = inside) -> uint8 label map
// (the highest such object index + 1, 0 = none) and the per-(image, object) pixel counts the sub-sampling rule needs, in one pass
__global__ __launch_bounds__(256) void mask_to_labels_kernel(const float* __restrict__ mask, int hw, int oc, uint8_t* __restrict__ labels,
                                                             int* __restrict__ counts) {
    __shared__ int cnt[256];
    const int n = blockIdx.y;
    if (threadIdx.x < oc) cnt[threadIdx.x] = 0;
    __syncthreads();
    const float* m = mask + (size_t)n * hw * oc;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
        int lab = 0;
        for (int k = 0; k < oc; ++k) {
            const bool in = m[(size_t)p * oc + k] > 0.5f;
            const unsigned long long b = __builtin_amdgcn_ballot_w64(in);
            if (in) lab = k + 1;
            if (b && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(b)) atomicAdd(&cnt[k], (int)__popcll(b));
        }
        labels[(size_t)n * hw + p] = (uint8_t)lab;
    }
    __syncthreads();
    if (threadIdx.x < oc && cnt[threadIdx.x]) atomicAdd(&counts[n * oc + threadIdx.x], cnt[threadIdx.x]);
}

}  // namespace

extern "C" int cp_pad_channels_3to4(const float* src, float* dst, long long pixels, void* stream) {
    CP_REQUIRE(src && dst && pixels > 0, "cp_pad_channels_3to4: bad arguments");
    CP_LAUNCH(pad3to4_kernel, dim3(grid_for(pixels)), dim3(THREADS), 0, (hipStream_t)stream, src, dst, pixels);
    return cp::check_launch("cp_pad_channels_3to4");
}

extern "C" int cp_maxpool3x3s2_f32(const float* src, int batch, int h, int w, int channels, const float* scale,
                                   const float* shift, int relu, float* dst, void* stream) {
    CP_REQUIRE(src && dst && batch > 0 && h > 0 && w > 0, "cp_maxpool3x3s2_f32: bad arguments");
    CP_REQUIRE(channels % 4 == 0, "cp_maxpool3x3s2_f32: channels must be a multiple of 4");
    CP_REQUIRE((scale == nullptr) == (shift == nullptr), "cp_maxpool3x3s2_f32: scale/shift come together");
    int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    long long total = (long long)batch * ho * wo * (channels / 4);
    CP_LAUNCH(maxpool_kernel, dim3(grid_for(total)), dim3(THREADS), 0, (hipStream_t)stream, src, batch, h, w,
                       channels, ho, wo, scale, shift, relu, dst);
    return cp::check_launch("cp_maxpool3x3s2_f32");
}

extern "C" int cp_upsample_bilinear_x2_f32(const float* src, int batch, int h, int w, int channels, float* dst, void* stream) {
    CP_REQUIRE(src && dst && batch > 0 && h > 0 && w > 0 && channels % 4 == 0, "cp_upsample_bilinear_x2_f32: bad arguments");
    long long total = (long long)batch * 4 * h * w * (channels / 4);
    CP_LAUNCH(bilinear_x2_kernel, dim3(grid_for(total)), dim3(THREADS), 0, (hipStream_t)stream, src, batch, h, w, channels, dst);
    return cp::check_launch("cp_upsample_bilinear_x2_f32");
}

extern "C" int cp_guided_upsample_x2_f32(const float* src, const uint8_t* sel, int batch, int h, int w, int channels,
                                         float* dst, void* stream) {
    CP_REQUIRE(src && sel && dst && batch > 0 && h > 0 && w > 0 && channels % 4 == 0, "cp_guided_upsample_x2_f32: bad arguments");
    long long total = (long long)batch * 4 * h * w * (channels / 4);
    CP_LAUNCH(guided_x2_kernel, dim3(grid_for(total)), dim3(THREADS), 0, (hipStream_t)stream, src, sel, batch, h, w, channels, dst);
    return cp::check_launch("cp_guided_upsample_x2_f32");
}

extern "C" int cp_mask_to_labels_f32(const float* mask, int batch, int h, int w, int objects, uint8_t* labels, int32_t* counts, void* stream) {
    CP_REQUIRE(mask && labels && counts && batch > 0 && h > 0 && w > 0 && objects > 0 && objects < 255, "cp_mask_to_labels_f32: bad arguments");
    CP_REQUIRE((long long)h * w < (1LL << 31), "cp_mask_to_labels_f32: image too large");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)batch * objects, st) != hipSuccess) return cp::check_launch("cp_mask_to_labels_f32 memset");
    const int hw = h * w;
    int gx = (hw + 255) / 256;
    if (gx > 512) gx = 512;
    CP_LAUNCH(mask_to_labels_kernel, dim3(gx, batch), dim3(256), 0, st, mask, hw, objects, labels, counts);
    return cp::check_launch("cp_mask_to_labels_f32");
}

extern "C" int cp_argmax_labels(const float* logits, int ld, int classes, long long pixels, uint8_t* labels, void* stream) {
    CP_REQUIRE(logits && labels && pixels > 0, "cp_argmax_labels: bad arguments");
    CP_REQUIRE(classes >= 1 && classes <= 255 && ld >= classes, "cp_argmax_labels: classes must be 1..255 and ld >= classes");
    const int nv = (classes + 3) / 4;
    if (ld % 4 == 0 && ((uintptr_t)logits & 15) == 0 && nv <= 3 && nv * 4 <= ld) {
        if (nv == 1) CP_LAUNCH(argmax_vec_kernel<1>, dim3(grid_for(pixels)), dim3(THREADS), 0, (hipStream_t)stream, logits, ld, classes, pixels, labels);
        else if (nv == 2) CP_LAUNCH(argmax_vec_kernel<2>, dim3(grid_for(pixels)), dim3(THREADS), 0, (hipStream_t)stream, logits, ld, classes, pixels, labels);
        else CP_LAUNCH(argmax_vec_kernel<3>, dim3(grid_for(pixels)), dim3(THREADS), 0, (hipStream_t)stream, logits, ld, classes, pixels, labels);
        return cp::check_launch("cp_argmax_labels");
    }
    CP_LAUNCH(argmax_kernel, dim3(grid_for(pixels)), dim3(THREADS), 0, (hipStream_t)stream, logits, ld, classes, pixels, labels);
    return cp::check_launch("cp_argmax_labels");
}

extern "C" int cp_label_pyramid(const uint8_t* labels0, int batch, int h, int w, uint8_t* const* labels,
                                float* const* pnorm, uint8_t* const* sel, void* stream) {
    CP_REQUIRE(labels0 && batch > 0 && h > 0 && w > 0, "cp_label_pyramid: bad arguments");
    CP_REQUIRE(labels, "cp_label_pyramid: labels[] array required (labels[1..3] receive the half-size maps)");
    hipStream_t st = (hipStream_t)stream;
    const uint8_t* lv[4] = {labels0, nullptr, nullptr, nullptr};
    int hs[4], ws[4];
    hs[0] = h; ws[0] = w;
    for (int l = 1; l < 4; ++l) { hs[l] = hs[l - 1] / 2; ws[l] = ws[l - 1] / 2; }
    for (int l = 1; l < 4; ++l) {
        if (!labels[l]) break;
        CP_REQUIRE(hs[l] > 0 && ws[l] > 0, "cp_label_pyramid: level %d is empty", l);
        long long total = (long long)batch * hs[l] * ws[l];
        CP_LAUNCH(half_labels_kernel, dim3(grid_for(total)), dim3(THREADS), 0, st, lv[l - 1], batch, hs[l - 1],
                           ws[l - 1], hs[l], ws[l], labels[l]);
        lv[l] = labels[l];
    }
    for (int l = 0; l < 4; ++l) {
        if (pnorm && pnorm[l]) {
            CP_REQUIRE(lv[l], "cp_label_pyramid: pnorm[%d] requested without labels[%d]", l, l);
            long long total = (long long)batch * hs[l] * ws[l];
            CP_LAUNCH(pnorm_kernel, dim3(grid_for(total)), dim3(THREADS), 0, st, lv[l], batch, hs[l], ws[l], pnorm[l]);
        }
        if (l < 3 && sel && sel[l]) {
            CP_REQUIRE(lv[l] && lv[l + 1], "cp_label_pyramid: sel[%d] needs labels[%d] and labels[%d]", l, l, l + 1);
            CP_REQUIRE(hs[l] == 2 * hs[l + 1] && ws[l] == 2 * ws[l + 1], "cp_label_pyramid: level %d size must be even", l);
            long long total = (long long)batch * hs[l] * ws[l];
            CP_LAUNCH(guided_sel_kernel, dim3(grid_for(total)), dim3(THREADS), 0, st, lv[l], lv[l + 1], batch, hs[l], ws[l], sel[l]);
        }
    }
    return cp::check_launch("cp_label_pyramid");
}
