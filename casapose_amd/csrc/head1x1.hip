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

int grid_for_groups(long long pixels) {
    const long long blocks = ((pixels + 31) / 32 + 3) / 4;
    return (int)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks));
}

}  // namespace

extern "C" int cp_head1x1_fwd_f32(const float* x, int ld_x, long long pixels, const float* w, int cout, float* out, int ld_out, void* stream) {
    CP_REQUIRE(x && w && out && pixels > 0 && cout > 0 && cout <= 32, "cp_head1x1_fwd_f32: bad arguments (32 input channels, 1 <= cout <= 32)");
    CP_REQUIRE(ld_x >= CIN && ld_x % 4 == 0 && ((uintptr_t)x & 15) == 0 && ld_out >= cout, "cp_head1x1_fwd_f32: x rows must be 16-byte aligned float4 rows of >= 32 channels");
    CP_LAUNCH(head_fwd_kernel, dim3(grid_for_groups(pixels)), dim3(256), 0, (hipStream_t)stream, x, ld_x, pixels, w, cout, out, ld_out);
    return cp::check_launch("cp_head1x1_fwd_f32");
}

extern "C" int cp_head1x1_dgrad_f32(const float* dy, int ld_dy, int dy_row_floats, long long pixels, const float* w, int cout, float* dx, int ld_dx,
                                    int accumulate, void* stream) {
    CP_REQUIRE(dy && w && dx && pixels > 0 && cout > 0 && cout <= 32 && ld_dy >= cout && dy_row_floats >= cout && ld_dx >= CIN, "cp_head1x1_dgrad_f32: bad arguments");
    if (dy_row_floats >= 32 && ld_dy % 4 == 0 && ((uintptr_t)dy & 15) == 0)
        CP_LAUNCH(head_dgrad_kernel<true>, dim3(grid_for_groups(pixels)), dim3(256), 0, (hipStream_t)stream, dy, ld_dy, pixels, w, cout, dx, ld_dx, accumulate);
    else
        CP_LAUNCH(head_dgrad_kernel<false>, dim3(grid_for_groups(pixels)), dim3(256), 0, (hipStream_t)stream, dy, ld_dy, pixels, w, cout, dx, ld_dx, accumulate);
    return cp::check_launch("cp_head1x1_dgrad_f32");
}

extern "C" int cp_head1x1_wgrad_f32(const float* x, int ld_x, const float* dy, int ld_dy, long long pixels, int cout, float* dw, int accumulate,
                                    void* stream) {
    CP_REQUIRE(x && dy && dw && pixels > 0 && cout > 0 && cout <= 32 && ld_x >= CIN && ld_dy >= cout, "cp_head1x1_wgrad_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate)
        if (hipMemsetAsync(dw, 0, sizeof(float) * CIN * cout, st) != hipSuccess) return cp::check_launch("cp_head1x1_wgrad_f32 memset");
    int blocks = grid_for_groups(pixels);
    if (blocks > 1024) blocks = 1024;
    CP_LAUNCH(head_wgrad_kernel, dim3(blocks), dim3(256), 0, st, x, ld_x, dy, ld_dy, pixels, cout, dw);
    return cp::check_launch("cp_head1x1_wgrad_f32");
}
