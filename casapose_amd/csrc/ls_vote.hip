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

namespace {

constexpr int ROWS = 16;        // rows per strip
constexpr int MAXKP = 9;        // compile-time bound for the private accumulators
constexpr int WAVES = 4;

template <int KP>
__global__ __launch_bounds__(256) void ls_accumulate_kernel(const float* __restrict__ field, int ld, int seg_off,
                                                            int dir_off, int conf_off, const uint8_t* __restrict__ labels,
                                                            int B, int H, int W, int objects, double* __restrict__ sums,
                                                            int strips_x, int strips_y) {
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
            if (lab > 0) {
                const float cy = ((float)y + 0.5f) / (float)H;
#pragma unroll
                for (int j = 0; j < KP; ++j) {
                    float dy = px[dir_off + 2 * j], dx = px[dir_off + 2 * j + 1];
                    float cf = px[conf_off + j];
                    float w = fmaxf(cf, 0.f) + log1pf(expf(-fabsf(cf)));  // softplus (:35)
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

}  // namespace

extern "C" size_t cp_ls_vote_workspace_bytes(int batch, int objects, int kp) {
    return (size_t)batch * objects * kp * 5 * sizeof(double);
}

extern "C" int cp_ls_vote_f32(const float* field, int ld, int seg_off, int dir_off, int conf_off, const uint8_t* labels,
                              int batch, int h, int w, int objects, int kp, double* sums_ws, float* keypoints,
                              void* stream) {
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
    CP_LAUNCH((ls_accumulate_kernel<MAXKP>), dim3(batch * blocks_per_img), dim3(256), lds, st, field, ld, seg_off,
                       dir_off, conf_off, labels, batch, h, w, objects, sums_ws, strips_x, strips_y);
    int total = batch * objects * kp;
    CP_LAUNCH(ls_solve_kernel, dim3((total + 255) / 256), dim3(256), 0, st, sums_ws, total, h, keypoints);
    return cp::check_launch("cp_ls_vote_f32");
}
