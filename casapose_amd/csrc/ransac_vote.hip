// PVNet-style RANSAC keypoint voting for all (image, object) pairs of a batch:
// ransac_voting_layer_all_masks -> ransac_voting_batch
// (casapose/pose_estimation/ransac_voting.py:276-368,447-484; hypothesis generation :197-227,
// inlier test :230-249).
//
// The reference serialises B x objects calls under tf.map_fn and materialises a
// [hyp, tn, kp] tensor per round.  Here:
//   1. the object's pixels are compacted in RASTER order (same order as tf.where, so injected
//      pixel-pair indices mean the same pixels) with a row-count scan + wave ballots;
//   2. hypotheses = two-ray intersections of the drawn pixel pairs;
//   3. voting: a lane owns one pixel (coalesced gather of its 18 direction floats), the
//      hypotheses of the block are broadcast from LDS, and every (hypothesis, keypoint) count
//      is the popcount of the wave's inlier ballot -- no per-pixel x hypothesis tensor exists;
//   4. per round: arg-max per keypoint, best-so-far update, the reference's stopping rule
//      1-(1-r_min^2)^hyps > confidence, evaluated on the device (finished objects skip the
//      remaining rounds);
//   5. refinement: fp64 normal-equation sums over the winners' inliers, 2x2 solve, with the
//      reference's "all keypoints invertible, cond < 1e6, else return the winners" rule.
// Random numbers: either the caller supplies 31-bit uniform draws (cp_ransac_vote_f32: tests inject them) or the library makes them from a 64-bit
// seed with a counter-based generator (cp_ransac_vote_seeded_f32, round 4: no draw tensor -- 94 MB per call at the reference's settings -- and the
// random thinning of objects above max_num pixels, ransac_voting.py:295-301, happens inside the compaction); a draw d selects pixel d % tn.
#include "common.h"

#include <type_traits>

namespace {

constexpr int KP = 9;
constexpr int HB = 16;  // granularity of the hypotheses per round (a voting block takes 16 or, when the count allows, 64 of them)

struct ObjState {        // one per (image, object)
    int tn;              // pixels (after min_num gate: 0 if skipped)
    int done;            // stopping rule reached
    int rounds;
    float hyp_num;
    float win_ratio[KP];
    float win_pts[KP][2];
};

// counter-based random numbers (splitmix64 finaliser of seed + counter * golden ratio): 64 well-mixed bits per (seed, counter), no state
__device__ __forceinline__ unsigned long long mix64(unsigned long long seed, unsigned long long ctr) {
    unsigned long long z = seed + (ctr + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// thinning predicate of one pixel: kept with probability thr / 2^32 (thr = 0xffffffff: always); the same decision in both compaction passes
__device__ __forceinline__ bool thin_keep(const unsigned* __restrict__ thr, size_t io, unsigned long long seed, unsigned long long pixel_id) {
    if (thr == nullptr) return true;
    const unsigned t = thr[io];
    return t == 0xffffffffu || (unsigned)(mix64(seed ^ 0x7468696E6E696E67ull, pixel_id) >> 32) < t;
}

// ---- 1. compaction -------------------------------------------------------------------------
__global__ void rowcount_kernel(const uint8_t* __restrict__ lab, int B, int H, int W, int objects, int* __restrict__ rowcnt,
                                const unsigned* __restrict__ thr, unsigned long long seed) {
    // one wave per (image,row); rowcnt[(img*objects+o)*H + y]
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B * H) return;
    const int img = row / H, y = row % H;
    const uint8_t* p = lab + (size_t)row * W;
    for (int o = 1 + 0; o <= objects; ++o) {
        int c = 0;
        for (int x0 = 0; x0 < W; x0 += 64) {
            int x = x0 + lane;
            bool m = x < W && p[x] == o && thin_keep(thr, (size_t)img * objects + (o - 1), seed, ((unsigned long long)img * H + y) * W + x);
            c += __popcll(__ballot(m));
        }
        if (lane == 0) rowcnt[((size_t)img * objects + (o - 1)) * H + y] = c;
    }
}

// thinning thresholds: one wave per (image, object) sums the row counts; above max_num pixels every pixel is kept with probability max_num / count
// (ransac_voting.py:295-301: selection = uniform < max_num / foreground_num)
// The min_num gate belongs to the UN-thinned count (ransac_voting.py:290-292 comes before :295-301): an object below it gets threshold 0 (no pixel
// survives, rowscan_kernel then sees zero pixels), and the seeded entry runs rowscan_kernel with a gate of one pixel.
__global__ void thin_threshold_kernel(const int* __restrict__ rowcnt, int H, int n_obj_total, int min_num, int max_num, unsigned* __restrict__ thr) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n_obj_total) return;
    int c = 0;
    for (int y = lane; y < H; y += 64) c += rowcnt[(size_t)i * H + y];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if (lane == 0) thr[i] = c < min_num ? 0u : (c > max_num ? (unsigned)((double)max_num / (double)c * 4294967296.0) : 0xffffffffu);
}

__global__ void rowscan_kernel(int* __restrict__ rowcnt, int H, int n_obj_total, int min_num, ObjState* __restrict__ st) {
    // one WAVE per (image, object): exclusive scan over rows, in place, 64 rows per step (a thread per object walked the rows one by one: 64 us)
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n_obj_total) return;
    int* r = rowcnt + (size_t)i * H;
    int run = 0;
    for (int y0 = 0; y0 < H; y0 += 64) {
        const int y = y0 + lane;
        const int c = y < H ? r[y] : 0;
        int x = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(x, off);
            if (lane >= off) x += t;
        }
        if (y < H) r[y] = run + x - c;
        run += __shfl(x, 63);
    }
    if (lane != 0) return;
    ObjState s;
    s.tn = (run < min_num) ? 0 : run;  // ransac_voting.py:290-292
    s.done = s.tn == 0;
    s.rounds = 0;
    s.hyp_num = 0.f;
    for (int v = 0; v < KP; ++v) {
        s.win_ratio[v] = 0.f;
        s.win_pts[v][0] = s.win_pts[v][1] = 0.f;
    }
    st[i] = s;
}

__global__ void compact_kernel(const uint8_t* __restrict__ lab, int B, int H, int W, int objects, const int* __restrict__ rowstart,
                               int* __restrict__ pixlist, int list_stride, const unsigned* __restrict__ thr, unsigned long long seed) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B * H) return;
    const int img = row / H, y = row % H;
    const uint8_t* p = lab + (size_t)row * W;
    for (int o = 1; o <= objects; ++o) {
        const size_t io = (size_t)img * objects + (o - 1);
        int base = rowstart[io * H + y];
        for (int x0 = 0; x0 < W; x0 += 64) {
            int x = x0 + lane;
            bool m = x < W && p[x] == o && thin_keep(thr, io, seed, ((unsigned long long)img * H + y) * W + x);
            unsigned long long bal = __ballot(m);
            if (m) {
                int rank = __popcll(bal & ((1ull << lane) - 1ull));
                pixlist[io * list_stride + base + rank] = y * W + x;
            }
            base += __popcll(bal);
        }
    }
}

// ---- 2. hypotheses ----------------------------------------------------------------------------
__device__ __forceinline__ void pixel_record(const float* __restrict__ vertex, int ld, int dir_off, int img, int H, int W, int pix, int v,
                                             float& cx, float& cy, float& dx, float& dy) {
    const int y = pix / W, x = pix - y * W;
    cx = (float)x + 0.5f;  // coords are (x, y) + 0.5 (:303-306)
    cy = (float)y + 0.5f;
    const float* p = vertex + ((size_t)img * H * W + pix) * ld + dir_off + 2 * v;
    dy = p[0];  // the field stores (dy, dx); the voter works in (dx, dy) (:308)
    dx = p[1];
}

__global__ void hypgen_kernel(const float* __restrict__ vertex, int ld, int dir_off, int H, int W, int objects, const int* __restrict__ pixlist,
                              int list_stride, const int32_t* __restrict__ draws, int hyp, const ObjState* __restrict__ st,
                              float* __restrict__ hyp_pts, int n_obj_total, int* __restrict__ counts, unsigned long long seed, int round) {
    // one thread per (image*object, h, v)
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long per = (long long)hyp * KP;
    if (i >= per * n_obj_total) return;
    const int io = (int)(i / per);
    const int hv = (int)(i % per);
    const ObjState& s = st[io];
    if (s.done) return;
    counts[i] = 0;   // this round's inlier counter of (object, hypothesis, keypoint): same index space (a separate zeroing launch before)
    const int img = io / objects;
    const int v = hv % KP;
    unsigned d0, d1;
    if (draws) {   // injected 31-bit draws
        const int32_t* d = draws + ((size_t)io * per + hv) * 2;
        d0 = (unsigned)d[0];
        d1 = (unsigned)d[1];
    } else {       // one 64-bit number per (round, object, hypothesis, keypoint): two 31-bit draws
        const unsigned long long z = mix64(seed, (unsigned long long)round * (unsigned long long)(per * n_obj_total) + (unsigned long long)i);
        d0 = (unsigned)(z >> 33);
        d1 = (unsigned)(z & 0x7fffffffull);
    }
    const int* pl = pixlist + (size_t)io * list_stride;
    const int p0 = pl[d0 % (unsigned)s.tn], p1 = pl[d1 % (unsigned)s.tn];
    float c0x, c0y, d0x, d0y, c1x, c1y, d1x, d1y;
    pixel_record(vertex, ld, dir_off, img, H, W, p0, v, c0x, c0y, d0x, d0y);
    pixel_record(vertex, ld, dir_off, img, H, W, p1, v, c1x, c1y, d1x, d1y);
    // generate_hypothesis (:219-226)
    const float det = d1x * d0y - d1y * d0x;
    const float u = ((c1y - c0y) * d1x - (c1x - c0x) * d1y) / det;
    float hx = c0x + d0x * u, hy = c0y + d0y * u;
    if (!(fabsf(det) > 1e-6f)) hx = hy = 0.f;
    hyp_pts[((size_t)io * per + hv) * 2 + 0] = hx;
    hyp_pts[((size_t)io * per + hv) * 2 + 1] = hy;
}

// ---- 3. voting --------------------------------------------------------------------------------
// the reference's expression (voting_for_hypothesis, :236-247) as written; used by the refinement pass over the winners' inliers
__device__ __forceinline__ bool inlier_test(float dx, float dy, float nd, float cx, float cy, float hx, float hy, float thresh) {
    const float ex = hx - cx, ey = hy - cy;
    const float nh = sqrtf(ex * ex + ey * ey);
    const bool valid = nd > 1e-6f && nh > 1e-6f && fabsf(hx + hy) > 1e-6f;
    const float ang = (dx * ex + dy * ey) / (nd * nh);
    return valid && ang > thresh;
}

// voting_for_hypothesis (:236-247): a pixel is an inlier of hypothesis h iff cos(angle between its direction d and the ray e = h - c) >
// thresh, with |d| > 1e-6, |e| > 1e-6 and |hx + hy| > 1e-6.  Almost every test is decided without the square root and the division:
// with s = d.e the inequality s / (|d||e|) > t is s > 0 and s^2 > t^2 |d|^2 |e|^2 (t >= 0); only when the two sides agree to 1e-5 (far
// more than the rounding of either form) the reference's own expression is evaluated, so the decision is the reference's in every case.
// Per pixel (dx, dy, |d|, t^2|d|^2) are prepared once for all hypotheses (|d| <= 1e-6: d = 0 and t^2|d|^2 = 1, never an inlier); an invalid
// hypothesis is stored as NaN (every comparison false).
template <int HB>   // hypotheses per block: every pixel record a block gathers (18 uncoalesced loads per pixel) is used for HB x 9 tests
__global__ __launch_bounds__(256) void vote_kernel(const float* __restrict__ vertex, int ld, int dir_off, int H, int W, int objects,
                                                   const int* __restrict__ pixlist, int list_stride, const float* __restrict__ hyp_pts,
                                                   int hyp, const ObjState* __restrict__ st, int* __restrict__ counts, float thresh,
                                                   int px_chunks) {
    // grid: (hyp / HB, px_chunks, image*object).  A block votes HB hypotheses x 9 keypoints over one chunk of pixels: a lane per pixel, the
    // wave's inlier count of a (hypothesis, keypoint) pair is popcount(ballot) and lands in lane `hypothesis` of counter register `keypoint`.
    __shared__ float2 hp[HB * KP];
    __shared__ int cnt[HB * KP];
    const int io = blockIdx.z;
    const ObjState& s = st[io];
    if (s.done) return;
    const int img = io / objects;
    const int h0 = blockIdx.x * HB;
    const int tid = threadIdx.x, lane = tid & 63;
    int near = 0;   // some hypothesis point within 1e-5 of a pixel centre (x + 0.5, y + 0.5)?  Only then can |e| ~ 0 occur for a pixel
    for (int i = tid; i < HB * KP; i += blockDim.x) {
        float hx = hyp_pts[(((size_t)io * hyp + h0) * KP + i) * 2], hy = hyp_pts[(((size_t)io * hyp + h0) * KP + i) * 2 + 1];
        if (!(fabsf(hx + hy) > 1e-6f)) hx = hy = __builtin_nanf("");
        hp[i] = make_float2(hx, hy);
        cnt[i] = 0;
        const float fx = hx - 0.5f - rintf(hx - 0.5f), fy = hy - 0.5f - rintf(hy - 0.5f);
        near |= (fabsf(fx) < 1e-5f && fabsf(fy) < 1e-5f) ? 1 : 0;
    }
    const bool check_e2 = __syncthreads_or(near) != 0;
    const int* pl = pixlist + (size_t)io * list_stride;
    const int per_chunk = (s.tn + px_chunks - 1) / px_chunks;
    const int t_begin = blockIdx.y * per_chunk, t_end = min(s.tn, t_begin + per_chunk);
    const bool exact_only = !(thresh >= 0.f);  // the squared form needs t >= 0
    const float t2 = thresh * thresh;
    int cntv[KP];  // lane hh: inliers of (hypothesis hh, keypoint v) seen by this wave
#pragma unroll
    for (int v = 0; v < KP; ++v) cntv[v] = 0;
    for (int t0 = t_begin + (tid & ~63); t0 < t_end; t0 += blockDim.x) {
        const int t = t0 + lane;
        const bool act = t < t_end;
        float cx = 0.f, cy = 0.f, dxv[KP], dyv[KP], ndv[KP], qv[KP];
#pragma unroll
        for (int v = 0; v < KP; ++v) { dxv[v] = dyv[v] = 0.f; ndv[v] = 1.f; qv[v] = 1.f; }
        if (act) {
            const int pix = pl[t];
            const int y = pix / W, x = pix - y * W;
            cx = (float)x + 0.5f;
            cy = (float)y + 0.5f;
            const float* p = vertex + ((size_t)img * H * W + pix) * ld + dir_off;
#pragma unroll
            for (int v = 0; v < KP; ++v) {
                const float dy = p[2 * v], dx = p[2 * v + 1];
                const float nd = sqrtf(dx * dx + dy * dy);
                if (nd > 1e-6f) { dxv[v] = dx; dyv[v] = dy; ndv[v] = nd; qv[v] = t2 * (nd * nd); }
            }
        }
        auto vote_tile = [&](auto check_c) {
            constexpr bool CHECK_E2 = decltype(check_c)::value;
#pragma unroll 1
            for (int hh = 0; hh < HB; ++hh) {
                // nine tests of hypothesis hh.  The wave's inlier counts stay in scalar registers (__builtin_amdgcn_ballot_w64 takes the comparison
                // mask as it is; HIP's __ballot(int) goes through v_cndmask + v_cmp_ne), the borderline margin |d| - 1e-5 rhs is a running
                // minimum that is checked ONCE per hypothesis (round 1 branched inside every test), and the counts land in lane hh with nine
                // adds under one exec mask.  |e|^2 < 1.1e-12 needs a hypothesis within 1e-6 of this pixel's centre: tested only in blocks that
                // hold such a hypothesis (CHECK_E2).
                int c[KP];
                float zmin = 1.f, e2min = 1.f;
#pragma unroll
                for (int v = 0; v < KP; ++v) {
                    const float2 h = hp[hh * KP + v];
                    const float ex = h.x - cx, ey = h.y - cy;
                    const float e2 = ex * ex + ey * ey;
                    const float sd = dxv[v] * ex + dyv[v] * ey;
                    const float rhs = qv[v] * e2;
                    const float d = __builtin_fmaf(sd, __builtin_fabsf(sd), -rhs);   // > 0  <=>  s > 0 and s^2 > t^2 |d|^2 |e|^2
                    // (no `&& act`: a lane without a pixel has d = 0 and t^2 |d|^2 = 1, i.e. d = -|e|^2 < 0 for every valid hypothesis, and NaN for an
                    //  invalid one -- the comparison is false by itself; the explicit mask cost a v_cndmask + v_cmp per test, 14 % of the loop's VALU)
                    c[v] = __popcll(__builtin_amdgcn_ballot_w64(d > 0.f));
                    zmin = fminf(zmin, __builtin_fmaf(-1e-5f, rhs, __builtin_fabsf(d)));   // NaN (invalid hypothesis) leaves the minimum alone
                    if constexpr (CHECK_E2) e2min = fminf(e2min, e2);
                }
                const bool suspect = (zmin <= 0.f || (CHECK_E2 && e2min < 1.1e-12f)) && act;
                if (exact_only || __builtin_amdgcn_ballot_w64(suspect)) {   // rare: those lanes take the reference's own expression
#pragma unroll
                    for (int v = 0; v < KP; ++v) {
                        const float2 h = hp[hh * KP + v];
                        const float ex = h.x - cx, ey = h.y - cy;
                        const float e2 = ex * ex + ey * ey;
                        const float sd = dxv[v] * ex + dyv[v] * ey;
                        const float rhs = qv[v] * e2;
                        const float d = __builtin_fmaf(sd, __builtin_fabsf(sd), -rhs);
                        const bool bl = (exact_only || __builtin_fabsf(d) <= 1e-5f * rhs || e2 < 1.1e-12f) && act;
                        const float nh = sqrtf(e2);
                        const bool in = nh > 1e-6f && (sd / (ndv[v] * nh)) > thresh && (dxv[v] != 0.f || dyv[v] != 0.f);
                        c[v] += __popcll(__builtin_amdgcn_ballot_w64(bl && in)) - __popcll(__builtin_amdgcn_ballot_w64(bl && d > 0.f));
                    }
                }
                if (lane == hh) {
#pragma unroll
                    for (int v = 0; v < KP; ++v) cntv[v] += c[v];
                }
            }
        };
        if (check_e2) vote_tile(std::true_type{});
        else vote_tile(std::false_type{});
    }
    if (lane < HB) {
#pragma unroll
        for (int v = 0; v < KP; ++v)
            if (cntv[v]) atomicAdd(&cnt[lane * KP + v], cntv[v]);
    }
    __syncthreads();
    for (int i = tid; i < HB * KP; i += blockDim.x)
        if (cnt[i]) atomicAdd(&counts[((size_t)io * hyp + h0) * KP + i], cnt[i]);
}

// ---- 4. round update --------------------------------------------------------------------------
__global__ void update_kernel(const int* __restrict__ counts, const float* __restrict__ hyp_pts, int hyp, ObjState* __restrict__ st,
                              int n_obj_total, float confidence, int max_iter) {
    // one wave per (image, object); lane-parallel arg-max over hypotheses for each keypoint
    const int io = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (io >= n_obj_total) return;
    ObjState s = st[io];
    if (s.done) return;
    float min_ratio = 1e30f;
    for (int v = 0; v < KP; ++v) {
        int best = -1, besth = 0x7fffffff;
        for (int h = lane; h < hyp; h += 64) {
            int c = counts[((size_t)io * hyp + h) * KP + v];
            if (c > best) { best = c; besth = h; }  // first maximum within the lane's stride
        }
        for (int off = 32; off; off >>= 1) {
            int ob = __shfl_xor(best, off), oh = __shfl_xor(besth, off);
            if (ob > best || (ob == best && oh < besth)) { best = ob; besth = oh; }  // tf.argmax: lowest index on ties
        }
        const float ratio = (float)best / (float)s.tn;  // :333
        if (s.win_ratio[v] < ratio) {                     // :336-338
            s.win_ratio[v] = ratio;
            s.win_pts[v][0] = hyp_pts[(((size_t)io * hyp + besth) * KP + v) * 2 + 0];
            s.win_pts[v][1] = hyp_pts[(((size_t)io * hyp + besth) * KP + v) * 2 + 1];
        }
        min_ratio = fminf(min_ratio, s.win_ratio[v]);
    }
    s.hyp_num += (float)hyp;
    s.rounds += 1;
    const float conf = 1.f - powf(1.f - min_ratio * min_ratio, s.hyp_num);  // :344-346
    if (conf > confidence || s.rounds >= max_iter) s.done = 1;
    if (lane == 0) st[io] = s;
}

__global__ void count_active_kernel(const ObjState* __restrict__ st, int n, int* __restrict__ out) {
    __shared__ int acc;
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    int a = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += st[i].done ? 0 : 1;
    if (a) atomicAdd(&acc, a);
    __syncthreads();
    if (threadIdx.x == 0) *out = acc;
}

// ---- 5. refinement ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void refine_accumulate_kernel(const float* __restrict__ vertex, int ld, int dir_off, int H, int W, int objects,
                                                                const int* __restrict__ pixlist, int list_stride,
                                                                const ObjState* __restrict__ st, double* __restrict__ sums, float thresh,
                                                                int px_chunks) {
    // grid (px_chunks, image*object); sums[io][v][5] = {a00, a01, a11, b0, b1}
    __shared__ double acc[KP * 5];
    const int io = blockIdx.y;
    const ObjState& s = st[io];
    if (s.tn == 0) return;
    const int img = io / objects;
    const int tid = threadIdx.x;
    for (int i = tid; i < KP * 5; i += blockDim.x) acc[i] = 0.0;
    __syncthreads();
    const int* pl = pixlist + (size_t)io * list_stride;
    const int per_chunk = (s.tn + px_chunks - 1) / px_chunks;
    const int t_begin = blockIdx.x * per_chunk, t_end = min(s.tn, t_begin + per_chunk);
    double a[KP][5];
#pragma unroll
    for (int v = 0; v < KP; ++v)
#pragma unroll
        for (int c = 0; c < 5; ++c) a[v][c] = 0.0;
    for (int t = t_begin + tid; t < t_end; t += blockDim.x) {
        const int pix = pl[t];
        const int y = pix / W, x = pix - y * W;
        const float cx = (float)x + 0.5f, cy = (float)y + 0.5f;
        const float* p = vertex + ((size_t)img * H * W + pix) * ld + dir_off;
#pragma unroll
        for (int v = 0; v < KP; ++v) {
            const float dy = p[2 * v], dx = p[2 * v + 1];
            const float nd = sqrtf(dx * dx + dy * dy);
            if (inlier_test(dx, dy, nd, cx, cy, s.win_pts[v][0], s.win_pts[v][1], thresh)) {
                const float nx = -dy, ny = dx;           // normal = reverse(direct * (1,-1)) (:349)
                const float b = nx * cx + ny * cy;       // (:359)
                a[v][0] += (double)(nx * nx);
                a[v][1] += (double)(nx * ny);
                a[v][2] += (double)(ny * ny);
                a[v][3] += (double)(nx * b);
                a[v][4] += (double)(ny * b);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < KP; ++v)
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            double x = a[v][c];
            for (int off = 32; off; off >>= 1) x += __shfl_xor(x, off);
            if ((tid & 63) == 0 && x != 0.0) atomicAdd(&acc[v * 5 + c], x);
        }
    __syncthreads();
    for (int i = tid; i < KP * 5; i += blockDim.x)
        if (acc[i] != 0.0) atomicAdd(&sums[(size_t)io * KP * 5 + i], acc[i]);
}

__global__ void refine_solve_kernel(const double* __restrict__ sums, const ObjState* __restrict__ st, int n_obj_total, float* __restrict__ out,
                                    int32_t* __restrict__ rounds_out) {
    const int io = blockIdx.x * blockDim.x + threadIdx.x;
    if (io >= n_obj_total) return;
    const ObjState& s = st[io];
    float* o = out + (size_t)io * KP * 2;
    if (rounds_out) rounds_out[io] = s.rounds;
    if (s.tn == 0) {
        for (int i = 0; i < KP * 2; ++i) o[i] = 0.f;
        return;
    }
    bool all_ok = true;
    double sol[KP][2];
    for (int v = 0; v < KP; ++v) {
        const double* q = sums + ((size_t)io * KP + v) * 5;
        // the reference forms ATA in fp32 (:361); singular values of the symmetric PSD 2x2
        const double a = (double)(float)q[0], b = (double)(float)q[1], c = (double)(float)q[2];
        const double ht = 0.5 * (a + c), hd = 0.5 * (a - c), rad = sqrt(hd * hd + b * b);
        const double l1 = fabs(ht + rad), l2 = fabs(ht - rad);
        const double smax = fmax(l1, l2), smin = fmin(l1, l2);
        const double cond = smax / smin;  // inf / nan when singular
        if (!(isfinite(cond) && cond < 1e6)) all_ok = false;  // is_invertible (:267-272), reduce_min over keypoints (:364)
        const double det = a * c - b * b;
        const double t0 = (double)(float)q[3], t1 = (double)(float)q[4];
        sol[v][0] = (c * t0 - b * t1) / det;
        sol[v][1] = (a * t1 - b * t0) / det;
    }
    for (int v = 0; v < KP; ++v) {
        o[2 * v + 0] = all_ok ? (float)sol[v][0] : s.win_pts[v][0];
        o[2 * v + 1] = all_ok ? (float)sol[v][1] : s.win_pts[v][1];
    }
}

struct Workspace {
    int* rowcnt;
    int* pixlist;
    ObjState* st;
    float* hyp_pts;
    int* counts;
    double* sums;
    int* nactive;
    unsigned* thr;
    size_t bytes;
};

Workspace carve(void* base, int batch, int h, int w, int objects, int hyp) {
    Workspace ws{};
    uintptr_t p = reinterpret_cast<uintptr_t>(base);
    auto take = [&](size_t n) {
        p = (p + 255) & ~(uintptr_t)255;
        uintptr_t r = p;
        p += n;
        return r;
    };
    const size_t no = (size_t)batch * objects;
    ws.rowcnt = reinterpret_cast<int*>(take(no * h * sizeof(int)));
    ws.pixlist = reinterpret_cast<int*>(take((size_t)batch * objects * h * w * sizeof(int)));
    ws.st = reinterpret_cast<ObjState*>(take(no * sizeof(ObjState)));
    ws.hyp_pts = reinterpret_cast<float*>(take(no * hyp * KP * 2 * sizeof(float)));
    ws.counts = reinterpret_cast<int*>(take(no * hyp * KP * sizeof(int)));
    ws.sums = reinterpret_cast<double*>(take(no * KP * 5 * sizeof(double)));
    ws.nactive = reinterpret_cast<int*>(take(sizeof(int)));
    ws.thr = reinterpret_cast<unsigned*>(take(no * sizeof(unsigned)));
    ws.bytes = p - reinterpret_cast<uintptr_t>(base) + 256;
    return ws;
}

}  // namespace

extern "C" size_t cp_ransac_workspace_bytes(int batch, int h, int w, int objects, int kp, int hyp) {
    (void)kp;
    return carve(nullptr, batch, h, w, objects, hyp).bytes;
}

namespace {
int ransac_vote(const uint8_t* labels, const float* vertex, int ld, int dir_off, int batch, int h, int w, int objects, int kp, const int32_t* idx,
                bool seeded, unsigned long long seed, int hyp, float inlier_thresh, float confidence, int max_iter, int min_num, int max_num, void* wsp,
                float* out, int32_t* rounds_out, void* stream);
}

extern "C" int cp_ransac_vote_f32(const uint8_t* labels, const float* vertex, int ld, int dir_off, int batch, int h, int w, int objects,
                                  int kp, const int32_t* idx, int hyp, float inlier_thresh, float confidence, int max_iter, int min_num,
                                  int max_num, void* wsp, float* out, int32_t* rounds_out, void* stream) {
    CP_REQUIRE(idx, "cp_ransac_vote_f32: null pointer");
    return ransac_vote(labels, vertex, ld, dir_off, batch, h, w, objects, kp, idx, false, 0ull, hyp, inlier_thresh, confidence, max_iter, min_num, max_num, wsp, out,
                       rounds_out, stream);
}

extern "C" int cp_ransac_vote_seeded_f32(const uint8_t* labels, const float* vertex, int ld, int dir_off, int batch, int h, int w, int objects,
                                         int kp, unsigned long long seed, int hyp, float inlier_thresh, float confidence, int max_iter, int min_num,
                                         int max_num, void* wsp, float* out, int32_t* rounds_out, void* stream) {
    CP_REQUIRE(max_num >= 1, "cp_ransac_vote_seeded_f32: max_num must be positive");
    return ransac_vote(labels, vertex, ld, dir_off, batch, h, w, objects, kp, nullptr, true, seed, hyp, inlier_thresh, confidence, max_iter, min_num, max_num, wsp, out,
                       rounds_out, stream);
}

namespace {
int ransac_vote(const uint8_t* labels, const float* vertex, int ld, int dir_off, int batch, int h, int w, int objects, int kp, const int32_t* idx,
                bool seeded, unsigned long long seed, int hyp, float inlier_thresh, float confidence, int max_iter, int min_num, int max_num, void* wsp,
                float* out, int32_t* rounds_out, void* stream) {
    CP_REQUIRE(labels && vertex && wsp && out, "cp_ransac_vote_f32: null pointer");
    CP_REQUIRE(kp == KP, "cp_ransac_vote_f32: built for %d keypoints (got %d)", KP, kp);
    CP_REQUIRE(batch > 0 && h > 0 && w > 0 && objects > 0 && objects < 255, "cp_ransac_vote_f32: bad sizes");
    CP_REQUIRE(hyp > 0 && hyp % HB == 0, "cp_ransac_vote_f32: hypotheses per round must be a multiple of %d", HB);
    CP_REQUIRE(max_iter >= 1 && dir_off >= 0 && dir_off + 2 * kp <= ld, "cp_ransac_vote_f32: bad max_iter / channel offsets");
    CP_REQUIRE((long long)h * w < (1LL << 31), "cp_ransac_vote_f32: image too large");
    // sub-sampling above max_num (:295-301) is random in the reference: with injected draws the caller applies it to `labels`; the seeded entry
    // thins inside the compaction (count, per-object keep probability, count and compact again with the same per-pixel decisions)
    hipStream_t st = (hipStream_t)stream;
    Workspace ws = carve(wsp, batch, h, w, objects, hyp);
    const int no = batch * objects;
    const int list_stride = h * w;
    const int rows = batch * h;
    (void)hipMemsetAsync(ws.sums, 0, (size_t)no * KP * 5 * sizeof(double), st);
    const unsigned* thr = nullptr;
    CP_LAUNCH(rowcount_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, labels, batch, h, w, objects, ws.rowcnt, thr, seed);
    if (seeded) {
        CP_LAUNCH(thin_threshold_kernel, dim3((no + 3) / 4), dim3(256), 0, st, ws.rowcnt, h, no, min_num, max_num, ws.thr);
        thr = ws.thr;
        CP_LAUNCH(rowcount_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, labels, batch, h, w, objects, ws.rowcnt, thr, seed);
    }
    CP_LAUNCH(rowscan_kernel, dim3((no + 3) / 4), dim3(256), 0, st, ws.rowcnt, h, no, seeded ? 1 : min_num, ws.st);   // (seeded: gated before the thinning, above)
    CP_LAUNCH(compact_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, labels, batch, h, w, objects, ws.rowcnt, ws.pixlist, list_stride, thr, seed);
    const int px_chunks = 8;
    const long long nhyp = (long long)no * hyp * KP;
    for (int r = 0; r < max_iter; ++r) {
        const int32_t* draws = idx ? idx + (size_t)r * nhyp * 2 : nullptr;
        CP_LAUNCH(hypgen_kernel, dim3((unsigned)((nhyp + 255) / 256)), dim3(256), 0, st, vertex, ld, dir_off, h, w, objects, ws.pixlist,
                  list_stride, draws, hyp, ws.st, ws.hyp_pts, no, ws.counts, seed, r);
        if (hyp % 64 == 0)
            CP_LAUNCH(vote_kernel<64>, dim3(hyp / 64, px_chunks, no), dim3(256), 0, st, vertex, ld, dir_off, h, w, objects, ws.pixlist, list_stride,
                      ws.hyp_pts, hyp, ws.st, ws.counts, inlier_thresh, px_chunks);
        else
            CP_LAUNCH(vote_kernel<HB>, dim3(hyp / HB, px_chunks, no), dim3(256), 0, st, vertex, ld, dir_off, h, w, objects, ws.pixlist, list_stride,
                      ws.hyp_pts, hyp, ws.st, ws.counts, inlier_thresh, px_chunks);
        CP_LAUNCH(update_kernel, dim3((no + 3) / 4), dim3(256), 0, st, ws.counts, ws.hyp_pts, hyp, ws.st, no, confidence, max_iter);
        // The stopping rule lives on the device; the remaining rounds of finished objects are empty launches (4 per round, ~25 us).  After
        // rounds 1, 2, 4, 8, 16 the host asks how many objects are still voting (one 4-byte copy + stream synchronisation, ~20 us) and stops
        // launching when none is: the usual case ends after one or two rounds of the reference's max_iter = 20.
        const int done_rounds = r + 1;
        if (done_rounds < max_iter && (done_rounds & (done_rounds - 1)) == 0) {
            CP_LAUNCH(count_active_kernel, dim3(1), dim3(256), 0, st, ws.st, no, ws.nactive);
            int active = 1;
            if (hipMemcpyAsync(&active, ws.nactive, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
                return cp::check_launch("cp_ransac_vote_f32 (round check)");
            if (active == 0) break;
        }
    }
    CP_LAUNCH(refine_accumulate_kernel, dim3(px_chunks, no), dim3(256), 0, st, vertex, ld, dir_off, h, w, objects, ws.pixlist, list_stride,
              ws.st, ws.sums, inlier_thresh, px_chunks);
    CP_LAUNCH(refine_solve_kernel, dim3((no + 63) / 64), dim3(64), 0, st, ws.sums, ws.st, no, out, rounds_out);
    return cp::check_launch("cp_ransac_vote_f32");
}
}  // namespace
