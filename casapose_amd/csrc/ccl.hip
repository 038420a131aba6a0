// Largest-connected-component filter of CoordLSVotingWeighted (filter_estimates=True):
// casapose/pose_estimation/voting_layers_2d.py:43-79, where the reference calls
// tfa.image.connected_components once per (image, object) map.
//
// Here ONE union-find (tile-local in LDS, then stitched along tile borders) labels the multi-class map (two 4-neighbours are connected iff they
// carry the same non-zero label), which yields every object's components at once.  Roots are
// the minimum linear index of a component, so "component id order" of the reference (raster
// order of the first pixel) is the order of the roots.
//
// Selection rule restated from :64-76 (including its quirks): per (image, object) build the
// histogram {bin 0 = all pixels not in the object, bin i = i-th component}, zero every bin below
// `min_size`, sort by (count desc, index asc) and keep the SECOND entry.  So: the largest
// component survives when the rest of the image is larger than it; if every component is below
// the threshold the first one in raster order survives; an object larger than the rest of the
// image is dropped.
#include "common.h"

namespace {

constexpr int THREADS = 256;

inline int grid_for(long long n) {
    long long b = (n + THREADS - 1) / THREADS;
    return (int)(b < 1 ? 1 : (b > 256 * 16 ? 256 * 16 : b));
}

__device__ __forceinline__ int find_root(const int* parent, int x) {
    int p = parent[x];
    while (p != x) {
        x = p;
        p = parent[x];
    }
    return x;
}

__device__ __forceinline__ void unite(int* parent, int a, int b) {
    while (true) {
        a = find_root(parent, a);
        b = find_root(parent, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }  // a > b: hang the larger root under the smaller
        int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ void ccl_init(int* __restrict__ count, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) count[i] = 0;
}

// Pass 1: union-find inside a 64x16 tile held in LDS (LDS atomics are ~10x cheaper than the global
// ones and never contend across blocks), then one global parent per pixel = its tile-local root.
constexpr int TW = 64, TH = 16;
__global__ __launch_bounds__(256) void ccl_local(const uint8_t* __restrict__ lab, int* __restrict__ parent, int H, int W, int tiles_x,
                                                 int tiles_y) {
    __shared__ uint8_t sl[TH][TW];
    __shared__ int sp[TH * TW];
    const int t = blockIdx.x;
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
    const size_t img = (size_t)n * H * W;
    for (int i = threadIdx.x; i < TH * TW; i += 256) {
        const int ly = i / TW, lx = i % TW;
        const int y = y0 + ly, x = x0 + lx;
        const uint8_t l = (y < H && x < W) ? lab[img + (size_t)y * W + x] : 0;
        sl[ly][lx] = l;
        sp[i] = l ? i : -1;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TH * TW; i += 256) {
        const int ly = i / TW, lx = i % TW;
        const uint8_t l = sl[ly][lx];
        if (!l) continue;
        if (lx > 0 && sl[ly][lx - 1] == l) unite(sp, i, i - 1);
        if (ly > 0 && sl[ly - 1][lx] == l) unite(sp, i, i - TW);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TH * TW; i += 256) {
        const int ly = i / TW, lx = i % TW;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= H || x >= W) continue;
        int r = -1;
        if (sl[ly][lx]) {
            const int lr = find_root(sp, i);
            r = (int)(img + (size_t)(y0 + lr / TW) * W + x0 + lr % TW);
        }
        parent[img + (size_t)y * W + x] = r;
    }
}

// Pass 2: stitch the tiles along their borders (global union-find, ~8 % of the pixels)
__global__ void ccl_border(const uint8_t* __restrict__ lab, int* __restrict__ parent, int H, int W, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int l = lab[i];
        if (!l) continue;
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        if (x > 0 && (x % TW) == 0 && lab[i - 1] == l) unite(parent, (int)i, (int)i - 1);
        if (y > 0 && (y % TH) == 0 && lab[i - W] == l) unite(parent, (int)i, (int)i - W);
    }
}

__global__ void ccl_flatten_count(const uint8_t* __restrict__ lab, int* __restrict__ parent, int* __restrict__ count, long long total) {
    // Sizes: one atomic per (wave, distinct root) instead of one per pixel -- a large component
    // would otherwise serialise hundreds of thousands of atomics on one address.
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n_iter = (total + stride - 1) / stride;
    const int lane = threadIdx.x & 63;
    for (long long it = 0; it < n_iter; ++it) {
        const long long i = it * stride + blockIdx.x * (long long)blockDim.x + threadIdx.x;
        int r = -1;
        if (i < total && lab[i]) {
            r = find_root(parent, (int)i);
            parent[i] = r;
        }
        unsigned long long todo = __ballot(r >= 0);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const int r0 = __shfl(r, leader);
            const unsigned long long same = __ballot(r == r0);
            if (lane == leader) atomicAdd(&count[r0], (int)__popcll(same));
            todo &= ~same;
        }
    }
}

// per (image, object): [0] foreground pixels, [1] best key, [2] second-best key
__global__ void ccl_zero_stats(unsigned long long* __restrict__ stats, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stats[i] = 0ull;
}

__global__ void ccl_fg_count(const uint8_t* __restrict__ lab, const int* __restrict__ parent, const int* __restrict__ count,
                             unsigned long long* __restrict__ stats, int objects, long long hw, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int l = lab[i];
        if (l && parent[i] == (int)i) atomicAdd(&stats[((i / hw) * objects + (l - 1)) * 3 + 0], (unsigned long long)count[i]);
    }
}

// key: (thresholded count << 32) | tie-break, larger key = earlier in the reference's top_k order.
// bin 0 gets tie-break 0xFFFFFFFF (index 0 wins ties), component with root r gets 0xFFFFFFFE - local r.
__device__ __forceinline__ unsigned long long comp_key(int cnt, int min_size, unsigned tie) {
    const unsigned c = cnt < min_size ? 0u : (unsigned)cnt;
    return ((unsigned long long)c << 32) | tie;
}

__global__ void ccl_best(const uint8_t* __restrict__ lab, const int* __restrict__ parent, const int* __restrict__ count,
                         unsigned long long* __restrict__ stats, int objects, long long hw, long long total, int min_size, int pass) {
    const long long nstat = (total / hw) * objects;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total + nstat; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long key;
        long long s;
        if (i < total) {
            const int l = lab[i];
            if (!l || parent[i] != (int)i) continue;
            s = (i / hw) * objects + (l - 1);
            key = comp_key(count[i], min_size, 0xFFFFFFFEu - (unsigned)(i % hw));
        } else {  // bin 0 of (image, object) s
            s = i - total;
            const long long bg = hw - (long long)stats[s * 3 + 0];
            key = comp_key((int)bg, min_size, 0xFFFFFFFFu);
        }
        if (pass == 0) atomicMax(&stats[s * 3 + 1], key);
        else if (key < stats[s * 3 + 1]) atomicMax(&stats[s * 3 + 2], key);
    }
}

__global__ void ccl_write(const uint8_t* __restrict__ lab, const int* __restrict__ parent, const unsigned long long* __restrict__ stats,
                          int objects, long long hw, long long total, uint8_t* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int l = lab[i];
        uint8_t o = 0;
        if (l) {
            const unsigned long long second = stats[((i / hw) * objects + (l - 1)) * 3 + 2];
            const unsigned tie = (unsigned)(second & 0xFFFFFFFFull);
            // `second` is 0 when the histogram had a single bin (never for l > 0: its own component exists);
            // tie 0xFFFFFFFF = bin 0 was second -> nothing of this object is kept
            if (second != 0ull && tie != 0xFFFFFFFFu) {
                const long long root_local = (long long)(0xFFFFFFFEu - tie);
                if ((long long)parent[i] == (i / hw) * hw + root_local) o = (uint8_t)l;
            }
        }
        out[i] = o;
    }
}

}  // namespace

extern "C" size_t cp_ccl_workspace_bytes(int batch, int h, int w, int objects) {
    size_t px = (size_t)batch * h * w;
    return px * sizeof(int) * 2 + (size_t)batch * objects * 3 * sizeof(unsigned long long) + 64;
}

extern "C" int cp_ccl_filter_labels(const uint8_t* labels_in, int batch, int h, int w, int objects, int min_size, void* ws,
                                    uint8_t* labels_out, void* stream) {
    CP_REQUIRE(labels_in && labels_out && ws, "cp_ccl_filter_labels: null pointer");
    CP_REQUIRE(batch > 0 && h > 0 && w > 0 && objects > 0 && objects < 255, "cp_ccl_filter_labels: bad sizes");
    const long long hw = (long long)h * w, total = hw * batch;
    CP_REQUIRE(total < (1LL << 31) && hw < 0xFFFFFFF0LL, "cp_ccl_filter_labels: too many pixels for 32-bit component ids");
    hipStream_t st = (hipStream_t)stream;
    int* parent = reinterpret_cast<int*>(ws);
    int* count = parent + total;
    uintptr_t sp = (reinterpret_cast<uintptr_t>(count + total) + 63) & ~(uintptr_t)63;
    unsigned long long* stats = reinterpret_cast<unsigned long long*>(sp);
    const int nstat = batch * objects;
    const int g = grid_for(total);
    CP_LAUNCH(ccl_init, dim3(g), dim3(THREADS), 0, st, count, total);
    CP_LAUNCH(ccl_zero_stats, dim3((nstat * 3 + THREADS - 1) / THREADS), dim3(THREADS), 0, st, stats, nstat * 3);
    const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH;
    CP_LAUNCH(ccl_local, dim3(batch * tiles_x * tiles_y), dim3(256), 0, st, labels_in, parent, h, w, tiles_x, tiles_y);
    CP_LAUNCH(ccl_border, dim3(g), dim3(THREADS), 0, st, labels_in, parent, h, w, total);
    CP_LAUNCH(ccl_flatten_count, dim3(g), dim3(THREADS), 0, st, labels_in, parent, count, total);
    CP_LAUNCH(ccl_fg_count, dim3(g), dim3(THREADS), 0, st, labels_in, parent, count, stats, objects, hw, total);
    CP_LAUNCH(ccl_best, dim3(g), dim3(THREADS), 0, st, labels_in, parent, count, stats, objects, hw, total, min_size, 0);
    CP_LAUNCH(ccl_best, dim3(g), dim3(THREADS), 0, st, labels_in, parent, count, stats, objects, hw, total, min_size, 1);
    CP_LAUNCH(ccl_write, dim3(g), dim3(THREADS), 0, st, labels_in, parent, stats, objects, hw, total, labels_out);
    return cp::check_launch("cp_ccl_filter_labels");
}
