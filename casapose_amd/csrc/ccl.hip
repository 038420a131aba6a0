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

// Pass 1: union-find inside a 64x16 tile held in LDS.  A wave owns a tile row: the horizontal runs come from one ballot (every pixel starts
// with its run's first pixel as parent -- no atomics), and two rows are joined once per OVERLAP SEGMENT of two runs instead of once per
// pixel (blob-shaped masks: ~50x fewer LDS atomics than the per-pixel version).  Output: one global parent per pixel = its tile-local
// root; tile-local roots (the only pixels that can end up as component roots) get their size counter zeroed here.
constexpr int TW = 64, TH = 16;
__global__ __launch_bounds__(256) void ccl_local(const uint8_t* __restrict__ lab, int* __restrict__ parent, int* __restrict__ count, int H, int W,
                                                 int tiles_x, int tiles_y) {
    __shared__ uint8_t sl[TH][TW];
    __shared__ int sp[TH * TW];
    __shared__ unsigned long long same_left[TH];
    const int t = blockIdx.x;
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
    const size_t img = (size_t)n * H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < TH / 4; ++r) {
        const int ly = wave * (TH / 4) + r, y = y0 + ly, x = x0 + lane;
        const int l = (y < H && x < W) ? lab[img + (size_t)y * W + x] : 0;
        const int left = __shfl_up(l, 1);
        const bool sl_ = lane > 0 && l != 0 && l == left;
        const unsigned long long SL = __ballot(sl_), FG = __ballot(l != 0);
        const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
        const unsigned long long starts = FG & ~SL & upto;   // run starts at or left of this lane (non-empty when l != 0)
        sl[ly][lane] = (uint8_t)l;
        sp[ly * TW + lane] = l ? ly * TW + (63 - __builtin_clzll(starts | 1ull)) : -1;
        if (lane == 0) same_left[ly] = SL;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TH / 4; ++r) {
        const int ly = wave * (TH / 4) + r;
        if (ly == 0) continue;
        const int l = sl[ly][lane];
        const bool up_same = l != 0 && l == sl[ly - 1][lane];
        const unsigned long long U = __ballot(up_same);
        // first pixel of an overlap segment: joined to the row above, but not (left neighbour joined AND same run in both rows)
        const unsigned long long first = U & ~((U << 1) & same_left[ly] & same_left[ly - 1]);
        if ((first >> lane) & 1ull) unite(sp, sp[ly * TW + lane], sp[(ly - 1) * TW + lane]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TH * TW; i += 256) {
        const int ly = i / TW, lx = i % TW;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= H || x >= W) continue;
        int r = -1;
        const size_t gi = img + (size_t)y * W + x;
        if (sl[ly][lx]) {
            const int lr = find_root(sp, i);
            r = (int)(img + (size_t)(y0 + lr / TW) * W + x0 + lr % TW);
            if (lr == i) count[gi] = 0;
        }
        parent[gi] = r;
    }
}

// Pass 2: stitch the tiles along their borders -- one thread per border pixel (6 % of the image), one global union per overlap segment
__global__ void ccl_border(const uint8_t* __restrict__ lab, int* __restrict__ parent, int B, int H, int W, int tiles_x, int tiles_y) {
    const long long nh = (long long)B * (tiles_y - 1) * W, nv = (long long)B * H * (tiles_x - 1);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nh + nv; i += (long long)gridDim.x * blockDim.x) {
        if (i < nh) {   // first row of a tile against the last row of the tile above
            const int x = (int)(i % W);
            const long long q = i / W;
            const int y = ((int)(q % (tiles_y - 1)) + 1) * TH, n = (int)(q / (tiles_y - 1));
            const long long g = ((long long)n * H + y) * W + x;
            const int l = lab[g];
            if (!l || lab[g - W] != l) continue;
            // the segment's first pixel does it -- inside one tile column only: there both rows' left links are tile-local and already made
            if ((x % TW) != 0 && lab[g - 1] == l && lab[g - W - 1] == l) continue;
            unite(parent, (int)g, (int)(g - W));
        } else {        // first column of a tile against the last column of the tile to its left
            const long long k = i - nh;
            const int y = (int)(k % H);
            const long long q = k / H;
            const int x = ((int)(q % (tiles_x - 1)) + 1) * TW, n = (int)(q / (tiles_x - 1));
            const long long g = ((long long)n * H + y) * W + x;
            const int l = lab[g];
            if (!l || lab[g - 1] != l) continue;
            if ((y % TH) != 0 && lab[g - W] == l && lab[g - W - 1] == l) continue;
            unite(parent, (int)g, (int)(g - 1));
        }
    }
}

// Pass 3: flatten, component sizes, and the list of component roots per image (what the selection pass walks instead of the image)
__global__ void ccl_flatten_count(const uint8_t* __restrict__ lab, int* __restrict__ parent, int* __restrict__ count, int* __restrict__ nroots,
                                  int* __restrict__ roots, long long hw, long long total) {
    // Sizes: one atomic per (wave, distinct root) instead of one per pixel -- a large component
    // would otherwise serialise hundreds of thousands of atomics on one address.
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n_iter = (total + stride - 1) / stride;
    const int lane = threadIdx.x & 63;
    __shared__ int blk_root, blk_cnt;
    for (long long it = 0; it < n_iter; ++it) {
        const long long i = it * stride + blockIdx.x * (long long)blockDim.x + threadIdx.x;
        int r = -1;
        bool is_root = false;
        if (i < total && lab[i]) {
            r = find_root(parent, (int)i);
            parent[i] = r;
            is_root = r == (int)i;
        }
        // component roots claim slots of their image's list with ONE atomic per (wave, image): on a salt-and-pepper label map (an untrained
        // network, bench.py's synthetic weights) every other pixel is a root and a per-root atomic on the image's counter serialised the pass
        // (152 us against 34 us on blob-shaped masks).  The order of the list is free: ccl_select takes maxima over keys that carry the index.
        unsigned long long rt = __ballot(is_root);
        while (rt) {
            const int leader = __builtin_ctzll(rt);
            const long long n0 = __shfl(is_root ? i / hw : -1, leader);
            const unsigned long long grp = __ballot(is_root && i / hw == n0);
            int base = 0;
            if (lane == leader) base = atomicAdd(&nroots[n0], (int)__popcll(grp));
            base = __shfl(base, leader);
            if (is_root && i / hw == n0) roots[n0 * hw + base + (int)__popcll(grp & ((1ull << lane) - 1ull))] = (int)i;
            rt &= ~grp;
        }
        // a component that covers a large part of the image (one class winning everywhere: again the untrained network) would still receive one
        // global atomic per wave on ONE address; the block therefore collects the pixels of ONE root per iteration -- the root of its first
        // labelled pixel -- in LDS and sends them with a single atomic
        if (threadIdx.x == 0) { blk_root = -1; blk_cnt = 0; }
        __syncthreads();
        if (r >= 0) atomicCAS(&blk_root, -1, r);
        __syncthreads();
        const int br = blk_root;
        unsigned long long todo = __ballot(r >= 0);
        for (int round = 0; todo; ++round) {
            if (round == 4) {   // many distinct roots in this wave (noise): the addresses differ, plain atomics do not collide
                if ((todo >> lane) & 1ull) atomicAdd(&count[r], 1);
                break;
            }
            const int leader = __builtin_ctzll(todo);
            const int r0 = __shfl(r, leader);
            const unsigned long long same = __ballot(r == r0);
            if (lane == leader) {
                if (r0 == br) atomicAdd(&blk_cnt, (int)__popcll(same));
                else atomicAdd(&count[r0], (int)__popcll(same));
            }
            todo &= ~same;
        }
        __syncthreads();
        if (threadIdx.x == 0 && blk_cnt) atomicAdd(&count[br], blk_cnt);
    }
}

// key: (thresholded count << 32) | tie-break, larger key = earlier in the reference's top_k order.
// bin 0 gets tie-break 0xFFFFFFFF (index 0 wins ties), component with root r gets 0xFFFFFFFE - local r.
__device__ __forceinline__ unsigned long long comp_key(int cnt, int min_size, unsigned tie) {
    const unsigned c = cnt < min_size ? 0u : (unsigned)cnt;
    return ((unsigned long long)c << 32) | tie;
}

// Pass 4 (one block per image, over the root list): per object the foreground size, then the best, the second-best and (rank 2:
// output_second_largest_component, voting_layers_2d.py:58-74, top_k with three bins) the third-best histogram entry
__global__ __launch_bounds__(256) void ccl_select(const uint8_t* __restrict__ lab, const int* __restrict__ count, const int* __restrict__ nroots,
                                                  const int* __restrict__ roots, unsigned long long* __restrict__ stats, int objects, long long hw,
                                                  int min_size, int rank) {
    __shared__ unsigned fg[256];
    __shared__ unsigned long long best[256], second[256], third[256];
    const int n = blockIdx.x, nr = nroots[n];
    const int* list = roots + (long long)n * hw;
    for (int o = threadIdx.x; o < objects; o += 256) { fg[o] = 0u; best[o] = 0ull; second[o] = 0ull; third[o] = 0ull; }
    __syncthreads();
    for (int k = threadIdx.x; k < nr; k += 256) {
        const int i = list[k];
        atomicAdd(&fg[lab[i] - 1], (unsigned)count[i]);
    }
    __syncthreads();
    for (int pass = 0; pass <= rank; ++pass) {
        for (int k = threadIdx.x; k < nr + objects; k += 256) {
            unsigned long long key;
            int o;
            if (k < nr) {
                const int i = list[k];
                o = lab[i] - 1;
                key = comp_key(count[i], min_size, 0xFFFFFFFEu - (unsigned)((long long)i - n * hw));
            } else {   // bin 0 of object o: every pixel that is not the object
                o = k - nr;
                key = comp_key((int)(hw - (long long)fg[o]), min_size, 0xFFFFFFFFu);
            }
            if (pass == 0) atomicMax(&best[o], key);
            else if (pass == 1) { if (key < best[o]) atomicMax(&second[o], key); }
            else if (key < second[o]) atomicMax(&third[o], key);
        }
        __syncthreads();
    }
    // the selected histogram entry: rank 1 = second in top_k order (rank 0 is assumed to be the background bin), rank 2 = third
    for (int o = threadIdx.x; o < objects; o += 256) stats[((long long)n * objects + o) * 3 + 2] = rank == 2 ? third[o] : second[o];
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
    size_t px = (size_t)batch * h * w;   // parent, size counters, root lists (hw slots per image), roots per image, 3 words per (image, object)
    return px * sizeof(int) * 3 + ((size_t)batch * sizeof(int) + 63) / 64 * 64 + (size_t)batch * objects * 3 * sizeof(unsigned long long) + 128;
}

extern "C" int cp_ccl_filter_labels(const uint8_t* labels_in, int batch, int h, int w, int objects, int min_size, int rank, void* ws,
                                    uint8_t* labels_out, void* stream) {
    CP_REQUIRE(labels_in && labels_out && ws, "cp_ccl_filter_labels: null pointer");
    CP_REQUIRE(rank == 1 || rank == 2, "cp_ccl_filter_labels: rank 1 (largest component) or 2 (second largest, output_second_largest_component)");
    CP_REQUIRE(batch > 0 && h > 0 && w > 0 && objects > 0 && objects < 255, "cp_ccl_filter_labels: bad sizes");
    const long long hw = (long long)h * w, total = hw * batch;
    CP_REQUIRE(total < (1LL << 31) && hw < 0xFFFFFFF0LL, "cp_ccl_filter_labels: too many pixels for 32-bit component ids");
    hipStream_t st = (hipStream_t)stream;
    int* parent = reinterpret_cast<int*>(ws);
    int* count = parent + total;
    int* roots = count + total;
    uintptr_t np = (reinterpret_cast<uintptr_t>(roots + total) + 63) & ~(uintptr_t)63;
    int* nroots = reinterpret_cast<int*>(np);
    uintptr_t sp = (np + (size_t)batch * sizeof(int) + 63) & ~(uintptr_t)63;
    unsigned long long* stats = reinterpret_cast<unsigned long long*>(sp);
    const int g = grid_for(total);
    if (hipMemsetAsync(nroots, 0, sizeof(int) * (size_t)batch, st) != hipSuccess) return cp::check_launch("cp_ccl_filter_labels memset");
    const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH;
    CP_LAUNCH(ccl_local, dim3(batch * tiles_x * tiles_y), dim3(256), 0, st, labels_in, parent, count, h, w, tiles_x, tiles_y);
    const long long nb = (long long)batch * (tiles_y - 1) * w + (long long)batch * h * (tiles_x - 1);
    if (nb > 0) CP_LAUNCH(ccl_border, dim3(grid_for(nb)), dim3(THREADS), 0, st, labels_in, parent, batch, h, w, tiles_x, tiles_y);
    CP_LAUNCH(ccl_flatten_count, dim3(g), dim3(THREADS), 0, st, labels_in, parent, count, nroots, roots, hw, total);
    CP_LAUNCH(ccl_select, dim3(batch), dim3(256), 0, st, labels_in, count, nroots, roots, stats, objects, hw, min_size, rank);
    CP_LAUNCH(ccl_write, dim3(g), dim3(THREADS), 0, st, labels_in, parent, stats, objects, hw, total, labels_out);
    return cp::check_launch("cp_ccl_filter_labels");
}
