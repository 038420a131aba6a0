// Convolution weight gradient, exact fp32 on v_mfma_f32_32x32x2_f32 (gfx950).
//
//   dWp[co][k] += sum over output pixels m of  dY[m][co] * A[m][k]
//
// with A the SAME implicit im2col matrix the forward kernel multiplies (conv_f32.hip: sources,
// taps, zero padding, partial-convolution tap mask) and k in the packed K order of cp_conv_ktot,
// so the result lands directly in the packed [cout][ktot] layout.  The reference obtains this
// product from tf.GradientTape (train_casapose.py:594-611) -> Conv2DBackpropFilter.
//
// GEMM view: the REDUCTION runs over pixels.  An MFMA 32x32x2 lane supplies A[i = lane&31][kk = lane>>5]
// and B[kk = lane>>5][j = lane&31]; with kk = a pair of consecutive pixels, i = co and j = k both operands
// are read from LDS tiles kept in their NATIVE [pixel][channel] layout with one conflict-free
// ds_read_b32 per lane (consecutive lanes -> consecutive channels): no transposes anywhere.
// A block owns a (WCO*32 co) x (WK*32 k) tile of dWp for one pixel range (split-K over pixels, fp32
// atomics at the end); 4 consumer waves each hold one 32x32 accumulator, 4 producer waves gather
// dY and im2col rows for the step after next while the current 32-pixel step is multiplied.
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int MAX_TAPS = 64;

struct WSrc {
    const float* data;
    int C, ld;
    unsigned bytes;
    int c4, cpt, nchunks;
};

struct WgK {
    WSrc s[2];
    const float* dy;
    int dy_ld;
    unsigned dy_bytes;
    int cout4;  // channels of dY that may be read (cout rounded up to 4, <= dy_ld)
    const float* row_scale;
    const uint8_t* tap_label;
    unsigned lab_bytes, rs_bytes;
    float* dw;
    int ktot, nchunks;
    int Hin, Win, Ho, Wo, Cout, KW, ntaps, stride, dil, pad;
    int M;
    int tiles_co, tiles_k, nsplit, steps_per_split;
    int q_begin; // first K chunk computed (cp::wgrad_f32_chunks: only the trailing chunks of the packed rows)
    int groups;  // grouped GEMM (Winograd planes): pixels [g*M, (g+1)*M) accumulate into dw + g*Cout*ktot; M = rows per group
};

template <int WCO, bool PARTIAL>
__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(const WgK p) {
    constexpr int WK = 4 / WCO;
    constexpr int DS = WCO * 32 + 4;  // dY tile row stride (floats)
    constexpr int AS = WK * 32 + 4;   // im2col tile row stride
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Dt = smem;                 // [2][32][DS]
    float* At = smem + 2 * 32 * DS;   // [2][32][AS]

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;

    const int ntiles = p.tiles_co * p.tiles_k;
    const int logical_all = cp::xcd_remap(blockIdx.x, ntiles * p.nsplit * p.groups);
    const int grp = logical_all / (ntiles * p.nsplit);
    const int logical = logical_all - grp * (ntiles * p.nsplit);
    const int gbase = grp * p.M;  // first pixel of this group
    const int split = logical / ntiles;
    const int t2 = logical - split * ntiles;
    const int tile_k = t2 / p.tiles_co, tile_co = t2 - tile_k * p.tiles_co;
    const int co0 = tile_co * (WCO * 32);
    const int q0 = p.q_begin + tile_k * WK;  // first K chunk of this tile
    const int step0 = split * p.steps_per_split;
    const int total_steps = (p.M + 31) >> 5;
    const int nsteps = min(p.steps_per_split, total_steps - step0);
    if (nsteps <= 0) return;

#define CP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    if (producer) {
        const int col4 = tid & 7;
        const int prow = tid >> 3;  // pixel row of the 32-pixel step handled by this thread
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.s[0].data, 0, p.s[0].bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.s[1].data ? p.s[1].data : p.s[0].data), 0,
                                                                              p.s[1].data ? p.s[1].bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(PARTIAL ? (const void*)p.tap_label : (const void*)p.dy), 0,
                                                                              PARTIAL ? p.lab_bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.row_scale ? (const void*)p.row_scale : (const void*)p.dy), 0,
                                                                              p.row_scale ? p.rs_bytes : 0u, 0x00020000);
        // per-chunk constants of this thread: source, tap offset, channel offset
        int c_si[WK], c_dy[WK], c_dx[WK], c_coff[WK];
#pragma unroll
        for (int c = 0; c < WK; ++c) {
            const int q = q0 + c;
            int si = (q >= p.s[0].nchunks) ? 1 : 0;
            const int ql = q - (si ? p.s[0].nchunks : 0);
            const int c4 = si ? p.s[1].c4 : p.s[0].c4;
            const int cpt = si ? p.s[1].cpt : p.s[0].cpt;
            int tap, coff;
            if (c4) { tap = ql * 8 + col4; coff = 0; }
            else { tap = ql / cpt; coff = (ql - tap * cpt) * 32 + col4 * 4; }
            const bool ok = q < p.nchunks && tap < p.ntaps;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            c_si[c] = si;
            c_dy[c] = ok ? ky * p.dil : 0x20000000;  // never in bounds
            c_dx[c] = kx * p.dil;
            c_coff[c] = coff;
        }
        unsigned d_coff[WCO];
#pragma unroll
        for (int j = 0; j < WCO; ++j) {
            const int co = co0 + j * 32 + col4 * 4;
            d_coff[j] = co < p.cout4 ? (unsigned)co * 4u : OOB;
        }
        // pixel state of the NEXT step to be issued
        int m = step0 * 32 + prow;
        int n = m / (p.Ho * p.Wo);
        int rem = m - n * (p.Ho * p.Wo);
        int oy = rem / p.Wo, ox = rem - oy * p.Wo;

        float4 areg[WK], dreg[WCO];
        int alab[WK], clab = 0;
        float rsv = 1.f;

        auto issue = [&]() {
            const bool valid = m < p.M;
            const int iy0 = valid ? oy * p.stride - p.pad : -0x10000000;
            const int ix0 = ox * p.stride - p.pad;
            const int pix0 = gbase + (n * p.Hin + iy0) * p.Win + ix0;  // (groups > 1 only with 1x1 / stride 1 geometry: pixel == row)
#pragma unroll
            for (int j = 0; j < WCO; ++j)
                dreg[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rsd, (int)((valid && d_coff[j] < OOB) ? (unsigned)(gbase + m) * (unsigned)p.dy_ld * 4u + d_coff[j] : OOB), 0, 0));
            rsv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsr, (int)(valid ? (unsigned)m * 4u : OOB), 0, 0));
            if constexpr (PARTIAL) clab = __builtin_amdgcn_raw_buffer_load_b8(rsl, valid ? (n * p.Hin + oy) * p.Win + ox : (int)OOB, 0, 0);
#pragma unroll
            for (int c = 0; c < WK; ++c) {
                const int iy = iy0 + c_dy[c], ix = ix0 + c_dx[c];
                const bool inb = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);
                const int gpix = pix0 + c_dy[c] * p.Win + c_dx[c];
                const int sld = c_si[c] ? p.s[1].ld : p.s[0].ld;
                if constexpr (PARTIAL) alab[c] = __builtin_amdgcn_raw_buffer_load_b8(rsl, inb ? gpix : (int)OOB, 0, 0);
                areg[c] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(c_si[c] ? rs1 : rs0,
                                                                                           (int)(inb ? (unsigned)((gpix * sld + c_coff[c]) * 4) : OOB), 0, 0));
            }
            // advance by one step (32 pixels)
            m += 32;
            ox += 32;
            while (ox >= p.Wo) { ox -= p.Wo; ++oy; }
            while (oy >= p.Ho) { oy -= p.Ho; ++n; }
        };
        const bool has_rs = p.row_scale != nullptr;
        auto store = [&](int buf) {
            float* d = Dt + buf * 32 * DS + prow * DS + col4 * 4;
            float* a = At + buf * 32 * AS + prow * AS + col4 * 4;
            const float f = has_rs ? rsv : 1.f;
#pragma unroll
            for (int j = 0; j < WCO; ++j) {
                float4 v = dreg[j];
                v.x *= f; v.y *= f; v.z *= f; v.w *= f;
                *reinterpret_cast<float4*>(d + j * 32) = v;
            }
#pragma unroll
            for (int c = 0; c < WK; ++c) {
                float4 v = areg[c];
                if constexpr (PARTIAL) {
                    if (alab[c] != clab) v = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                *reinterpret_cast<float4*>(a + c * 32) = v;
            }
        };

        issue();
        store(0);
        if (nsteps > 1) issue();
        CP_BARRIER();
        for (int q = 0; q < nsteps; ++q) {
            if (q + 1 < nsteps) {
                store((q + 1) & 1);
                if (q + 2 < nsteps) issue();
            }
            CP_BARRIER();
        }
        return;
    }

    // ------------------------------- consumers ------------------------------------------------
    const int wco = (wave & 3) % WCO, wk = (wave & 3) / WCO;
    const int lrow = lane & 31, half = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    CP_BARRIER();  // stage 0 ready
    for (int q = 0; q < nsteps; ++q) {
        const float* d = Dt + (q & 1) * 32 * DS + half * DS + wco * 32 + lrow;
        const float* a = At + (q & 1) * 32 * AS + half * AS + wk * 32 + lrow;
        float fa[16], fb[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            fa[s] = d[2 * s * DS];
            fb[s] = a[2 * s * AS];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc, 0, 0, 0);
        CP_BARRIER();
    }
#undef CP_BARRIER

    // ---- accumulate into dWp: row = co, column = k (consecutive lanes -> consecutive k) --------
    const int k = (q0 + wk) * BK + lrow;
    if ((q0 + wk) < p.nchunks) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < p.Cout) atomicAdd(p.dw + ((size_t)grp * p.Cout + co) * p.ktot + k, acc[r]);
        }
    }
}

// Grouped 1x1 problem of the Winograd weight gradient (dU[g][co][k] = sum over the group's rows of M[g][row][co] * V[g][row][k], 36 groups):
// the same MFMA scheme with a 128 co x 128 k tile per block -- each consumer wave holds 2 x 2 accumulators, so an LDS operand value feeds
// two MFMAs and a block reads half the bytes per FLOP of the 64 x 64 tile above (these layers were L2-bandwidth-bound there: 7.4 GB per
// stage-4 layer at 5 TB/s).  Rows in steps of 32, two LDS stages, 4 producer waves (8 x 16-byte loads per thread and step).
struct GemmT {
    const float* a;   // [groups][rows][K]   (V)
    const float* d;   // [groups][rows][Cout] (M)
    float* dw;        // [groups][Cout][K]
    int K, Cout, lda, ldd;
    unsigned a_bytes, d_bytes;
    int rows, groups, tiles_co, tiles_k, spt;   // spt = 32-row steps per tile
    long long total;                            // groups * tiles * spt: the step stream the persistent blocks share out evenly
};

__global__ __launch_bounds__(512, 2) void wgrad_gemm128_kernel(const GemmT p) {
    constexpr int TS = 128 + 4;   // LDS row stride (floats)
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Dt = smem;                  // [2][32][TS]
    float* At = smem + 2 * 32 * TS;    // [2][32][TS]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = threadIdx.x & 255, lane = tid & 63;
    // a block takes a contiguous range of the (tile, step) stream: every block the same number of steps (+-1), at most two tiles
    // touched partially -> at most two more flushes than tiles, and no tail of half-empty rounds (576 tiles on 512 block slots)
    const long long s0 = p.total * blockIdx.x / gridDim.x, s1 = p.total * (blockIdx.x + 1) / gridDim.x;
    const int nsteps = (int)(s1 - s0);
    if (nsteps <= 0) return;
    const int ntiles = p.tiles_co * p.tiles_k;
    int tile0 = (int)(s0 / p.spt), st0 = (int)(s0 - (long long)tile0 * p.spt);
#define CP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    if (wave >= 4) {
        const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.d, 0, p.d_bytes, 0x00020000);
        const int col4 = tid & 31, prow = tid >> 5;   // 32 float4 columns x 8 rows per pass, 4 passes
        float4 areg[4], dreg[4];
        int tile = tile0, st = st0;
        auto issue = [&]() {
            const int grp = tile / ntiles, t2 = tile - grp * ntiles;
            const int tile_k = t2 / p.tiles_co, tile_co = t2 - tile_k * p.tiles_co;
            const int kc = tile_k * 128 + col4 * 4, cc = tile_co * 128 + col4 * 4;
            const unsigned acol = (unsigned)kc * 4u | (kc < p.K ? 0u : OOB), dcol = (unsigned)cc * 4u | (cc < p.Cout ? 0u : OOB);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = st * 32 + prow + i * 8;
                const unsigned oob = r < p.rows ? 0u : OOB;
                const unsigned gr = (unsigned)(grp * p.rows + r);
                areg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsa, (int)((gr * (unsigned)p.lda * 4u + acol) | oob), 0, 0));
                dreg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsd, (int)((gr * (unsigned)p.ldd * 4u + dcol) | oob), 0, 0));
            }
            if (++st == p.spt) { st = 0; ++tile; }
        };
        auto store = [&](int buf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<float4*>(At + buf * 32 * TS + (prow + i * 8) * TS + col4 * 4) = areg[i];
                *reinterpret_cast<float4*>(Dt + buf * 32 * TS + (prow + i * 8) * TS + col4 * 4) = dreg[i];
            }
        };
        issue();
        store(0);
        if (nsteps > 1) issue();
        CP_BARRIER();
        for (int q = 0; q < nsteps; ++q) {
            if (q + 1 < nsteps) {
                store((q + 1) & 1);
                if (q + 2 < nsteps) issue();
            }
            CP_BARRIER();
        }
        return;
    }
    const int wco = wave & 1, wk = wave >> 1;
    const int lrow = lane & 31, half = lane >> 5;
    f32x16 acc[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    auto flush = [&](int tile) {
        const int grp = tile / ntiles, t2 = tile - grp * ntiles;
        const int tile_k = t2 / p.tiles_co, tile_co = t2 - tile_k * p.tiles_co;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = tile_k * 128 + wk * 64 + j * 32 + lrow;
                if (k >= p.K) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = tile_co * 128 + wco * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < p.Cout) atomicAdd(p.dw + ((size_t)grp * p.Cout + co) * p.K + k, acc[i][j][r]);
                }
            }
        zero();
    };
    zero();
    int tile = tile0, st = st0;
    CP_BARRIER();
    for (int q = 0; q < nsteps; ++q) {
        const float* d = Dt + (q & 1) * 32 * TS + half * TS + wco * 64 + lrow;
        const float* a = At + (q & 1) * 32 * TS + half * TS + wk * 64 + lrow;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float fd0 = d[2 * s * TS], fd1 = d[2 * s * TS + 32];
            const float fa0 = a[2 * s * TS], fa1 = a[2 * s * TS + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd0, fa0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd0, fa1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd1, fa0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd1, fa1, acc[1][1], 0, 0, 0);
        }
        CP_BARRIER();
        if (++st == p.spt) {
            flush(tile);
            st = 0;
            ++tile;
        }
    }
#undef CP_BARRIER
    if (st != 0) flush(tile);
}

int launch_gemm128(GemmT k, hipStream_t st) {
    k.tiles_co = (k.Cout + 127) / 128;
    k.tiles_k = (k.K + 127) / 128;
    k.spt = (k.rows + 31) / 32;
    k.total = (long long)k.groups * k.tiles_co * k.tiles_k * k.spt;
    const long long grid = std::min<long long>(512, (k.total + 7) / 8);   // two blocks per CU; >= 8 steps per block
    const size_t lds = (size_t)4 * 32 * (128 + 4) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_gemm128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    CP_LAUNCH(wgrad_gemm128_kernel, dim3((unsigned)std::max<long long>(grid, 1)), dim3(512), lds, st, k);
    return cp::check_launch("cp_conv2d_wgrad_f32 (grouped)");
}

int chunks_for(int taps, int C) { return (C == 4) ? (taps + 7) / 8 : taps * (C / 32); }

template <int WCO, bool PARTIAL>
int launch_wgrad(WgK k, hipStream_t st) {
    constexpr int WK = 4 / WCO;
    k.tiles_co = (k.Cout + WCO * 32 - 1) / (WCO * 32);
    k.tiles_k = (k.nchunks - k.q_begin + WK - 1) / WK;
    const int total_steps = (k.M + 31) / 32;
    const int tiles = k.tiles_co * k.tiles_k;
    // enough blocks to fill 256 CUs x 2 several times over, but >= 8 steps per block so the pipeline fill amortises
    int nsplit = (256 * 8 + tiles * k.groups - 1) / (tiles * k.groups);
    if (nsplit > (total_steps + 7) / 8) nsplit = (total_steps + 7) / 8;
    if (nsplit < 1) nsplit = 1;
    k.steps_per_split = (total_steps + nsplit - 1) / nsplit;
    k.nsplit = (total_steps + k.steps_per_split - 1) / k.steps_per_split;
    const size_t lds = (size_t)2 * 32 * ((WCO * 32 + 4) + (WK * 32 + 4)) * sizeof(float);
    CP_LAUNCH((conv_wgrad_kernel<WCO, PARTIAL>), dim3((unsigned)(tiles * k.nsplit * k.groups)), dim3(512), lds, st, k);
    return cp::check_launch("cp_conv2d_wgrad_f32");
}

}  // namespace

static int wgrad_f32_impl(const cp_conv_desc* d, const float* dy, int dy_ld, float* dw_packed, int accumulate, int first_chunk, void* stream) {
    CP_REQUIRE_DESC(d, "cp_conv2d_wgrad_f32");
    CP_REQUIRE(dy && dw_packed, "cp_conv2d_wgrad_f32: null pointer");
    CP_REQUIRE(d->num_sources == 1 || d->num_sources == 2, "cp_conv2d_wgrad_f32: num_sources must be 1 or 2");
    CP_REQUIRE(d->kh * d->kw <= MAX_TAPS && d->kh > 0 && d->kw > 0, "cp_conv2d_wgrad_f32: unsupported kernel %dx%d", d->kh, d->kw);
    CP_REQUIRE(d->stride >= 1 && d->dilation >= 1 && d->pad >= 0, "cp_conv2d_wgrad_f32: bad stride/dilation/pad");
    const int eh = (d->kh - 1) * d->dilation + 1, ew = (d->kw - 1) * d->dilation + 1;
    CP_REQUIRE(d->out_h == (d->in_h + 2 * d->pad - eh) / d->stride + 1 && d->out_w == (d->in_w + 2 * d->pad - ew) / d->stride + 1,
               "cp_conv2d_wgrad_f32: out size inconsistent with the input geometry");
    CP_REQUIRE(!d->tap_label || d->stride == 1, "cp_conv2d_wgrad_f32: tap_label needs stride 1");
    const int cout4 = (d->cout + 3) & ~3;
    CP_REQUIRE(dy_ld >= cout4 && dy_ld % 4 == 0 && ((uintptr_t)dy & 15) == 0, "cp_conv2d_wgrad_f32: dy_ld must be a multiple of 4 and >= cout rounded up to 4");
    WgK k{};
    int chans[2] = {0, 0};
    for (int s = 0; s < d->num_sources; ++s) {
        const cp_conv_source& in = d->src[s];
        CP_REQUIRE(in.data, "cp_conv2d_wgrad_f32: source %d has null data", s);
        CP_REQUIRE(in.mode == CP_SRC_DIRECT && !in.pre_scale, "cp_conv2d_wgrad_f32: sources must be materialised (CP_SRC_DIRECT, no pre-affine)");
        CP_REQUIRE(in.channels == 4 || (in.channels > 0 && in.channels % 32 == 0), "cp_conv2d_wgrad_f32: source %d channels must be 4 or a multiple of 32", s);
        CP_REQUIRE(in.ld >= in.channels && in.ld % 4 == 0 && ((uintptr_t)in.data & 15) == 0, "cp_conv2d_wgrad_f32: source %d ld/alignment", s);
        const long long nbytes = (long long)d->batch * d->in_h * d->in_w * in.ld * 4;
        CP_REQUIRE(nbytes < (1LL << 31), "cp_conv2d_wgrad_f32: source %d spans >= 2 GiB", s);
        WSrc& o = k.s[s];
        o.data = in.data; o.C = in.channels; o.ld = in.ld; o.bytes = (unsigned)nbytes;
        o.c4 = in.channels == 4; o.cpt = o.c4 ? 1 : in.channels / 32; o.nchunks = chunks_for(d->kh * d->kw, in.channels);
        chans[s] = in.channels;
    }
    k.ktot = cp_conv_ktot(d->kh, d->kw, d->num_sources, chans);
    k.nchunks = k.ktot / BK;
    const long long M = (long long)d->batch * d->out_h * d->out_w;
    CP_REQUIRE(M * dy_ld * 4 < (1LL << 31), "cp_conv2d_wgrad_f32: dy spans >= 2 GiB");
    k.dy = dy; k.dy_ld = dy_ld; k.dy_bytes = (unsigned)(M * dy_ld * 4); k.cout4 = cout4;
    // dy is the gradient of the UN-normalised sum: the normalisation backward (cp_bn_act_bwd_apply_f32) has already
    // multiplied it by the partial convolution's row_scale, for the data gradient as well as for this kernel
    k.row_scale = nullptr; k.rs_bytes = (unsigned)(M * 4);
    k.tap_label = d->tap_label; k.lab_bytes = (unsigned)((long long)d->batch * d->in_h * d->in_w);
    k.dw = dw_packed;
    k.Hin = d->in_h; k.Win = d->in_w; k.Ho = d->out_h; k.Wo = d->out_w; k.Cout = d->cout; k.KW = d->kw; k.ntaps = d->kh * d->kw;
    k.stride = d->stride; k.dil = d->dilation; k.pad = d->pad; k.M = (int)M;
    k.groups = 1;
    k.q_begin = first_chunk;
    CP_REQUIRE(first_chunk >= 0 && first_chunk < k.nchunks && (first_chunk == 0 || !d->group_rows), "cp_conv2d_wgrad_f32: bad first chunk");
    if (d->group_rows) {  // the 36 Winograd planes in one launch: a 1x1 problem per group of rows, dw_packed is [groups][cout][ktot]
        CP_REQUIRE(d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->num_sources == 1 && !d->tap_label && d->batch * d->in_h == 1,
                   "cp_conv2d_wgrad_f32: grouped mode is a plain 1x1 problem laid out as one row of pixels (batch = in_h = 1)");
        CP_REQUIRE(d->group_rows % 32 == 0 && M % d->group_rows == 0, "cp_conv2d_wgrad_f32: group_rows must be a multiple of 32 dividing the pixel count");
        k.groups = (int)(M / d->group_rows);
        k.M = d->group_rows;
        k.Win = k.Wo = d->group_rows;
    }
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate)
        if (hipMemsetAsync(dw_packed, 0, sizeof(float) * (size_t)k.groups * d->cout * k.ktot, st) != hipSuccess) return cp::check_launch("cp_conv2d_wgrad_f32 memset");
    if (d->group_rows && k.ktot % 128 == 0 && d->cout % 64 == 0 && d->cout >= 128 && !getenv("CP_WGRAD_GEMM64")) {
        GemmT g{};
        g.a = d->src[0].data; g.d = dy; g.dw = dw_packed;
        g.K = k.ktot; g.Cout = d->cout; g.lda = d->src[0].ld; g.ldd = dy_ld;
        g.a_bytes = k.s[0].bytes; g.d_bytes = k.dy_bytes;
        g.rows = k.M; g.groups = k.groups;
        return launch_gemm128(g, st);
    }
    if (d->cout <= 32) return d->tap_label ? launch_wgrad<1, true>(k, st) : launch_wgrad<1, false>(k, st);
    return d->tap_label ? launch_wgrad<2, true>(k, st) : launch_wgrad<2, false>(k, st);
}

extern "C" int cp_conv2d_wgrad_f32(const cp_conv_desc* d, const float* dy, int dy_ld, float* dw_packed, int accumulate, void* stream) {
    return wgrad_f32_impl(d, dy, dy_ld, dw_packed, accumulate, 0, stream);
}

// the packed columns [first_chunk * 32, ktot) only, accumulated into dw_packed (conv_wgrad_split.hip: the 4-channel image source)
int cp::wgrad_f32_chunks(const cp_conv_desc* d, const float* dy, int dy_ld, float* dw_packed, int first_chunk, hipStream_t st) {
    return wgrad_f32_impl(d, dy, dy_ld, dw_packed, 1, first_chunk, (void*)st);
}
