// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions on the bf16 matrix pipe (gfx950), the backward partner of conv_hsplit.hip:
//
//   dWp[co][k(tap, ci)] += sum over output pixels p of  dY[p][co] * X[p + tap][ci]        (X zero outside the image; partial convolutions:
//                                                                                          X[p + tap] counts only where label[p + tap] == label[p])
//
// in the packed [cout][ktot] layout of cp_conv2d_wgrad_f32, which this replaces for those layers (the reference obtains the product from
// tf.GradientTape, train_casapose.py:594-611 -> Conv2DBackpropFilter).
//   NP = 3  fp32-EQUIVALENT: both operands are split exactly into three bf16 terms, six products per fp32 product, fp32 accumulation
//   NP = 1  operands rounded to bf16 (BASELINE.json configs[2])
//   NP = 2  (round 6, planes = CP_PLANES_F16X2) both operands as fp16 pairs (split_f16.h): three exact products per fp32 product.  Both operands
//           must sit inside fp16's band as they are: X is what the forward converted (its monitor watches it), dY carries the power of two the
//           training plan puts on the loss (train_engine.train_bwd_f16x2).
//
// The reduction runs over PIXELS, so both MFMA operands are needed pixel-major ("transposed") while the tensors are [pixel][channel] in
// HBM.  v_mfma_f32_32x32x16_bf16 wants 8 consecutive k (= pixels) of one row (= channel) per lane; ds_read_b64_tr_b16 delivers exactly
// that from an LDS image kept in the native layout: each 16-lane group reads a [4 pixels][16 channels] block and every lane receives one
// channel's 4 pixels.  With [pixel][32 channels] rows of 64 bytes the 32 lanes of a read group cover 256 contiguous bytes: conflict-free
// at every tap offset, and a tap is just a pixel offset (immediate) -- no transposes, no shuffles.
//
// Streaming: a block owns a (MB*32 ci) x (NB*32 co) tile of dW and walks a strip of 64 image columns row by row.  LDS holds a ring of four
// input rows (66 pixels: one halo column each side) and two dY rows per plane; every input row is fetched from HBM once per block and used
// by three output rows x three taps.  4 loader waves fetch fp32 rows two or three steps ahead into registers, split them and store the
// planes (one 16-byte store per plane for 8 channels of a pixel); 4 consumer waves hold 9 taps x 32x32 accumulators each (144 registers)
// and issue 6 MFMAs per (tap, 16 pixels).  One barrier per row.  The tile is flushed with fp32 atomics when the block moves to another
// (ci, co) tile (summation order not fixed, as in cp_conv2d_wgrad_f32).
#include "common.h"
#include "split_f16.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int SW = 64;            // output columns per row step
constexpr int XC = SW + 2;        // input columns staged per row (one halo column each side)
constexpr int XROWB = XC * 64;    // bytes of one (row, 32-channel block, plane): [66 pixels][32 bf16]
constexpr int DROWB = SW * 64;
constexpr int NXS = 4, NDS = 2;   // ring depths: input rows y-1, y, y+1 + the one being written; dY rows y + the one being written

struct WsSrc {
    const float* data;
    int ld, C, blocks, kbase;
    unsigned bytes;
};

struct WSplitK {
    WsSrc s[2];
    const float* dy;
    int dy_ld;
    unsigned dy_bytes;
    const uint8_t* label;
    unsigned lab_bytes;
    float* dw;
    int ktot;
    const float* img;       // optional trailing 4-channel source (the image), IMG instantiations: [B][H][W][img_ld]
    int img_ld, img_kbase;
    unsigned img_bytes;
    int B, H, W, Cout;
    int cblocks;            // 32-channel blocks of the sources handled here
    int tiles_m, tiles_n;   // (ci, co) tiles
    int strips, chunks, rc; // column strips per image, row chunks per strip, rows per chunk
    int J, U;               // pixel jobs per tile, units = tiles * J
};

struct Unit {
    int tm, tn, n, x0, ya, yb, pair;
};

__device__ __forceinline__ Unit decode(const WSplitK& p, int u) {
    Unit r;
    r.pair = u / p.J;
    int j = u - r.pair * p.J;
    r.tm = r.pair / p.tiles_n;
    r.tn = r.pair - r.tm * p.tiles_n;
    const int ch = j % p.chunks;
    j /= p.chunks;
    const int sx = j % p.strips;
    r.n = j / p.strips;
    r.x0 = sx * SW;
    r.ya = ch * p.rc;
    r.yb = min(p.H, r.ya + p.rc);
    return r;
}

#define WS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ unsigned pack_hi16(unsigned a_lo, unsigned b_hi) { return __builtin_amdgcn_perm(b_hi, a_lo, 0x07060302u); }

// exact three-way split of 8 floats into packed bf16 (see conv_hsplit.hip / wino_gemm_split.hip)
__device__ __forceinline__ void split8(const float4 v0, const float4 v1, uint4& hi, uint4& mid, uint4& lo) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h[e] = __builtin_bit_cast(unsigned, x[e]);
        const float r1 = x[e] - __builtin_bit_cast(float, h[e] & 0xffff0000u);
        m[e] = __builtin_bit_cast(unsigned, r1);
        const float r2 = r1 - __builtin_bit_cast(float, m[e] & 0xffff0000u);
        l[e] = __builtin_bit_cast(unsigned, r2);
    }
    hi = make_uint4(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]), pack_hi16(h[4], h[5]), pack_hi16(h[6], h[7]));
    mid = make_uint4(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]), pack_hi16(m[4], m[5]), pack_hi16(m[6], m[7]));
    lo = make_uint4(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]), pack_hi16(l[4], l[5]), pack_hi16(l[6], l[7]));
}

__device__ __forceinline__ uint4 round8(const float4 v0, const float4 v1) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned u = __builtin_bit_cast(unsigned, x[e]);
        r[e] = u + 0x7fffu + ((u >> 16) & 1u);
    }
    return make_uint4(pack_hi16(r[0], r[1]), pack_hi16(r[2], r[3]), pack_hi16(r[4], r[5]), pack_hi16(r[6], r[7]));
}

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& mid, uint2& lo) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = __builtin_bit_cast(unsigned, x[e]);
        const float r1 = x[e] - __builtin_bit_cast(float, h[e] & 0xffff0000u);
        m[e] = __builtin_bit_cast(unsigned, r1);
        const float r2 = r1 - __builtin_bit_cast(float, m[e] & 0xffff0000u);
        l[e] = __builtin_bit_cast(unsigned, r2);
    }
    hi = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
    mid = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
    lo = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
}

__device__ __forceinline__ uint2 round4(const float4 v) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned u = __builtin_bit_cast(unsigned, x[e]);
        r[e] = u + 0x7fffu + ((u >> 16) & 1u);
    }
    return make_uint2(pack_hi16(r[0], r[1]), pack_hi16(r[2], r[3]));
}

template <int NP>
__device__ __forceinline__ void store_planes(unsigned char* dst, int plane_stride, const float4 v0, const float4 v1) {
    if constexpr (NP == 2) {   // fp16 pair: hi, lo
        uint2 h0, l0, h1, l1;
        cp::split4h(v0, h0, l0);
        cp::split4h(v1, h1, l1);
        *reinterpret_cast<uint4*>(dst) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4*>(dst + plane_stride) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    } else if constexpr (NP == 3) {
        uint4 h, m, l;
        split8(v0, v1, h, m, l);
        *reinterpret_cast<uint4*>(dst) = h;
        *reinterpret_cast<uint4*>(dst + plane_stride) = m;
        *reinterpret_cast<uint4*>(dst + 2 * plane_stride) = l;
    } else {
        *reinterpret_cast<uint4*>(dst) = round8(v0, v1);
    }
}

// 8 consecutive pixels (k) of this lane's channel: two transpose reads of 4 pixels each
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* a) {
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * 64));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// IMG (MB = NB = 1 only): the trailing 4-channel image source is a 36-column block of its own -- columns (tap, channel), taps 0..7 in one
// 32x32 accumulator and tap 8 in the first four columns of a second -- fed by transpose reads of an [pixel][4 bf16] image ring in which
// the SUPPLYING lane picks its tap's row and column offset: 12 more MFMAs per 16 pixels instead of a second pass over dY by the fp32 kernel.
template <int NP, int MB, int NB, bool PARTIAL, bool IMG>
__global__ __launch_bounds__(512, 2) void wgrad_split_kernel(const WSplitK p) {
    static_assert(!IMG || (MB == 1 && NB == 1), "the image block needs the one-tile configuration");
    constexpr int IPLANE = XC * 8, ISLOT = NP * IPLANE;   // image ring: [slot][plane][66 pixels][4 bf16]
    constexpr unsigned OOB = 0x80000000u;
    constexpr int XPLANE = MB * XROWB, XSLOT = NP * XPLANE;
    constexpr int DPLANE = NB * DROWB, DSLOT = NP * DPLANE;
    constexpr int PH = 4 / (MB * NB);   // consumer waves sharing one 32x32 tile: they split the four 16-pixel steps of a row
    constexpr int KSW = 4 / PH;
#ifdef WS_D3
    constexpr int D = 3;
#else
    constexpr int D = (MB * NB == 4) ? 2 : 3;   // register sets of the loaders = rows in flight
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Xs = smem;
    unsigned char* Ds = smem + NXS * XSLOT;
    unsigned short* Ms = reinterpret_cast<unsigned short*>(Ds + NDS * DSLOT);   // [NDS][9 taps][SW] AND-masks 0xffff / 0 (PARTIAL)
    unsigned char* Is = reinterpret_cast<unsigned char*>(Ms) + (PARTIAL ? NDS * 9 * SW * 2 : 0);   // [NXS][ISLOT] (IMG)

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int G = gridDim.x;
    const int u0 = (int)((long long)blockIdx.x * p.U / G), u1 = (int)((long long)(blockIdx.x + 1) * p.U / G);
    if (u0 >= u1) return;
    int NT = 0;
    for (int u = u0; u < u1; ++u) {
        const Unit t = decode(p, u);
        NT += t.yb - t.ya + 2;
    }
    const int NTP = (NT + D - 1) / D * D;

    if (wave >= 4) {
        if constexpr (NP == 2) cp::f16_overflow_clamps();
        // ------------------------------------------------ loaders ---------------------------------------------------------------
        const int lw = wave - 4;
        const int oct = lane & 3;
        constexpr int XPR = 16 * (4 / MB), XR = (XC + XPR - 1) / XPR;   // pixels per round / rounds of an input row
        constexpr int DPR = 16 * (4 / NB), DR = SW / DPR;
        const int x_mb = lw % MB, x_px0 = (lane >> 2) + 16 * (lw / MB);
        const int d_nb = lw % NB, d_px0 = (lane >> 2) + 16 * (lw / NB);
        const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(PARTIAL ? (const void*)p.label : (const void*)p.dy), 0,
                                                                              PARTIAL ? p.lab_bytes : 0u, 0x00020000);
        float4 xr[D][XR][2], dr[D][DR][2], ir[D];
        const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)(IMG ? (const void*)p.img : (const void*)p.dy), 0, IMG ? p.img_bytes : 0u, 0x00020000);
        const int i_px = lw * 17 + lane;   // IMG: 17 pixels of the 66-pixel image row per loader wave
        int lb[D][9];   // PARTIAL, loader wave 0: the 3x3 label neighbourhood of output column `lane`
        unsigned lbok[D];

        int iu = u0;
        Unit un = decode(p, iu);
        int iy = un.ya - 2;

        // Every call issues the SAME loads in one basic block (out-of-range work gets out-of-bounds offsets, which fetch nothing): the
        // compiler's wait-count pass then knows exactly how many younger loads are in flight when a set is consumed -- with a branch
        // around any of them it has to assume the worst and waits for the newest set too, which exposes the memory latency once per row.
        bool live = true;   // false once the cursor has run past this block's last unit
        auto issue = [&](int d) {
            // input row y + 1 of the step at the issue cursor, its dY row y and (PARTIAL) the labels its tap masks need
#ifdef WS_NOLOAD
            const int y = p.ktot >= 0 ? -100000 : iy;
#else
            const int y = iy;
#endif
            const int cbi = un.tm * MB + x_mb;
            const bool s1 = cbi >= p.s[0].blocks;
            const int blk = s1 ? cbi - p.s[0].blocks : cbi;
            const int ld = s1 ? p.s[1].ld : p.s[0].ld;
            const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(s1 ? p.s[1].data : p.s[0].data), 0, s1 ? p.s[1].bytes : p.s[0].bytes, 0x00020000);
            const bool row_ok = live && cbi < p.cblocks && (unsigned)(y + 1) < (unsigned)p.H;
            const int rowpix = (un.n * p.H + y + 1) * p.W;
#pragma unroll
            for (int r = 0; r < XR; ++r) {
                const int px = x_px0 + r * XPR;
                const int xx = un.x0 - 1 + px;
                const bool ok = row_ok && px < XC && (unsigned)xx < (unsigned)p.W;
                const unsigned off = (unsigned)(((rowpix + xx) * ld + blk * 32 + oct * 8) * 4) | (ok ? 0u : OOB);   // OR, not select: no branch around the load
                xr[d][r][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)off, 0, 0));
                xr[d][r][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)(off + 16u), 0, 0));
            }
            const int co = (un.tn * NB + d_nb) * 32 + oct * 8;
            const bool drow_ok = live && y >= un.ya && co < p.Cout;
            const int dpix = (un.n * p.H + y) * p.W;
#pragma unroll
            for (int r = 0; r < DR; ++r) {
                const int x = un.x0 + d_px0 + r * DPR;
                const bool ok = drow_ok && x < p.W;
                const unsigned off = (unsigned)(((dpix + x) * p.dy_ld + co) * 4) | (ok ? 0u : OOB);
                dr[d][r][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsd, (int)off, 0, 0));
                dr[d][r][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsd, (int)(off + 16u), 0, 0));
            }
            if constexpr (IMG) {
                const int xx = un.x0 - 1 + i_px;
                const bool ok = live && lane < 17 && i_px < XC && (unsigned)(y + 1) < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                ir[d] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsi, (int)((unsigned)((rowpix + xx) * p.img_ld * 4) | (ok ? 0u : OOB)), 0, 0));
            }
            if constexpr (PARTIAL) {
                // lane = output column (loader wave 0 only; the others issue the same nine loads out of bounds); the comparison happens in
                // write(): nothing here may wait for a load.  Bit t of lbok = neighbour t is inside the image.
                const int x = un.x0 + lane;
                const bool cok = live && lw == 0 && y >= un.ya && x < p.W;
                unsigned okb = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    const bool ok = cok && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                    lb[d][t] = (int)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rsl, (int)((unsigned)((un.n * p.H + yy) * p.W + xx) | (ok ? 0u : OOB)), 0, 0);
                    okb |= ok ? (1u << t) : 0u;
                }
                lbok[d] = okb;
            }
            // advance the issue cursor
            if (++iy >= un.yb) {
                if (++iu < u1) {
                    un = decode(p, iu);
                    iy = un.ya - 2;
                } else {
                    live = false;
                }
            }
        };
        auto write = [&](int d, int T) {
#ifdef WS_NOWRITE
            if (p.ktot >= 0) return;
#endif
            unsigned char* xb = Xs + (T & 3) * XSLOT + x_mb * XROWB + oct * 16;
#pragma unroll
            for (int r = 0; r < XR; ++r) {
                const int px = x_px0 + r * XPR;
                if (px < XC) store_planes<NP>(xb + px * 64, XPLANE, xr[d][r][0], xr[d][r][1]);
            }
            unsigned char* db = Ds + (T & 1) * DSLOT + d_nb * DROWB + oct * 16;
#pragma unroll
            for (int r = 0; r < DR; ++r) store_planes<NP>(db + (d_px0 + r * DPR) * 64, DPLANE, dr[d][r][0], dr[d][r][1]);
            if constexpr (IMG) {
                if (lane < 17 && i_px < XC) {
                    unsigned char* ib = Is + (T & 3) * ISLOT + i_px * 8;
                    if constexpr (NP == 2) {
                        uint2 h, l;
                        cp::split4h(ir[d], h, l);
                        *reinterpret_cast<uint2*>(ib) = h;
                        *reinterpret_cast<uint2*>(ib + IPLANE) = l;
                    } else if constexpr (NP == 3) {
                        uint2 h, m, l;
                        split4(ir[d], h, m, l);
                        *reinterpret_cast<uint2*>(ib) = h;
                        *reinterpret_cast<uint2*>(ib + IPLANE) = m;
                        *reinterpret_cast<uint2*>(ib + 2 * IPLANE) = l;
                    } else {
                        *reinterpret_cast<uint2*>(ib) = round4(ir[d]);
                    }
                }
            }
            if constexpr (PARTIAL) {
                if (lw == 0) {
#pragma unroll
                    for (int t = 0; t < 9; ++t)   // out-of-image neighbours never match (their X is zero anyway)
                        Ms[((T & 1) * 9 + t) * SW + lane] = (((lbok[d] >> t) & 1u) && lb[d][t] == lb[d][4]) ? (unsigned short)0xffffu : (unsigned short)0;
                }
            }
        };

#ifdef WS_CONSUMER_ONLY   // timing ablation: the loaders fill the rings once with pseudo-random bits and leave; the consumers run alone
        if (p.ktot >= 0) {
            unsigned* w = reinterpret_cast<unsigned*>(smem);
            const int nw = (NXS * XSLOT + NDS * DSLOT) / 4;
            for (int i = threadIdx.x - 256; i < nw; i += 256) w[i] = (0x3f803f80u ^ ((unsigned)i * 2654435761u)) & 0x3fff3fffu;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            return;
        }
#endif
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d);
        for (int T = 0; T < NTP; T += D) {   // NTP = NT rounded up to a multiple of D: the padding steps move nothing and the consumers only meet their barriers
#pragma unroll
            for (int d = 0; d < D; ++d) {
                write(d, T + d);
                issue(d);
                WS_BARRIER();
            }
        }
        return;
    }

    // ---------------------------------------------------- consumers -------------------------------------------------------------
#ifdef WS_PRIO
    __builtin_amdgcn_s_setprio(WS_PRIO);
#endif
    const int mb = wave % MB, nb = (wave / MB) % NB, ph = wave / (MB * NB);
    const int kg = lane >> 5, half = (lane >> 4) & 1, li = lane & 15;
    const int lane_off = (8 * kg + (li >> 2)) * 64 + (16 * half + 4 * (li & 3)) * 2;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x16 acci[IMG ? 2 : 1];   // IMG: columns (tap, channel) of the image source: taps 0..7 | tap 8
#pragma unroll
    for (int b = 0; b < (IMG ? 2 : 1); ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acci[b][r] = 0.f;

    auto flush = [&](const Unit& un) {
        const int cbi = un.tm * MB + mb;
#ifdef WS_NOFLUSH   // timing ablations (tools/build_variant.sh), never in the shipped library
        if (cbi < p.cblocks && p.ktot < 0) {
#else
        if (cbi < p.cblocks) {
#endif
            const bool s1 = cbi >= p.s[0].blocks;
            const int blk = s1 ? cbi - p.s[0].blocks : cbi;
            const int C = s1 ? p.s[1].C : p.s[0].C;
            const int kb = (s1 ? p.s[1].kbase : p.s[0].kbase) + blk * 32 + (lane & 31);
            const int co0 = (un.tn * NB + nb) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (r & 3) + 8 * (r >> 2);
                    if (co < p.Cout) atomicAdd(p.dw + (size_t)co * p.ktot + kb + t * C, acc[t][r]);
                }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        if constexpr (IMG) {
            const int co0 = un.tn * 32 + 4 * (lane >> 5), n = lane & 31;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (r & 3) + 8 * (r >> 2);
                    if (co < p.Cout && (b == 0 || n < 4)) atomicAdd(p.dw + (size_t)co * p.ktot + p.img_kbase + b * 32 + n, acci[b][r]);
                    acci[b][r] = 0.f;
                }
        }
    };

    int T = 0;
    Unit un = decode(p, u0);
    for (int u = u0; u < u1; ++u) {
        const Unit nu = decode(p, u);
        if (nu.pair != un.pair) flush(un);
        un = nu;
        for (int y = un.ya - 2; y < un.yb; ++y, ++T) {
            WS_BARRIER();
            if (y < un.ya) continue;
#ifdef WS_NOMFMA
            if (p.ktot >= 0) continue;
#endif
            // software pipeline over the row's tap steps (16 pixels x one tap): the fragments of step ts + 1 are read while the six MFMAs of
            // step ts issue; one scheduling fence per step keeps that order (and the register count: two fragment sets, 144 accumulators)
            const unsigned char* dbase = Ds + (T & 1) * DSLOT + nb * DROWB + lane_off + ph * KSW * 1024;
            const unsigned char* xb[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) xb[ky] = Xs + ((T + 2 + ky) & 3) * XSLOT + mb * XROWB + lane_off + ph * KSW * 1024;
            const unsigned char* mbase = reinterpret_cast<const unsigned char*>(Ms) + (T & 1) * (9 * SW * 2) + (16 * ph * KSW + 8 * kg) * 2;
            constexpr int NTS = 9 * KSW;
            bf16x8 a[2][NP], b[2][NP];
            uint4 mw[2];
            auto load_a = [&](bf16x8(&dst)[NP], int j) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) dst[pl] = frag_tr(dbase + pl * DPLANE + j * 1024);
            };
            auto load_b = [&](bf16x8(&dst)[NP], uint4& m, int ts) {
                const int j = ts / 9, t = ts % 9;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) dst[pl] = frag_tr(xb[t / 3] + pl * XPLANE + j * 1024 + (t % 3) * 64);
                if constexpr (PARTIAL) m = *reinterpret_cast<const uint4*>(mbase + t * (SW * 2) + j * 32);
            };
            // 16-pixel steps of this wave that lie inside the image (a strip's tail may be empty: W = 112 fills 7 of its 8 steps)
            const int nks = min(KSW, (p.W - un.x0 - 16 * ph * KSW + 15) >> 4);
            if (nks <= 0) continue;
            load_a(a[0], 0);
            load_b(b[0], mw[0], 0);
            static_for<0, NTS>([&](auto tsc) {
                constexpr int ts = decltype(tsc)::value;
                constexpr int j = ts / 9, t = ts % 9;
                if (j >= nks) return;
                if (ts + 1 < NTS) load_b(b[(ts + 1) & 1], mw[(ts + 1) & 1], ts + 1);
                if (t == 6 && j + 1 < KSW) load_a(a[(j + 1) & 1], j + 1);
                bf16x8(&bb)[NP] = b[ts & 1];
                bf16x8(&aa)[NP] = a[j & 1];
                if constexpr (PARTIAL) {   // zero the pixels whose tap neighbour carries another label: 8 x 16-bit masks of this lane's pixels
                    const uint4 m = mw[ts & 1];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        uint4 bv = __builtin_bit_cast(uint4, bb[pl]);
                        bv.x &= m.x; bv.y &= m.y; bv.z &= m.z; bv.w &= m.w;
                        bb[pl] = __builtin_bit_cast(bf16x8, bv);
                    }
                }
                f32x16& c = acc[t];
                if constexpr (NP == 2) {   // lo * hi, hi * lo, hi * hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cp::f16x8_t, aa[1]), __builtin_bit_cast(cp::f16x8_t, bb[0]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cp::f16x8_t, aa[0]), __builtin_bit_cast(cp::f16x8_t, bb[1]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cp::f16x8_t, aa[0]), __builtin_bit_cast(cp::f16x8_t, bb[0]), c, 0, 0, 0);
                } else if constexpr (NP == 3) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[2], bb[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[0], bb[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[1], bb[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[1], bb[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[0], bb[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[0], bb[0], c, 0, 0, 0);
                } else {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[0], bb[0], c, 0, 0, 0);
                }
#ifndef WS_NO_SCHED_BARRIER
                if constexpr (NP == 3) {   // (one MFMA per step with NP = 1: the scheduler needs the freedom there)
                    // order inside the step: the next step's transpose reads go out between the first MFMAs, so they have landed (>= 96 cycles)
                    // when the next step starts -- left alone the scheduler sinks them behind the last use of the registers they reuse, i.e.
                    // to the end of the step, and every step then begins with an LDS-latency stall
                    constexpr int NR = (ts + 1 < NTS ? 2 * NP : 0) + ((ts % 9) == 6 && ts / 9 + 1 < KSW ? 2 * NP : 0) + ((PARTIAL && ts + 1 < NTS) ? 1 : 0);
                    constexpr int R0 = (NR + 2) / 3, R1 = (NR - R0 + 1) / 2, R2 = NR - R0 - R1;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (R0 > 0) __builtin_amdgcn_sched_group_barrier(0x100, R0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (R1 > 0) __builtin_amdgcn_sched_group_barrier(0x100, R1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (R2 > 0) __builtin_amdgcn_sched_group_barrier(0x100, R2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
            });
            if constexpr (IMG) {   // KSW = 1: this wave's 16 pixels are step `ph`; a[0] still holds their dY fragments
                const int kgp = 16 * ph + 8 * kg + (li >> 2);   // this lane's SUPPLIED pixel (transpose read: row li >> 2 of the group's 4)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int tap_s = b == 0 ? 4 * half + (li & 3) : 8;    // the tap whose 4 channels this lane supplies
                    const int tap_r = b == 0 ? 4 * half + (li >> 2) : 8;   // the tap of the column this lane receives
                    const unsigned char* ib = Is + ((T + 2 + tap_s / 3) & 3) * ISLOT + (kgp + tap_s % 3) * 8;
                    bf16x8 bi[NP];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        typedef s16x4 __attribute__((address_space(3))) * lds_p;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ib + pl * IPLANE));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ib + pl * IPLANE + 4 * 8));
                        bi[pl] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                    if constexpr (PARTIAL) {
                        const uint4 m = *reinterpret_cast<const uint4*>(mbase + tap_r * (SW * 2));
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) {
                            uint4 bv = __builtin_bit_cast(uint4, bi[pl]);
                            bv.x &= m.x; bv.y &= m.y; bv.z &= m.z; bv.w &= m.w;
                            bi[pl] = __builtin_bit_cast(bf16x8, bv);
                        }
                    }
                    f32x16& c = acci[b];
                    if constexpr (NP == 2) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cp::f16x8_t, a[0][1]), __builtin_bit_cast(cp::f16x8_t, bi[0]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cp::f16x8_t, a[0][0]), __builtin_bit_cast(cp::f16x8_t, bi[1]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cp::f16x8_t, a[0][0]), __builtin_bit_cast(cp::f16x8_t, bi[0]), c, 0, 0, 0);
                    } else if constexpr (NP == 3) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][2], bi[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], bi[2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], bi[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], bi[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], bi[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], bi[0], c, 0, 0, 0);
                    } else {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], bi[0], c, 0, 0, 0);
                    }
                }
            }
        }
    }
    for (; T < NTP; ++T) WS_BARRIER();
    flush(un);
}

template <int NP, int MB, int NB, bool PARTIAL, bool IMG = false>
int launch(WSplitK k, hipStream_t st) {
    k.tiles_m = (k.cblocks + MB - 1) / MB;
    k.tiles_n = (k.Cout / 32 + NB - 1) / NB;
    const int P = k.tiles_m * k.tiles_n;
    k.strips = (k.W + SW - 1) / SW;
    const int J0 = k.B * k.strips;
    // rows per unit: every unit costs its rows + 2 fill steps, and a block walks ceil(U / G) units -- take the split of the image height
    // that minimises the longest block (ties: fewer, taller units)
    long long best = -1;
    for (int c = 1; c <= std::max(1, k.H / 8); ++c) {
        const int rc = (k.H + c - 1) / c, cc = (k.H + rc - 1) / rc;
        const long long U = (long long)P * J0 * cc, G = std::min<long long>(256, U);
        const long long cost = ((U + G - 1) / G) * (rc + 2);
        if (best < 0 || cost < best) {
            best = cost;
            k.rc = rc;
            k.chunks = cc;
        }
    }
    k.J = J0 * k.chunks;
    const long long U = (long long)P * k.J;
    if (U >= (1LL << 30)) {
        cp::set_error("cp_conv2d_wgrad_split: too many work units");
        return CP_ERR_INVALID;
    }
    k.U = (int)U;
    const int G = (int)std::min<long long>(256, U);
    const size_t lds = (size_t)NXS * NP * MB * XROWB + (size_t)NDS * NP * NB * DROWB + (PARTIAL ? NDS * 9 * SW * 2 : 0) + (IMG ? NXS * NP * XC * 8 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_split_kernel<NP, MB, NB, PARTIAL, IMG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    CP_LAUNCH((wgrad_split_kernel<NP, MB, NB, PARTIAL, IMG>), dim3((unsigned)G), dim3(512), lds, st, k);
    return cp::check_launch("cp_conv2d_wgrad_split");
}

template <int NP, bool PARTIAL>
int launch_shape(const WSplitK& k, hipStream_t st) {
    const int nbk = k.Cout / 32;
    if (k.cblocks >= 2) return nbk >= 2 ? launch<NP, 2, 2, PARTIAL>(k, st) : launch<NP, 2, 1, PARTIAL>(k, st);
    if (nbk >= 2) return launch<NP, 1, 2, PARTIAL>(k, st);
    return k.img ? launch<NP, 1, 1, PARTIAL, true>(k, st) : launch<NP, 1, 1, PARTIAL>(k, st);
}

bool applicable(const cp_conv_desc* d) {
    if (!d || d->struct_size != (uint32_t)sizeof(cp_conv_desc)) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->dilation != 1 || d->pad != 1) return false;
    if (d->num_sources < 1 || d->num_sources > 2 || d->group_rows) return false;
    if (d->cout <= 0 || d->cout % 32) return false;
    if (d->out_h != d->in_h || d->out_w != d->in_w) return false;
    int big = 0;
    for (int s = 0; s < d->num_sources; ++s) {
        const cp_conv_source& in = d->src[s];
        if (!in.data || in.mode != CP_SRC_DIRECT || in.pre_scale) return false;
        if (in.channels == 4) {
            if (s != d->num_sources - 1) return false;   // the image: handled by the fp32 kernel, must be the trailing source
            continue;
        }
        if (in.channels <= 0 || in.channels % 32 || in.ld < in.channels || in.ld % 4 || ((uintptr_t)in.data & 15)) return false;
        if ((long long)d->batch * d->in_h * d->in_w * in.ld * 4 >= (1LL << 31)) return false;
        ++big;
    }
    return big >= 1;
}

}  // namespace

extern "C" int cp_conv_wgrad_split_applicable(const cp_conv_desc* d) { return applicable(d) ? 1 : 0; }

extern "C" int cp_conv2d_wgrad_split(const cp_conv_desc* d, const float* dy, int dy_ld, float* dw_packed, int accumulate, int planes, void* stream) {
    CP_REQUIRE_DESC(d, "cp_conv2d_wgrad_split");
    CP_REQUIRE(dy && dw_packed, "cp_conv2d_wgrad_split: null pointer");
    CP_REQUIRE(planes == 1 || planes == 3 || planes == CP_PLANES_F16X2, "cp_conv2d_wgrad_split: planes must be 3 (exact split), 1 (bf16) or CP_PLANES_F16X2");
    CP_REQUIRE(applicable(d), "cp_conv2d_wgrad_split: descriptor not covered (3x3 / stride 1 / pad 1, direct sources with 32-multiple channels + "
                              "an optional trailing 4-channel source, cout a multiple of 32); see cp_conv_wgrad_split_applicable");
    CP_REQUIRE(dy_ld >= d->cout && dy_ld % 4 == 0 && ((uintptr_t)dy & 15) == 0, "cp_conv2d_wgrad_split: dy_ld must be a multiple of 4 and >= cout, dy 16-byte aligned");
    const long long M = (long long)d->batch * d->out_h * d->out_w;
    CP_REQUIRE(M * dy_ld * 4 < (1LL << 31), "cp_conv2d_wgrad_split: dy spans >= 2 GiB");
    WSplitK k{};
    int chans[2] = {0, 0};
    int kbase = 0, first_small_chunk = -1;
    for (int s = 0; s < d->num_sources; ++s) {
        const cp_conv_source& in = d->src[s];
        chans[s] = in.channels;
        if (in.channels == 4) {
            first_small_chunk = kbase / 32;
            continue;
        }
        WsSrc& o = k.s[s];
        o.data = in.data; o.ld = in.ld; o.C = in.channels; o.blocks = in.channels / 32; o.kbase = kbase;
        o.bytes = (unsigned)((long long)d->batch * d->in_h * d->in_w * in.ld * 4);
        k.cblocks += o.blocks;
        kbase += 9 * in.channels;
    }
    k.ktot = cp_conv_ktot(3, 3, d->num_sources, chans);
    k.dy = dy; k.dy_ld = dy_ld; k.dy_bytes = (unsigned)(M * dy_ld * 4);
    k.label = d->tap_label; k.lab_bytes = (unsigned)((long long)d->batch * d->in_h * d->in_w);
    k.dw = dw_packed;
    k.B = d->batch; k.H = d->in_h; k.W = d->in_w; k.Cout = d->cout;
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate)
        if (hipMemsetAsync(dw_packed, 0, sizeof(float) * (size_t)d->cout * k.ktot, st) != hipSuccess) return cp::check_launch("cp_conv2d_wgrad_split memset");
    // the 4-channel image source (K = 36 of a few hundred): inside the kernel as a column block of its own when the rest is one 32 x 32 tile
    // (decoder blocks 5 / 10), otherwise its columns of dWp come from the fp32 kernel restricted to those chunks
    const cp_conv_source& last = d->src[d->num_sources - 1];
    const bool img_inside = first_small_chunk >= 0 && k.cblocks == 1 && d->cout == 32 && last.ld % 4 == 0 && ((uintptr_t)last.data & 15) == 0 &&
                            (long long)d->batch * d->in_h * d->in_w * last.ld * 4 < (1LL << 31) && !getenv("CP_WGRAD_IMG_F32");
    if (img_inside) {
        const cp_conv_source& in = last;
        k.img = in.data; k.img_ld = in.ld; k.img_kbase = first_small_chunk * 32;
        k.img_bytes = (unsigned)((long long)d->batch * d->in_h * d->in_w * in.ld * 4);
    }
    int rc;
    if (planes == CP_PLANES_F16X2) rc = d->tap_label ? launch_shape<2, true>(k, st) : launch_shape<2, false>(k, st);
    else if (planes == 3) rc = d->tap_label ? launch_shape<3, true>(k, st) : launch_shape<3, false>(k, st);
    else rc = d->tap_label ? launch_shape<1, true>(k, st) : launch_shape<1, false>(k, st);
    if (rc != CP_OK) return rc;
    if (first_small_chunk >= 0 && !img_inside) return cp::wgrad_f32_chunks(d, dy, dy_ld, dw_packed, first_small_chunk, st);
    return CP_OK;
}
