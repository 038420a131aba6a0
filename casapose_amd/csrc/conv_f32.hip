// Fused implicit-GEMM convolution forward, exact fp32 on v_mfma_f32_32x32x2_f32 (gfx950).
//
// GEMM view: M = batch*out_h*out_w output pixels, N = cout, K = sum over sources of
// taps*channels.  A (pixels x K) is gathered on the fly from the NHWC activations --
// with the source's spatial mode (direct / guided-nearest x2 / bilinear x2), the optional
// per-channel affine and the partial-convolution tap mask applied while the tile is in
// registers -- and staged through LDS; B (cout x K) comes from the pre-packed weights.
// A block of 256 threads (4 waves, one per SIMD) owns a BM x BN output tile; each wave
// owns TM x TN accumulators of 32x32.  K advances in chunks of 32 with register-staged
// double buffering: global loads for chunk q+1 are in flight while chunk q is multiplied.
//
// Operand trick: an MFMA 32x32x2 consumes ONE f32 of A and B per lane (lane l supplies
// k = l>>5).  The order of K inside a GEMM is free as long as A and B agree, so both
// tiles are stored [row][k] with k contiguous and read with one ds_read_b128 per 8 k:
// half-wave 0 takes k 0..3, half-wave 1 takes k 4..7, and MFMA step j multiplies the
// pair (j, 4+j).  Row stride 36 floats (9 x 16 B) makes those reads conflict-free.
//
// Reference call sites replaced: see include/casapose_hip.h (cp_conv2d_fwd_f32).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_STRIDE = 36;  // floats per tile row (32 + 4 pad)
constexpr int MAX_TAPS = 64;

struct SrcK {
    const float* data;
    const uint8_t* sel;
    const float* pre_scale;
    const float* pre_shift;
    int C, ld, mode, Hs, Ws;
    int c4;       // 1: channels == 4, eight taps per chunk
    int cpt;      // chunks per tap (C/32) when !c4
    int nchunks;  // K chunks contributed by this source
};

struct ConvK {
    SrcK s[2];
    const float* W;
    int ktot;
    int B, Hin, Win, Ho, Wo, Cout, KH, KW, stride, dil, pad;
    int M;
    const uint8_t* tap_label;
    const float* row_scale;
    const float* residual;
    int res_ld;
    const float* scale;
    const float* shift;
    const uint8_t* epi_label;
    int act;
    float* out_raw;
    int raw_ld;
    float* out_act;
    int act_ld;
    int tiles_m, tiles_n, nchunks;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int WGM, int WGN, int TM, int TN>
__global__ __launch_bounds__(256, 2) void conv_f32_kernel(const ConvK p) {
    constexpr int BM = WGM * TM * 32;
    constexpr int BN = WGN * TN * 32;
    constexpr int RM = BM / 32;  // A rows staged per thread
    constexpr int RN = BN / 32;  // B rows staged per thread
    static_assert(WGM * WGN == 4, "4 waves per block");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                            // [2][BM][LDS_STRIDE]
    float* Bs = smem + 2 * BM * LDS_STRIDE;      // [2][BN][LDS_STRIDE]
    int* tapoff = reinterpret_cast<int*>(Bs + 2 * BN * LDS_STRIDE);  // [MAX_TAPS] (dy<<16)|(dx & 0xffff)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    const int logical = cp::xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tile_n = logical % p.tiles_n;
    const int tile_m = logical / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const int ntaps = p.KH * p.KW;
    if (tid < MAX_TAPS) {
        int ky = tid / p.KW, kx = tid - ky * p.KW;
        tapoff[tid] = ((ky * p.dil) << 16) | ((kx * p.dil) & 0xffff);
    }

    // ---- per-thread staging coordinates -------------------------------------------------
    const int col4 = tid & 7;   // which float4 of the 32-wide K chunk
    const int rbase = tid >> 3; // 0..31
    int r_n[RM], r_iy0[RM], r_ix0[RM], r_clab[RM];
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        int m = m0 + rbase + 32 * i;
        if (m < p.M) {
            int n = m / (p.Ho * p.Wo);
            int rem = m - n * (p.Ho * p.Wo);
            int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            r_n[i] = n;
            r_iy0[i] = oy * p.stride - p.pad;
            r_ix0[i] = ox * p.stride - p.pad;
            r_clab[i] = p.tap_label ? (int)p.tap_label[((size_t)n * p.Hin + oy) * p.Win + ox] : 0;
        } else {
            r_n[i] = 0;
            r_iy0[i] = -0x10000000;  // always out of bounds (also when added to the invalid-tap dy)
            r_ix0[i] = 0;
            r_clab[i] = -1;
        }
    }
    __syncthreads();  // tapoff visible

    float4 areg[RM], breg[RN];

    auto load_chunk = [&](int q) {
        // ---- B: packed weights, always in bounds along K --------------------------------
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            int co = n0 + rbase + 32 * j;
            breg[j] = (co < p.Cout) ? ld4(p.W + (size_t)co * p.ktot + q * BK + col4 * 4) : make_float4(0, 0, 0, 0);
        }
        // ---- A: gather ------------------------------------------------------------------
        const int si = (q >= p.s[0].nchunks) ? 1 : 0;
        const SrcK& s = p.s[si];
        const int ql = q - (si ? p.s[0].nchunks : 0);
        int tap, coff;
        if (s.c4) {
            tap = ql * 8 + col4;
            coff = 0;
        } else {
            tap = ql / s.cpt;
            coff = (ql - tap * s.cpt) * 32 + col4 * 4;
        }
        const int to = tapoff[tap < MAX_TAPS ? tap : MAX_TAPS - 1];
        const int dy = (tap < ntaps) ? (to >> 16) : 0x20000000;
        const int dx = (int)(short)(to & 0xffff);
        float4 ps = make_float4(1, 1, 1, 1), pb = make_float4(0, 0, 0, 0);
        const bool has_pre = s.pre_scale != nullptr;
        if (has_pre) {
            ps = ld4(s.pre_scale + coff);
            pb = ld4(s.pre_shift + coff);
        }
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            const int iy = r_iy0[i] + dy, ix = r_ix0[i] + dx;
            bool inb = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);
            const size_t gpix = ((size_t)r_n[i] * p.Hin + (inb ? iy : 0)) * p.Win + (inb ? ix : 0);
            if (p.tap_label) {
                int lab = inb ? (int)p.tap_label[gpix] : -2;
                inb = inb && (lab == r_clab[i]);
            }
            float4 v = make_float4(0, 0, 0, 0);
            if (inb) {
                if (s.mode == CP_SRC_DIRECT) {
                    v = ld4(s.data + gpix * s.ld + coff);
                } else if (s.mode == CP_SRC_NEAREST_SEL) {
                    int sl = s.sel[gpix];
                    int sy = (iy >> 1) + (sl >> 1), sx = (ix >> 1) + (sl & 1);
                    v = ld4(s.data + (((size_t)r_n[i] * s.Hs + sy) * s.Ws + sx) * s.ld + coff);
                } else {  // CP_SRC_BILINEAR_X2, half-pixel centres
                    int y0 = (iy >> 1) - ((iy & 1) ? 0 : 1), x0 = (ix >> 1) - ((ix & 1) ? 0 : 1);
                    float fy = (iy & 1) ? 0.25f : 0.75f, fx = (ix & 1) ? 0.25f : 0.75f;
                    int y1 = min(y0 + 1, s.Hs - 1), x1 = min(x0 + 1, s.Ws - 1);
                    y0 = max(y0, 0);
                    x0 = max(x0, 0);
                    const float* base = s.data + (size_t)r_n[i] * s.Hs * s.Ws * s.ld + coff;
                    float4 v00 = ld4(base + ((size_t)y0 * s.Ws + x0) * s.ld);
                    float4 v01 = ld4(base + ((size_t)y0 * s.Ws + x1) * s.ld);
                    float4 v10 = ld4(base + ((size_t)y1 * s.Ws + x0) * s.ld);
                    float4 v11 = ld4(base + ((size_t)y1 * s.Ws + x1) * s.ld);
                    float gx = 1.f - fx, gy = 1.f - fy;
                    v.x = (v00.x * gx + v01.x * fx) * gy + (v10.x * gx + v11.x * fx) * fy;
                    v.y = (v00.y * gx + v01.y * fx) * gy + (v10.y * gx + v11.y * fx) * fy;
                    v.z = (v00.z * gx + v01.z * fx) * gy + (v10.z * gx + v11.z * fx) * fy;
                    v.w = (v00.w * gx + v01.w * fx) * gy + (v10.w * gx + v11.w * fx) * fy;
                }
                if (has_pre) {
                    v.x = v.x * ps.x + pb.x;
                    v.y = v.y * ps.y + pb.y;
                    v.z = v.z * ps.z + pb.z;
                    v.w = v.w * ps.w + pb.w;
                }
            }
            areg[i] = v;
        }
    };

    auto store_chunk = [&](int buf) {
        float* a = As + buf * BM * LDS_STRIDE;
        float* b = Bs + buf * BN * LDS_STRIDE;
#pragma unroll
        for (int i = 0; i < RM; ++i)
            *reinterpret_cast<float4*>(a + (rbase + 32 * i) * LDS_STRIDE + col4 * 4) = areg[i];
#pragma unroll
        for (int j = 0; j < RN; ++j)
            *reinterpret_cast<float4*>(b + (rbase + 32 * j) * LDS_STRIDE + col4 * 4) = breg[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lrow = lane & 31;
    const int khalf = (lane >> 5) * 4;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    for (int q = 0; q < p.nchunks; ++q) {
        const int buf = q & 1;
        const bool more = (q + 1) < p.nchunks;
        if (more) load_chunk(q + 1);

        const float* a = As + buf * BM * LDS_STRIDE + (wm * TM * 32 + lrow) * LDS_STRIDE + khalf;
        const float* b = Bs + buf * BN * LDS_STRIDE + (wn * TN * 32 + lrow) * LDS_STRIDE + khalf;
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            float4 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const float4*>(a + i * 32 * LDS_STRIDE + k8 * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const float4*>(b + j * 32 * LDS_STRIDE + k8 * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----------------------------------------------------------------------
    const int hi4 = (lane >> 5) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + hi4;
            if (m >= p.M) continue;
            const float rs = p.row_scale ? p.row_scale[m] : 1.f;
            const int lab = p.epi_label ? (int)p.epi_label[m] : 0;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = n0 + (wn * TN + j) * 32 + lrow;
                if (co >= p.Cout) continue;
                float v = acc[i][j][r] * rs;
                if (p.residual) v += p.residual[(size_t)m * p.res_ld + co];
                if (p.out_raw) p.out_raw[(size_t)m * p.raw_ld + co] = v;
                if (p.out_act) {
                    float t = v;
                    if (p.scale) t = t * p.scale[lab * p.Cout + co] + p.shift[lab * p.Cout + co];
                    if (p.act == CP_ACT_RELU) t = fmaxf(t, 0.f);
                    else if (p.act == CP_ACT_LEAKY01) t = fmaxf(t, 0.f) - fmaxf(-0.1f * t, 0.f);
                    p.out_act[(size_t)m * p.act_ld + co] = t;
                }
            }
        }
    }
}

template <int WGM, int WGN, int TM, int TN>
int launch(const ConvK& k, hipStream_t st) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    ConvK kk = k;
    kk.tiles_m = (k.M + BM - 1) / BM;
    kk.tiles_n = (k.Cout + BN - 1) / BN;
    size_t lds = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(float) + MAX_TAPS * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f32_kernel<WGM, WGN, TM, TN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid(kk.tiles_m * kk.tiles_n);
    CP_LAUNCH((conv_f32_kernel<WGM, WGN, TM, TN>), grid, dim3(256), lds, st, kk);
    return cp::check_launch("cp_conv2d_fwd_f32");
}

int chunks_for(int taps, int C) { return (C == 4) ? (taps + 7) / 8 : taps * (C / 32); }

// Pick the tile that wastes the least MFMA work once padding of N, padding of M and the
// last partial round over the 256 CUs are accounted for; ties go to the larger tile.
int pick_tile(long long M, int N) {
    struct T { int id, bm, bn; };
    static const T tiles[] = {{CP_TILE_128x128, 128, 128}, {CP_TILE_64x128, 64, 128}, {CP_TILE_128x64, 128, 64},
                              {CP_TILE_64x64, 64, 64},     {CP_TILE_256x32, 256, 32}, {CP_TILE_128x32, 128, 32}};
    double best = 1e30;
    int best_id = CP_TILE_128x128;
    for (const T& t : tiles) {
        long long tm = (M + t.bm - 1) / t.bm, tn = (N + t.bn - 1) / t.bn;
        long long nt = tm * tn;
        long long rounds = (nt + 255) / 256;
        double cost = (double)rounds * 256.0 * t.bm * t.bn;  // MFMA work the chip spends
        // small tiles pay relatively more staging per MFMA: mild penalty
        cost *= 1.0 + 2.0 / (t.bm < t.bn ? t.bm : t.bn);
        if (cost < best * 0.999) {
            best = cost;
            best_id = t.id;
        }
    }
    return best_id;
}

}  // namespace

extern "C" int cp_conv_ktot(int kh, int kw, int num_sources, const int* channels) {
    int chunks = 0;
    for (int s = 0; s < num_sources; ++s) chunks += chunks_for(kh * kw, channels[s]);
    return chunks * BK;
}

extern "C" int cp_conv_pack_weights_host(const float* w, int layout, int kh, int kw, int cout, int num_sources,
                                         const int* channels, const int* real_channels, float* dst) {
    CP_REQUIRE(w && dst && channels && real_channels, "cp_conv_pack_weights_host: null pointer");
    CP_REQUIRE(num_sources == 1 || num_sources == 2, "cp_conv_pack_weights_host: num_sources must be 1 or 2");
    const int taps = kh * kw;
    int cin = 0;
    for (int s = 0; s < num_sources; ++s) {
        CP_REQUIRE(channels[s] == 4 || (channels[s] > 0 && channels[s] % 32 == 0),
                   "cp_conv_pack_weights_host: source channels must be 4 or a multiple of 32 (got %d)", channels[s]);
        CP_REQUIRE(real_channels[s] > 0 && real_channels[s] <= channels[s], "cp_conv_pack_weights_host: bad real_channels");
        cin += real_channels[s];
    }
    const int ktot = cp_conv_ktot(kh, kw, num_sources, channels);
    for (size_t i = 0; i < (size_t)cout * ktot; ++i) dst[i] = 0.f;
    int kbase = 0, cbase = 0;
    for (int s = 0; s < num_sources; ++s) {
        const int C = channels[s], Cr = real_channels[s];
        for (int t = 0; t < taps; ++t) {
            const int ky = t / kw, kx = t % kw;
            for (int c = 0; c < Cr; ++c) {
                const int k = kbase + t * C + c;  // both modes: tap-major, channel-minor
                const int ci = cbase + c;
                for (int co = 0; co < cout; ++co) {
                    size_t src = (layout == 0) ? ((((size_t)ky * kw + kx) * cin + ci) * cout + co)
                                               : ((((size_t)ci * kh + ky) * kw + kx) * cout + co);
                    dst[(size_t)co * ktot + k] = w[src];
                }
            }
        }
        kbase += chunks_for(taps, C) * BK;
        cbase += Cr;
    }
    return CP_OK;
}

extern "C" int cp_conv_selected_tile(const cp_conv_desc* d) {
    if (!d) return CP_ERR_INVALID;
    return d->tile_hint ? d->tile_hint : pick_tile((long long)d->batch * d->out_h * d->out_w, d->cout);
}

extern "C" int cp_conv2d_fwd_f32(const cp_conv_desc* d, void* stream) {
    CP_REQUIRE(d, "cp_conv2d_fwd_f32: null descriptor");
    CP_REQUIRE(d->num_sources == 1 || d->num_sources == 2, "cp_conv2d_fwd_f32: num_sources must be 1 or 2");
    CP_REQUIRE(d->kh * d->kw <= MAX_TAPS && d->kh > 0 && d->kw > 0, "cp_conv2d_fwd_f32: unsupported kernel %dx%d", d->kh, d->kw);
    CP_REQUIRE(d->stride >= 1 && d->dilation >= 1 && d->pad >= 0, "cp_conv2d_fwd_f32: bad stride/dilation/pad");
    CP_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cout > 0, "cp_conv2d_fwd_f32: empty tensor");
    const int eh = (d->kh - 1) * d->dilation + 1, ew = (d->kw - 1) * d->dilation + 1;
    CP_REQUIRE(d->out_h == (d->in_h + 2 * d->pad - eh) / d->stride + 1 && d->out_w == (d->in_w + 2 * d->pad - ew) / d->stride + 1,
               "cp_conv2d_fwd_f32: out size %dx%d inconsistent with input %dx%d k%d s%d d%d p%d", d->out_h, d->out_w,
               d->in_h, d->in_w, d->kh, d->stride, d->dilation, d->pad);
    CP_REQUIRE(d->weights, "cp_conv2d_fwd_f32: null weights");
    CP_REQUIRE(d->out_raw || d->out_act, "cp_conv2d_fwd_f32: no output requested");
    CP_REQUIRE(!d->tap_label || d->stride == 1, "cp_conv2d_fwd_f32: tap_label needs stride 1");
    CP_REQUIRE((d->scale == nullptr) == (d->shift == nullptr), "cp_conv2d_fwd_f32: scale and shift come together");
    CP_REQUIRE(!d->epi_label || d->scale, "cp_conv2d_fwd_f32: epi_label needs a scale/shift table");
    CP_REQUIRE((long long)d->batch * d->out_h * d->out_w < (1LL << 31), "cp_conv2d_fwd_f32: too many output pixels");

    ConvK k{};
    int chans[2] = {0, 0};
    for (int s = 0; s < d->num_sources; ++s) {
        const cp_conv_source& in = d->src[s];
        CP_REQUIRE(in.data, "cp_conv2d_fwd_f32: source %d has null data", s);
        CP_REQUIRE(in.channels == 4 || (in.channels > 0 && in.channels % 32 == 0),
                   "cp_conv2d_fwd_f32: source %d channels must be 4 or a multiple of 32 (got %d)", s, in.channels);
        CP_REQUIRE(in.ld >= in.channels && in.ld % 4 == 0, "cp_conv2d_fwd_f32: source %d ld must be >= channels and a multiple of 4", s);
        CP_REQUIRE(((uintptr_t)in.data & 15) == 0, "cp_conv2d_fwd_f32: source %d not 16-byte aligned", s);
        CP_REQUIRE(in.mode >= 0 && in.mode <= 2, "cp_conv2d_fwd_f32: source %d bad mode", s);
        CP_REQUIRE(in.mode != CP_SRC_NEAREST_SEL || in.sel, "cp_conv2d_fwd_f32: source %d needs a sel map", s);
        CP_REQUIRE(in.mode == CP_SRC_DIRECT || (d->in_h % 2 == 0 && d->in_w % 2 == 0),
                   "cp_conv2d_fwd_f32: x2 source modes need even in_h/in_w");
        CP_REQUIRE((in.pre_scale == nullptr) == (in.pre_shift == nullptr), "cp_conv2d_fwd_f32: pre_scale/pre_shift come together");
        SrcK& o = k.s[s];
        o.data = in.data;
        o.sel = in.sel;
        o.pre_scale = in.pre_scale;
        o.pre_shift = in.pre_shift;
        o.C = in.channels;
        o.ld = in.ld;
        o.mode = in.mode;
        o.Hs = (in.mode == CP_SRC_DIRECT) ? d->in_h : d->in_h / 2;
        o.Ws = (in.mode == CP_SRC_DIRECT) ? d->in_w : d->in_w / 2;
        o.c4 = in.channels == 4;
        o.cpt = o.c4 ? 1 : in.channels / 32;
        o.nchunks = chunks_for(d->kh * d->kw, in.channels);
        chans[s] = in.channels;
    }
    k.W = d->weights;
    k.ktot = cp_conv_ktot(d->kh, d->kw, d->num_sources, chans);
    k.nchunks = k.ktot / BK;
    k.B = d->batch; k.Hin = d->in_h; k.Win = d->in_w; k.Ho = d->out_h; k.Wo = d->out_w; k.Cout = d->cout;
    k.KH = d->kh; k.KW = d->kw; k.stride = d->stride; k.dil = d->dilation; k.pad = d->pad;
    k.M = d->batch * d->out_h * d->out_w;
    k.tap_label = d->tap_label; k.row_scale = d->row_scale;
    k.residual = d->residual; k.res_ld = d->residual_ld;
    k.scale = d->scale; k.shift = d->shift; k.epi_label = d->epi_label; k.act = d->act;
    k.out_raw = d->out_raw; k.raw_ld = d->out_raw_ld; k.out_act = d->out_act; k.act_ld = d->out_act_ld;
    CP_REQUIRE(!d->out_raw || d->out_raw_ld >= d->cout, "cp_conv2d_fwd_f32: out_raw_ld < cout");
    CP_REQUIRE(!d->out_act || d->out_act_ld >= d->cout, "cp_conv2d_fwd_f32: out_act_ld < cout");
    CP_REQUIRE(!d->residual || d->residual_ld >= d->cout, "cp_conv2d_fwd_f32: residual_ld < cout");

    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int tile = d->tile_hint ? d->tile_hint : pick_tile(k.M, k.Cout);
    switch (tile) {
        case CP_TILE_128x128: return launch<2, 2, 2, 2>(k, st);
        case CP_TILE_64x128: return launch<2, 2, 1, 2>(k, st);
        case CP_TILE_128x64: return launch<2, 2, 2, 1>(k, st);
        case CP_TILE_64x64: return launch<2, 2, 1, 1>(k, st);
        case CP_TILE_128x32: return launch<4, 1, 1, 1>(k, st);
        case CP_TILE_256x32: return launch<4, 1, 2, 1>(k, st);
        default: cp::set_error("cp_conv2d_fwd_f32: unknown tile_hint %d", tile); return CP_ERR_INVALID;
    }
}
