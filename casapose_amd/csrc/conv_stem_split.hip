// The ResNet stem (`conv0`, resnet.py:248-249: 7x7 / stride 2 / pad 3, 4-channel image -> 64 channels, input batch-norm as a per-channel affine on
// the real pixels, bn0 + ReLU in the epilogue) on the bf16 matrix pipe -- the arithmetic modes of conv_hsplit.hip:
//   NP = 3  fp32-EQUIVALENT: every fp32 operand split exactly into three bf16 terms, six products accumulated in fp32;
//   NP = 2  fp32-LEVEL: the fp16 two-way split of split_f16.h, three products on v_mfma_f32_32x32x16_f16 (weights pre-multiplied by a power of two,
//           accumulators multiplied by StemSK::descale);
//   NP = 1  bf16 operands (round to nearest even), fp32 accumulation.
// conv_stem.hip runs the layer on v_mfma_f32_32x32x2_f32: 200 MFMAs of 64 cycles per 32-pixel row and 64 output channels.  Here K = 49 taps x 4
// channels is walked in 13 steps of FOUR taps (k = 8 * half + 4 * (tap & 1) + channel: lane half 0 takes taps 4s, 4s+1, half 1 taps 4s+2, 4s+3;
// taps 49-51 do not exist: zero weights), i.e. 13 x 2 x 6 = 156 MFMAs of 32 cycles -- 2.56x less matrix-pipe time.
//   * a persistent block owns 4 x 32 output pixels; its four producer waves stage the 13 x 69 input halo (affine applied, then split) as NP planes
//     of [row][column parity][35][4 bf16]: the stride-2 reads of a tap (input column 2 x + kx) are then CONTIGUOUS 8-byte reads over the lanes;
//   * the whole weight set (13 steps x 2 cout blocks x NP planes x 1 KB = 78 KB with three planes) is fragment-major and RESIDENT in LDS for the
//     life of the block -- six 16-byte fragments per step and wave from L1 would need the full L1 rate of a CU;
//   * accumulators transposed (MFMA A = weights, B = pixels): lane = pixel, 16-byte epilogue accesses (conv_stem.hip's epilogue).
#include "common.h"
#include <cmath>
#include "split_f16.h"

#include <algorithm>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 4, TW = 32;                   // output tile
constexpr int HR = 2 * TH + 5, HC = 2 * TW + 5;  // input halo: 13 x 69 pixels
constexpr int HCP = (HC + 1) / 2;                // 35 pixels per column parity
constexpr int ROW_B = 2 * HCP * 8;               // bytes of one halo row of one plane
constexpr int PLANE_B = HR * ROW_B;              // 7280 B
constexpr int HP = HR * HC;
constexpr int NSTEP = 13;                        // ceil(49 / 4) four-tap steps
constexpr int NFILL = (HP + 255) / 256;

struct StemSK {
    const float* img;        // [B,H,W,4]
    unsigned img_bytes;
    const float* pre_scale;  // [4] or null
    const float* pre_shift;
    const unsigned char* W;  // [13 steps][2 cout blocks][NP planes][64 lanes][8 bf16]
    const float* scale;      // [64] or null
    const float* shift;
    int act;
    float* out_raw; int raw_ld;
    float* out_act; int act_ld;
    int B, H, Wd, Ho, Wo, tiles_y, tiles_x, ntiles;
    float descale;           // NP = 2: 1 / (the power of two the weights were multiplied by); 1 otherwise
    uint32_t* mon;           // f16x2 range monitor slot (common.h) or null
};

#define CP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ unsigned ss_pack_hi16(unsigned a_lo, unsigned b_hi) { return __builtin_amdgcn_perm(b_hi, a_lo, 0x07060302u); }

// exact three-way split of four floats into packed bf16 pairs (wino_gemm_split.hip)
__device__ __forceinline__ void ss_split4(const float4 v, uint2& hi, uint2& mid, uint2& lo) {
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
    hi = make_uint2(ss_pack_hi16(h[0], h[1]), ss_pack_hi16(h[2], h[3]));
    mid = make_uint2(ss_pack_hi16(m[0], m[1]), ss_pack_hi16(m[2], m[3]));
    lo = make_uint2(ss_pack_hi16(l[0], l[1]), ss_pack_hi16(l[2], l[3]));
}

__device__ __forceinline__ uint2 ss_round4(const float4 v) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned u = __builtin_bit_cast(unsigned, x[e]);
        r[e] = u + 0x7fffu + ((u >> 16) & 1u);
    }
    return make_uint2(ss_pack_hi16(r[0], r[1]), ss_pack_hi16(r[2], r[3]));
}

template <int NP>
__global__ __launch_bounds__(512, 2) void conv_stem_split_kernel(const StemSK p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 stages][NP][PLANE_B] halo planes | [13][2][NP][1 KB] weights
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NPROD = (NP == 3) ? 6 : (NP == 2) ? 3 : 1;
    constexpr unsigned W_BYTES = NSTEP * 2 * NP * 1024u;
    unsigned char* halo = smem;
    unsigned char* wl = smem + 2 * NP * PLANE_B;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;
    const int bid = cp::xcd_remap(blockIdx.x, gridDim.x);
    const int my_tiles = (p.ntiles - bid + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;

    auto tile_pos = [&](int k, int& n, int& y0, int& x0) {
        int t = bid + k * (int)gridDim.x;
        x0 = (t % p.tiles_x) * TW;
        t /= p.tiles_x;
        y0 = (t % p.tiles_y) * TH;
        n = t / p.tiles_y;
    };

    if (producer) {
        if constexpr (NP == 2) cp::f16_overflow_clamps();
        const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)p.img, 0, p.img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, W_BYTES, 0x00020000);
        for (unsigned o = (unsigned)tid * 16u; o < W_BYTES; o += 256u * 16u)   // the resident weight fragments
            *reinterpret_cast<u32x4*>(wl + o) = __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)o, 0, 0);
        float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.pre_scale) {
            ps = *reinterpret_cast<const float4*>(p.pre_scale);
            pb = *reinterpret_cast<const float4*>(p.pre_shift);
        }
        int e_hy[NFILL], e_hx[NFILL];
        unsigned e_lds[NFILL];
#pragma unroll
        for (int i = 0; i < NFILL; ++i) {
            const int e = tid + 256 * i;
            e_hy[i] = e < HP ? e / HC : 0x4000;  // 0x4000: never in bounds
            e_hx[i] = e % HC;
            e_lds[i] = (unsigned)((e / HC) * ROW_B + ((e_hx[i] & 1) * HCP + (e_hx[i] >> 1)) * 8);
        }
        float l_amax = 0.f;
        auto fill = [&](int k, int stage) {
            int n, y0, x0;
            tile_pos(k, n, y0, x0);
            float4 v[NFILL];
            bool inb[NFILL];
#pragma unroll
            for (int i = 0; i < NFILL; ++i) {
                const int y = 2 * y0 - 3 + e_hy[i], x = 2 * x0 - 3 + e_hx[i];
                inb[i] = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd;
                v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsi, inb[i] ? (int)((((unsigned)n * p.H + y) * p.Wd + x) * 16u) : (int)OOB, 0, 0));
            }
            unsigned char* h = halo + stage * (NP * PLANE_B);
#pragma unroll
            for (int i = 0; i < NFILL; ++i) {
                if (e_hy[i] >= 0x4000) continue;
                float4 r = v[i];
                if (inb[i]) {  // the affine applies to real pixels only: padding stays exactly zero
                    r.x = r.x * ps.x + pb.x; r.y = r.y * ps.y + pb.y; r.z = r.z * ps.z + pb.z; r.w = r.w * ps.w + pb.w;
                }
                if (NP == 2 && p.mon) l_amax = cp::amax4(l_amax, r);   // (uniform) the image through the input affine, as converted
                if constexpr (NP == 3) {
                    uint2 a, b, c;
                    ss_split4(r, a, b, c);
                    *reinterpret_cast<uint2*>(h + e_lds[i]) = a;
                    *reinterpret_cast<uint2*>(h + PLANE_B + e_lds[i]) = b;
                    *reinterpret_cast<uint2*>(h + 2 * PLANE_B + e_lds[i]) = c;
                } else if constexpr (NP == 2) {
                    uint2 a, b;
                    cp::split4h(r, a, b);
                    *reinterpret_cast<uint2*>(h + e_lds[i]) = a;
                    *reinterpret_cast<uint2*>(h + PLANE_B + e_lds[i]) = b;
                } else {
                    *reinterpret_cast<uint2*>(h + e_lds[i]) = ss_round4(r);
                }
            }
        };
        fill(0, 0);
        CP_BARRIER();
        for (int k = 0; k < my_tiles; ++k) {
            if (k + 1 < my_tiles) fill(k + 1, (k + 1) & 1);   // stage (k+1)&1 was last read for tile k-1
            CP_BARRIER();
        }
        if constexpr (NP == 2) {
            if (p.mon) {
                cp::monitor_flush(p.mon, l_amax);
                cp::monitor_count_launch(p.mon, tid == 0);
            }
        }
        return;
    }

    // ------------------------------------ consumers: wave w owns output row w of the tile ------------------------------------
    const int wy = wave;
    const int lrow = lane & 31, half = lane >> 5;
    const unsigned npix = (unsigned)(p.B * p.Ho * p.Wo);
    const __amdgpu_buffer_rsrc_t r_tab_s = __builtin_amdgcn_make_buffer_rsrc((void*)(p.scale ? (const void*)p.scale : (const void*)p.W), 0, p.scale ? 256u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_tab_b = __builtin_amdgcn_make_buffer_rsrc((void*)(p.scale ? (const void*)p.shift : (const void*)p.W), 0, p.scale ? 256u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_raw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out_raw ? (void*)p.out_raw : (void*)p.W), 0,
                                                                            p.out_raw ? npix * (unsigned)p.raw_ld * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_act = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out_act ? (void*)p.out_act : (void*)p.W), 0,
                                                                            p.out_act ? npix * (unsigned)p.act_ld * 4u : 0u, 0x00020000);
    // LDS byte offset (inside a plane) of this lane's two taps of step s: taps 4s + 2 half and 4s + 2 half + 1 (clamped to tap 48: zero weights beyond)
    unsigned toff[NSTEP][2];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int t = min(4 * s + 2 * half + q, 48);
            const int ky = t / 7, kx = t % 7;
            toff[s][q] = (unsigned)((2 * wy + ky) * ROW_B + ((kx & 1) * HCP + (kx >> 1) + lrow) * 8);
        }
    const unsigned wlane = (unsigned)lane * 16u;
    f32x16 acc[2];
    bf16x8 fa[2][NP], fw[2][2][NP];   // [slot][plane] pixel fragments, [slot][cout block][plane] weight fragments
    CP_BARRIER();  // halo of tile 0 and the weights are in LDS
    for (int k = 0; k < my_tiles; ++k) {
        int n, y0, x0;
        tile_pos(k, n, y0, x0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const unsigned char* hb = halo + (k & 1) * (NP * PLANE_B);
        auto read_step = [&](int s, int slot) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const uint2 a = *reinterpret_cast<const uint2*>(hb + pl * PLANE_B + toff[s][0]);
                const uint2 b = *reinterpret_cast<const uint2*>(hb + pl * PLANE_B + toff[s][1]);
                fa[slot][pl] = __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
#pragma unroll
                for (int j = 0; j < 2; ++j) fw[slot][j][pl] = *reinterpret_cast<const bf16x8*>(wl + (unsigned)((s * 2 + j) * NP + pl) * 1024u + wlane);
            }
        };
        read_step(0, 0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s + 1 < NSTEP) read_step(s + 1, (s + 1) & 1);
#pragma unroll
            for (int t = 0; t < NPROD; ++t) {
                // (weight plane, pixel plane): lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi -- smallest terms first
                // NP = 2: lo*hi, hi*lo, hi*hi
                const int sw = (NP == 1) ? 0 : (NP == 2) ? (t == 0 ? 1 : 0) : ((t == 0) ? 2 : (t == 1) ? 0 : (t == 2) ? 1 : (t == 3) ? 1 : 0);
                const int sp = (NP == 1) ? 0 : (NP == 2) ? (t == 1 ? 1 : 0) : ((t == 0) ? 0 : (t == 1) ? 2 : (t == 2) ? 1 : (t == 3) ? 0 : (t == 4) ? 1 : 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (NP == 2)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cp::f16x8_t, fw[s & 1][j][sw]), __builtin_bit_cast(cp::f16x8_t, fa[s & 1][sp]), acc[j], 0, 0, 0);
                    else
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[s & 1][j][sw], fa[s & 1][sp], acc[j], 0, 0, 0);
                }
            }
        }
        // ---- epilogue: lane = pixel; register r of block j = channel j*32 + (r&3) + 8*(r>>2) + 4*half (conv_stem.hip) ----
        const int y = y0 + wy, x = x0 + lrow;
        const bool pok = y < p.Ho && x < p.Wo;
        const unsigned pix = (unsigned)((n * p.Ho + y) * p.Wo + x);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float4 sc[4], sh[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int ch = j * 32 + g4 * 8 + half * 4;
                sc[g4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_tab_s, ch * 4, 0, 0));
                sh[g4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_tab_b, ch * 4, 0, 0));
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int ch = j * 32 + g4 * 8 + half * 4;
                float4 v = make_float4(acc[j][g4 * 4 + 0], acc[j][g4 * 4 + 1], acc[j][g4 * 4 + 2], acc[j][g4 * 4 + 3]);
                if constexpr (NP == 2) { v.x *= p.descale; v.y *= p.descale; v.z *= p.descale; v.w *= p.descale; }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_raw, (int)(pok ? (pix * (unsigned)p.raw_ld + (unsigned)ch) * 4u : OOB), 0, 0);
                float4 t = v;
                if (p.scale) {
                    t.x = v.x * sc[g4].x + sh[g4].x; t.y = v.y * sc[g4].y + sh[g4].y; t.z = v.z * sc[g4].z + sh[g4].z; t.w = v.w * sc[g4].w + sh[g4].w;
                }
                if (p.act == CP_ACT_RELU) {
                    t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
                } else if (p.act == CP_ACT_LEAKY01) {
                    t.x = fmaxf(t.x, 0.f) - fmaxf(-0.1f * t.x, 0.f); t.y = fmaxf(t.y, 0.f) - fmaxf(-0.1f * t.y, 0.f);
                    t.z = fmaxf(t.z, 0.f) - fmaxf(-0.1f * t.z, 0.f); t.w = fmaxf(t.w, 0.f) - fmaxf(-0.1f * t.w, 0.f);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), r_act, (int)(pok ? (pix * (unsigned)p.act_ld + (unsigned)ch) * 4u : OOB), 0, 0);
            }
        }
        CP_BARRIER();  // tile done: its halo stage may be refilled
    }
}

template <int NP>
int launch_stem_split(StemSK k, hipStream_t st) {
    const size_t lds = (size_t)2 * NP * PLANE_B + (size_t)NSTEP * 2 * NP * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem_split_kernel<NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int grid = std::min(cp::persistent_blocks(), k.ntiles);
    CP_LAUNCH(conv_stem_split_kernel<NP>, dim3(grid), dim3(512), lds, st, k);
    return cp::check_launch("cp_conv2d_fwd_stem_split");
}

}  // namespace

namespace cp {
bool stem_applicable(const cp_conv_desc* d);   // conv_stem.hip: the same range (7x7 / stride 2 / pad 3, one 4-channel source, cout 64)
}

extern "C" int cp_conv_stem_split_weight_floats(void) { return NSTEP * 2 * 64 * 8; }

// HOST: 7x7 kernel (layout 0 = HWIO, 1 = IHWO) -> the fp32 image of the fragment stream, [13 steps][2 cout blocks][64 lanes][8]:
//   lane (i = l & 31, half = l >> 5), element e  <-  W[co = 32 j + i][channel e & 3] at tap 4 s + 2 half + (e >> 2)   (taps >= 49, channel 3: zero)
// cp_conv_split_weights_f32 turns it into the bf16 planes on the device.
extern "C" int cp_conv_pack_weights_stem_split_host(const float* w, int layout, int real_channels, float* dst) {
    CP_REQUIRE(w && dst && real_channels >= 1 && real_channels <= 4 && (layout == 0 || layout == 1), "cp_conv_pack_weights_stem_split_host: bad arguments");
    for (int s = 0; s < NSTEP; ++s)
        for (int j = 0; j < 2; ++j)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int co = 32 * j + (l & 31), t = 4 * s + 2 * (l >> 5) + (e >> 2), c = e & 3;
                    float v = 0.f;
                    if (t < 49 && c < real_channels) {
                        const int ky = t / 7, kx = t % 7;
                        v = (layout == 0) ? w[(((size_t)ky * 7 + kx) * real_channels + c) * 64 + co] : w[(((size_t)c * 7 + ky) * 7 + kx) * 64 + co];
                    }
                    dst[(((size_t)s * 2 + j) * 64 + l) * 8 + e] = v;
                }
    return CP_OK;
}

extern "C" int cp_conv2d_fwd_stem_split(const cp_conv_desc* d, const void* weights_split, int planes, void* stream) {
    CP_REQUIRE(planes == 1 || planes == 3, "cp_conv2d_fwd_stem_split: planes must be 1 or 3 (CP_PLANES_F16X2 needs cp_conv2d_fwd_stem_split_scaled)");
    return cp_conv2d_fwd_stem_split_scaled(d, weights_split, planes, 1.f, stream);
}

extern "C" int cp_conv2d_fwd_stem_split_scaled(const cp_conv_desc* d, const void* weights_split, int planes, float w_descale, void* stream) {
    CP_REQUIRE_DESC(d, "cp_conv2d_fwd_stem_split");
    CP_REQUIRE(weights_split && (planes == 1 || planes == 3 || planes == CP_PLANES_F16X2), "cp_conv2d_fwd_stem_split: bad arguments");
    CP_REQUIRE(planes == CP_PLANES_F16X2 ? (w_descale > 0.f && std::isfinite(w_descale)) : w_descale == 1.f,
               "cp_conv2d_fwd_stem_split_scaled: a descale factor other than 1 goes with CP_PLANES_F16X2 only");
    cp_conv_desc probe = *d;
    probe.weights_halo = reinterpret_cast<const float*>(weights_split);   // stem_applicable only asks that a second packing exists
    CP_REQUIRE(cp::stem_applicable(&probe), "cp_conv2d_fwd_stem_split: not the stem convolution (7x7 / stride 2 / pad 3, one 4-channel source, cout 64, no labels / residual / head)");
    CP_REQUIRE(d->out_raw || d->out_act, "cp_conv2d_fwd_stem_split: no output");
    StemSK k{};
    k.img = d->src[0].data;
    const long long nbytes = (long long)d->batch * d->in_h * d->in_w * 16;
    CP_REQUIRE(nbytes < (1LL << 31), "cp_conv2d_fwd_stem_split: the image batch spans >= 2 GiB");
    const int max_ld = std::max(d->out_raw ? d->out_raw_ld : 0, d->out_act ? d->out_act_ld : 0);
    CP_REQUIRE((long long)d->batch * d->out_h * d->out_w * max_ld * 4 < (1LL << 31), "cp_conv2d_fwd_stem_split: an output spans >= 2 GiB");
    k.img_bytes = (unsigned)nbytes;
    k.pre_scale = d->src[0].pre_scale; k.pre_shift = d->src[0].pre_shift;
    k.W = reinterpret_cast<const unsigned char*>(weights_split);
    k.scale = d->scale; k.shift = d->shift; k.act = d->act;
    k.out_raw = d->out_raw; k.raw_ld = d->out_raw_ld; k.out_act = d->out_act; k.act_ld = d->out_act_ld;
    k.B = d->batch; k.H = d->in_h; k.Wd = d->in_w; k.Ho = d->out_h; k.Wo = d->out_w;
    k.tiles_y = (k.Ho + TH - 1) / TH; k.tiles_x = (k.Wo + TW - 1) / TW;
    k.ntiles = k.B * k.tiles_y * k.tiles_x;
    k.descale = w_descale;
    k.mon = planes == CP_PLANES_F16X2 ? cp::f16x2_monitor() : nullptr;
    if (planes == CP_PLANES_F16X2) return launch_stem_split<2>(k, (hipStream_t)stream);
    return planes == 3 ? launch_stem_split<3>(k, (hipStream_t)stream) : launch_stem_split<1>(k, (hipStream_t)stream);
}
