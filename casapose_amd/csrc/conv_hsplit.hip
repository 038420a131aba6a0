// 3x3 / stride 1 / pad 1 convolution for the shallow, high-resolution layers (cout <= 64) on the bf16 matrix pipe:
//   NP = 3  fp32-EQUIVALENT: every fp32 operand is split exactly into three bf16 terms (hi, mid, lo: 8 + 8 + 8 significand bits) and the
//           six products lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (the scheme of
//           wino_gemm_split.hip; the three dropped products are <= 2^-24 of the full product -- the rounding an fp32 multiply makes anyway);
//   NP = 2  fp32-LEVEL with half the MFMAs: the fp16 two-way split of split_f16.h (hi = rn_f16(x), lo = rn_f16(x - hi): the operand to one fp32 ulp;
//           products lo*hi, hi*lo, hi*hi on v_mfma_f32_32x32x16_f16).  The weights arrive multiplied by a power of two (their low parts stay
//           normal numbers); the epilogue multiplies the accumulators by its inverse (HSplitK::descale, exact).
//   NP = 1  plain bf16 operands (round to nearest even), fp32 accumulation -- "bf16 convolutions" of BASELINE.json configs[2].
// Activations and outputs stay fp32 in HBM; the split / conversion happens once per staged halo pixel.
//
// Why not conv_halo.hip with another instruction: six bf16 MFMAs of K = 16 take 6 x 32 cycles where the fp32 MFMA needs 8 x 64, so at the
// old tile shape the weight fragments (1.5x the bytes per k, 2.7x less time) would need ~64 B/clk/CU from L1.  Here
//   * a block owns 8 rows x 32 columns of output; each of its 4 consumer waves (one per SIMD) computes TWO rows, so every weight fragment
//     fetched from L2 feeds two pixel fragments (31 B/clk/CU), and every pixel fragment read from LDS feeds TN cout blocks;
//   * 4 loader waves fetch the next 16-channel slice's halo (10 x 34 pixels), split it and store it into the other LDS stage -- one barrier
//     per slice.  (A first version let all 8 waves load AND multiply: the vmcnt counter retires loads in order, so every wait for a weight
//     fragment also waited for the halo loads issued before it and the HBM latency was exposed once per slice: 2.5x the MFMA time.)
//   * halo planes are [pixel][16 bf16] = 32-byte rows with the two 16-byte slots swapped for pixels with bit 3 set: the 16 lanes of a
//     ds_read_b128 group hit 16 different slots at any tap offset (no padding: 3 planes x 2 stages = 65 KB, + 16 KB image halo);
//   * the 4-channel image source is three K = 16 steps with k = tap * 4 + channel.
// Accumulators are transposed as in conv_halo.hip (MFMA A = weights, B = pixels): lane = pixel, so the partial-conv tap mask, 9/count,
// the CLADE table row, residual and stores are per lane with four consecutive channels in four consecutive registers.
#include "common.h"
#include "split_f16.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int COLS = 34;             // 32 output columns + 2 halo columns
constexpr int TH = 8;                // output rows per tile (2 per consumer wave)
constexpr int HR = TH + 2;
constexpr int HP = HR * COLS;        // 340 halo pixels
constexpr int PLANE_B = HP * 32;     // bytes of one halo plane (16 bf16 per pixel)
constexpr int IPLANE_B = HP * 8;     // bytes of one image-halo plane (4 bf16 per pixel)
constexpr int NIT = (HP * 4 + 255) / 256;   // float4 halo elements per loader thread and slice
constexpr int NIMG = (HP + 255) / 256;      // image-halo pixels per loader thread

// HS_HEADK (round 5): the layer is a "head layer" -- 32 output channels, a normalisation table, leaky ReLU, a fused 1x1 head and NOTHING else (no
// residual, no raw / activated output): blocks 5 and 10 of the decoders, 17 % of the forward.  Their epilogue is compiled without the operands it
// does not have (three buffer descriptors and a dozen uniform flags fewer: the generic form reloads 350 spilled scalars per tile) and with the
// per-channel table held in registers for the whole kernel.
enum : int { HS_BILINEAR = 2, HS_PARTIAL = 4, HS_SEL = 8, HS_HEADK = 16, HS_PREFIX = 32 };   // HS_PREFIX: a head layer that writes whole output records (head_pre_n)

struct SSrc {
    const float* data;
    const uint8_t* sel;
    int C, ld, Hs, Ws;
    unsigned bytes;
};

struct HSplitK {
    SSrc s[2];
    const float* img;
    unsigned img_bytes;
    const unsigned char* W;   // [pass][step][cout block][plane][64 lanes][8 bf16]; steps = 9 per 16-channel slice, then 3 image steps
    unsigned w_bytes;
    int B, H, Wd, Cout;
    int nch0, nch;            // 16-channel slices of source 0 / of both sources
    int tiles_y, tiles_x, ntiles;   // ntiles = passes * B * tiles_y * tiles_x: the tile list is walked once per pass of 32*TN output channels
    int passes, tiles_per_pass;
    const uint8_t* label;
    unsigned lab_bytes;
    const float* residual;
    int res_ld;
    const float* scale;
    const float* shift;
    int clade, norm, act;
    float* out_raw;
    int raw_ld;
    float* out_act;
    int act_ld;
    const unsigned char* head_w;   // fused 1x1 head (cout == 32): [2 steps][plane][64 lanes][8 bf16], cp_conv_pack_head_split_host
    float* head_out;
    int head_cout, head_ld;
    uint8_t* head_lab;   // optional arg-max of the first head_lab_classes head channels
    int head_lab_classes;
    float descale, head_descale;   // NP = 2: 1 / (power of two the conv / head weights were multiplied by); 1 otherwise
    int head_pre_n;                // HS_HEADK only: > 0 = head_out addresses whole output RECORDS; floats [0, head_pre_n) of a pixel's record are copied
                                   // from the dense rows `residual` (row length res_ld: a head layer has no residual), the head's columns follow them
    int w_res;                     // 1: head layers keep their whole weight stream LDS-resident where it fits (A/B switch CASAPOSE_HS_WRES)
    int epi_split;                 // 1: the loader wave w + 4 runs the epilogue of row 1 of consumer wave w's rows (accumulators handed over through LDS)
    uint32_t* mon;                 // f16x2 range monitor slot (common.h) or null: max |x| of what the loaders convert -> [0], of the fused head's operand -> [2]
};

#ifdef HS_TRACE
// -DHS_TRACE (a variant build): block 0's consumer wave 0 and loader wave 4 stamp the shader clock in front of and behind every barrier into hs_trace[]
// ([role][2048] entries: even = arrival, odd = release; entry 0 of each role = s_memrealtime at kernel start for calibration), read back through
// cp_hs_trace_read (exported by that variant only): the time line of the two roles, phase by phase.
__device__ unsigned long long hs_trace[2][2048];
#define HST(tag) do { asm volatile("; HST " #tag ::: "memory"); if (hst_on && hst_i < 2046) hs_trace[hst_role][hst_i++] = ((unsigned long long)(tag) << 56) | (__builtin_readcyclecounter() & 0xffffffffffffffull); } while (0)
#define CP_BARRIER() do { HST(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); HST(1); } while (0)
#else
#define HST(tag)
#define CP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
#ifndef HS_EPI_AUX
#define HS_EPI_AUX 0   // cache policy bits of the generic epilogue's stores (variant builds: 2 = nt, 1 = sc0, 16 = sc1)
#endif

// -DHS_PROFILE (a variant build, tools/build_variant.sh): shader-clock time per section of the consumer / loader waves, summed over all waves into
// hs_prof[] and read back through cp_hs_profile_read (exported by that variant only).  Sections: consumers 0 tile setup, 1 slice MFMA loops,
// 2 image block, 3 epilogue, 4 waiting at barriers; loaders 8 phase work before the barrier, 9 waiting at barriers.
#ifdef HS_PROFILE
__device__ unsigned long long hs_prof[16];
#define HSP_DECL unsigned long long hsp_t = __builtin_readcyclecounter(); unsigned hsp_acc[6] = {0, 0, 0, 0, 0, 0}
#define HSP(i) do { const unsigned long long hsp_now = __builtin_readcyclecounter(); hsp_acc[i] += (unsigned)(hsp_now - hsp_t); hsp_t = hsp_now; } while (0)
#define HSP_FLUSH(base) do { if (lane == 0) { for (int hsp_i = 0; hsp_i < 6; ++hsp_i) atomicAdd(&hs_prof[(base) + hsp_i], (unsigned long long)hsp_acc[hsp_i]); } } while (0)
#else
#define HSP_DECL
#define HSP(i)
#define HSP_FLUSH(base)
#endif

__device__ __forceinline__ unsigned pack_hi16(unsigned a_lo, unsigned b_hi) { return __builtin_amdgcn_perm(b_hi, a_lo, 0x07060302u); }

// exact three-way split of four floats into packed bf16 pairs (see wino_gemm_split.hip)
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

// round-to-nearest-even bf16 of four floats (finite inputs; NaN payloads are not preserved bit for bit, which no caller needs)
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

// planes of one float4 at `dst`, `dst + stride`, ...: the exact bf16 split (3), the fp16 two-way split (2) or rounded bf16 (1)
template <int NP>
__device__ __forceinline__ void store_planes(unsigned char* dst, unsigned stride, const float4 val) {
    if constexpr (NP == 3) {
        uint2 a, b, c;
        split4(val, a, b, c);
        *reinterpret_cast<uint2*>(dst) = a;
        *reinterpret_cast<uint2*>(dst + stride) = b;
        *reinterpret_cast<uint2*>(dst + 2 * stride) = c;
    } else if constexpr (NP == 2) {
        uint2 a, b;
        cp::split4h(val, a, b);
        *reinterpret_cast<uint2*>(dst) = a;
        *reinterpret_cast<uint2*>(dst + stride) = b;
    } else {
        *reinterpret_cast<uint2*>(dst) = round4(val);
    }
}

// (weight plane, pixel plane) of product t, smallest terms first.  NP = 3: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi; NP = 2: lo*hi, hi*lo, hi*hi
template <int NP> __device__ __forceinline__ constexpr int prod_w(int t) {
    return NP == 1 ? 0 : NP == 2 ? (t == 0 ? 1 : 0) : ((t == 0) ? 2 : (t == 1) ? 0 : (t == 2) ? 1 : (t == 3) ? 1 : 0);
}
template <int NP> __device__ __forceinline__ constexpr int prod_p(int t) {
    return NP == 1 ? 0 : NP == 2 ? (t == 1 ? 1 : 0) : ((t == 0) ? 0 : (t == 1) ? 2 : (t == 2) ? 1 : (t == 3) ? 0 : (t == 4) ? 1 : 0);
}
template <int NP>
__device__ __forceinline__ f32x16 mfma_np(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    if constexpr (NP == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cp::f16x8_t, a), __builtin_bit_cast(cp::f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int NP>
__global__ void hsplit_weights_kernel(const float* __restrict__ src, long long nfrag, float scale, unsigned char* __restrict__ dst) {
    // src: [fragment][64 lanes][8 floats] -> dst: [fragment][plane][64 lanes][8 bf16]
    const long long total = nfrag * 64;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long f = i >> 6;
        const int lane = (int)(i & 63);
        const float4 v0 = *reinterpret_cast<const float4*>(src + i * 8), v1 = *reinterpret_cast<const float4*>(src + i * 8 + 4);
        unsigned char* d = dst + (f * NP) * 1024 + lane * 16;
        if constexpr (NP == 3) {
            uint2 h0, m0, l0, h1, m1, l1;
            split4(v0, h0, m0, l0);
            split4(v1, h1, m1, l1);
            *reinterpret_cast<uint4*>(d) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            *reinterpret_cast<uint4*>(d + 1024) = make_uint4(m0.x, m0.y, m1.x, m1.y);
            *reinterpret_cast<uint4*>(d + 2048) = make_uint4(l0.x, l0.y, l1.x, l1.y);
        } else if constexpr (NP == 2) {
            uint2 h0, l0, h1, l1;
            cp::split4h(make_float4(v0.x * scale, v0.y * scale, v0.z * scale, v0.w * scale), h0, l0);
            cp::split4h(make_float4(v1.x * scale, v1.y * scale, v1.z * scale, v1.w * scale), h1, l1);
            *reinterpret_cast<uint4*>(d) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            *reinterpret_cast<uint4*>(d + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
        } else {
            const uint2 a = round4(v0), b = round4(v1);
            *reinterpret_cast<uint4*>(d) = make_uint4(a.x, a.y, b.x, b.y);
        }
    }
}

// relu(t) - relu(-0.1 t) (casa_layer's LeakyReLU, casapose.py:98-105) in two instructions: max(t, 0.1 t).  Same value for every finite t -- t > 0: t
// either way; t < 0: -fl(-0.1f * t) = fl(0.1f * t), rounding to nearest is symmetric -- only the sign of a zero result can differ (-0 for t = -0).
__device__ __forceinline__ float leaky01(float t) { return fmaxf(t, 0.1f * t); }

template <int I, int N, typename F>
__device__ __forceinline__ void hs_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        hs_static_for<I + 1, N>(f);
    }
}

// Taps per weight group (= per barrier phase).  32 output channels: the whole slice.  64 channels: the whole slice too where the LDS holds two
// 9-tap groups (one or two operand planes: 36 / 72 KB of weight stages) -- round 5: with three taps per group (round 2-4) a slice was one loader-heavy
// phase (halo stores + requests) followed by two light ones, and the consumers waited in the first while the loaders waited in the other two; with
// three planes (exact split) three taps stay.  128 channels per pass: one tap.
#ifdef HS_GT3   // (variant build for A/B measurements: round 4's three-tap groups)
__host__ __device__ constexpr int hs_group_taps(int tn, int np) { return tn == 1 ? 9 : (tn == 2 ? 3 : 1); }
#else
__host__ __device__ constexpr int hs_group_taps(int tn, int np) { return tn == 1 ? 9 : (tn == 2 ? (np <= 2 ? 9 : 3) : 1); }
#endif

template <int TN, int NP, int MODE>
__global__ __launch_bounds__(512, 2) void conv_hsplit_kernel(const HSplitK p) {
    constexpr bool PARTIAL = (MODE & HS_PARTIAL) != 0;
    constexpr bool BILINEAR = (MODE & HS_BILINEAR) != 0;   // source 0 is read at half resolution through a x2 half-pixel bilinear filter
    constexpr bool SEL = (MODE & HS_SEL) != 0;             // source 0 is read at half resolution through the guided-upsampling selection map
    constexpr bool HEADK = (MODE & HS_HEADK) != 0;         // head layer: table + leaky ReLU + fused 1x1 head, no other output (TN == 1)
    constexpr bool PREFIX = (MODE & HS_PREFIX) != 0;       // ... that copies head_pre_n floats in front of its columns (whole output records)
    static_assert(!PREFIX || HEADK, "only head layers write output records");
    static_assert(!HEADK || TN == 1, "a fused head needs 32 output channels");
    constexpr int NV = 1;   // (round 3 fetched the four bilinear taps of source 0 from global memory: NV = 4; now a low-resolution tile is staged in LDS)
    constexpr unsigned OOB = 0x80000000u;
    static_assert(NP == 1 || NP == 2 || NP == 3, "1 = bf16, 2 = fp16 two-way split, 3 = exact bf16 split");
    constexpr int NPROD = (NP == 3) ? 6 : (NP == 2) ? 3 : 1;
    constexpr unsigned FRAG_B = NP * 1024u;          // all planes of one (step, cout block) fragment
    constexpr int GT = hs_group_taps(TN, NP);               // taps per weight group: a whole slice for the 32-channel layers (one barrier per slice),
                                                            // a third of it for the 64-channel ones, one tap for 128 channels per pass (LDS budget)
    constexpr int GPS = 9 / GT;                      // groups per slice
    constexpr int GSUB = GT * TN;                    // (tap, cout block) sub-steps of a group; the image block is a group of 3 * TN sub-steps
    constexpr unsigned GROUP_B = GSUB * FRAG_B;      // bytes of a weight group: 27 KB (TN = 1) / 18 KB (TN = 2) with three planes
    constexpr unsigned IGROUP_B = 3 * TN * FRAG_B;   // bytes of the image block's group (layers with the image source have TN = 1)
    static_assert(TN == 1 || TN == 2 || TN == 4, "32, 64 or 128 output channels per pass");
    constexpr int NWL = (int)((GROUP_B / 16 + 255) / 256);   // 16-byte pieces of a group per loader thread

    const bool has_img = p.img != nullptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* halo = smem;                              // [2 stages][NP][HP][32 B]
    unsigned char* imgh = smem + 2 * NP * PLANE_B;           // [2 tile parities][NP][HP][8 B]
    unsigned short* labh = reinterpret_cast<unsigned short*>(smem + 2 * NP * PLANE_B + 2 * NP * IPLANE_B);   // [2 tile parities][HP]: label | 0xff00 outside the image
    unsigned char* wst = smem + 2 * NP * PLANE_B + 2 * NP * IPLANE_B + 4 * HP;                                // [2 stages][GROUP_B] weight groups
    unsigned char* hwl = wst + ((HEADK && NP <= 2) ? 2 * GROUP_B + IGROUP_B : 2 * GROUP_B);   // (three planes leave no room for the resident form)                                                                   // [2 steps][NP][1 KB] fused-head weights
    // BILINEAR: source 0 is stored at half resolution; the (TH/2 + 2) x (32/2 + 2) source pixels a tile's halo interpolates from are staged
    // as fp32 ([2 stages][LOW_P pixels][16 channels]) and the four taps of every halo pixel come from there -- a slice costs 432 16-byte
    // global loads per block instead of 5440 (round 3: four taps per halo pixel from L2, which is what made the fused form slower than
    // a materialised upsampled tensor on the 32-channel layers)
    constexpr int LOW_R = HR / 2 + 1, LOW_C = COLS / 2 + 1, LOW_P = LOW_R * LOW_C;   // 6 x 18 = 108 source pixels
    constexpr int LOW_B = LOW_P * 64;
    unsigned char* lowb = hwl + 2 * NP * 1024;   // [2 stages][LOW_P][64 B]
    // Round 5: the per-channel normalisation table of a layer WITHOUT class-adaptive rows, staged once: [scale | shift][TAB_C floats].  The generic
    // epilogue then has no global loads unless the layer has a residual, so nothing of it waits on vmcnt -- where stores and loads share one
    // in-order counter, every table load of row r + 1 also waited for the write acknowledgements of row r's stores (26 k of the 53 k cycles a
    // 64-channel tile of stage 1 takes, tools/debug/hs_profile.py)
    constexpr int TAB_C = 512;
    float* tabl = reinterpret_cast<float*>(lowb + (BILINEAR ? 2 * LOW_B : 0));
    const bool tab_lds = !HEADK && p.scale != nullptr && !p.clade && p.Cout <= TAB_C;
    if (tab_lds) {
        for (int i = (int)threadIdx.x; i < TAB_C; i += 512) {
            tabl[i] = i < p.Cout ? p.scale[i] : 0.f;
            tabl[TAB_C + i] = i < p.Cout ? p.shift[i] : 0.f;
        }
    }
    const bool head = HEADK || ((TN == 1) && p.head_out != nullptr);
    // Epilogue split (round 5).  The epilogue is serial code on the consumer waves -- a quarter of all conv_hsplit time with the matrix pipe idle --
    // while the loader waves of most layers wait at the barrier.  With epi_split a consumer wave keeps row 0 of its two rows and hands the
    // accumulators of row 1 to its loader twin through `accst` ([wave][TN][4][64 lanes][16 B], the register layout as it is); the twin runs the
    // same epilogue code for that row during the first phase of the NEXT tile, beside the consumers' MFMAs.
    unsigned char* accst = reinterpret_cast<unsigned char*>(tabl + 2 * TAB_C);
    constexpr bool CAN_SPLIT = NP <= 2 && TN == 1;   // (the hand-over buffer fits beside neither three operand planes nor the 64-channel kernels' 9-tap weight groups)
    constexpr unsigned ACCST_B = 4u * TN * 4u * 1024u;   // one hand-over buffer: 16 KB per 32 output channels
    // WHEN the twin works (a phase = one weight group, a barrier at its end): a head layer (three phases a tile: slice, slice, image + epilogue) takes
    // the whole row during the image phase of the NEXT tile -- the one phase in which the loaders have nothing else to do -- out of the buffer of the
    // tile's parity (two buffers: the consumers fill the other one at the end of that very phase); any other layer takes its 4 TN channel groups a
    // few per phase over the phases 0 ... ngroups_tile - 2 of the next tile (one buffer: it is refilled in the last phase)

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave >= 4;
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;
    const int lrow = lane & 31, kh = lane >> 5;

#ifdef HS_TRACE
    const int hst_role = wave >= 4 ? 1 : 0;
    const bool hst_on = blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0;
    int hst_i = 2;
    if (hst_on) { hs_trace[hst_role][0] = __builtin_amdgcn_s_memrealtime(); hs_trace[hst_role][1] = __builtin_readcyclecounter(); }
#endif
    const int bid = cp::xcd_remap(blockIdx.x, gridDim.x);
    const int g = (int)gridDim.x;
    const int my_tiles = (p.ntiles - bid + g - 1) / g;
    if (my_tiles <= 0) return;
    // nb = n % B (the image), pass = n / B (the group of 32 TN output channels), kept by increments: round 6 found the loaders' per-tile requests spending
    // 1.7-2 k cycles EACH (selection bytes, image halo, label halo, element offsets: 7 k of the 19 k cycles of a tile of block 10, tools/debug/hs_trace.py) --
    // `inb ? ((tp.n % p.B) * H + y) * W + x : OOB` compiled into a branch per element with the 30-instruction emulated modulo inside it
    struct TilePos { int tx, ty, n, nb, pass; };
    TilePos first;
    {
        int t = bid;
        first.tx = t % p.tiles_x;
        t /= p.tiles_x;
        first.ty = t % p.tiles_y;
        first.n = t / p.tiles_y;
        first.pass = first.n / p.B;
        first.nb = first.n - first.pass * p.B;
    }
    const int d_tx = g % p.tiles_x, d_ty = (g / p.tiles_x) % p.tiles_y, d_n = g / (p.tiles_x * p.tiles_y);
    const int d_pass = d_n / p.B, d_nb = d_n - d_pass * p.B;
    auto next_tile = [&](TilePos& t) {
        t.tx += d_tx;
        int cy = 0;
        if (t.tx >= p.tiles_x) { t.tx -= p.tiles_x; cy = 1; }
        t.ty += d_ty + cy;
        int cn = 0;
        if (t.ty >= p.tiles_y) { t.ty -= p.tiles_y; cn = 1; }
        t.n += d_n + cn;
        t.nb += d_nb + cn;
        t.pass += d_pass;
        if (t.nb >= p.B) { t.nb -= p.B; ++t.pass; }
    };
    // TilePos.n runs over passes * B: pass = n / B selects 32*TN output channels (and their weight stream), n % B the image
    const int tile_w_bytes = (p.nch * 9 + (has_img ? 3 : 0)) * TN * (int)FRAG_B;   // one pass's weight stream
    const int nslices = p.nch;                       // LDS-staged slices per tile (the image halo rides with slice 0)
    const int total_slices = my_tiles * nslices;
    const int ngroups_tile = nslices * GPS + (has_img ? 1 : 0);   // weight groups per tile: GPS per slice + the image block
    const int total_groups = my_tiles * ngroups_tile;
    // Round 6: RESIDENT weights for the head layers (blocks 5 / 10: two slices + the image block = 42 KB with two operand planes).  Every tile re-staged
    // the same fragment stream through the loaders' registers -- three requests and three register -> LDS store phases a tile, in the role that
    // bounds these layers (tools/debug/hs_trace.py: loaders 16-19 k cycles of work a tile against 10-12 k for the consumers).  Staged once per block.
    constexpr unsigned WRES_B = 2 * GROUP_B + IGROUP_B;
    const bool wres = HEADK && NP <= 2 && p.w_res != 0 && p.passes == 1 && (unsigned)tile_w_bytes <= WRES_B;
    // Measured (bs 16, A/B in one call): head layers 0.518 -> 0.484 / 0.588 -> 0.555 ms with the split; the generic layers LOSE 10-20 % with their channel
    // groups dealt over the phases (the loaders' phases are the critical ones there) -- so only head layers split unless epi_split == 2 forces it
    // (a fused head outside the HS_HEADK form needs its whole row at once: never dealt)
    const bool esplit = CAN_SPLIT && (HEADK ? p.epi_split != 0 : (p.epi_split == 2 && !head)) && nslices >= 2;   // (the twin latches the tile's labels one phase after the tile)

    // ---- epilogue machinery, shared by both roles (round 5): a consumer wave w and the loader wave w + 4 can each take rows of the same tile ----
    const int ew = wave & 3;
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned npix = (unsigned)(p.B * p.H * p.Wd);
    const bool has_lab = PARTIAL || p.clade;
    const unsigned tab_b = p.scale ? (unsigned)((p.clade ? 256 : 1) * p.Cout * 4) : 0u;
    const __amdgpu_buffer_rsrc_t r_tab_s = __builtin_amdgcn_make_buffer_rsrc((void*)(p.scale ? (const void*)p.scale : (const void*)p.W), 0, tab_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_tab_b = __builtin_amdgcn_make_buffer_rsrc((void*)(p.scale ? (const void*)p.shift : (const void*)p.W), 0, tab_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual ? (const void*)p.residual : (const void*)p.W), 0,
                                                                            p.residual ? npix * (unsigned)p.res_ld * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_raw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out_raw ? (void*)p.out_raw : (void*)p.W), 0,
                                                                            p.out_raw ? npix * (unsigned)p.raw_ld * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_act = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out_act ? (void*)p.out_act : (void*)p.W), 0,
                                                                            p.out_act ? npix * (unsigned)p.act_ld * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_head = __builtin_amdgcn_make_buffer_rsrc((void*)(head ? (void*)p.head_out : (void*)p.W), 0,
                                                                             head ? npix * (unsigned)p.head_ld * 4u : 0u, 0x00020000);
    int pmask[2] = {0x1ff, 0x1ff}, clab[2] = {0, 0};
    // labels of a tile come from the label halo the loaders staged with the tile's first slice: no global latency, no registers held
    auto read_labels = [&](int parity) {
        const unsigned short* lh = labh + parity * HP;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int centre = (2 * ew + r + 1) * COLS + lrow + 1;
            const int lc = lh[centre];
            if constexpr (PARTIAL) {
                int m = 0;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) m |= ((int)lh[centre + (tp / 3 - 1) * COLS + (tp % 3 - 1)] == lc) ? (1 << tp) : 0;
                pmask[r] = (lc & 0xff00) ? 0 : m;
            }
            clab[r] = lc & 0xff;
        }
    };

    // Epilogue (round 5: straight-line).  Round 4's form cost 0.31 ms of block 5's 0.73 (tools/debug/hs_ablate.sh) -- not for its arithmetic: every
    // `pok && ch < Cout ? offset : OOB`, every `if (nq >= 4) ... else if ...` store ladder and the short-circuit arg-max compiled into
    // s_and_saveexec / s_cbranch_execz regions (87 of them in the two rows of a tile, each a dozen issue slots with the matrix pipe idle).  Now:
    // conditions are combined bitwise and select an offset (an out-of-range buffer offset drops the access), operands that do not exist are
    // skipped by UNIFORM branches only, and the head's stores are chosen by uniform comparisons with head_cout.  Same expressions, same results.
    const __amdgpu_buffer_rsrc_t r_hlab = __builtin_amdgcn_make_buffer_rsrc((void*)((head && p.head_lab) ? (void*)p.head_lab : (void*)p.W), 0,
                                                                             (head && p.head_lab) ? npix : 0u, 0x00020000);
#if defined(HS_EPI_NOSTORE)   // timing experiments (variant builds only)
    const bool has_res = p.residual != nullptr, has_tab = p.scale != nullptr, has_raw = p.out_raw != nullptr && p.B < 0, has_act = p.out_act != nullptr && p.B < 0;
#elif defined(HS_EPI_NORES)
    const bool has_res = p.residual != nullptr && p.B < 0, has_tab = p.scale != nullptr, has_raw = p.out_raw != nullptr, has_act = p.out_act != nullptr;
#elif defined(HS_EPI_NORAW)
    const bool has_res = p.residual != nullptr, has_tab = p.scale != nullptr, has_raw = p.out_raw != nullptr && p.B < 0, has_act = p.out_act != nullptr;
#else
    const bool has_res = p.residual != nullptr, has_tab = p.scale != nullptr, has_raw = p.out_raw != nullptr, has_act = p.out_act != nullptr;
#endif
    // HEADK: the per-channel table of a layer without CLADE, loaded once; where no partial-convolution factor exists the weights' power-of-two
    // descale is folded into its scale column (exact: a power of two commutes with the rounding of the product)
    float4 hk_sc[4], hk_sh[4];
    if constexpr (HEADK) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            hk_sc[g4] = make_float4(0.f, 0.f, 0.f, 0.f);
            hk_sh[g4] = hk_sc[g4];
            if (!p.clade) {
                hk_sc[g4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_tab_s, (g4 * 8 + kh * 4) * 4, 0, 0));
                hk_sh[g4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_tab_b, (g4 * 8 + kh * 4) * 4, 0, 0));
                if constexpr (NP == 2 && !PARTIAL) {
                    hk_sc[g4].x *= p.descale; hk_sc[g4].y *= p.descale; hk_sc[g4].z *= p.descale; hk_sc[g4].w *= p.descale;
                }
            }
        }
    }
    float e_amax = 0.f, l_amax = 0.f;   // f16x2 range monitor (p.mon): this thread's maxima over the head operand / over what it staged as a loader
    auto flush_mon = [&]() __attribute__((always_inline)) {
#ifdef HS_TRACE
        if (hst_on) { hs_trace[hst_role][2046] = __builtin_amdgcn_s_memrealtime(); hs_trace[hst_role][2047] = ((unsigned long long)hst_i << 48) | (__builtin_readcyclecounter() & 0xffffffffffffull); }
#endif
        if constexpr (NP == 2) {
            if (p.mon) {   // uniform
                cp::monitor_flush(p.mon, l_amax);
                cp::monitor_flush(p.mon + 2, e_amax);
                cp::monitor_count_launch(p.mon, threadIdx.x == 0);
            }
        }
    };
    // pieces: bit j * 4 + g4 set = this call handles that group of four channels (generic form; a head layer's row is one piece)
    auto epilogue = [&](f32x16 (&acc)[2][TN], int r_begin, int r_end, int n, int y0, int x0, int cbase, unsigned pieces) __attribute__((always_inline)) {
        asm volatile("" : "+s"(n), "+s"(y0), "+s"(x0), "+s"(cbase));   // (pins the tile-dependent arithmetic below to this point: see epilogue_t)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r < r_begin || r >= r_end) continue;   // (uniform: the rows of a tile can be divided between a consumer wave and its loader twin)
            const int y = y0 + 2 * ew + r, x = x0 + lrow;
            const bool pok = (y < p.H) & (x < p.Wd);
            const unsigned pix = (unsigned)((n * p.H + y) * p.Wd + x);
            // whole output records (head_pre_n > 0, HS_HEADK): the record's first floats come from dense rows another head wrote.  Requested here, stored
            // behind the head's own columns: lane half kh moves floats 4 kh .. 4 kh + 3, half 0 also floats 8 .. head_pre_n - 1 (8 <= head_pre_n <= 12)
            u32x4 pre4 = {0u, 0u, 0u, 0u}, pre8 = {0u, 0u, 0u, 0u};
            if constexpr (PREFIX) {
                {
                    const unsigned po = pix * (unsigned)p.res_ld * 4u;
                    pre4 = __builtin_amdgcn_raw_buffer_load_b128(r_res, (int)(pok ? po + 16u * (unsigned)kh : OOB), 0, 0);
                    if (p.head_pre_n > 8) pre8 = __builtin_amdgcn_raw_buffer_load_b128(r_res, (int)((pok & (kh == 0)) ? po + 32u : OOB), 0, 0);
                }
            }
            float f = 1.f;
            if constexpr (PARTIAL) f = p.norm ? 9.0f / (float)max(__popc(pmask[r]), 1) : 1.0f;
            if constexpr (NP == 2) f *= p.descale;   // the weights' power-of-two scale, undone exactly
            const unsigned tab_row = (unsigned)(clab[r] * (p.clade ? p.Cout : 0));
            float4 keep[4];
            if constexpr (HEADK) {
                // t = leaky((acc * f) * scale + shift), the generic form's expressions without the operands this layer does not have; pixels
                // outside the image keep whatever they computed (a pixel is a column of the head's product and is not stored)
                const bool fold = (NP == 2 && !PARTIAL);   // f == descale, already inside hk_sc
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float4 sc = hk_sc[g4], sh = hk_sh[g4];
                    if (p.clade) {
                        const unsigned to = (tab_row + (unsigned)(g4 * 8 + kh * 4)) * 4u;
                        sc = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_tab_s, (int)to, 0, 0));
                        sh = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_tab_b, (int)to, 0, 0));
                        if (fold) { sc.x *= p.descale; sc.y *= p.descale; sc.z *= p.descale; sc.w *= p.descale; }
                    }
                    float4 t;
                    if (fold) {
                        t.x = acc[r][0][g4 * 4 + 0] * sc.x + sh.x;
                        t.y = acc[r][0][g4 * 4 + 1] * sc.y + sh.y;
                        t.z = acc[r][0][g4 * 4 + 2] * sc.z + sh.z;
                        t.w = acc[r][0][g4 * 4 + 3] * sc.w + sh.w;
                    } else {
                        t.x = (acc[r][0][g4 * 4 + 0] * f) * sc.x + sh.x;
                        t.y = (acc[r][0][g4 * 4 + 1] * f) * sc.y + sh.y;
                        t.z = (acc[r][0][g4 * 4 + 2] * f) * sc.z + sh.z;
                        t.w = (acc[r][0][g4 * 4 + 3] * f) * sc.w + sh.w;
                    }
                    keep[g4] = make_float4(leaky01(t.x), leaky01(t.y), leaky01(t.z), leaky01(t.w));
                }
            } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float4 res[4], esc[4], esh[4];
                // the 128-channel kernels have no registers to park a whole row's operands: they fetch per group of four channels
                constexpr int PRE = (TN == 4) ? 1 : 4;
#pragma unroll
                for (int g0 = 0; g0 < 4; g0 += PRE) {
#pragma unroll
                for (int g4 = g0; g4 < g0 + PRE; ++g4) {
                    if (!((pieces >> (j * 4 + g4)) & 1u)) continue;
                    const int ch = cbase + j * 32 + g4 * 8 + kh * 4;
                    const bool cok = ch < p.Cout;
                    res[g4] = make_float4(0.f, 0.f, 0.f, 0.f);
                    esc[g4] = res[g4];
                    esh[g4] = res[g4];
                    if (has_res)   // uniform branches: a layer without these operands issues no loads and waits for none
                        res[g4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_res, (int)((pok & cok) ? (pix * (unsigned)p.res_ld + (unsigned)ch) * 4u : OOB), 0, 0));
                    if (tab_lds) {
                        const int cl = ch < TAB_C - 3 ? ch : 0;   // (channels past Cout are not stored; keep the read inside the table)
                        esc[g4] = *reinterpret_cast<const float4*>(tabl + cl);
                        esh[g4] = *reinterpret_cast<const float4*>(tabl + TAB_C + cl);
                    } else if (has_tab) {
                        const unsigned to = cok ? (tab_row + (unsigned)ch) * 4u : OOB;
                        esc[g4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_tab_s, (int)to, 0, 0));
                        esh[g4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_tab_b, (int)to, 0, 0));
                    }
                }
#pragma unroll
                for (int g4 = g0; g4 < g0 + PRE; ++g4) {
                    if (!((pieces >> (j * 4 + g4)) & 1u)) continue;
                    const int ch = cbase + j * 32 + g4 * 8 + kh * 4;
                    const bool ok = pok & (ch < p.Cout);
                    float4 v;
                    v.x = acc[r][j][g4 * 4 + 0] * f + res[g4].x;
                    v.y = acc[r][j][g4 * 4 + 1] * f + res[g4].y;
                    v.z = acc[r][j][g4 * 4 + 2] * f + res[g4].z;
                    v.w = acc[r][j][g4 * 4 + 3] * f + res[g4].w;
                    if (has_raw)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_raw, (int)(ok ? (pix * (unsigned)p.raw_ld + (unsigned)ch) * 4u : OOB), 0, HS_EPI_AUX);
                    float4 t = v;
                    if (has_tab) {
                        t.x = v.x * esc[g4].x + esh[g4].x;
                        t.y = v.y * esc[g4].y + esh[g4].y;
                        t.z = v.z * esc[g4].z + esh[g4].z;
                        t.w = v.w * esc[g4].w + esh[g4].w;
                    }
                    if (p.act == CP_ACT_RELU) {
                        t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
                    } else if (p.act == CP_ACT_LEAKY01) {
                        t.x = leaky01(t.x); t.y = leaky01(t.y); t.z = leaky01(t.z); t.w = leaky01(t.w);
                    }
                    if (has_act)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), r_act, (int)(ok ? (pix * (unsigned)p.act_ld + (unsigned)ch) * 4u : OOB), 0, HS_EPI_AUX);
                    if (j == 0) {
                        keep[g4].x = ok ? t.x : 0.f;
                        keep[g4].y = ok ? t.y : 0.f;
                        keep[g4].z = ok ? t.z : 0.f;
                        keep[g4].w = ok ? t.w : 0.f;
                    }
                }
                }
            }
            }   // !HEADK
            if constexpr (TN == 1) {
                if (head) {
                    if (NP == 2 && p.mon) {   // (uniform) the head's operand as it is converted below; pixels outside the image compute with padding and are not stored
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) e_amax = pok ? cp::amax4(e_amax, keep[g4]) : e_amax;
                    }
                    // Fused 1x1 head: out[q][pixel] = sum_c Wh[c][q] * t[c][pixel] on the same matrix pipe.  The order of K is free, so step m
                    // takes, from lane half kh, the eight channels this lane already holds: 8*(2m) + 4*kh + 0..3 and 8*(2m+1) + 4*kh + 0..3
                    // (the head weights are packed in that order); the activated values are split / rounded in registers.
                    f32x16 a2;
#pragma unroll
                    for (int e = 0; e < 16; ++e) a2[e] = 0.f;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        bf16x8 px[NP];
                        if constexpr (NP == 3) {
                            uint2 h0, m0, l0, h1, m1, l1;
                            split4(keep[2 * m], h0, m0, l0);
                            split4(keep[2 * m + 1], h1, m1, l1);
                            px[0] = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
                            px[1] = __builtin_bit_cast(bf16x8, make_uint4(m0.x, m0.y, m1.x, m1.y));
                            px[2] = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
                        } else if constexpr (NP == 2) {
                            uint2 h0, l0, h1, l1;
                            cp::split4h(keep[2 * m], h0, l0);
                            cp::split4h(keep[2 * m + 1], h1, l1);
                            px[0] = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
                            px[1] = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
                        } else {
                            const uint2 a = round4(keep[2 * m]), b = round4(keep[2 * m + 1]);
                            px[0] = __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
                        }
                        bf16x8 hw[NP];
#pragma unroll
                        for (int sidx = 0; sidx < NP; ++sidx) hw[sidx] = *reinterpret_cast<const bf16x8*>(hwl + (unsigned)(m * NP + sidx) * 1024u + wlane);
#pragma unroll
                        for (int t6 = 0; t6 < NPROD; ++t6) {
                            a2 = mfma_np<NP>(hw[prod_w<NP>(t6)], px[prod_p<NP>(t6)], a2);
                        }
                    }
                    if constexpr (NP == 2) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) a2[e] *= p.head_descale;
                    }
                    // stores: register g4 * 4 + e of lane half kh is head channel q = 8 g4 + 4 kh + e.  A group of eight channels that lies wholly
                    // below head_cout goes out as one 16-byte store per lane; the group that straddles it as single dwords, one store per e that
                    // ANY lane half still owns -- which stores exist is decided by uniform comparisons, which lanes take part by the offset
                    const unsigned hbase = (pix * (unsigned)p.head_ld + (unsigned)(kh * 4 + (PREFIX ? p.head_pre_n : 0))) * 4u;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const unsigned o = hbase + (unsigned)(g4 * 32);
                        if (p.head_cout >= g4 * 8 + 8) {
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__builtin_bit_cast(unsigned, (float)a2[g4 * 4 + 0]), __builtin_bit_cast(unsigned, (float)a2[g4 * 4 + 1]),
                                                                         __builtin_bit_cast(unsigned, (float)a2[g4 * 4 + 2]), __builtin_bit_cast(unsigned, (float)a2[g4 * 4 + 3])},
                                                                   r_head, (int)(pok ? o : OOB), 0, 0);
                        } else if (p.head_cout > g4 * 8) {
                            const int left = p.head_cout - g4 * 8 - kh * 4;   // channels of this group this lane half still owns (<= 0: none)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (p.head_cout > g4 * 8 + e)   // lane half 0 owns q = 8 g4 + e
                                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)a2[g4 * 4 + e]), r_head, (int)((pok & (e < left)) ? o + 4u * e : OOB), 0, 0);
                        }
                    }
                    if constexpr (PREFIX) {
                        {   // the copied floats complete the record's lines (each role copies the rows it finishes: handing row 1's copy to the consumer
                            // wave as well was measured slower, 0.56 -> 0.66 ms for block 10)
                            const unsigned ro = pix * (unsigned)p.head_ld * 4u;
                            __builtin_amdgcn_raw_buffer_store_b128(pre4, r_head, (int)(pok ? ro + 16u * (unsigned)kh : OOB), 0, 0);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (8 + e < p.head_pre_n) __builtin_amdgcn_raw_buffer_store_b32(pre8[e], r_head, (int)((pok & (kh == 0)) ? ro + 32u + 4u * e : OOB), 0, 0);
                        }
                    }
                    if (p.head_lab) {   // the hard label map straight from the head's registers: first maximum wins (cp_argmax_labels)
                        float best = -__builtin_inff();
                        int bi = 0x7fffffff;
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int q = g4 * 8 + kh * 4 + e;
                                const float vq = a2[g4 * 4 + e];
                                const bool take = (q < p.head_lab_classes) & (vq > best);
                                best = take ? vq : best;
                                bi = take ? q : bi;
                            }
                        const float ob = __shfl_xor(best, 32);
                        const int oi = __shfl_xor(bi, 32);
                        const bool other = (ob > best) | ((ob == best) & (oi < bi));
                        bi = other ? oi : bi;
                        bi = (bi == 0x7fffffff) ? 0 : bi;
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bi, r_hlab, (int)(((kh == 0) & pok) ? pix : OOB), 0, 0);
                    }
                }
            }
        }
    };

    if constexpr (NP == 2) cp::f16_overflow_clamps();
    if (loader) {
        // ------------------------------------------------------------------ loaders ------------------------------------------------------
#ifdef HS_LOADER_IDLE
        if (p.B > 0) {   // timing experiment: the consumers alone (LDS holds whatever it held)
            if constexpr (BILINEAR) CP_BARRIER();
            CP_BARRIER();
            for (int gg = 0; gg < total_groups; ++gg) CP_BARRIER();
            return;
        }
#endif
        TilePos etile = first;   // the tile whose row-1 epilogue this wave owes (epi_split): the one BEFORE the tile the consumers are multiplying
        int e_n = 0, e_y0 = 0, e_x0 = 0, e_cbase = 0, e_k = 0;
        constexpr int NPIECE = HEADK ? 1 : 4 * TN;
        const int e_q = HEADK ? 1 : (NPIECE + max(ngroups_tile - 1, 1) - 1) / max(ngroups_tile - 1, 1);   // channel groups per phase
        // phase = index of the current phase inside the consumers' tile; last = the call after the block's last tile (everything that is left)
        auto loader_epilogue = [&](int phase, bool last) __attribute__((always_inline)) {
            if constexpr (CAN_SPLIT) {
                if (phase == 0 || last) {   // latch the owed tile's position and labels (its label halo is overwritten later in this tile)
                    const int pass = etile.pass;
                    e_n = etile.nb; e_y0 = etile.ty * TH; e_x0 = etile.tx * 32; e_cbase = pass * 32 * TN;
                    if (has_lab) read_labels(e_k & 1);
                }
                unsigned pieces;
                if (last) pieces = 0xffffffffu;
                else if (HEADK) pieces = (phase == ngroups_tile - 1) ? 1u : 0u;
                else {
                    const int b0 = phase * e_q, b1 = min(NPIECE, b0 + e_q);
                    pieces = b0 < b1 ? ((b1 >= 32 ? 0xffffffffu : ((1u << b1) - 1u)) & ~((1u << b0) - 1u)) : 0u;
                }
                if (pieces) {
                    f32x16 accl[2][TN];
                    const unsigned char* src = accst + (unsigned)(HEADK ? (e_k & 1) * ACCST_B : 0);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (HEADK || ((pieces >> (j * 4 + g4)) & 1u)) v = *reinterpret_cast<const float4*>(src + (unsigned)(((ew * TN + j) * 4 + g4) * 1024) + wlane);
                            accl[1][j][g4 * 4 + 0] = v.x; accl[1][j][g4 * 4 + 1] = v.y; accl[1][j][g4 * 4 + 2] = v.z; accl[1][j][g4 * 4 + 3] = v.w;
                            accl[0][j][g4 * 4 + 0] = v.x; accl[0][j][g4 * 4 + 1] = v.y; accl[0][j][g4 * 4 + 2] = v.z; accl[0][j][g4 * 4 + 3] = v.w;
                        }
                    epilogue(accl, 1, 2, e_n, e_y0, e_x0, e_cbase, pieces);
                }
                if (last || phase == ngroups_tile - 1) {   // this tile's row is done: the next one is owed
                    next_tile(etile);
                    ++e_k;
                }
            }
        };
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.s[0].data, 0, p.s[0].bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.s[1].data ? p.s[1].data : p.s[0].data), 0,
                                                                              p.s[1].data ? p.s[1].bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)(has_img ? p.img : p.s[0].data), 0, has_img ? p.img_bytes : 0u, 0x00020000);
        // halo element `it` of a thread: float4 number it*256 + tid = (pixel, channel quad)
        int e_hy[NIT], e_hx[NIT];
        unsigned e_lds[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 256 + tid, pix = idx >> 2, q = idx & 3;
            e_hy[it] = pix < HP ? pix / COLS : 0x4000;
            e_hx[it] = pix % COLS;
            e_lds[it] = (unsigned)(pix * 32 + (((q >> 1) ^ ((pix >> 3) & 1)) * 16) + (q & 1) * 8);
        }
        const int q4 = (tid & 3) * 4;
        // Round 6: per-tile requests by strength reduction.  An element's pixel is (tile base) + (its own offset inside the halo): e_pof = hy * W + hx for
        // the full-resolution sources and maps, e_sof = ((hy - 1) >> 1) * Ws + ((hx - 1) >> 1) for the half-resolution source of a guided x2 input (a
        // tile starts at an odd image row / column minus ... : y = 8 ty - 1 + hy, so y >> 1 = 4 ty + ((hy - 1) >> 1)); a tile that does not touch the
        // image border (most) needs no range test per element.  Before, every request recomputed ((n H + y) W + x) and its range test per element
        // with three integer multiplies each: ~700 instructions per tile in the loaders of a head layer, 7 k of its 19 k cycles (tools/debug/hs_trace.py).
        int e_pof[NIT], e_sof[NIT];
        bool e_ok[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            e_ok[it] = e_hy[it] < 0x4000;
            e_pof[it] = e_ok[it] ? e_hy[it] * p.Wd + e_hx[it] : 0;
            e_sof[it] = (SEL && e_ok[it]) ? ((e_hy[it] - 1) >> 1) * p.s[0].Ws + ((e_hx[it] - 1) >> 1) : 0;
        }
        auto tile_interior = [&](const TilePos& tp) { return tp.ty > 0 && tp.tx > 0 && tp.ty * TH + TH + 1 <= p.H && tp.tx * 32 + 33 <= p.Wd; };
        float4 lv[NIT][NV];
        float4 liv[NIMG];
        int selb[NIT];       // SEL: the selection byte of this element's pixel (constant over the slices of a tile)
        const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc((void*)(SEL ? (const void*)p.s[0].sel : (const void*)p.W), 0,
                                                                              SEL ? p.lab_bytes : 0u, 0x00020000);
        // Round 6: the selection bytes of a tile are requested ONE TILE AHEAD (selb_nx).  They feed the tile's element offsets, so requesting them with
        // the tile itself put a whole memory round trip into the loaders' path once per tile -- 4-6 k cycles of the 19 k a tile of block 10 takes,
        // with the consumers waiting at the barrier meanwhile (tools/debug/hs_trace.py: the "issue" section of that phase 6.0 k against 0.6-0.9 k).
        int selb_nx[NIT];
        auto issue_sel = [&](const TilePos& tp) {
            if constexpr (SEL) {
                const int tb = (tp.nb * p.H + tp.ty * TH - 1) * p.Wd + tp.tx * 32 - 1;   // (uniform)
                if (tile_interior(tp)) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) selb_nx[it] = __builtin_amdgcn_raw_buffer_load_b8(rss, e_ok[it] ? tb + e_pof[it] : (int)OOB, 0, 0);
                } else {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int y = tp.ty * TH - 1 + e_hy[it], x = tp.tx * 32 - 1 + e_hx[it];
                        const bool inb = ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.Wd);
                        selb_nx[it] = __builtin_amdgcn_raw_buffer_load_b8(rss, inb ? tb + e_pof[it] : (int)OOB, 0, 0);
                    }
                }
            }
        };
        // byte offsets of this thread's halo elements for the tile being fetched (channel 0 of the slice's 16; OOB outside the image), computed
        // once per tile: a slice only adds its uniform channel offset through the load's scalar offset -- no per-slice address arithmetic
        unsigned eo0[NIT][NV], eo1[NIT];
        auto tile_offsets = [&](const TilePos& tp) {
            const int y0 = tp.ty * TH, x0 = tp.tx * 32;
            const int tb = (tp.nb * p.H + y0 - 1) * p.Wd + x0 - 1;                                           // (uniform) pixel of the halo's corner
            const int sb = SEL ? (tp.nb * p.s[0].Hs + (TH / 2) * tp.ty) * p.s[0].Ws + 16 * tp.tx : 0;        // ... and of its half-resolution source
            const unsigned ld0b = (unsigned)p.s[0].ld * 4u, ld1b = (unsigned)p.s[1].ld * 4u, q4b = (unsigned)q4 * 4u;
            const bool interior = tile_interior(tp);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                bool inb = e_ok[it];
                if (!interior) {   // (uniform)
                    const int y = y0 - 1 + e_hy[it], x = x0 - 1 + e_hx[it];
                    inb = ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.Wd);
                }
                const unsigned pixel = (unsigned)(tb + e_pof[it]);
                eo1[it] = inb ? pixel * ld1b + q4b : OOB;
                if constexpr (BILINEAR) {
                    eo0[it][0] = OOB;   // source 0 goes through the low-resolution LDS tile (bilinear loader below)
                } else if constexpr (SEL) {
                    const int sl = selb[it];
                    const unsigned sp = (unsigned)(sb + e_sof[it] + ((sl & 2) ? p.s[0].Ws : 0) + (sl & 1));
                    eo0[it][0] = inb ? sp * ld0b + q4b : OOB;
                } else {
                    eo0[it][0] = inb ? pixel * ld0b + q4b : OOB;
                }
            }
        };
        // part / nparts: only the elements it with it % nparts == part (the generic loop of the 64-channel kernels deals a slice's halo over the
        // GPS weight-group phases of the slice before it; part = -1: all of them)
        auto issue_slice = [&](const TilePos& tp, int c, int part = -1, int nparts = 1) {
            (void)tp;
            const int si = c >= p.nch0 ? 1 : 0;
            const int cs = (c - (si ? p.nch0 : 0)) * 64;   // uniform byte offset of the slice's first channel
#ifdef HS_NOLOAD
            if (p.B > 0) return;   // timing experiment
#endif
            if (si == 0) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    if (part >= 0 && it % nparts != part) continue;
#pragma unroll
                    for (int v = 0; v < NV; ++v) lv[it][v] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)eo0[it][v], cs, 0));
                }
            } else {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    if (part >= 0 && it % nparts != part) continue;
                    lv[it][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)eo1[it], cs, 0));
                }
            }
        };
        auto store_slice = [&](int stage, int part = -1, int nparts = 1) {
#ifdef HS_NOSTORE
            if (p.B > 0) return;   // timing experiment
#endif
            unsigned char* h = halo + stage * (NP * PLANE_B);
            if (NP == 2 && p.mon) {   // (uniform; out-of-range elements loaded zeros)
#pragma unroll
                for (int it = 0; it < NIT; ++it) l_amax = cp::amax4(l_amax, lv[it][0]);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (part >= 0 && it % nparts != part) continue;
                if (e_hy[it] >= 0x4000) continue;
                store_planes<NP>(h + e_lds[it], PLANE_B, lv[it][0]);
            }
        };
        // pixel elements (image halo, label halo): one per halo pixel, same strength reduction as the channel elements above
        int i_hy[NIMG], i_hx[NIMG], i_pof[NIMG];
        bool i_ok[NIMG];
#pragma unroll
        for (int it = 0; it < NIMG; ++it) {
            const int pix = it * 256 + tid;
            i_ok[it] = pix < HP;
            i_hy[it] = pix / COLS;
            i_hx[it] = pix % COLS;
            i_pof[it] = i_ok[it] ? i_hy[it] * p.Wd + i_hx[it] : 0;
        }
        auto pixel_inb = [&](const TilePos& tp, int it, bool interior) -> bool {
            if (interior) return i_ok[it];
            const int y = tp.ty * TH - 1 + i_hy[it], x = tp.tx * 32 - 1 + i_hx[it];
            return i_ok[it] & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.Wd);
        };
        auto issue_img = [&](const TilePos& tp) {
            const int tb = (tp.nb * p.H + tp.ty * TH - 1) * p.Wd + tp.tx * 32 - 1;
            const bool interior = tile_interior(tp);
#pragma unroll
            for (int it = 0; it < NIMG; ++it) {
                const bool inb = pixel_inb(tp, it, interior);
                liv[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsi, (int)(inb ? (unsigned)(tb + i_pof[it]) * 16u : OOB), 0, 0));
            }
        };
        auto store_img = [&](int parity) {
            unsigned char* h = imgh + parity * (NP * IPLANE_B);
            if (NP == 2 && p.mon) {
#pragma unroll
                for (int it = 0; it < NIMG; ++it) l_amax = cp::amax4(l_amax, liv[it]);
            }
#pragma unroll
            for (int it = 0; it < NIMG; ++it) {
                const int pix = it * 256 + tid;
                if (pix >= HP) continue;
                store_planes<NP>(h + pix * 8, IPLANE_B, liv[it]);
            }
        };
        const bool has_lab_l = PARTIAL || p.clade;
        const __amdgpu_buffer_rsrc_t rsl_l = __builtin_amdgcn_make_buffer_rsrc((void*)(has_lab_l ? (const void*)p.label : (const void*)p.W), 0,
                                                                                has_lab_l ? p.lab_bytes : 0u, 0x00020000);
        int llab[NIMG], llab_out[NIMG];
        auto issue_lab = [&](const TilePos& tp) {
            const int tb = (tp.nb * p.H + tp.ty * TH - 1) * p.Wd + tp.tx * 32 - 1;
            const bool interior = tile_interior(tp);
#pragma unroll
            for (int it = 0; it < NIMG; ++it) {
                const bool inb = pixel_inb(tp, it, interior);
                llab[it] = __builtin_amdgcn_raw_buffer_load_b8(rsl_l, inb ? tb + i_pof[it] : (int)OOB, 0, 0);
                llab_out[it] = inb ? 0 : 0xff00;   // (round 6: OR-ed in by store_lab -- here the OR waited for the load just issued, a round trip per tile)
            }
        };
        auto store_lab = [&](int parity) {
#pragma unroll
            for (int it = 0; it < NIMG; ++it) {
                const int pix = it * 256 + tid;
                if (pix < HP) labh[parity * HP + pix] = (unsigned short)(llab[it] | llab_out[it]);
            }
        };
        // weight groups: the tile's fragment stream is contiguous in memory, GROUP_B bytes per group; every tile reads the same stream
        const __amdgpu_buffer_rsrc_t rsw_l = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.w_bytes, 0x00020000);
        u32x4 lw[NWL];
        // the weight cursor: groups are requested strictly in order (0, 1, 2, ...), so (group within the tile, pass of the tile) advance by
        // increments -- round 4 divided by ngroups_tile and by tiles_per_pass for every group (two emulated integer divisions per phase)
        int w_lg = 0, w_pass = bid / p.tiles_per_pass, w_rem = bid % p.tiles_per_pass;
        const int g_q = g / p.tiles_per_pass, g_r = g % p.tiles_per_pass;
        auto issue_w = [&](int gg) {   // global group index -> group of the tile (wide groups first, the image block's group last)
            (void)gg;   // (groups are requested in order)
            if (wres) return;   // (uniform) the whole stream is resident
            const int lgw = w_lg, pass = w_pass;   // this block's current tile belongs to that pass of output channels
            if (++w_lg == ngroups_tile) {
                w_lg = 0;
                w_pass += g_q;
                w_rem += g_r;
                if (w_rem >= p.tiles_per_pass) { w_rem -= p.tiles_per_pass; ++w_pass; }
            }
            const unsigned base = (unsigned)(pass * tile_w_bytes) + (unsigned)lgw * GROUP_B;
            const unsigned len = (lgw < nslices * GPS) ? GROUP_B : IGROUP_B;
#pragma unroll
            for (int it = 0; it < NWL; ++it) {
                const unsigned o = (unsigned)(it * 256 + tid) * 16u;
                lw[it] = __builtin_amdgcn_raw_buffer_load_b128(rsw_l, (int)(o < len ? base + o : OOB), 0, 0);
            }
        };
        auto store_w = [&](int stage) {
            if (wres) return;
#pragma unroll
            for (int it = 0; it < NWL; ++it) {
                const unsigned o = (unsigned)(it * 256 + tid) * 16u;
                if (o < GROUP_B) *reinterpret_cast<u32x4*>(wst + stage * GROUP_B + o) = lw[it];
            }
        };
        // While the consumers multiply group gg: the weights of group gg+1 go into the other weight stage (and group gg+2 is requested);
        // during the FIRST group of a slice the halo of the next slice goes into the other halo stage (and the slice after it is requested).
        TilePos ftile = first;
        int fc = 0, fk = 0;
        bool img_pending = false;
        auto advance = [&]() {      // move the fetch cursor to the next slice
            if (++fc == nslices) {
                fc = 0;
                ++fk;
                next_tile(ftile);
            }
        };
        auto issue_tile_extras = [&]() {   // with slice 0 of a tile: its element offsets, image halo, label halo; the NEXT tile's selection bytes
            if (fc != 0) return;
            if constexpr (SEL) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) selb[it] = selb_nx[it];   // requested a tile ago
            }
            tile_offsets(ftile);
            if constexpr (SEL) {
                TilePos nt = ftile;
                next_tile(nt);       // (past the block's last tile: a valid address of some other tile, or out of range -- never used)
                issue_sel(nt);
            }
            if (has_img) issue_img(ftile);
            if (has_lab_l) issue_lab(ftile);
            img_pending = true;
        };
        auto store_tile_extras = [&]() {   // stages of tile fk: read last during tile fk - 2
            if (!img_pending) return;
            if (has_img) store_img(fk & 1);
            if (has_lab_l) store_lab(fk & 1);
            img_pending = false;
        };
        if constexpr (HEADK) {
            if (wres) {   // the tile's whole fragment stream, staged once (group g at g * GROUP_B, the image block's group behind the slices')
                for (unsigned o = (unsigned)tid * 16u; o < (unsigned)tile_w_bytes; o += 256u * 16u)
                    *reinterpret_cast<u32x4*>(wst + o) = __builtin_amdgcn_raw_buffer_load_b128(rsw_l, (int)o, 0, 0);
            }
        }
        if (head) {   // the fused head's weights: 2 steps x NP KB, staged once
            const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc((void*)p.head_w, 0, 2u * NP * 1024u, 0x00020000);
            for (unsigned o = (unsigned)tid * 16u; o < 2u * NP * 1024u; o += 256u * 16u)
                *reinterpret_cast<u32x4*>(hwl + o) = __builtin_amdgcn_raw_buffer_load_b128(rsh, (int)o, 0, 0);
        }
        if constexpr (BILINEAR) {
            // ---------------------------------------------- loaders, source 0 at half resolution -----------------------------------------
            // Slice s (global index) is multiplied in phase s.  Its halo stage is written in phase s - 1: a DIRECT slice (source 1) from the
            // registers its loads were issued into in phase s - 2 (as in the generic loader); a BILINEAR slice by interpolation from the
            // low-resolution stage s & 1, which was stored in phase s - 2 from loads issued in phase s - 3.  Three cursors walk the block's
            // slice list one, two and three slices ahead of the consumers.
            constexpr int NLO = (LOW_P * 4 + 255) / 256;
            struct Cur { TilePos t; int c, k, s; };   // tile, slice within the tile, tile index of this block, global slice index
            auto step = [&](Cur& u) {
                ++u.s;
                if (++u.c == nslices) { u.c = 0; ++u.k; next_tile(u.t); }
            };
            bool l_ok[NLO];
            int l_r[NLO], l_c[NLO];
            unsigned l_lds[NLO];
#pragma unroll
            for (int it = 0; it < NLO; ++it) {
                const int idx = it * 256 + tid, pix = idx >> 2;
                l_ok[it] = pix < LOW_P;
                l_r[it] = pix / LOW_C;
                l_c[it] = pix % LOW_C;
                l_lds[it] = (unsigned)(pix * 64 + (idx & 3) * 16);
            }
            float4 llow[NLO];
            unsigned elo[NLO];
            int elo_k = -1, eo1_k = -1;
            auto issue_low = [&](const Cur& u) {
                if (u.k != elo_k) {   // first low-resolution slice of a tile: the tile's source offsets (edge-clamped, as the reference's resize)
                    elo_k = u.k;
                    const int n = u.t.nb, ys0 = u.t.ty * (TH / 2) - 1, xs0 = u.t.tx * 16 - 1;
                    const int Hs = p.s[0].Hs, Ws = p.s[0].Ws;
#pragma unroll
                    for (int it = 0; it < NLO; ++it) {
                        const int ys = min(max(ys0 + l_r[it], 0), Hs - 1), xs = min(max(xs0 + l_c[it], 0), Ws - 1);
                        elo[it] = l_ok[it] ? (unsigned)((((n * Hs + ys) * Ws + xs) * p.s[0].ld + q4) * 4) : OOB;
                    }
                }
#ifdef HS_NOLOAD
                if (p.B > 0) return;   // timing experiment
#endif
#pragma unroll
                for (int it = 0; it < NLO; ++it) llow[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)elo[it], u.c * 64, 0));
            };
            auto store_low = [&](int stage) {
                if (NP == 2 && p.mon) {   // the low-resolution source bounds its x2 interpolation (a convex combination)
#pragma unroll
                    for (int it = 0; it < NLO; ++it) l_amax = l_ok[it] ? cp::amax4(l_amax, llow[it]) : l_amax;
                }
#pragma unroll
                for (int it = 0; it < NLO; ++it)
                    if (l_ok[it]) *reinterpret_cast<float4*>(lowb + stage * LOW_B + l_lds[it]) = llow[it];
            };
            // Interpolation by 2 x 2 BLOCKS (round 5).  Half-pixel centres: halo rows 2 by, 2 by + 1 (image rows y0 - 1 + ..., y0 - 1 odd) both lie between
            // source rows by and by + 1 of the stage -- weights 0.75 / 0.25 and 0.25 / 0.75 -- and likewise the columns, so the four halo pixels of a
            // block share ONE 2 x 2 source neighbourhood: a task = (block, channel quad) reads four 16-byte source values and produces four halo
            // values (round 4: a task per halo value, four reads each: 4x the LDS reads, 1.5x the arithmetic).  The horizontal blends are shared by the
            // two rows; arithmetic on float pairs (v_pk_mul_f32 / v_pk_fma_f32: the loader waves carry no MFMAs for them to disturb).
            // 5 x 17 blocks x 4 quads = 340 tasks on 256 threads.
            constexpr int NBX = COLS / 2, NTASK = (HR / 2) * NBX * 4, NT = (NTASK + 255) / 256;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            unsigned t_low[NT], t_out[NT][4];
            int t_by[NT], t_bx[NT];
#pragma unroll
            for (int sl = 0; sl < NT; ++sl) {
                const int tk = sl * 256 + tid, blk = tk >> 2, q = tk & 3;
                t_by[sl] = tk < NTASK ? blk / NBX : -1;
                t_bx[sl] = blk % NBX;
                t_low[sl] = (unsigned)(((blk / NBX) * LOW_C + blk % NBX) * 64 + q * 16);
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int pix = (2 * (blk / NBX) + (o >> 1)) * COLS + 2 * (blk % NBX) + (o & 1);
                    t_out[sl][o] = (unsigned)(pix * 32 + (((q >> 1) ^ ((pix >> 3) & 1)) * 16) + (q & 1) * 8);
                }
            }
            auto interp = [&](const Cur& u) {   // low-resolution stage u.s & 1 -> halo stage u.s & 1
                const unsigned char* lo = lowb + (u.s & 1) * LOW_B;
                unsigned char* h = halo + (u.s & 1) * (NP * PLANE_B);
                const int y0 = u.t.ty * TH - 1, x0 = u.t.tx * 32 - 1;
                // zero padding of the convolution applies to the UPSAMPLED map: only tiles on the image border have halo pixels outside it
                const bool edge = (y0 < 0) | (x0 < 0) | (y0 + HR > p.H) | (x0 + COLS > p.Wd);
#pragma unroll
                for (int sl = 0; sl < NT; ++sl) {
                    if (t_by[sl] < 0) continue;
#ifdef HS_NOINTERP
                    if (p.B > 0) { for (int o = 0; o < 4; ++o) store_planes<NP>(h + t_out[sl][o], PLANE_B, make_float4(1.f, 2.f, 3.f, 4.f)); continue; }   // timing experiment
#endif
                    const float4 v00 = *reinterpret_cast<const float4*>(lo + t_low[sl]), v01 = *reinterpret_cast<const float4*>(lo + t_low[sl] + 64);
                    const float4 v10 = *reinterpret_cast<const float4*>(lo + t_low[sl] + LOW_C * 64), v11 = *reinterpret_cast<const float4*>(lo + t_low[sl] + LOW_C * 64 + 64);
                    f32x2 out[4][2];   // [2 dy + dx][channel pair]
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const f32x2 a00 = c ? f32x2{v00.z, v00.w} : f32x2{v00.x, v00.y}, a01 = c ? f32x2{v01.z, v01.w} : f32x2{v01.x, v01.y};
                        const f32x2 a10 = c ? f32x2{v10.z, v10.w} : f32x2{v10.x, v10.y}, a11 = c ? f32x2{v11.z, v11.w} : f32x2{v11.x, v11.y};
                        // (v_left * gx + v_right * fx): even halo column fx = 0.25, odd 0.75 -- the expression of the per-value form
                        const f32x2 te = a00 * 0.75f + a01 * 0.25f, to = a00 * 0.25f + a01 * 0.75f;   // upper source row
                        const f32x2 be = a10 * 0.75f + a11 * 0.25f, bo = a10 * 0.25f + a11 * 0.75f;   // lower source row
                        out[0][c] = te * 0.75f + be * 0.25f;   // even halo row: fy = 0.25
                        out[1][c] = to * 0.75f + bo * 0.25f;
                        out[2][c] = te * 0.25f + be * 0.75f;   // odd halo row: fy = 0.75
                        out[3][c] = to * 0.25f + bo * 0.75f;
                    }
                    if (edge) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            const int y = y0 + 2 * t_by[sl] + (o >> 1), x = x0 + 2 * t_bx[sl] + (o & 1);
                            const bool inb = ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.Wd);
                            out[o][0] = inb ? out[o][0] : f32x2{0.f, 0.f};
                            out[o][1] = inb ? out[o][1] : f32x2{0.f, 0.f};
                        }
                    }
#pragma unroll
                    for (int o = 0; o < 4; ++o) store_planes<NP>(h + t_out[sl][o], PLANE_B, make_float4(out[o][0].x, out[o][0].y, out[o][1].x, out[o][1].y));
                }
            };
            auto issue_direct = [&](const Cur& u) {
                if (u.k != eo1_k) {
                    eo1_k = u.k;
                    tile_offsets(u.t);
                }
                issue_slice(u.t, u.c);
            };
            // A-step (one phase before): the halo stage
            auto step_a = [&](const Cur& u) {
                if (u.s >= total_slices) return;
                if (u.c < p.nch0) interp(u);
                else store_slice(u.s & 1);
                if (u.c == 0) {
                    if (has_img) store_img(u.k & 1);
                    if (has_lab_l) store_lab(u.k & 1);
                }
            };
            auto step_c = [&](const Cur& u) {
                if (u.s < total_slices && u.c < p.nch0) issue_low(u);
            };
            Cur ca{first, 0, 0, 0};
            issue_low(ca);          // slice 0 is always a source-0 slice
            if (has_img) issue_img(ca.t);
            if (has_lab_l) issue_lab(ca.t);
            store_low(0);
            Cur cb = ca;
            step(cb);               // slice 1
            step_c(cb);
            issue_w(0);
            CP_BARRIER();           // (the consumers run the same extra barrier) low-resolution stage 0 is complete
            step_a(ca);             // interpolate slice 0, store tile 0's image / label halo
            store_w(0);
            if (cb.s < total_slices) {   // slice 1: its B-step without the extras of slice 0's tile being issued twice
                if (cb.c == 0) {
                    if (has_img) issue_img(cb.t);
                    if (has_lab_l) issue_lab(cb.t);
                }
                if (cb.c < p.nch0) store_low(1);
                else issue_direct(cb);
            }
            Cur cc = cb;
            step(cc);               // slice 2
            step_c(cc);
            if (total_groups > 1) issue_w(1);
            CP_BARRIER();
            // from here: ca = slice gs + 1, cb = gs + 2, cc = gs + 3 at the first group of slice gs
            ca = cb;
            cb = cc;
            step(cc);
            int lg = 0;
            HSP_DECL;
            // (stores before issues inside a phase, as in the generic loop below: a wait then only meets loads that are a phase old)
            for (int gg = 0; gg < total_groups; ++gg) {
                const bool more_w = gg + 1 < total_groups;
                if (more_w) store_w((gg + 1) & 1);
                const bool slice_phase = lg < nslices * GPS && lg % GPS == 0;   // first group of a slice
                // Order inside a phase (round 5): (A) everything that CONSUMES the loads of the phase before -- register -> LDS stores; (B) every
                // ISSUE for the phases to come; (C) the interpolation, LDS -> LDS and by far the longest piece, LAST: the loads of (B) are in flight
                // behind it and have landed when the next phase's (A) asks for them.  Round 4 issued after the interpolation, so every phase began
                // with a full memory latency (the loader's "work" time did not move when the interpolation itself got 2x cheaper).
                const bool a_live = slice_phase && ca.s < total_slices;
                if (a_live) {
                    if (ca.c >= p.nch0) store_slice(ca.s & 1);
                    if (ca.c == 0) {
                        if (has_img) store_img(ca.k & 1);
                        if (has_lab_l) store_lab(ca.k & 1);
                    }
                }
                if (slice_phase && cb.s < total_slices && cb.c < p.nch0) store_low(cb.s & 1);
                HSP(0);
                HST(2);
                if (more_w && gg + 2 < total_groups) issue_w(gg + 2);
                if (slice_phase) {
                    if (cb.s < total_slices) {
                        if (cb.c == 0) {
                            if (has_img) issue_img(cb.t);
                            if (has_lab_l) issue_lab(cb.t);
                        }
                        if (cb.c >= p.nch0) issue_direct(cb);
                    }
                    step_c(cc);
                    HSP(2);
                    HST(3);
                    if (a_live && ca.c < p.nch0) interp(ca);
                    HSP(3);
                    HST(4);
                    step(ca);
                    step(cb);
                    step(cc);
                }
                if (esplit && gg >= ngroups_tile) loader_epilogue(lg, false);   // row 1 of the tile before, beside the consumers' work on this one
                if (++lg == ngroups_tile) lg = 0;
                HSP(2);
                HST(5);
                CP_BARRIER();
                HSP(1);
            }
            if (esplit) loader_epilogue(0, true);   // the last tile's
            HSP_FLUSH(8);
            flush_mon();
            return;
        }
        issue_sel(first);
        issue_tile_extras();
        issue_slice(ftile, 0);
        issue_w(0);
        store_slice(0);
        store_tile_extras();
        store_w(0);
        int issued = 1;
        if (total_slices > 1) {
            advance();
            issue_tile_extras();
            issue_slice(ftile, fc);
            issued = 2;
        }
        if (total_groups > 1) issue_w(1);
        CP_BARRIER();
        int gs = 0, lg = 0;   // global slice counter / group index within the tile of the group being multiplied
        // Order inside a phase (round 4): every STORE (registers -> LDS) first, every ISSUE (global -> registers) after them, so that a vmcnt wait
        // before a store can only meet loads that are a whole phase old (the loads sit behind uniform branches; if the compiler cannot count
        // them it waits with vmcnt(0)).  Measured against the old order (weight loads of group gg + 2 issued before the halo store): 1509 vs
        // 1506-1514 images/s -- no difference, the hand-over was not exposed; the order is kept because it cannot be the worse one.
        constexpr bool kStoresFirst = true;
        HSP_DECL;
        for (int gg = 0; gg < total_groups; ++gg) {
            const bool more_w = gg + 1 < total_groups;
            if (more_w) {
                store_w((gg + 1) & 1);                             // weight stage read last during group gg - 1
                if (!kStoresFirst && gg + 2 < total_groups) issue_w(gg + 2);
            }
            bool fetch = false;
            // (round 5, tried and dropped: dealing the halo store / request of a slice over the GPS weight-group phases of the 64-channel kernels --
            //  every phase then stores registers whose neighbours were requested one phase ago, the compiler's vmcnt(0) in front of the store waits
            //  for those too, and the latency is exposed in every phase instead of one in three: 10-35 % slower)
            if (lg < nslices * GPS && lg % GPS == 0) {             // first group of slice gs
                if (gs + 1 < total_slices) {
                    store_slice((gs + 1) & 1);                     // halo stage read last during slice gs - 1
                    store_tile_extras();
                    fetch = issued < total_slices;
                }
                ++gs;
            }
            HST(2);
            if (fetch) {   // (round 6: in front of this phase's weight request -- everything the new tile's requests wait for is then a phase old)
                advance();
                issue_tile_extras();
                issue_slice(ftile, fc);
                ++issued;
            }
            if (kStoresFirst && more_w && gg + 2 < total_groups) issue_w(gg + 2);
            HST(3);
            if (esplit && gg >= ngroups_tile) loader_epilogue(lg, false);   // row 1 of the tile before, beside the consumers' work on this one
            HST(5);
            if (++lg == ngroups_tile) lg = 0;
            HSP(0);
            CP_BARRIER();
            HSP(1);
        }
        if (esplit) loader_epilogue(0, true);   // the last tile's
        HSP_FLUSH(8);
        flush_mon();
        return;
    }

    // ------------------------------------------------------------------ consumers: 4 waves x 2 rows --------------------------------------
#ifdef HS_CONSUMER_IDLE
    if (p.B > 0) {   // timing experiment: the loaders alone
        if constexpr (BILINEAR) CP_BARRIER();
        CP_BARRIER();
        for (int gg = 0; gg < total_groups; ++gg) CP_BARRIER();
        return;
    }
#endif
    bf16x8 fw[2][NP];   // [slot][plane] weight fragments of one sub-step, read from the staged group
    auto ldw = [&](const unsigned char* wg, int sub, int slot) {
#pragma unroll
        for (int s = 0; s < NP; ++s) fw[slot][s] = *reinterpret_cast<const bf16x8*>(wg + (unsigned)(sub * NP + s) * 1024u + wlane);
    };
    // LDS byte offsets of this lane's pixel fragments: rows 2*wave + r, tap (ky, kx)
    unsigned aoff[2][9];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int pix = (2 * wave + r + t / 3) * COLS + lrow + t % 3;
            aoff[r][t] = (unsigned)(pix * 32 + ((kh ^ ((pix >> 3) & 1)) * 16));
        }
    f32x16 acc[2][TN];
    constexpr int FAS = (TN == 4) ? 1 : 2;   // pixel-fragment slots: double-buffered except in the 128-channel kernels (register budget)
    bf16x8 fa[FAS][2][NP];   // [slot][row][plane]
    int dump_k = 0;   // tiles handed over so far (head layers alternate between two hand-over buffers)
    auto finish_tile = [&](int n, int y0, int x0, int cbase) __attribute__((always_inline)) {
        if constexpr (CAN_SPLIT) {
            if (esplit) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        *reinterpret_cast<float4*>(accst + (unsigned)(HEADK ? (dump_k & 1) * ACCST_B : 0) + (unsigned)(((ew * TN + j) * 4 + g4) * 1024) + wlane) =
                            make_float4(acc[1][j][g4 * 4 + 0], acc[1][j][g4 * 4 + 1], acc[1][j][g4 * 4 + 2], acc[1][j][g4 * 4 + 3]);
                ++dump_k;
            }
        }
        epilogue(acc, 0, esplit ? 1 : 2, n, y0, x0, cbase, 0xffffffffu);
    };

    // one (tap, cout block) sub-step: NPROD x 2 MFMAs; smallest terms first
    auto mfma_sub = [&](int aslot, int wslot, int j) {
#pragma unroll
        for (int t = 0; t < NPROD; ++t) {
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[r][j] = mfma_np<NP>(fw[wslot][prod_w<NP>(t)], fa[aslot][r][prod_p<NP>(t)], acc[r][j]);
        }
    };

    // ------------------------------------------------------------------ pipeline ---------------------------------------------------------
    TilePos ctile = first;
    if constexpr (BILINEAR) CP_BARRIER();   // the loaders' hand-over of the first low-resolution stage (their interpolation reads other threads' stores)
    CP_BARRIER();   // slice 0, the first weight group (and the first tile's image / label halo) are in LDS
    HSP_DECL;
    int gs = 0, gg = 0;   // global slice / group counters
    for (int k = 0; k < my_tiles; ++k) {
        const int pass = ctile.pass;
        const int n = ctile.nb, y0 = ctile.ty * TH, x0 = ctile.tx * 32;
        const int cbase = pass * 32 * TN;   // first output channel of this pass
        next_tile(ctile);
        if (has_lab) read_labels(k & 1);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[r][j][e] = 0.f;
        HSP(0);
        for (int c = 0; c < nslices; ++c, ++gs) {
            const unsigned char* hb = halo + (gs & 1) * (NP * PLANE_B);
            auto read_a = [&](int t, int slot) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int s = 0; s < NP; ++s) {
                        bf16x8 v = *reinterpret_cast<const bf16x8*>(hb + s * PLANE_B + aoff[r][t]);
                        if constexpr (PARTIAL) {
                            if (!((pmask[r] >> t) & 1)) v = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
                        }
                        fa[slot][r][s] = v;
                    }
            };
#pragma unroll
            for (int g3 = 0; g3 < GPS; ++g3, ++gg) {   // GPS groups of GT taps, a barrier after each (the weight stage flips)
                const unsigned char* wg = wst + (wres ? (unsigned)(c * GPS + g3) : (unsigned)(gg & 1)) * GROUP_B;
                read_a(g3 * GT, 0);
                ldw(wg, 0, 0);
                hs_static_for<0, GT>([&](auto stc) {
                    constexpr int st = decltype(stc)::value;
                    if (FAS == 2 && st + 1 < GT) read_a(g3 * GT + st + 1, (st + 1) & 1);
                    hs_static_for<0, TN>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        constexpr int sub = st * TN + j;
                        if (sub + 1 < GSUB) ldw(wg, sub + 1, (sub + 1) & 1);
                        mfma_sub(st & (FAS - 1), sub & 1, j);
#ifndef HS_NO_SCHED
                        if constexpr (NP == 3) {
                            // order inside the sub-step (2 x 6 MFMAs): the LDS reads of the NEXT sub-step / tap go out between the first MFMAs,
                            // so they have landed when it starts; left alone the scheduler sinks them behind the last use of the registers
                            // they overwrite, i.e. to the end, and every sub-step begins with an LDS-latency stall
                            constexpr int NR = (sub + 1 < GSUB ? NP : 0) + ((FAS == 2 && j == 0 && st + 1 < GT) ? 2 * NP : 0);
                            constexpr int R0 = (NR + 2) / 3, R1 = (NR - R0 + 1) / 2, R2 = NR - R0 - R1;
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            if constexpr (R0 > 0) __builtin_amdgcn_sched_group_barrier(0x100, R0, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            if constexpr (R1 > 0) __builtin_amdgcn_sched_group_barrier(0x100, R1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            if constexpr (R2 > 0) __builtin_amdgcn_sched_group_barrier(0x100, R2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        } else if constexpr (NP == 2) {
                            // 2 x 3 MFMAs: the next sub-step's / tap's reads between the first and the second pair
                            constexpr int NR = (sub + 1 < GSUB ? NP : 0) + ((FAS == 2 && j == 0 && st + 1 < GT) ? 2 * NP : 0);
                            constexpr int R0 = (NR + 1) / 2, R1 = NR - R0;
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            if constexpr (R0 > 0) __builtin_amdgcn_sched_group_barrier(0x100, R0, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            if constexpr (R1 > 0) __builtin_amdgcn_sched_group_barrier(0x100, R1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#endif
                    });
                    if (FAS == 1 && st + 1 < GT) read_a(g3 * GT + st + 1, 0);
                });
#ifdef HS_NOEPI
                if (g3 == GPS - 1 && c + 1 == nslices && !has_img && p.B < 0) finish_tile(n, y0, x0, cbase);   // timing experiment: never true, keeps the accumulators alive
#else
                HSP(1);
                HST(2);
                if (g3 == GPS - 1 && c + 1 == nslices && !has_img) { finish_tile(n, y0, x0, cbase); HSP(3); HST(4); }
#endif
                CP_BARRIER();
                HSP(4);
            }
        }
        // ---- image block: K = 9 taps x 4 channels (+12 zero) = 3 steps, lane half kh covers taps 4s+2kh, +1 ----
        if (has_img) {
            const unsigned char* ib = imgh + (k & 1) * (NP * IPLANE_B);
            const unsigned char* wg = wst + (wres ? (unsigned)(nslices * GPS) : (unsigned)(gg & 1)) * GROUP_B;
            ldw(wg, 0, 0);
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int s = 0; s < NP; ++s) {
                        uint2 lo2 = make_uint2(0u, 0u), hi2 = make_uint2(0u, 0u);
                        const int ta = 4 * s3, tb = 4 * s3 + 2;   // first tap of the lower / upper lane half (compile-time)
                        const int t0 = kh ? tb : ta;
                        if (t0 < 9) {
                            const unsigned o0 = ((kh ? aoff[r][tb < 9 ? tb : 0] : aoff[r][ta < 9 ? ta : 0]) & ~31u) >> 2;   // pixel * 8
                            lo2 = *reinterpret_cast<const uint2*>(ib + s * IPLANE_B + o0);
                            if constexpr (PARTIAL) {
                                if (!((pmask[r] >> t0) & 1)) lo2 = make_uint2(0u, 0u);
                            }
                        }
                        if (t0 + 1 < 9) {
                            const unsigned o1 = ((kh ? aoff[r][tb + 1 < 9 ? tb + 1 : 0] : aoff[r][ta + 1 < 9 ? ta + 1 : 0]) & ~31u) >> 2;
                            hi2 = *reinterpret_cast<const uint2*>(ib + s * IPLANE_B + o1);
                            if constexpr (PARTIAL) {
                                if (!((pmask[r] >> (t0 + 1)) & 1)) hi2 = make_uint2(0u, 0u);
                            }
                        }
                        fa[s3 & 1][r][s] = __builtin_bit_cast(bf16x8, make_uint4(lo2.x, lo2.y, hi2.x, hi2.y));
                    }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int sub = s3 * TN + j;
                    if (sub + 1 < 3 * TN) ldw(wg, sub + 1, (sub + 1) & 1);
                    mfma_sub(s3 & (FAS - 1), sub & 1, j);
                }
            }
            HSP(2);
            HST(3);
#ifdef HS_NOEPI
            if (p.B < 0)
#endif
            finish_tile(n, y0, x0, cbase);
            HSP(3);
            HST(4);
            ++gg;
            CP_BARRIER();
            HSP(4);
        }
    }
    HSP_FLUSH(0);
    flush_mon();
}


template <int TN, int NP, int MODE>
int launch_hsplit(HSplitK k, hipStream_t st) {
    k.tiles_y = (k.H + TH - 1) / TH;
    k.tiles_x = (k.Wd + 31) / 32;
    k.passes = (k.Cout + 32 * TN - 1) / (32 * TN);
    k.tiles_per_pass = k.B * k.tiles_y * k.tiles_x;
    k.ntiles = k.passes * k.tiles_per_pass;
    // halo 65 KB + image halo 16 KB + labels 1.4 KB + weight groups 54 / 36 KB (three planes): one block of 8 waves per CU
    size_t lds = (size_t)2 * NP * PLANE_B + (size_t)2 * NP * IPLANE_B + (size_t)2 * HP * 2 + (size_t)2 * hs_group_taps(TN, NP) * TN * NP * 1024 +
                 (size_t)2 * NP * 1024;   // + the fused head's weights
    if ((MODE & HS_HEADK) && NP <= 2) lds += (size_t)3 * TN * NP * 1024;   // + room for the image block's group behind two slice groups (resident weights)
    if (MODE & HS_BILINEAR) lds += (size_t)2 * ((HR / 2 + 1) * (COLS / 2 + 1)) * 64;   // + two low-resolution stages of source 0
    lds += (size_t)2 * 512 * 4;                                                         // + the per-channel normalisation table (TAB_C)
    if (NP <= 2 && TN == 1) lds += (size_t)4 * TN * 4 * 1024 * ((MODE & HS_HEADK) ? 2 : 1);   // + the accumulator hand-over(s) of the epilogue split
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_hsplit_kernel<TN, NP, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int grid = std::min(cp::persistent_blocks(), k.ntiles);
    CP_LAUNCH((conv_hsplit_kernel<TN, NP, MODE>), dim3(grid), dim3(512), lds, st, k);
    return cp::check_launch("cp_conv2d_fwd_split");
}

// output channels per pass: 32 or 64.  (A 128-channel pass -- TN = 4, 128 accumulator registers -- was compiled and spills ~75 registers at the
// 256-register budget of two waves per SIMD; wider layers simply take more passes over the tile list, re-staging the halo each time.)
inline int split_tn(int cout) { return cout <= 32 ? 1 : 2; }

int split_fragments(int cout, int num_sources, const int* channels) {
    const int tn = split_tn(cout), passes = (cout + 32 * tn - 1) / (32 * tn);
    int steps = 0;
    for (int s = 0; s < num_sources; ++s) steps += (channels[s] == 4) ? 3 : (channels[s] / 16) * 9;
    return steps * tn * passes;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------------------------
extern "C" int cp_conv_split_applicable(const cp_conv_desc* d) {
    if (!d || d->struct_size != (uint32_t)sizeof(cp_conv_desc)) return 0;   // another revision of the header: not ours to read
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->dilation != 1 || d->pad != 1) return 0;
    if (d->cout > 512 || d->cout % 4 != 0 || d->group_rows) return 0;
    if (d->head_out && (d->cout != 32 || !d->head_weights || d->head_cout < 1 || d->head_cout > 32)) return 0;
    if ((d->out_raw && d->out_raw_ld % 4) || (d->out_act && d->out_act_ld % 4) || (d->residual && d->residual_ld % 4)) return 0;
    if ((((uintptr_t)d->out_raw) | ((uintptr_t)d->out_act) | ((uintptr_t)d->residual)) & 15) return 0;
    if (d->num_sources < 1 || d->num_sources > 2) return 0;
    for (int s = 0; s < d->num_sources; ++s) {
        const cp_conv_source& in = d->src[s];
        if (in.pre_scale || in.pre_shift) return 0;
        if (s == 0 ? (in.mode != CP_SRC_DIRECT && in.mode != CP_SRC_BILINEAR_X2 && in.mode != CP_SRC_NEAREST_SEL) : in.mode != CP_SRC_DIRECT) return 0;
        if (in.mode != CP_SRC_DIRECT && ((d->in_h | d->in_w) & 1)) return 0;
        if (in.channels == 4) {
            if (s != d->num_sources - 1 || s == 0 || in.ld != 4 || d->cout > 32) return 0;   // the image source comes last, after a 16-multiple source (32-channel layers)
        } else if (in.channels % 16 != 0 || in.ld % 4 != 0 || (((uintptr_t)in.data) & 15)) {
            return 0;
        }
    }
    if (d->src[0].channels == 4) return 0;
    if (d->tap_label && d->epi_label && d->tap_label != d->epi_label) return 0;
    if (!d->tap_label && d->row_scale) return 0;
    if (d->src[0].mode == CP_SRC_BILINEAR_X2 && d->tap_label) return 0;   // the instantiated combinations: those the network uses
    if (d->src[0].mode == CP_SRC_NEAREST_SEL && (!d->tap_label || !d->src[0].sel)) return 0;
    return 1;
}

extern "C" int cp_conv_split_weight_floats(int cout, int num_sources, const int* channels) {
    return split_fragments(cout, num_sources, channels) * 64 * 8;
}

extern "C" size_t cp_conv_split_weight_bytes(int cout, int num_sources, const int* channels, int planes) {
    return (size_t)split_fragments(cout, num_sources, channels) * (size_t)(planes & 15) * 1024;   // CP_PLANES_F16X2 = 0x12: two planes
}

// HOST: Keras-layout kernel -> the fp32 image of the fragment stream, [step][cout block][64 lanes][8 k]:
//   16-channel slice c of a source, tap t: lane (i = l & 31, kh = l >> 5), element e  <-  W[co = 32*j + i][channel 16*c + 8*kh + e] at tap t
//   image source, step s3:                                                             <-  W[co][channel e & 3] at tap 4*s3 + 2*kh + (e >> 2)
// (channels / taps / output channels that do not exist: zero).  cp_conv_split_weights_f32 turns this image into the bf16 planes on the
// device, so a training step re-packs with one gather + one split launch.
extern "C" int cp_conv_pack_weights_split_host(const float* w, int layout, int cout, int num_sources, const int* channels, const int* real_channels,
                                               float* dst) {
    CP_REQUIRE(w && dst && cout > 0 && cout <= 512 && num_sources >= 1 && num_sources <= 2, "cp_conv_pack_weights_split_host: bad arguments");
    const int tn = split_tn(cout), passes = (cout + 32 * tn - 1) / (32 * tn);
    int cin = 0;
    for (int s = 0; s < num_sources; ++s) cin += real_channels[s];
    const int total = cp_conv_split_weight_floats(cout, num_sources, channels);
    for (int i = 0; i < total; ++i) dst[i] = 0.f;
    auto src_index = [&](int ci, int t, int co) {
        const int ky = t / 3, kx = t % 3;
        return (layout == 0) ? ((((size_t)ky * 3 + kx) * cin + ci) * cout + co) : ((((size_t)ci * 3 + ky) * 3 + kx) * cout + co);
    };
    int steps_pass = 0;
    for (int s = 0; s < num_sources; ++s) steps_pass += (channels[s] == 4) ? 3 : (channels[s] / 16) * 9;
    for (int ps = 0; ps < passes; ++ps) {   // one complete fragment stream per pass of 32 * tn output channels
        size_t step = (size_t)ps * steps_pass;
        int cbase = 0;
        for (int s = 0; s < num_sources; ++s) {
            const int C = channels[s], Cr = real_channels[s];
            if (C == 4) {
                for (int s3 = 0; s3 < 3; ++s3, ++step)
                    for (int j = 0; j < tn; ++j)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                const int co = 32 * (ps * tn + j) + (l & 31), t = 4 * s3 + 2 * (l >> 5) + (e >> 2), ch = e & 3;
                                if (co < cout && t < 9 && ch < Cr) dst[((step * tn + j) * 64 + l) * 8 + e] = w[src_index(cbase + ch, t, co)];
                            }
            } else {
                for (int c = 0; c < C / 16; ++c)
                    for (int t = 0; t < 9; ++t, ++step)
                        for (int j = 0; j < tn; ++j)
                            for (int l = 0; l < 64; ++l)
                                for (int e = 0; e < 8; ++e) {
                                    const int co = 32 * (ps * tn + j) + (l & 31), ch = 16 * c + 8 * (l >> 5) + e;
                                    if (co < cout && ch < Cr) dst[((step * tn + j) * 64 + l) * 8 + e] = w[src_index(cbase + ch, t, co)];
                                }
            }
            cbase += Cr;
        }
    }
    return CP_OK;
}

extern "C" int cp_conv_split_weights_scaled_f32(const float* packed, long long floats, int planes, float scale, void* out, void* stream) {
    CP_REQUIRE(packed && out && floats > 0 && floats % 512 == 0 && (planes == 1 || planes == 3 || planes == CP_PLANES_F16X2), "cp_conv_split_weights_f32: bad arguments");
    CP_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)out & 15) == 0, "cp_conv_split_weights_f32: pointers must be 16-byte aligned");
    CP_REQUIRE(planes == CP_PLANES_F16X2 ? (scale > 0.f && std::isfinite(scale)) : scale == 1.f, "cp_conv_split_weights_scaled_f32: a scale other than 1 goes with CP_PLANES_F16X2 only");
    const long long nfrag = floats / 512;
    const int blocks = (int)std::min<long long>((nfrag * 64 + 255) / 256, 4096);
    if (planes == 3)
        CP_LAUNCH(hsplit_weights_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, packed, nfrag, 1.f, reinterpret_cast<unsigned char*>(out));
    else if (planes == CP_PLANES_F16X2)
        CP_LAUNCH(hsplit_weights_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, packed, nfrag, scale, reinterpret_cast<unsigned char*>(out));
    else
        CP_LAUNCH(hsplit_weights_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, packed, nfrag, 1.f, reinterpret_cast<unsigned char*>(out));
    return cp::check_launch("cp_conv_split_weights_f32");
}

extern "C" int cp_conv_split_weights_f32(const float* packed, long long floats, int planes, void* out, void* stream) {
    CP_REQUIRE(planes == 1 || planes == 3, "cp_conv_split_weights_f32: planes must be 1 or 3 (CP_PLANES_F16X2 needs cp_conv_split_weights_scaled_f32)");
    return cp_conv_split_weights_scaled_f32(packed, floats, planes, 1.f, out, stream);
}

// HOST: [1][1][32][head_cout] (HWIO) 1x1 kernel -> the fp32 image of the fused head's two fragments, [2 steps][64 lanes][8]:
//   step m, lane (q = l & 31, kh = l >> 5), element e  <-  Wh[channel 8*(2m + (e >> 2)) + 4*kh + (e & 3)][q]   (the K order the epilogue uses)
// 1024 floats; cp_conv_split_weights_f32 turns it into the bf16 planes.
extern "C" int cp_conv_pack_head_split_host(const float* w, int head_cout, float* dst) {
    CP_REQUIRE(w && dst && head_cout >= 1 && head_cout <= 32, "cp_conv_pack_head_split_host: bad arguments");
    for (int m = 0; m < 2; ++m)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
                const int q = l & 31, c = 8 * (2 * m + (e >> 2)) + 4 * (l >> 5) + (e & 3);
                dst[(m * 64 + l) * 8 + e] = q < head_cout ? w[c * head_cout + q] : 0.f;
            }
    return CP_OK;
}

extern "C" int cp_conv2d_fwd_split(const cp_conv_desc* d, const void* weights_split, const void* head_weights_split, int planes, void* stream) {
    CP_REQUIRE(planes == 1 || planes == 3, "cp_conv2d_fwd_split: planes must be 1 or 3 (CP_PLANES_F16X2 needs cp_conv2d_fwd_split_scaled)");
    return cp_conv2d_fwd_split_scaled(d, weights_split, head_weights_split, planes, 1.f, 1.f, stream);
}

extern "C" int cp_conv2d_fwd_split_scaled(const cp_conv_desc* d, const void* weights_split, const void* head_weights_split, int planes, float w_descale,
                                          float head_descale, void* stream) {
    CP_REQUIRE_DESC(d, "cp_conv2d_fwd_split");
    CP_REQUIRE(weights_split && (planes == 1 || planes == 3 || planes == CP_PLANES_F16X2), "cp_conv2d_fwd_split: bad arguments");
    CP_REQUIRE(planes == CP_PLANES_F16X2 ? (w_descale > 0.f && std::isfinite(w_descale) && head_descale > 0.f && std::isfinite(head_descale))
                                         : (w_descale == 1.f && head_descale == 1.f),
               "cp_conv2d_fwd_split_scaled: descale factors other than 1 go with CP_PLANES_F16X2 only");
    CP_REQUIRE(cp_conv_split_applicable(d), "cp_conv2d_fwd_split: this convolution is outside the kernel's range (3x3 / stride 1 / pad 1, cout <= 512 and a multiple of 4, sources "
                                            "of 16-multiple channels + optional trailing 4-channel source; source 0 direct, bilinear x2 or guided x2)");
    CP_REQUIRE(d->out_raw || d->out_act || d->head_out, "cp_conv2d_fwd_split: no output");
    CP_REQUIRE(!d->head_out || head_weights_split, "cp_conv2d_fwd_split: a fused head needs its split weights");
    HSplitK k{};
    int nch = 0;
    for (int s = 0; s < d->num_sources; ++s) {
        const cp_conv_source& in = d->src[s];
        const int Hs = (in.mode == CP_SRC_DIRECT) ? d->in_h : d->in_h / 2, Ws = (in.mode == CP_SRC_DIRECT) ? d->in_w : d->in_w / 2;
        const long long nbytes = (long long)d->batch * Hs * Ws * in.ld * 4;
        CP_REQUIRE(nbytes < (1LL << 31), "cp_conv2d_fwd_split: source %d spans %lld bytes; 32-bit range-checked addressing needs < 2 GiB", s, nbytes);
        if (in.channels == 4) {
            k.img = in.data;
            k.img_bytes = (unsigned)nbytes;
            continue;
        }
        k.s[s].data = in.data;
        k.s[s].sel = in.sel;
        k.s[s].C = in.channels;
        k.s[s].ld = in.ld;
        k.s[s].Hs = Hs;
        k.s[s].Ws = Ws;
        k.s[s].bytes = (unsigned)nbytes;
        if (s == 0) k.nch0 = in.channels / 16;
        nch += in.channels / 16;
    }
    k.nch = nch;
    int chans[2] = {d->src[0].channels, d->num_sources > 1 ? d->src[1].channels : 0};
    k.W = reinterpret_cast<const unsigned char*>(weights_split);
    k.w_bytes = (unsigned)cp_conv_split_weight_bytes(d->cout, d->num_sources, chans, planes);
    k.B = d->batch; k.H = d->in_h; k.Wd = d->in_w; k.Cout = d->cout;
    const int max_ld = std::max(std::max(d->out_raw ? d->out_raw_ld : 0, d->out_act ? d->out_act_ld : 0), d->head_out ? d->head_out_ld : 0);
    const long long out_bytes = (long long)d->batch * d->in_h * d->in_w * max_ld * 4;
    CP_REQUIRE(out_bytes < (1LL << 32), "cp_conv2d_fwd_split: output spans >= 4 GiB");
    k.label = d->tap_label ? d->tap_label : d->epi_label;
    k.lab_bytes = (unsigned)((size_t)d->batch * d->in_h * d->in_w);
    k.residual = d->residual; k.res_ld = d->residual_ld;
    k.scale = d->scale; k.shift = d->shift; k.clade = d->epi_label != nullptr; k.act = d->act;
    k.norm = d->row_scale != nullptr;
    k.out_raw = d->out_raw; k.raw_ld = d->out_raw_ld; k.out_act = d->out_act; k.act_ld = d->out_act_ld;
    k.head_w = d->head_out ? reinterpret_cast<const unsigned char*>(head_weights_split) : nullptr;
    k.head_out = d->head_out; k.head_cout = d->head_cout; k.head_ld = d->head_out_ld;
    k.head_lab = d->head_out ? d->head_label_out : nullptr; k.head_lab_classes = d->head_label_classes;
    k.descale = w_descale; k.head_descale = head_descale;
    k.mon = planes == CP_PLANES_F16X2 ? cp::f16x2_monitor() : nullptr;
    {   // epilogue split: on by default (CASAPOSE_HS_EPI_SPLIT=0 keeps both rows on the consumer waves: A/B measurements)
        static const int split_env = getenv("CASAPOSE_HS_EPI_SPLIT") ? atoi(getenv("CASAPOSE_HS_EPI_SPLIT")) : 1;
        k.epi_split = split_env;
    }
    {
        static const int wres_env = getenv("CASAPOSE_HS_WRES") ? atoi(getenv("CASAPOSE_HS_WRES")) : 1;
        k.w_res = wres_env;
    }
    const int np = planes & 15;
    const int tn = split_tn(d->cout);
    const bool headk = tn == 1 && d->head_out && !d->residual && !d->out_raw && !d->out_act && d->scale && d->act == CP_ACT_LEAKY01 && d->cout == 32;
    if (d->head_out && d->head_prefix) {   // whole output records: the head-only instantiations carry the copy
        CP_REQUIRE(headk, "cp_conv2d_fwd_split: head_prefix needs a head-only layer (32 channels, table, leaky ReLU, no other output)");
        CP_REQUIRE(d->head_prefix_n >= 8 && d->head_prefix_n <= 12 && d->head_prefix_ld >= d->head_prefix_n && d->head_out_ld >= d->head_prefix_n + d->head_cout &&
                   ((uintptr_t)d->head_out & 15) == 0 && d->head_out_ld % 4 == 0,
                   "cp_conv2d_fwd_split: 8 <= head_prefix_n <= 12, prefix rows of >= head_prefix_n floats, 16-byte aligned records of >= head_prefix_n + head_cout floats");
        k.residual = d->head_prefix;
        k.res_ld = d->head_prefix_ld;
        k.head_pre_n = d->head_prefix_n;
    }
    const int mode = (d->tap_label ? HS_PARTIAL : 0) | (d->src[0].mode == CP_SRC_BILINEAR_X2 ? HS_BILINEAR : 0) | (d->src[0].mode == CP_SRC_NEAREST_SEL ? HS_SEL : 0) |
                     (headk ? HS_HEADK : 0) | (k.head_pre_n ? HS_PREFIX : 0);
    hipStream_t st = (hipStream_t)stream;
#define CP_HS(TN_, NP_, M_) if (tn == TN_ && np == NP_ && mode == (M_)) return launch_hsplit<TN_, NP_, (M_)>(k, st);
#define CP_HS4(TN_, NP_) CP_HS(TN_, NP_, 0) CP_HS(TN_, NP_, HS_BILINEAR) CP_HS(TN_, NP_, HS_PARTIAL) CP_HS(TN_, NP_, HS_PARTIAL | HS_SEL)
#define CP_HSK(NP_) CP_HS(1, NP_, HS_HEADK) CP_HS(1, NP_, HS_HEADK | HS_BILINEAR) CP_HS(1, NP_, HS_HEADK | HS_PARTIAL) CP_HS(1, NP_, HS_HEADK | HS_PARTIAL | HS_SEL)
    CP_HS4(1, 3)
    CP_HS4(2, 3)
    CP_HS4(1, 2)
    CP_HS4(2, 2)
    CP_HS4(1, 1)
    CP_HS4(2, 1)
    CP_HSK(3)
    CP_HSK(2)
    CP_HSK(1)
    // whole output records: the last head of the network sits behind a partial convolution (decoder 2, casapose.py:107-142) -- those two forms only
    CP_HS(1, 3, HS_HEADK | HS_PREFIX | HS_PARTIAL) CP_HS(1, 3, HS_HEADK | HS_PREFIX | HS_PARTIAL | HS_SEL)
    CP_HS(1, 2, HS_HEADK | HS_PREFIX | HS_PARTIAL) CP_HS(1, 2, HS_HEADK | HS_PREFIX | HS_PARTIAL | HS_SEL)
#undef CP_HSK
#undef CP_HS4
#undef CP_HS
    cp::set_error("cp_conv2d_fwd_split: operand mode %d is not instantiated", mode);
    return CP_ERR_INVALID;
}

#ifdef HS_TRACE
extern "C" int cp_hs_trace_read(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(hs_trace), sizeof(unsigned long long) * 2 * 2048) == hipSuccess ? 0 : -1;
}
#endif

#ifdef HS_PROFILE
extern "C" int cp_hs_profile_read(unsigned long long* host_out, int reset) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(hs_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(hs_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
