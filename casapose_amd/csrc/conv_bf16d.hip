// Direct 3x3 / stride-1 convolution with bf16 OPERANDS for the deep layers (BASELINE.json configs[2] "bf16 convs"; round-2 verdict item 7):
// dilation D = pad in {1, 2, 4}, up to two concatenated NHWC fp32 sources (channels multiples of 16), 128 output channels per pass, fp32
// accumulation on v_mfma_f32_32x32x16_bf16, fp32 tensors in HBM (operands are rounded to bf16 -- round to nearest even -- while they are staged).
//
// Why a kernel of its own: conv_hsplit.hip with one plane walks a wide layer in passes of 64 output channels and
// re-stages the input halo for each of them -- 19 B/clk per CU from L2 at 36 MFMAs per slice -- and its 2 rows x 64 channels per wave need
// 128 B/clk of LDS reads.  Here
//   * a block owns 512 output pixels (16 x 32 or 8 x 64) x 128 output channels: the halo of a 16-channel slice is staged ONCE for four times
//     the MFMA work (144 MFMAs per wave and slice), 960 / 1152 halo pixels for 512 outputs even at dilation 4;
//   * a wave owns 128 pixels x 128 channels = 4 x 4 accumulator tiles (256 registers): per tap 4 pixel + 4 weight fragments for 16 MFMAs,
//     64 B/clk of LDS reads per CU;
//   * that accumulator block leaves room for ONE wave per SIMD, so the four waves of a block both load and multiply: the global loads of
//     slice s + 1 (halo elements and weight fragments, ~100 registers) are issued before the MFMAs of slice s and converted / stored into
//     the other LDS stage behind its last taps; one barrier per slice.
// Weights: the fragment stream of conv_hsplit.hip for 64-channel passes ([pass][slice][tap][2 cout blocks][64 lanes][8 bf16], one plane:
// cp_conv_pack_weights_split_host + cp_conv_split_weights_f32(planes = 1)); a 128-channel pass here reads the streams of passes 2q, 2q + 1.
// Accumulators are transposed (MFMA A = weights, B = pixels: lane = pixel, four consecutive channels in four consecutive registers).
#include "common.h"

#include <algorithm>
#include <type_traits>

// hand-counted waits (see wino_gemm_wide.hip; checked in the ISA by tests/test_asm_invariants.py; -DCP_SAFE_WAITS: vmcnt(0))
#ifdef CP_SAFE_WAITS
#define CP_WAIT_VM_N(N) asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define CP_WAIT_VM_N(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct DSrc {
    const float* data;
    int C, ld;
    unsigned bytes;
};

struct DeepK {
    DSrc s[2];
    const unsigned char* W;   // conv_hsplit.hip's one-plane fragment stream for 64-channel passes
    unsigned w_bytes;
    int B, H, Wd, Cout;
    int nch0, nch;            // 16-channel slices of source 0 / of both sources
    int tiles_y, tiles_x, ntiles, passes, tiles_per_pass;
    const float* residual;
    int res_ld;
    const float* scale;       // per-channel affine of the activated output (or null)
    const float* shift;
    int act;
    float* out_raw;
    int raw_ld;
    float* out_act;
    int act_ld;
};

#define CP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ unsigned pack_hi16(unsigned a_lo, unsigned b_hi) { return __builtin_amdgcn_perm(b_hi, a_lo, 0x07060302u); }

// round to nearest even, two v_cvt_pk_bf16_f32 (gfx950)
__device__ __forceinline__ uint2 round4(const float4 v) {
    const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
    return make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2)), __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2)));
}

template <int I, int N, typename F>
__device__ __forceinline__ void deep_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        deep_static_for<I + 1, N>(f);
    }
}

template <int D, int TW>
__global__ __launch_bounds__(256, 1) void conv_bf16d_kernel(const DeepK p) {
    constexpr int TH = 512 / TW;                 // output rows of a tile
    constexpr int HW = TW + 2 * D, HH = TH + 2 * D;
    constexpr int HP = HH * HW;                  // halo pixels
    constexpr int PLANE_B = HP * 16;             // one k-half of a stage: [pixel][8 bf16] -- 16 consecutive pixels are 256 contiguous bytes, so a
                                                 // ds_read_b128 lane group is conflict-free at every tap offset and a tap is an IMMEDIATE offset
    constexpr int HALO_B = 2 * PLANE_B;          // one stage: [k-half][pixel][8 bf16]
    constexpr int WSL_B = 9 * 4 * 1024;          // one stage of weights: [tap][4 cout blocks][64 lanes][16 B]
    constexpr int RPW = TH / 4;                  // output rows per wave
    constexpr int FPR = TW / 32;                 // pixel fragments per row
    constexpr unsigned OOB = 0x80000000u;
    static_assert(RPW * FPR == 4, "four pixel fragments per wave");

    // two separate LDS objects: the weights arrive by LDS-DMA, and with ONE object the compiler orders every ds_write of the halo behind the
    // outstanding DMAs (s_waitcnt vmcnt(5..9) in front of each store, i.e. it drains the halo prefetch as well)
    __shared__ __attribute__((aligned(16))) unsigned char halo[2 * HALO_B];   // [2][k-half][pixel][8 bf16]
    __shared__ __attribute__((aligned(16))) unsigned char wst[2 * WSL_B];     // [2][tap][4 cout blocks][64 lanes][16 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, kh = lane >> 5;

    const int bid = cp::xcd_remap(blockIdx.x, gridDim.x);
    const int g = (int)gridDim.x;
    const int my_tiles = (p.ntiles - bid + g - 1) / g;
    if (my_tiles <= 0) return;
    const int nsl = p.nch;
    const int total = my_tiles * nsl;            // (tile, slice) steps of this block
    const unsigned pass_w_bytes = (unsigned)(p.nch * 9 * 2) * 1024u;   // one 64-channel pass of the stream

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.s[0].data, 0, p.s[0].bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.s[1].data ? p.s[1].data : p.s[0].data), 0, p.s[1].data ? p.s[1].bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.w_bytes, 0x00020000);

    // ---- staging side ---------------------------------------------------------------------------------------------------------------------
    // Halo: ITEM hy = one halo ROW; thread (slot = tid >> 2, q = tid & 3) loads channel quad q of halo pixel (hy, slot) -- HW <= 64 slots, the
    // rest idle.  The row part of an address is SCALAR (image row, bounds, slice channel offset), the lane part ((slot * ld + 4 q) * 4 bytes, the
    // column bound) is computed once per tile / source: an item costs a select, the load, two v_cvt_pk_bf16_f32 and a ds_write_b64 with an
    // immediate row offset.  (The first version enumerated float4 elements linearly: ~45 instructions each for the pixel's row / column / bounds /
    // offset, ~1000 per slice against 144 MFMAs of 32 cycles -- the counters showed the waves issuing instructions 45 % of the time and the
    // matrix pipe busy 35 %.)
    // Weights: already bf16 in the layout the consumers read, so they go global -> LDS directly (buffer_load_dwordx4 ... lds, 1 KB per wave
    // instruction, nine per wave and slice, scalar addresses): no registers, no VALU, no ds_write.
    static_assert(HW <= 64, "one halo row per item needs TW + 2 D <= 64");
    constexpr int NIT = HH;                      // items = halo rows
    // (threads whose slot lies beyond the row repeat its last pixel: same load, same LDS bytes, no branch in the tap's instruction stream)
    const int slot = (tid >> 2) < HW ? (tid >> 2) : HW - 1, q = tid & 3;
    const unsigned lds_lane = (unsigned)((q >> 1) * PLANE_B + slot * 16 + (q & 1) * 8);
    float4 lv[NIT];

    auto tile_of = [&](int k, int& pass, int& n, int& ty, int& tx) {
        int t = bid + k * g;
        pass = t / p.tiles_per_pass;
        t -= pass * p.tiles_per_pass;
        tx = t % p.tiles_x;
        t /= p.tiles_x;
        ty = t % p.tiles_y;
        n = t / p.tiles_y;
    };
    struct StepPos { int pass, n, ty, tx, c, si, cs; unsigned voff; };
    auto step_pos = [&](int step) {   // a step past the end repeats step 0 (its loads land in registers / a stage nobody reads)
        StepPos sp;
        const int st = step < total ? step : 0;
        const int k = st / nsl;
        sp.c = st - k * nsl;
        tile_of(k, sp.pass, sp.n, sp.ty, sp.tx);
        sp.si = sp.c >= p.nch0 ? 1 : 0;
        sp.cs = (sp.c - (sp.si ? p.nch0 : 0)) * 64;
        const int x = sp.tx * TW - D + slot;
        const unsigned ld = (unsigned)(sp.si ? p.s[1].ld : p.s[0].ld);
        sp.voff = ((unsigned)x < (unsigned)p.Wd) ? ((unsigned)slot * ld + (unsigned)(q * 4)) * 4u : OOB;   // lane part of every row's address
        return sp;
    };
    auto issue_item = [&](auto ic, const StepPos& sp) {
        constexpr int hy = decltype(ic)::value;
        const int y = sp.ty * TH - D + hy;                                  // scalar
        const unsigned rowoob = ((unsigned)y < (unsigned)p.H) ? 0u : OOB;   // arithmetic, not a branch: a tap stays ONE basic block, so its
                                                                            // instructions can be dealt between the MFMAs (sched_group_barrier)
        const int ld = sp.si ? p.s[1].ld : p.s[0].ld;
        const int rowbase = (((sp.n * p.H + y) * p.Wd + sp.tx * TW - D) * ld) * 4 + sp.cs;   // may be "negative" at the borders: added to voff mod 2^32
        const unsigned vo = (sp.voff + (unsigned)rowbase) | ((sp.voff | rowoob) & OOB);   // bit 31 set = beyond every buffer (< 2 GiB): reads zero
        lv[hy] = sp.si ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)vo, 0, 0))
                       : __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)vo, 0, 0));
    };
    auto store_item = [&](auto ic, unsigned char* hst) {   // hst = stage base + this lane's part
        constexpr int hy = decltype(ic)::value;
        *reinterpret_cast<uint2*>(hst + hy * HW * 16) = round4(lv[hy]);
    };
    auto weights_dma = [&](const StepPos& sp, int stage) {   // fragments wave, wave + 4, ... of the slice's 36: (tap, cout block) = (f >> 2, f & 3)
        const unsigned sbase = (unsigned)(2 * sp.pass) * pass_w_bytes + (unsigned)(sp.c * 18) * 1024u;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int f = wave + 4 * i, tap = f >> 2, j4 = f & 3;
            const unsigned src = sbase + (unsigned)(j4 >> 1) * pass_w_bytes + (unsigned)((tap * 2 + (j4 & 1)) * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(wst + stage * WSL_B + f * 1024), 16, (int)(lane * 16), (int)src, 0, 0);
        }
    };
    constexpr int NITEM = NIT;
    constexpr int IPT = (NITEM + 8) / 9;   // items per tap

    // ---- multiplying side ------------------------------------------------------------------------------------------------------------------
    // pixel fragment f of this wave: output row wave*RPW + f / FPR, columns (f % FPR) * 32 + lrow; halo pixel index of its tap (0, 0)
    unsigned pbase[4];   // LDS byte offset of the fragment's tap (0, 0) inside a stage
#pragma unroll
    for (int f = 0; f < 4; ++f) pbase[f] = (unsigned)(kh * PLANE_B + ((wave * RPW + f / FPR) * HW + (f % FPR) * 32 + lrow) * 16);
    const unsigned wlane = (unsigned)lane * 16u;
    f32x16 acc[4][4];   // [pixel fragment][cout block]

    const unsigned npix = (unsigned)(p.B * p.H * p.Wd);
    const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual ? (const void*)p.residual : (const void*)p.W), 0,
                                                                            p.residual ? npix * (unsigned)p.res_ld * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_raw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out_raw ? (void*)p.out_raw : (void*)p.W), 0,
                                                                            p.out_raw ? npix * (unsigned)p.raw_ld * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_act = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out_act ? (void*)p.out_act : (void*)p.W), 0,
                                                                            p.out_act ? npix * (unsigned)p.act_ld * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_sc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.scale ? (const void*)p.scale : (const void*)p.W), 0, p.scale ? (unsigned)p.Cout * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_sh = __builtin_amdgcn_make_buffer_rsrc((void*)(p.scale ? (const void*)p.shift : (const void*)p.W), 0, p.scale ? (unsigned)p.Cout * 4u : 0u, 0x00020000);

    // Straight-line code (absent operands are out-of-range buffer accesses: loads return zero, stores are dropped; the activation is a pair of
    // selects) with a scheduling barrier per cout block: with branches in it the compiler moved the whole 256-register accumulator block into
    // arch registers at the top of the epilogue and spilled everything else around it.
    const float has_sc = p.scale ? 1.f : 0.f;
    auto epilogue = [&](int pass, int n, int ty, int tx) {
        const int cbase = pass * 128;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int y = ty * TH + wave * RPW + f / FPR, x = tx * TW + (f % FPR) * 32 + lrow;
            const bool pok = y < p.H && x < p.Wd;
            const unsigned pix = (unsigned)((n * p.H + y) * p.Wd + x);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int ch = cbase + j * 32 + g4 * 8 + kh * 4;
                    const bool ok = pok && ch < p.Cout;
                    const unsigned tab = ch < p.Cout ? (unsigned)ch * 4u : OOB;
                    const float4 r = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_res, (int)(ok ? (pix * (unsigned)p.res_ld + (unsigned)ch) * 4u : OOB), 0, 0));
                    const float4 sc = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_sc, (int)tab, 0, 0));
                    const float4 sh = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_sh, (int)tab, 0, 0));
                    float v[4] = {acc[f][j][g4 * 4 + 0] + r.x, acc[f][j][g4 * 4 + 1] + r.y, acc[f][j][g4 * 4 + 2] + r.z, acc[f][j][g4 * 4 + 3] + r.w};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(v[0], v[1], v[2], v[3])), r_raw,
                                                           (int)(ok ? (pix * (unsigned)p.raw_ld + (unsigned)ch) * 4u : OOB), 0, 0);
                    const float scs[4] = {sc.x, sc.y, sc.z, sc.w}, shs[4] = {sh.x, sh.y, sh.z, sh.w};
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = v[e] * (has_sc != 0.f ? scs[e] : 1.f) + shs[e];   // shift reads zero without a table
                        const float pos = fmaxf(u, 0.f), neg = fmaxf(-0.1f * u, 0.f);
                        t[e] = p.act == CP_ACT_RELU ? pos : (p.act == CP_ACT_LEAKY01 ? pos - neg : u);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(t[0], t[1], t[2], t[3])), r_act,
                                                           (int)(ok ? (pix * (unsigned)p.act_ld + (unsigned)ch) * 4u : OOB), 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- pipeline: step = (tile, slice); stage = step & 1 -------------------------------------------------------------------------------
    {   // prologue: step 0 goes through the registers into stage 0 (its weights straight into stage 0), step 1 is left in flight
        const StepPos s0 = step_pos(0), s1 = step_pos(1);
        weights_dma(s0, 0);
        deep_static_for<0, NITEM>([&](auto ic) { issue_item(ic, s0); });
        unsigned char* hst = halo + lds_lane;
        deep_static_for<0, NITEM>([&](auto ic) {
            store_item(ic, hst);
            issue_item(ic, s1);
        });
        CP_WAIT_VM_N(NITEM);   // the nine weight DMAs of step 0 (older than step 1's halo loads) have landed
    }
    CP_BARRIER();
    int step = 0;
    for (int k = 0; k < my_tiles; ++k) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[f][j][e] = 0.f;
        for (int c = 0; c < nsl; ++c, ++step) {
            const unsigned char* hb = halo + (step & 1) * HALO_B;
            const unsigned char* wb = wst + (step & 1) * WSL_B + wlane;
            const StepPos n1 = step_pos(step + 1), n2 = step_pos(step + 2);
            const int nst = (step + 1) & 1;
            unsigned char* hst = halo + nst * HALO_B + lds_lane;
            weights_dma(n1, nst);   // stage nst was read last in step - 1: free since the barrier that ended it
            // fragments double-buffered in registers: the eight reads of tap t + 1 are issued before the sixteen MFMAs of tap t
            bf16x8 fa[2][4], fw[2][4];
            auto read_tap = [&](auto tc, int slot_) {
                constexpr int t = decltype(tc)::value;
                constexpr int toff = (((t / 3) * D) * HW + (t % 3) * D) * 16;
#pragma unroll
                for (int f = 0; f < 4; ++f) fa[slot_][f] = *reinterpret_cast<const bf16x8*>(hb + pbase[f] + toff);
#pragma unroll
                for (int j = 0; j < 4; ++j) fw[slot_][j] = *reinterpret_cast<const bf16x8*>(wb + (t * 4 + j) * 1024);
            };
            read_tap(std::integral_constant<int, 0>{}, 0);
            deep_static_for<0, 9>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if constexpr (t + 1 < 9) read_tap(std::integral_constant<int, t + 1>{}, (t + 1) & 1);
                // this tap's share of the staging: rows of step + 1 into the other stage, their registers re-armed with step + 2
                deep_static_for<t * IPT, (t * IPT + IPT < NITEM ? t * IPT + IPT : NITEM)>([&](auto ic) {
                    store_item(ic, hst);
                    issue_item(ic, n2);
                });
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[t & 1][j], fa[t & 1][f], acc[f][j], 0, 0, 0);
#ifndef DEEP_NO_SCHED
                // the tap's other instructions (8 fragment reads, ~3 rows of staging: select, load, 2 cvt, store, ~15 scalar) dealt between its 16
                // MFMAs: with one wave per SIMD whatever stands between two MFMA groups runs with the matrix pipe idle
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);   // scalar
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // a DS read
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // a DS write
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a VMEM read
                }
#endif
                __builtin_amdgcn_sched_barrier(0);   // nothing crosses a tap
            });
            CP_WAIT_VM_N(NITEM);   // the weight DMAs of step + 1, issued before this step's NITEM halo loads
            if (c + 1 < nsl) CP_BARRIER();
        }
        int pass, n, ty, tx;
        tile_of(k, pass, n, ty, tx);
        epilogue(pass, n, ty, tx);
        CP_BARRIER();
    }
}

template <int D, int TW>
int launch_deep(DeepK k, hipStream_t st) {
    constexpr int TH = 512 / TW;
    k.tiles_y = (k.H + TH - 1) / TH;
    k.tiles_x = (k.Wd + TW - 1) / TW;
    k.passes = (k.Cout + 127) / 128;
    k.tiles_per_pass = k.B * k.tiles_y * k.tiles_x;
    k.ntiles = k.passes * k.tiles_per_pass;
    const int grid = std::min(256, k.ntiles);
    CP_LAUNCH((conv_bf16d_kernel<D, TW>), dim3(grid), dim3(256), 0, st, k);
    return cp::check_launch("cp_conv2d_fwd_bf16_deep");
}

}  // namespace

extern "C" int cp_conv_bf16_deep_applicable(const cp_conv_desc* d) {
    if (!d || d->struct_size != (uint32_t)sizeof(cp_conv_desc)) return 0;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != d->dilation || (d->dilation != 1 && d->dilation != 2 && d->dilation != 4)) return 0;
    if (d->cout % 128 != 0 || d->cout > 1024 || d->group_rows || d->head_out || d->tap_label || d->row_scale || d->epi_label) return 0;
    if ((!d->out_raw && !d->out_act) || (d->out_raw && d->out_raw_ld % 4) || (d->out_act && d->out_act_ld % 4) || (d->residual && d->residual_ld % 4)) return 0;
    if ((((uintptr_t)d->out_raw) | ((uintptr_t)d->out_act) | ((uintptr_t)d->residual)) & 15) return 0;
    if (d->num_sources < 1 || d->num_sources > 2) return 0;
    for (int s = 0; s < d->num_sources; ++s) {
        const cp_conv_source& in = d->src[s];
        if (in.pre_scale || in.pre_shift || in.mode != CP_SRC_DIRECT) return 0;
        if (in.channels % 16 != 0 || in.ld % 4 != 0 || (((uintptr_t)in.data) & 15)) return 0;
        if ((long long)d->batch * d->in_h * d->in_w * in.ld * 4 >= (1LL << 31)) return 0;
    }
    const int max_ld = std::max(std::max(d->out_raw ? d->out_raw_ld : 0, d->out_act ? d->out_act_ld : 0), d->residual ? d->residual_ld : 0);
    if ((long long)d->batch * d->in_h * d->in_w * max_ld * 4 >= (1LL << 31)) return 0;   // 32-bit range-checked addressing with 0x80000000 as "absent"
    return 1;
}

extern "C" int cp_conv2d_fwd_bf16_deep(const cp_conv_desc* d, const void* weights_bf16, void* stream) {
    CP_REQUIRE_DESC(d, "cp_conv2d_fwd_bf16_deep");
    CP_REQUIRE(weights_bf16, "cp_conv2d_fwd_bf16_deep: null weights");
    CP_REQUIRE(cp_conv_bf16_deep_applicable(d), "cp_conv2d_fwd_bf16_deep: outside the kernel's range (3x3 / stride 1 / pad = dilation in {1, 2, 4}, cout a multiple of 128, direct "
                                                "sources of 16-multiple channels, raw and / or per-channel activated output, no labels)");
    DeepK k{};
    int nch = 0;
    for (int s = 0; s < d->num_sources; ++s) {
        const cp_conv_source& in = d->src[s];
        const long long nbytes = (long long)d->batch * d->in_h * d->in_w * in.ld * 4;
        CP_REQUIRE(nbytes < (1LL << 31), "cp_conv2d_fwd_bf16_deep: source %d spans %lld bytes; 32-bit range-checked addressing needs < 2 GiB", s, nbytes);
        k.s[s].data = in.data; k.s[s].C = in.channels; k.s[s].ld = in.ld; k.s[s].bytes = (unsigned)nbytes;
        if (s == 0) k.nch0 = in.channels / 16;
        nch += in.channels / 16;
    }
    k.nch = nch;
    k.W = reinterpret_cast<const unsigned char*>(weights_bf16);
    k.w_bytes = (unsigned)((size_t)(d->cout / 64) * nch * 9 * 2 * 1024);
    k.B = d->batch; k.H = d->in_h; k.Wd = d->in_w; k.Cout = d->cout;
    // absent lanes / absent tensors are addressed at byte offset 0x80000000 (OOB in the kernel): that is only out of range for tensors below 2 GiB
    const int max_ld = std::max(std::max(d->out_raw ? d->out_raw_ld : 0, d->out_act ? d->out_act_ld : 0), d->residual ? d->residual_ld : 0);
    CP_REQUIRE((long long)d->batch * d->in_h * d->in_w * max_ld * 4 < (1LL << 31), "cp_conv2d_fwd_bf16_deep: an output or the residual spans >= 2 GiB");
    k.residual = d->residual; k.res_ld = d->residual_ld;
    k.scale = d->scale; k.shift = d->shift; k.act = d->act;
    k.out_raw = d->out_raw; k.raw_ld = d->out_raw_ld; k.out_act = d->out_act; k.act_ld = d->out_act_ld;
    hipStream_t st = (hipStream_t)stream;
    // tile shape by the image width: 8 x 64 where 64 columns divide it better than 32 (56 -> 64 of 64 against 64 of 56 ... both 87.5 %; 80 -> 96 vs 128)
    const int w32 = (d->in_w + 31) / 32 * 32, w64 = (d->in_w + 63) / 64 * 64;
    const int h16 = (d->in_h + 15) / 16 * 16, h8 = (d->in_h + 7) / 8 * 8;
    (void)w32; (void)w64; (void)h16; (void)h8;   // (an 8 x 64 tile shape existed in the first version; one halo row per item needs TW + 2 D <= 64)
#define CP_DEEP(D_) if (d->dilation == D_) return launch_deep<D_, 32>(k, st);
    CP_DEEP(1) CP_DEEP(2) CP_DEEP(4)
#undef CP_DEEP
    return CP_ERR_INVALID;
}
