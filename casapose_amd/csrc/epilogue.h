// Shared, branch-free convolution epilogue (both conv kernels).
//
// The first version walked the 16 accumulator rows of a 32x32 MFMA tile one by one:
// label load -> wait -> CLADE table loads -> wait -> store, i.e. ~4 serialised memory latencies per
// row, ~25 us per tile -- several times the MFMA time of a shallow layer.  Here every fetch of a
// batch of rows is issued back to back (range-checked buffer loads: an out-of-range offset returns
// 0 / drops the store, so there is no control flow), then the dependent table look-ups as a second
// batch, then the arithmetic and the stores.
#pragma once
#include "common.h"

namespace cp {

typedef float f32x16_t __attribute__((ext_vector_type(16)));

struct EpiArgs {
    const float* row_scale;  // [pixels] or null
    const uint8_t* label;    // [pixels] or null (CLADE table index)
    const float* residual;   // [pixels][res_ld] or null
    const float* scale;      // [cout] or [classes][cout] or null
    const float* shift;
    float* out_raw;
    float* out_act;
    int res_ld, raw_ld, act_ld, cout, act;
    unsigned npix;           // pixels in the output tensor (range checks)
};

struct EpiRsrc {
    __amdgpu_buffer_rsrc_t rs, lab, res, raw, actb;
    bool has_rs, has_lab, has_res, has_raw, has_act, has_aff;
};

__device__ __forceinline__ EpiRsrc epi_make(const EpiArgs& e, const void* any_valid) {
    EpiRsrc r;
    r.has_rs = e.row_scale != nullptr;
    r.has_lab = e.label != nullptr;
    r.has_res = e.residual != nullptr;
    r.has_raw = e.out_raw != nullptr;
    r.has_act = e.out_act != nullptr;
    r.has_aff = e.scale != nullptr;
    // a null feature gets an EMPTY descriptor: every access is out of range (loads 0, stores dropped)
    r.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(r.has_rs ? (const void*)e.row_scale : any_valid), 0, r.has_rs ? e.npix * 4u : 0u, 0x00020000);
    r.lab = __builtin_amdgcn_make_buffer_rsrc((void*)(r.has_lab ? (const void*)e.label : any_valid), 0, r.has_lab ? e.npix : 0u, 0x00020000);
    r.res = __builtin_amdgcn_make_buffer_rsrc((void*)(r.has_res ? (const void*)e.residual : any_valid), 0,
                                              r.has_res ? e.npix * (unsigned)e.res_ld * 4u : 0u, 0x00020000);
    r.raw = __builtin_amdgcn_make_buffer_rsrc((void*)(r.has_raw ? (void*)e.out_raw : (void*)any_valid), 0,
                                              r.has_raw ? e.npix * (unsigned)e.raw_ld * 4u : 0u, 0x00020000);
    r.actb = __builtin_amdgcn_make_buffer_rsrc((void*)(r.has_act ? (void*)e.out_act : (void*)any_valid), 0,
                                               r.has_act ? e.npix * (unsigned)e.act_ld * 4u : 0u, 0x00020000);
    return r;
}

// One 32x32 accumulator block per j (TN blocks side by side along channels).  `rowpix(r)` maps
// accumulator row r (0..15 of this lane) to the output pixel index, or -1 if the row is outside
// the tensor.  `co[j]` is this lane's output channel in block j.  `rs_pre[r]` is an extra per-row
// factor computed by the caller (partial-conv 9/count) -- pass nullptr when unused.  `keep` receives the
// activated values [16 rows][TN] of this lane (0 for rows/channels outside the tensor).
template <int TN, int RB, typename RowPix>
__device__ __forceinline__ void epilogue_block(const f32x16_t (&acc)[TN], const int (&co)[TN], const EpiArgs& e, const EpiRsrc& rr,
                                               RowPix rowpix, const float* rs_pre, float (&keep)[16][TN]) {
    constexpr unsigned OOB = 0x80000000u;
    static_assert(16 % RB == 0, "rows per batch must divide 16");  // RB trades registers for memory round trips
    unsigned coff[TN];                       // OOB when this lane's channel does not exist
    float sc0[TN], sh0[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const bool ok = co[j] < e.cout;
        coff[j] = ok ? (unsigned)co[j] : OOB;
        // per-channel affine (no label): one load per lane
        sc0[j] = (rr.has_aff && !rr.has_lab && ok) ? e.scale[co[j]] : 1.f;
        sh0[j] = (rr.has_aff && !rr.has_lab && ok) ? e.shift[co[j]] : 0.f;
    }
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += RB) {
        int pix[RB];
        float rs[RB];
        int lab[RB];
        float res[RB][TN];
        // ---- batch 1: everything that depends only on the pixel ------------------------------
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            pix[q] = rowpix(r0 + q);
            const unsigned po = pix[q] >= 0 ? (unsigned)pix[q] : OOB;
            rs[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr.rs, (int)(po < OOB ? po * 4u : OOB), 0, 0));
            lab[q] = __builtin_amdgcn_raw_buffer_load_b8(rr.lab, (int)po, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned o = (po < OOB && coff[j] < OOB) ? (po * (unsigned)e.res_ld + coff[j]) * 4u : OOB;
                res[q][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr.res, (int)o, 0, 0));
            }
        }
        // ---- batch 2: class-adaptive tables (depend on the labels) -----------------------------
        float sc[RB][TN], sh[RB][TN];
#pragma unroll
        for (int q = 0; q < RB; ++q)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (rr.has_aff && rr.has_lab) {
                    const int t = (lab[q] & 0xff) * e.cout + (co[j] < e.cout ? co[j] : 0);
                    sc[q][j] = e.scale[t];
                    sh[q][j] = e.shift[t];
                } else {
                    sc[q][j] = sc0[j];
                    sh[q][j] = sh0[j];
                }
            }
        // ---- arithmetic + stores ----------------------------------------------------------------
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const unsigned po = pix[q] >= 0 ? (unsigned)pix[q] : OOB;
            float f = rr.has_rs ? rs[q] : 1.f;
            if (rs_pre) f *= rs_pre[r0 + q];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v = acc[j][r0 + q] * f + res[q][j];
                const bool ok = po < OOB && coff[j] < OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rr.raw,
                                                      (int)(ok ? (po * (unsigned)e.raw_ld + coff[j]) * 4u : OOB), 0, 0);
                float t = v * sc[q][j] + sh[q][j];
                if (e.act == CP_ACT_RELU) t = fmaxf(t, 0.f);
                else if (e.act == CP_ACT_LEAKY01) t = fmaxf(t, 0.f) - fmaxf(-0.1f * t, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), rr.actb,
                                                      (int)(ok ? (po * (unsigned)e.act_ld + coff[j]) * 4u : OOB), 0, 0);
                keep[r0 + q][j] = ok ? t : 0.f;  // for a fused 1x1 head: the caller may multiply the activated tile again
            }
        }
    }
}

}  // namespace cp
