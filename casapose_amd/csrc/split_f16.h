// Two-way fp16 split of fp32 operands for the bf16/fp16 matrix pipe ("f16x2": three products instead of the six of the exact bf16 split).
//     hi = rn_f16(x),   lo = rn_f16(x - hi)            (x - hi is exact in fp32: at most 13 significant bits, |x - hi| <= 2^-11 |x|)
// lo keeps 11 of those 13 bits, so at most ONE unit of x's last place is lost: |x - hi - lo| <= 2^-23 |x| (one fp32 ulp), 0.75 * 2^-24 rms, and three
// fp32 operands in four are reproduced exactly (tests/test_f16x2_arith.py) -- as long as `lo` is a NORMAL fp16 number (|x| >= 2^-2); below that
// it is absolute (fp16 subnormals, spacing 2^-24: honoured by v_cvt_pk_f16_f32 and by v_mfma_f32_32x32x16_f16 on gfx950, measured with
// tools/debug/f16_probe.hip).  The products hi*hi, hi*lo, lo*hi are exact in the fp32 accumulator (11 x 11 bits); the dropped lo*lo is
// <= 2^-22 of the product.  Measured against fp64 the result is at or below the fp32 MFMA's error (tests/test_gpu_f16x2.py).  Range: fp16 ends at 65504; with MODE.FP16_OVFL set a conversion clamps
// instead of producing inf and lo takes 11 bits of the remainder: no inf / NaN, the error of such an operand grows to <= 2^-12 of it (beyond
// 131008: saturation).  Graceful, not fp32-level: the callers keep their operands inside the range.
// WEIGHTS are multiplied by a power of two first (cp_f16x2_weight_scale: max |w| -> [2^11, 2^12)) so that their low parts are normal numbers;
// the kernels multiply the accumulators by the inverse (exact).  Activations are used as they are: every convolution input of this network
// is a normalised (BN / CLADE + activation) tensor or the normalised image.
#pragma once
#include <hip/hip_runtime.h>

namespace cp {

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// clamp fp16 overflows to +-65504 for the rest of the wave's life (conversions only; the MFMA accumulates in fp32)
__device__ __forceinline__ void f16_overflow_clamps() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1" ::: "memory"); }

// Round 5: three instructions per PAIR of operands instead of six.  The low half lo = rn_f16(x - hi) is ONE mixed-precision instruction per element --
// v_fma_mixlo_f16 / v_fma_mixhi_f16 evaluate x * 1.0 - (float)hi in fp32 (exact: the difference has at most 13 significant bits) and round the result
// to fp16 into the low / high half of the destination -- where the compiler's own form takes two v_cvt_f32_f16, a v_pk_add_f32 and a v_cvt_pk_f16_f32
// (it folds fma(h, -1, x) back into a subtraction, so the instruction is written out).  Bit-identical to that form on 4.2 M operands from 1e-9 to 3e5,
// signed zeros, subnormal halves and, with MODE.FP16_OVFL set, beyond 65504 (tools/debug/f16_mix_probe.hip, run on the MI355X).
__device__ __forceinline__ void split2h(float a, float b, unsigned& hi, unsigned& lo) {
    const f32x2_t x = {a, b};
    const f16x2_t h = __builtin_convertvector(x, f16x2_t);
    hi = __builtin_bit_cast(unsigned, h);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CP_SPLIT2H_PLAIN)
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(l) : "v"(a), "v"(b), "v"(hi));
    lo = l;
#else
    const f32x2_t r = x - __builtin_convertvector(h, f32x2_t);
    const f16x2_t l = __builtin_convertvector(r, f16x2_t);
    lo = __builtin_bit_cast(unsigned, l);
#endif
}

__device__ __forceinline__ void split4h(const float4 v, uint2& hi, uint2& lo) {
    split2h(v.x, v.y, hi.x, lo.x);
    split2h(v.z, v.w, hi.y, lo.y);
}

}  // namespace cp
