// Stand-alone probe (hipcc --offload-arch=gfx950 -O2 f16_mix_probe.hip -o f16_mix_probe): is the 3-instruction low half of the fp16 two-way split
// (v_fma_mixlo_f16 / v_fma_mixhi_f16: fp32 subtraction of the up-converted high half, ONE rounding to fp16) bit-identical to the 5-instruction form
// (v_cvt_f32_f16 x 2, v_pk_add_f32, v_cvt_pk_f16_f32) -- over random magnitudes from 1e-9 to 3e5, signs, and with MODE.FP16_OVFL set (clamping)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ void split_old(float a, float b, unsigned& hi, unsigned& lo) {
    const f32x2_t x = {a, b};
    const f16x2_t h = __builtin_convertvector(x, f16x2_t);
    const f32x2_t r = x - __builtin_convertvector(h, f32x2_t);
    const f16x2_t l = __builtin_convertvector(r, f16x2_t);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ void split_mix(float a, float b, unsigned& hi, unsigned& lo) {
    const f32x2_t x = {a, b};
    const f16x2_t h = __builtin_convertvector(x, f16x2_t);
    hi = __builtin_bit_cast(unsigned, h);
    unsigned l = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(l) : "v"(a), "v"(b), "v"(hi));
#endif
    lo = l;
}
__global__ void k(const float* in, int n, int ovfl, unsigned* out) {
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1" ::: "memory");
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned h0, l0, h1, l1;
    split_old(in[2 * i], in[2 * i + 1], h0, l0);
    split_mix(in[2 * i], in[2 * i + 1], h1, l1);
    out[4 * i] = h0; out[4 * i + 1] = l0; out[4 * i + 2] = h1; out[4 * i + 3] = l1;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    srand(7);
    for (int i = 0; i < n; ++i) {
        const double e = -9.0 + 14.5 * (rand() / (double)RAND_MAX);
        h[i] = (float)((rand() & 1 ? -1.0 : 1.0) * pow(10.0, e) * (0.5 + rand() / (double)RAND_MAX));
    }
    h[0] = 0.f; h[1] = -0.f; h[2] = 65504.f; h[3] = 65520.f; h[4] = 131008.f; h[5] = 6.1e-5f; h[6] = 5.96e-8f; h[7] = 1e-10f;
    float* d; unsigned* o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 2 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> r(2 * n);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        k<<<n / 2 / 256, 256>>>(d, n, ovfl, o);
        hipMemcpy(r.data(), o, n * 2 * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n / 2; ++i)
            if (r[4 * i] != r[4 * i + 2] || r[4 * i + 1] != r[4 * i + 3]) {
                if (bad < 5) printf("  mismatch at %g %g: old %08x %08x mix %08x %08x\n", h[2 * i], h[2 * i + 1], r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
                ++bad;
            }
        printf("FP16_OVFL %d: %ld mismatching pairs of %d\n", ovfl, bad, n / 2);
    }
    return 0;
}
