// Probe (gfx950): do v_cvt_f16_f32 and v_mfma_f32_32x32x16_f16 honour fp16 subnormals?  Build: hipcc --offload-arch=gfx950 -O2 -o variants/f16_probe tools/debug/f16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(float* out, float a_val, float b_val) {
    const _Float16 ha = (_Float16)a_val, hb = (_Float16)b_val;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (e == 0 && threadIdx.x < 32) ? ha : (_Float16)0.f; b[e] = (e == 0 && threadIdx.x < 32) ? hb : (_Float16)0.f; }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = (float)ha; out[1] = (float)hb; out[2] = acc[0]; out[3] = (float)(ha * hb); }
}
int main() {
    float* d; hipMalloc(&d, 64);
    const float tests[][2] = {{1e-6f, 1024.f}, {3e-5f, 2.f}, {6.2e-5f, 1.f}, {1.0f, 1e-7f}, {70000.f, 1.f}, {-1e6f, 1.f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("a=%g b=%g : cvt(a)=%g cvt(b)=%g  mfma a*b=%g  (exact %g)  valu half mul %g\n", t[0], t[1], h[0], h[1], h[2], (double)h[0] * h[1], h[3]);
    }
    return 0;
}
