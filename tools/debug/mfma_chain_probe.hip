// Probe: issue rate of v_mfma_f32_32x32x16_bf16 when consecutive MFMAs accumulate into 1 / 2 / 3 / 9 different accumulators (one wave per SIMD).
// hipcc --offload-arch=gfx950 -O3 mfma_chain_probe.hip -o mfma_chain_probe && ./mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(float)(threadIdx.x + e); y[e] = (__bf16)(float)(threadIdx.x * 2 + e); }
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 36 / NACC; ++rep)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC> void run(float* d, long long* c) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<256, 256>>>(d, 10, c);
    hipEventRecord(e0);
    k<NACC><<<256, 256>>>(d, iters, c);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * (36 / NACC) * NACC;
    printf("accumulators %d: %.1f ns per MFMA per SIMD (%.2f PFLOP/s chip), s_memtime ticks per MFMA %.1f\n", NACC, ms * 1e6 / n,
           n * 32768.0 * 1024 / (ms * 1e-3) / 1e15, (double)h / n);
}
int main() {
    float* d; long long* c; hipMalloc(&d, 256 * 256 * 4); hipMalloc(&c, 8);
    run<1>(d, c); run<2>(d, c); run<3>(d, c); run<4>(d, c); run<9>(d, c);
    return 0;
}
