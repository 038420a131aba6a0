// Probe of ds_read_b64_tr_b16 on gfx950: which LDS element each (lane, j) receives.  Run: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int row_stride_elems) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    // lane l supplies the address of 4 contiguous elements: row (l&15)/4 ... see host print; here: generic per-lane element offset
    const int g = l >> 4, i = l & 15;
    const int elem = (g * 4 + (i >> 2)) * row_stride_elems + (i & 3) * 4;   // 16-lane group g covers rows 4g..4g+3, lane i: row i/4, cols 4(i%4)..+3
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + elem));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)r[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {16, 32}) {
        probe<<<1, 64>>>(d, stride);
        uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row stride %d elements: lane -> 4 values as (row,col)\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l * 4 + j] / stride, h[l * 4 + j] % stride);
            printf("\n");
        }
    }
    return 0;
}
