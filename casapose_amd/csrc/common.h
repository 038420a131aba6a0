// Shared helpers for the gfx950 kernels behind include/casapose_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "casapose_hip.h"

namespace cp {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return CP_ERR_LAUNCH;
    }
    return CP_OK;
}

// hipGetLastError() is sticky per thread: clear whatever an earlier (possibly foreign, e.g.
// PyTorch's device probing) call left behind so check_launch() reports THIS launch only.
#define CP_LAUNCH(...)               \
    do {                             \
        (void)hipGetLastError();     \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

#define CP_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            cp::set_error(__VA_ARGS__);  \
            return CP_ERR_INVALID;       \
        }                                \
    } while (0)

// Bijective XCD-aware remap of a 1-D block id: blocks b, b+8, b+16, ... land on the same
// XCD (observed dispatch, MI355X_MICROARCH.md), so give each XCD one contiguous run of
// logical tiles to keep operand panels in that XCD's L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int nx = 8;
    int q = nblocks / nx, r = nblocks % nx;
    int xcd = bid % nx, idx = bid / nx;
    int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// conv_wgrad.hip: the fp32 weight-gradient kernel over the packed K chunks [first_chunk, ktot / 32) only, accumulating
int wgrad_f32_chunks(const cp_conv_desc* d, const float* dy, int dy_ld, float* dw_packed, int first_chunk, hipStream_t st);

}  // namespace cp
