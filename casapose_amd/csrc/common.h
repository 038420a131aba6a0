// Shared helpers for the gfx950 kernels behind include/casapose_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

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

// a descriptor built against another revision of the header (cp_conv_desc grew at its tail between rounds): refuse it by size
#define CP_REQUIRE_DESC(d, fn)                                                                                                        \
    CP_REQUIRE((d) && (d)->struct_size == (uint32_t)sizeof(cp_conv_desc),                                                             \
               fn ": descriptor is null or its struct_size (%u) is not this library's sizeof(cp_conv_desc) = %u (ABI %d); rebuild the " \
                  "caller against include/casapose_hip.h",                                                                           \
               (d) ? (unsigned)(d)->struct_size : 0u, (unsigned)sizeof(cp_conv_desc), CP_ABI_VERSION)

// Blocks of a persistent launch (one block per CU: 256).  cp_set_persistent_blocks / CASAPOSE_PERSIST_BLOCKS = a smaller multiple of 8 leaves CUs
// free for a kernel of ANOTHER stream: the two-stream forward (engine.py) runs the HBM-bound passes of one half-batch beside the matrix-pipe
// kernels of the other.
int& persistent_blocks_ref();   // capi.hip
inline int persistent_blocks() { return persistent_blocks_ref(); }

// ---- f16x2 range monitor (round 6) -------------------------------------------------------------------------------------------------
// A monitor slot is four 32-bit words of device memory: [0] = bits of max |operand| over every fp32 value a launch converted to an fp16 pair
// (atomic max on the bit pattern: monotone for non-negative floats, inf included), [1] = launches that reported, [2] = the same maximum for the
// operand of a fused 1x1 head (the activated map that exists in registers only), [3] reserved.  cp_f16x2_monitor_set() arms the slot for the
// CALLING THREAD's next launches; every kernel that converts activations (conv_hsplit, conv_stem_split, the split GEMMs) or writes the Winograd
// planes another kernel converts (wino_in, wino_out_in) reads it at launch time.  A null slot costs one uniform branch per staged slice.
uint32_t*& f16x2_monitor_ref();   // capi.hip (thread-local)
inline uint32_t* f16x2_monitor() { return f16x2_monitor_ref(); }

__device__ __forceinline__ float amax4(float acc, const float4 v) {
    return fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), acc), fmaxf(fabsf(v.z), fabsf(v.w)));   // (v_max3_f32 with |.| modifiers; NaN operands are skipped)
}

// every lane of the wave calls it (wave-uniform control flow); word 0 or 2 of the slot
__device__ __forceinline__ void monitor_flush(uint32_t* word, float amax) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
    if ((threadIdx.x & 63) == 0 && amax > 0.f) {
        // Look before the atomic: the slot is a sticky maximum, so after the first waves of the first armed launch almost no wave has anything to add --
        // and tens of thousands of same-address atomics at the end of every transform launch cost 0.28 ms of an 7.8 ms forward (measured, A/B in one
        // call: 2040 against 1970 images/s).  A stale (smaller) value read here only means an atomic that was not needed.
        const uint32_t bits = __builtin_bit_cast(uint32_t, amax);
        if (bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, bits);
    }
}
// one thread of the launch (the caller names it: the reporting waves need not include thread 0)
__device__ __forceinline__ void monitor_count_launch(uint32_t* slot, bool leader) {
    if (blockIdx.x == 0 && leader) atomicAdd(slot + 1, 1u);
}

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

// Arg-max over the first `classes` head channels of this lane's pixel from the fused head's accumulator (conv_halo.hip / conv_hsplit.hip):
// register g4 * 4 + e of lane half `half` holds channel q = 8 g4 + 4 half + e, so a pixel's 32 channels sit in lanes l and l ^ 32.
// First maximum wins (cp_argmax_labels); all lanes of the wave must call it.
template <typename V>
__device__ __forceinline__ int head_argmax(const V& a2, int half, int classes) {
    float best = -__builtin_inff();
    int bi = 0x7fffffff;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = g4 * 8 + half * 4 + e;
            const float v = a2[g4 * 4 + e];
            if (q < classes && v > best) { best = v; bi = q; }
        }
    const float ob = __shfl_xor(best, 32);
    const int oi = __shfl_xor(bi, 32);
    if (ob > best || (ob == best && oi < bi)) bi = oi;
    return bi == 0x7fffffff ? 0 : bi;
}

// conv_wgrad.hip: the fp32 weight-gradient kernel over the packed K chunks [first_chunk, ktot / 32) only, accumulating
int wgrad_f32_chunks(const cp_conv_desc* d, const float* dy, int dy_ld, float* dw_packed, int first_chunk, hipStream_t st);

}  // namespace cp
