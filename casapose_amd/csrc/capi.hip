// Library-level entry points of include/casapose_hip.h: error reporting and probing.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>

namespace {
thread_local std::string g_last_error;
}

namespace cp {
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
}  // namespace cp

namespace cp {
int& persistent_blocks_ref() {
    static int n = [] {
        const char* e = getenv("CASAPOSE_PERSIST_BLOCKS");
        const int v = e ? atoi(e) : 256;
        return (v < 8 || v > 256 || v % 8) ? 256 : v;
    }();
    return n;
}
}  // namespace cp

namespace cp {
uint32_t*& f16x2_monitor_ref() {
    thread_local uint32_t* slot = nullptr;
    return slot;
}
}  // namespace cp

extern "C" int cp_f16x2_monitor_set(uint32_t* slot) {
    CP_REQUIRE(((uintptr_t)slot & 15) == 0, "cp_f16x2_monitor_set: a slot is four 32-bit words, 16-byte aligned");
    cp::f16x2_monitor_ref() = slot;
    return CP_OK;
}
extern "C" uint32_t* cp_f16x2_monitor_get(void) { return cp::f16x2_monitor_ref(); }

// The band check of the f16x2 range guard, for any caller: amax = what a monitor slot (or cp_amax_f32) measured on the operand AS CONVERTED (with
// whatever power of two is already applied).  Returns 0 in the band [lo, hi] (or amax == 0: nothing to judge), 1 with *rescale = the power of two
// that brings amax into [2^10, 2^11) (32x headroom to 65504, low halves normal down to 2^-12 of the maximum), 2 when no power of two in
// [2^-24, 2^24] does or amax is not finite: run that layer on the exact three-way bf16 split (no range condition).
extern "C" int cp_f16x2_range_check(float amax, float lo, float hi, float* rescale) {
    if (rescale) *rescale = 1.f;
    if (amax == 0.f) return 0;
    if (!(amax > 0.f) || !std::isfinite(amax)) return 2;
    if (amax >= lo && amax <= hi) return 0;
    int e = 0;
    (void)std::frexp(amax, &e);          // amax = m * 2^e, m in [0.5, 1)  ->  amax * 2^(11 - e) in [2^10, 2^11)
    const int k = 11 - e;
    if (k < -24 || k > 24) return 2;
    if (rescale) *rescale = std::ldexp(1.f, k);
    return 1;
}

namespace {
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long groups, long long group_stride, long long count, uint32_t* slot) {
    // 16-byte loads where the group layout allows them (count and stride multiples of 4, base aligned: checked by the launcher through `vec`)
    float a = 0.f;
    const long long per = count >> 2, total = groups * per;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long g = i / per, j = i - g * per;
        a = cp::amax4(a, *reinterpret_cast<const float4*>(x + g * group_stride + j * 4));
    }
    cp::monitor_flush(slot, a);
    cp::monitor_count_launch(slot, threadIdx.x == 0);
}
__global__ __launch_bounds__(256) void amax_scalar_kernel(const float* __restrict__ x, long long groups, long long group_stride, long long count, uint32_t* slot) {
    float a = 0.f;
    const long long total = groups * count;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long g = i / count, j = i - g * count;
        a = fmaxf(a, fabsf(x[g * group_stride + j]));
    }
    cp::monitor_flush(slot, a);
    cp::monitor_count_launch(slot, threadIdx.x == 0);
}
}  // namespace

// max |x| over `groups` runs of `count` floats, run g starting at x + g * group_stride, folded into slot[0] (atomic max of the bit pattern; the
// caller zeroes the slot).  What the f16x2 calibration needs where no converting kernel reports by itself (a caller's own tensors).
extern "C" int cp_amax_f32(const float* x, long long groups, long long group_stride, long long count, uint32_t* slot, void* stream) {
    CP_REQUIRE(x && slot && groups > 0 && count > 0 && group_stride >= 0, "cp_amax_f32: bad arguments");
    const bool vec = ((uintptr_t)x & 15) == 0 && count % 4 == 0 && group_stride % 4 == 0;
    const long long items = vec ? groups * (count / 4) : groups * count;
    const int blocks = (int)std::min<long long>((items + 255) / 256, 2048);
    if (vec) CP_LAUNCH(amax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, groups, group_stride, count, slot);
    else CP_LAUNCH(amax_scalar_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, groups, group_stride, count, slot);
    return cp::check_launch("cp_amax_f32");
}

extern "C" int cp_set_persistent_blocks(int blocks) {
    CP_REQUIRE(blocks >= 8 && blocks <= 256 && blocks % 8 == 0, "cp_set_persistent_blocks: a multiple of 8 in [8, 256] (one block per CU, whole XCD rows)");
    cp::persistent_blocks_ref() = blocks;
    return CP_OK;
}
extern "C" int cp_get_persistent_blocks(void) { return cp::persistent_blocks_ref(); }

extern "C" const char* cp_last_error(void) { return g_last_error.c_str(); }
extern "C" int cp_version(void) { return CP_ABI_VERSION; }
extern "C" size_t cp_conv_desc_size(void) { return sizeof(cp_conv_desc); }
extern "C" size_t cp_conv_source_size(void) { return sizeof(cp_conv_source); }
extern "C" int cp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// ---- matrix-pipe probe -------------------------------------------------------------------------------------------------------------
// A bare stream of MFMAs on every SIMD (one wave per SIMD, nine accumulators in rotation, no memory traffic): what the part SUSTAINS under
// its power limit, as opposed to the datasheet peak the roofline entries are priced against.  bench.py times it with HIP events and prints
// the result beside the roofline fraction.  which = 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_bf16.
namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int WHICH>
__global__ __launch_bounds__(256) void mfma_probe_kernel(float* out, int iters) {
    f32x16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float x = 1.0f + (float)(threadIdx.x & 7) * 0.125f, y = 0.5f + (float)(threadIdx.x & 3) * 0.25f;   // non-trivial operands: zeros would raise the clock
    bf16x8 bx, by;
#pragma unroll
    for (int e = 0; e < 8; ++e) { bx[e] = (__bf16)(x + 0.0625f * e); by[e] = (__bf16)(y - 0.03125f * e); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int a = 0; a < 9; ++a) {
                if constexpr (WHICH == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
                else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, acc[a], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

extern "C" size_t cp_mfma_probe_workspace_bytes(void) { return (size_t)256 * 256 * sizeof(float); }

// launches the probe once; returns the FLOPs it executes in *flops (2*M*N*K per MFMA x 36 per iteration x 1024 waves)
extern "C" int cp_mfma_probe(int which, int iters, void* ws, double* flops, void* stream) {
    CP_REQUIRE(ws && flops && iters > 0 && (which == 0 || which == 1), "cp_mfma_probe: bad arguments");
    float* out = reinterpret_cast<float*>(ws);
    if (which == 0) CP_LAUNCH(mfma_probe_kernel<0>, dim3(256), dim3(256), 0, (hipStream_t)stream, out, iters);
    else CP_LAUNCH(mfma_probe_kernel<1>, dim3(256), dim3(256), 0, (hipStream_t)stream, out, iters);
    *flops = (double)iters * 36.0 * 1024.0 * (which == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16);
    return cp::check_launch("cp_mfma_probe");
}
