// Shared helpers for liboess translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define OESS_HIP(expr)                                   \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return OESS_ELAUNCH;       \
    } while (0)

namespace oess {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// compute units of the current device (cached per process; persistent kernels size their grid from it)
static inline int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even float -> bf16 (NaN preserved as quiet NaN)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// two floats -> packed bf16x2 (low half = a) with the gfx950 hardware conversion (round-to-nearest-even, identical to
// f32_to_bf16 for every non-NaN input; NaNs come out as a quiet NaN).  One VALU instruction instead of ~10.
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// eight floats -> one 16-byte vector of bf16
__device__ __forceinline__ uint4 pack_bf16x8(const float (&v)[8]) {
    return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace oess
