// ATen's bilinear index rule (area_pixel_compute_source_index, fp32), shared by the resize kernels and the pooling matrix:
//   src = align_corners ? dst * (in - 1) / (out - 1) : max((dst + 0.5) * in / out - 0.5, 0)
//   i0 = (int)src, i1 = min(i0 + 1, in - 1), lambda = src - i0
#pragma once
#include <hip/hip_runtime.h>

namespace oess {

struct Axis { float scale; int align; int in, out; };

__device__ __forceinline__ void src_index(const Axis& ax, int dst, int& i0, int& i1, float& lam) {
    float s = ax.align ? ax.scale * (float)dst : fmaxf(ax.scale * ((float)dst + 0.5f) - 0.5f, 0.0f);
    i0 = (int)s;
    if (i0 > ax.in - 1) i0 = ax.in - 1;
    i1 = (i0 < ax.in - 1) ? i0 + 1 : i0;
    lam = s - (float)i0;
}
// candidate output range whose footprint can touch input index i (generous by 2 on both sides; exact test follows)
__device__ __forceinline__ void candidates(const Axis& ax, int i, int& lo, int& hi) {
    if (ax.scale <= 0.f) { lo = 0; hi = ax.out - 1; return; }
    const float inv = 1.0f / ax.scale;
    float a, b;
    if (ax.align) { a = ((float)i - 1.0f) * inv; b = ((float)i + 1.0f) * inv; }
    else { a = ((float)i - 0.5f) * inv - 0.5f; b = ((float)i + 1.5f) * inv - 0.5f; }
    lo = (int)floorf(a) - 2; hi = (int)ceilf(b) + 2;
    if (lo < 0) lo = 0;
    if (hi > ax.out - 1) hi = ax.out - 1;
}
__device__ __forceinline__ float weight_for(const Axis& ax, int dst, int i) {
    int i0, i1; float lam;
    src_index(ax, dst, i0, i1, lam);
    return (i0 == i ? 1.0f - lam : 0.0f) + (i1 == i ? lam : 0.0f);
}

static inline Axis make_axis(int in, int out, int align) {
    Axis a; a.in = in; a.out = out; a.align = align;
    a.scale = align ? (out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f) : (float)in / (float)out;
    return a;
}

}  // namespace oess
