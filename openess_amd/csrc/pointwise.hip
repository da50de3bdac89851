// HBM-bound pointwise kernels around the MFMA convolutions (gfx950): ConvLSTM gate fusion and the
// fused EventPreprocessor-apply + NCHW->NHWC(8) bf16 layout change.  16-byte accesses per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "oess.h"
#include "oess_common.h"

namespace {
using namespace oess;
constexpr int THREADS = 256;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// gates: [P][4*C] bf16, channel blocks (in, remember, out, cell)  -- e2vid/model/submodules.py:205
// cell:  [P][C] fp32 (state, updated in place; prev == nullptr means zero state)
// hidden: bf16, pixel stride hs (may be a channel slice of the cat(x, h) buffer)
__global__ __launch_bounds__(THREADS) void convlstm_kernel(const uint16_t* __restrict__ gates, int64_t gs,
                                                           const float* __restrict__ prev_cell, float* __restrict__ cell,
                                                           uint16_t* __restrict__ hidden, int64_t hs, int64_t P, int C) {
    const int c8 = C >> 3;                         // 8 channels per thread
    const int64_t total = P * c8;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
        const int64_t p = i / c8;
        const int c0 = (int)(i - p * c8) * 8;
        const uint16_t* g = gates + p * gs + c0;
        union U { uint4 q; uint16_t h[8]; } gi, gr, go, gc, hout;
        gi.q = *reinterpret_cast<const uint4*>(g);
        gr.q = *reinterpret_cast<const uint4*>(g + C);
        go.q = *reinterpret_cast<const uint4*>(g + 2 * C);
        gc.q = *reinterpret_cast<const uint4*>(g + 3 * C);
        float pc[8];
        if (prev_cell) {
            const float4 a = *reinterpret_cast<const float4*>(prev_cell + p * C + c0);
            const float4 b = *reinterpret_cast<const float4*>(prev_cell + p * C + c0 + 4);
            pc[0] = a.x; pc[1] = a.y; pc[2] = a.z; pc[3] = a.w; pc[4] = b.x; pc[5] = b.y; pc[6] = b.z; pc[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) pc[k] = 0.0f;
        }
        float nc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float ig = sigmoidf_(bf16_to_f32(gi.h[k]));
            const float rg = sigmoidf_(bf16_to_f32(gr.h[k]));
            const float og = sigmoidf_(bf16_to_f32(go.h[k]));
            const float cg = tanhf(bf16_to_f32(gc.h[k]));
            nc[k] = rg * pc[k] + ig * cg;                         // submodules.py:211
            hout.h[k] = f32_to_bf16(og * tanhf(nc[k]));           // submodules.py:212
        }
        *reinterpret_cast<float4*>(cell + p * C + c0) = make_float4(nc[0], nc[1], nc[2], nc[3]);
        *reinterpret_cast<float4*>(cell + p * C + c0 + 4) = make_float4(nc[4], nc[5], nc[6], nc[7]);
        *reinterpret_cast<uint4*>(hidden + p * hs + c0) = hout.q;
    }
}

// EventPreprocessor apply (inference_utils.py:80-85) fused with the NCHW fp32 -> NHWC bf16 (8 channel,
// zero padded) layout change that feeds the E2VID head convolution.  stats = {sum, sumsq, nnz}.
__global__ __launch_bounds__(THREADS) void norm_to_nhwc8_kernel(const float* __restrict__ in, int B, int Ctot, int c0,
                                                                int Cs, int64_t HW, const double* __restrict__ stats,
                                                                int normalize, uint16_t* __restrict__ out) {
    const double nnz = stats ? stats[2] : 0.0;
    const bool active = normalize && nnz > 0.0;
    float mean = 0.f, stdv = 1.f;
    if (active) {
        const float nf = (float)nnz;
        mean = (float)stats[0] / nf;
        stdv = sqrtf(__fsub_rn((float)stats[1] / nf, __fmul_rn(mean, mean)));
    }
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
        const int64_t b = i / HW, r = i - b * HW;
        union { uint4 q; uint16_t h[8]; } o;
        o.q = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c < Cs) {
                float a = in[(b * Ctot + c0 + c) * HW + r];
                if (active) a = __fmul_rn((a != 0.0f) ? 1.0f : 0.0f, __fsub_rn(a, mean)) / stdv;
                o.h[c] = f32_to_bf16(a);
            }
        }
        *reinterpret_cast<uint4*>(out + i * 8) = o.q;
    }
}

}  // namespace

extern "C" {

int oess_convlstm_gates_bf16(const void* gates, long long gates_pix_stride, const float* prev_cell, float* cell,
                             void* hidden, long long hidden_pix_stride, long long n_pixels, int C, oess_stream_t stream) {
    if (!gates || !cell || !hidden || n_pixels <= 0 || C <= 0 || (C & 7) || (gates_pix_stride & 7) ||
        (hidden_pix_stride & 7) || gates_pix_stride < 4 * C || hidden_pix_stride < C)
        return OESS_EINVAL;
    int64_t work = n_pixels * (C >> 3);
    int grid = (int)((work + THREADS - 1) / THREADS);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(convlstm_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, (const uint16_t*)gates,
                       (int64_t)gates_pix_stride, prev_cell, cell, (uint16_t*)hidden, (int64_t)hidden_pix_stride,
                       (int64_t)n_pixels, C);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_event_slice_to_nhwc8_bf16(const float* in, int B, int Ctot, int c0, int Cs, long long HW, const double* stats,
                                   int normalize, void* out_nhwc8, oess_stream_t stream) {
    if (!in || !out_nhwc8 || B <= 0 || Ctot <= 0 || c0 < 0 || Cs <= 0 || Cs > 8 || c0 + Cs > Ctot || HW <= 0 ||
        (normalize && !stats))
        return OESS_EINVAL;
    int64_t work = (int64_t)B * HW;
    int grid = (int)((work + THREADS - 1) / THREADS);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(norm_to_nhwc8_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, in, B, Ctot, c0, Cs,
                       (int64_t)HW, stats, normalize, (uint16_t*)out_nhwc8);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // extern "C"
