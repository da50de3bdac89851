// HBM-bound pointwise kernels around the MFMA convolutions (gfx950): ConvLSTM gate fusion and the
// fused EventPreprocessor-apply + NCHW->NHWC(8) bf16 layout change.  16-byte accesses per lane.
#include <mutex>
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


// ---------------------------------------------------------------------------------------------
// Linear probe: nn.Conv2d(K, K, 1) on the fp32 logits (models/style_networks.py:169-170, models/deeplabv3.py:186-187), K <= 32.
// Dense NHWC fp32 rows [P][K] (K = 11: 44-byte rows), so everything moves through LDS tiles of LP_TP pixels with coalesced
// 4-byte accesses.  MIOpen resolved this layer's weight gradient to a naive kernel (565 ms on first use) and its bias gradient
// to a 2.9 ms ATen reduction at 8 x 440 x 640.
//   forward : y[p][j] = b[j] + sum_i W[j][i] x[p][i]
//   backward: gx[p][i] = sum_j W[j][i] g[p][j];  gW[j][i] = sum_p g[p][j] x[p][i];  gb[j] = sum_p g[p][j]
//             -- per-workgroup partial rows [K*K + K] in double, added in a fixed order by the finalize kernel (bit-repeatable).
// ---------------------------------------------------------------------------------------------
constexpr int LP_TP = 256;                  // pixels per tile = threads per workgroup
constexpr int LP_KMAX = 32;
constexpr int LP_MAX_ROWS = 1024;           // backward workgroups = partial rows

__global__ __launch_bounds__(LP_TP) void probe_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, int64_t P, int K, float* __restrict__ y) {
    extern __shared__ float lp_s[];                      // [K*K] weights, [K] bias, [LP_TP * K] tile (in place: x then y)
    float* sw = lp_s; float* sb = sw + K * K; float* tile = sb + K;
    for (int i = threadIdx.x; i < K * K; i += LP_TP) sw[i] = w[i];
    for (int i = threadIdx.x; i < K; i += LP_TP) sb[i] = b ? b[i] : 0.f;
    for (int64_t p0 = (int64_t)blockIdx.x * LP_TP; p0 < P; p0 += (int64_t)gridDim.x * LP_TP) {
        const int np = (int)((P - p0 < LP_TP) ? P - p0 : LP_TP);
        __syncthreads();
        for (int i = threadIdx.x; i < np * K; i += LP_TP) tile[i] = x[p0 * K + i];
        __syncthreads();
        float out[LP_KMAX];
        if ((int)threadIdx.x < np) {
            const float* r = tile + threadIdx.x * K;
            for (int j = 0; j < K; ++j) {
                float acc = sb[j];
                for (int i = 0; i < K; ++i) acc += sw[j * K + i] * r[i];
                out[j] = acc;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < np)
            for (int j = 0; j < K; ++j) tile[threadIdx.x * K + j] = out[j];
        __syncthreads();
        for (int i = threadIdx.x; i < np * K; i += LP_TP) y[p0 * K + i] = tile[i];
    }
}

__global__ __launch_bounds__(LP_TP) void probe_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ w, int64_t P, int K, float* __restrict__ gx,
                                                          double* __restrict__ part) {
    extern __shared__ float lp_s[];                      // [K*K] weights, [LP_TP * K] x tile, [LP_TP * K] g tile (gx written over it)
    float* sw = lp_s; float* tx = sw + K * K; float* tg = tx + LP_TP * K;
    for (int i = threadIdx.x; i < K * K; i += LP_TP) sw[i] = w[i];
    const int nacc = K * K + K;                          // pair t < K*K: (j, i) = (t / K, t % K); t >= K*K: bias j = t - K*K
    double acc[(LP_KMAX * LP_KMAX + LP_KMAX + LP_TP - 1) / LP_TP];
    for (int a = 0; a < (int)(sizeof(acc) / sizeof(acc[0])); ++a) acc[a] = 0.0;
    for (int64_t p0 = (int64_t)blockIdx.x * LP_TP; p0 < P; p0 += (int64_t)gridDim.x * LP_TP) {
        const int np = (int)((P - p0 < LP_TP) ? P - p0 : LP_TP);
        __syncthreads();
        for (int i = threadIdx.x; i < np * K; i += LP_TP) { tx[i] = x[p0 * K + i]; tg[i] = g[p0 * K + i]; }
        __syncthreads();
        // weight / bias gradient partials of this tile: thread t owns accumulators t, t + 256, ...; pixels in order
        for (int a = 0, t = threadIdx.x; t < nacc; t += LP_TP, ++a) {
            // four interleaved running sums (pixels p % 4), joined in a fixed order: independent LDS reads in flight
            const bool pair = t < K * K;
            const int j = pair ? t / K : t - K * K;
            const float* pg = tg + j;
            const float* px = pair ? tx + (t - j * K) : nullptr;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int p = 0;
            if (pair) {
                for (; p + 4 <= np; p += 4) {
                    s0 += pg[(p + 0) * K] * px[(p + 0) * K];
                    s1 += pg[(p + 1) * K] * px[(p + 1) * K];
                    s2 += pg[(p + 2) * K] * px[(p + 2) * K];
                    s3 += pg[(p + 3) * K] * px[(p + 3) * K];
                }
                for (; p < np; ++p) s0 += pg[p * K] * px[p * K];
            } else {
                for (; p + 4 <= np; p += 4) {
                    s0 += pg[(p + 0) * K]; s1 += pg[(p + 1) * K]; s2 += pg[(p + 2) * K]; s3 += pg[(p + 3) * K];
                }
                for (; p < np; ++p) s0 += pg[p * K];
            }
            acc[a] += (double)((s0 + s1) + (s2 + s3));
        }
        if (!gx) continue;                               // frozen producer (the linear-probe protocol): no input gradient
        float out[LP_KMAX];
        if ((int)threadIdx.x < np) {
            const float* r = tg + threadIdx.x * K;
            for (int i = 0; i < K; ++i) {
                float v = 0.f;
                for (int j = 0; j < K; ++j) v += sw[j * K + i] * r[j];
                out[i] = v;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < np)
            for (int i = 0; i < K; ++i) tg[threadIdx.x * K + i] = out[i];
        __syncthreads();
        for (int i = threadIdx.x; i < np * K; i += LP_TP) gx[p0 * K + i] = tg[i];
    }
    for (int a = 0, t = threadIdx.x; t < nacc; t += LP_TP, ++a) part[(size_t)blockIdx.x * nacc + t] = acc[a];
}

// one wave per accumulator: lane l adds rows l, l + 64, ... in order, then a fixed butterfly over the 64 lanes
__global__ __launch_bounds__(256) void probe_bwd_finalize_kernel(const double* __restrict__ part, int rows, int K,
                                                                 float* __restrict__ gw, float* __restrict__ gb) {
    const int nacc = K * K + K;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= nacc) return;
    double s = 0.0;
    for (int r = lane; r < rows; r += 64) s += part[(size_t)r * nacc + t];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) { if (t < K * K) gw[t] = (float)s; else gb[t - K * K] = (float)s; }
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


size_t oess_linear_probe_partials_bytes(int K) { return (K > 0 && K <= LP_KMAX) ? (size_t)LP_MAX_ROWS * (K * K + K) * sizeof(double) : 0; }

int oess_linear_probe_fwd_f32(const float* x, const float* w, const float* bias, long long P, int K, float* y, oess_stream_t stream) {
    if (!x || !w || !y || P <= 0 || K <= 0 || K > LP_KMAX) return OESS_EINVAL;
    long long g = (P + LP_TP - 1) / LP_TP;
    if (g > 4096) g = 4096;
    const size_t lds = ((size_t)K * K + K + (size_t)LP_TP * K) * sizeof(float);
    hipLaunchKernelGGL(probe_fwd_kernel, dim3((unsigned)g), dim3(LP_TP), lds, (hipStream_t)stream, x, w, bias, (int64_t)P, K, y);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_linear_probe_bwd_f32(const float* x, const float* grad_y, const float* w, long long P, int K, float* grad_x, float* grad_w,
                              float* grad_bias, void* partials, size_t partials_bytes, oess_stream_t stream) {
    if (!x || !grad_y || !w || !grad_w || !grad_bias || !partials || P <= 0 || K <= 0 || K > LP_KMAX) return OESS_EINVAL;
    if (partials_bytes < oess_linear_probe_partials_bytes(K)) return OESS_ENOMEM;
    long long g = (P + LP_TP - 1) / LP_TP;
    if (g > LP_MAX_ROWS) g = LP_MAX_ROWS;
    const size_t lds = ((size_t)K * K + (size_t)2 * LP_TP * K) * sizeof(float);
    if (lds > 64 * 1024) {          // K = 31, 32: 67-70 KB, above the 64 KB default dynamic-LDS limit
        // the attribute is PER DEVICE: once per device of this process (a process-wide once-flag left every device but the first at 64 KB)
        static bool set_on[64] = {false};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !set_on[dev]) {
            (void)hipFuncSetAttribute((const void*)&probe_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (dev >= 0 && dev < 64) set_on[dev] = true;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(probe_bwd_kernel, dim3((unsigned)g), dim3(LP_TP), lds, st, x, grad_y, w, (int64_t)P, K, grad_x, (double*)partials);
    hipLaunchKernelGGL(probe_bwd_finalize_kernel, dim3((unsigned)((K * K + K + 3) / 4)), dim3(256), 0, st, (const double*)partials, (int)g, K, grad_w, grad_bias);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // extern "C"
