// Superpixel mean of a bilinearly UPSAMPLED feature map through its pooling matrix.
//   deeplabv3_resnet50.forward: feats = F.interpolate(feats, size=input, bilinear, align_corners=False)   models/deeplabv3.py:184
//   training/pretrain_trainer.py:445-465 (frame2recon): k = scatter_mean(feats, superpixels)
// Both maps are linear, so for a low-resolution map y [B x h x w x C] and full-resolution ids [B x Ho x Wo]
//   k[s] = (sum_q M[s][q] y[q]) / (n[s] + 1e-6),   M[s][q] = sum over the pixels p of superpixel row s of the bilinear weight w(p, q),
// and the backward is  dy[q] = sum_s M[s][q] gk[s] / (n[s] + 1e-6).  M depends on the ids and the geometry only: it is built once per
// step from one pass over the ids (18 MB at 8 x 440 x 640) and serves the forward, the backward and every map pooled over the
// same superpixels; the 8 x 256 x 440 x 640 feature tensor (1.15 GB in bf16), its scatter, its gradient and the bilinear adjoint
// are never formed.  At output stride 16 a column q is a 28 x 40 cell: M is [S <= 800] x [B h w = 8 960].
// Determinism: the weights of a run of pixels are added in pixel order in fp32, every run's four products enter M as 2^-40
// fixed-point integers (integer atomics: the sum does not depend on the order), the two products walk q / s in ascending order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "oess.h"
#include "oess_common.h"
#include "bilinear_axis.h"

namespace {
using namespace oess;
constexpr int THREADS = 256;
typedef unsigned long long u64_t;
constexpr double PM_FIX = 1099511627776.0;            // 2^40: a weight sum is at most the pixel count of a superpixel (< 2^23)
constexpr float PM_UNFIX = 1.0f / 1099511627776.0f;

__device__ __forceinline__ void pm_add(u64_t* dst, float v) {
    const long long q = __double2ll_rn((double)v * PM_FIX);
    if (q != 0) atomicAdd(dst, (u64_t)q);
}
__device__ __forceinline__ float pm_value(u64_t m) { return (float)m * PM_UNFIX; }

// work item = (sample, output row oy, input column ix): the output pixels of that row whose LEFT corner is ix, merged by id
__global__ __launch_bounds__(THREADS) void poolmat_scatter_kernel(const int64_t* __restrict__ ids, int B, int sps, int S, Axis ay, Axis ax,
                                                                  u64_t* __restrict__ M, int* __restrict__ cnt) {
    const int w = ax.in, h = ay.in;
    const int64_t BQ = (int64_t)B * h * w;
    const int64_t items = (int64_t)B * ay.out * w;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < items; i += (int64_t)gridDim.x * THREADS) {
        const int ix = (int)(i % w);
        const int64_t t = i / w;                               // b * Ho + oy
        const int b = (int)(t / ay.out), oy = (int)(t - (int64_t)b * ay.out);
        int y0, y1; float ly;
        src_index(ay, oy, y0, y1, ly);
        const int x1c = (ix < w - 1) ? ix + 1 : ix;
        int lo, hi;
        candidates(ax, ix, lo, hi);
        int64_t cur = 0;
        float a = 0.f, bq = 0.f;
        int n = 0;
        auto flush = [&]() {
            if (n == 0) return;
            const int64_t gid = cur + (int64_t)b * sps;
            if (gid >= 0 && gid < S) {
                u64_t* row = M + gid * BQ + (int64_t)b * h * w;
                pm_add(row + y0 * w + ix, (1.0f - ly) * a);
                pm_add(row + y0 * w + x1c, (1.0f - ly) * bq);
                pm_add(row + y1 * w + ix, ly * a);
                pm_add(row + y1 * w + x1c, ly * bq);
                atomicAdd(cnt + gid, n);
            }
        };
        for (int ox = lo; ox <= hi; ++ox) {
            int x0, x1; float lx;
            src_index(ax, ox, x0, x1, lx);
            if (x0 != ix) continue;
            const int64_t id = ids[t * ax.out + ox];
            if (n != 0 && id != cur) { flush(); a = 0.f; bq = 0.f; n = 0; }
            cur = id;
            a += 1.0f - lx;
            bq += lx;
            n += 1;
        }
        flush();
    }
}

template <bool BF16>
__device__ __forceinline__ float pm_load(const void* base, int64_t idx) {
    if (BF16) return bf16_to_f32(((const uint16_t*)base)[idx]);
    return ((const float*)base)[idx];
}

// one workgroup per row s; thread = channel (+ THREADS, ...); 256-column chunks of the row staged in LDS, all-zero chunks skipped
template <bool BF16>
__global__ __launch_bounds__(THREADS) void poolmat_fwd_kernel(const u64_t* __restrict__ M, const int* __restrict__ cnt, const void* __restrict__ y,
                                                              int64_t ys, int64_t BQ, int C, float* __restrict__ k, float* __restrict__ count) {
    __shared__ float mrow[THREADS];
    const int s = blockIdx.x;
    const u64_t* row = M + (int64_t)s * BQ;
    constexpr int CPT = 4;                                  // channels per thread: C <= 1024
    float acc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[j] = 0.f;
    for (int64_t q0 = 0; q0 < BQ; q0 += THREADS) {
        const int64_t q = q0 + threadIdx.x;
        const float m = (q < BQ) ? pm_value(row[q]) : 0.f;
        __syncthreads();                                    // the previous chunk has been read
        mrow[threadIdx.x] = m;
        if (!__syncthreads_or(m != 0.f)) continue;
        const int nq = (BQ - q0 < THREADS) ? (int)(BQ - q0) : THREADS;
        for (int qq = 0; qq < nq; ++qq) {
            const float mv = mrow[qq];
            if (mv == 0.f) continue;
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int c = threadIdx.x + j * THREADS;
                if (c < C) acc[j] += mv * pm_load<BF16>(y, (q0 + qq) * ys + c);
            }
        }
    }
    const float cn = (float)cnt[s];
    const float d = __fadd_rn(cn, 1e-6f);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = threadIdx.x + j * THREADS;
        if (c < C) k[(int64_t)s * C + c] = acc[j] / d;
    }
    if (threadIdx.x == 0) count[s] = cn;
}

// one workgroup per column q; 256-row chunks of the column staged in LDS (strided reads), all-zero chunks skipped
template <bool BF16>
__global__ __launch_bounds__(THREADS) void poolmat_bwd_kernel(const u64_t* __restrict__ M, const int* __restrict__ cnt, const float* __restrict__ gk,
                                                              int64_t BQ, int C, int S, void* __restrict__ gy, int64_t gys) {
    __shared__ float mcol[THREADS];
    const int64_t q = blockIdx.x;
    constexpr int CPT = 4;
    float acc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[j] = 0.f;
    for (int s0 = 0; s0 < S; s0 += THREADS) {
        const int s = s0 + threadIdx.x;
        const float m = (s < S) ? pm_value(M[(int64_t)s * BQ + q]) : 0.f;
        __syncthreads();
        mcol[threadIdx.x] = m;
        if (!__syncthreads_or(m != 0.f)) continue;
        const int ns = (S - s0 < THREADS) ? S - s0 : THREADS;
        for (int ss = 0; ss < ns; ++ss) {
            const float mv = mcol[ss];
            if (mv == 0.f) continue;
            const float d = __fadd_rn((float)cnt[s0 + ss], 1e-6f);
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int c = threadIdx.x + j * THREADS;
                if (c < C) acc[j] += mv * (gk[(int64_t)(s0 + ss) * C + c] / d);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = threadIdx.x + j * THREADS;
        if (c < C) {
            if (BF16) ((uint16_t*)gy)[q * gys + c] = f32_to_bf16(acc[j]);
            else ((float*)gy)[q * gys + c] = acc[j];
        }
    }
}

size_t pm_matrix_words(int S, int B, int h, int w) { return (size_t)S * B * h * w; }

}  // namespace

extern "C" {

size_t oess_pool_matrix_bytes(int S, int B, int h, int w) {
    if (S <= 0 || B <= 0 || h <= 0 || w <= 0) return 0;
    return pm_matrix_words(S, B, h, w) * sizeof(u64_t) + align_up((size_t)S * sizeof(int), 16);
}

int oess_pool_matrix_build(const int64_t* ids, int B, int Ho, int Wo, int h, int w, int align_corners, int superpixel_size, int S,
                           void* matrix, size_t matrix_bytes, oess_stream_t stream) {
    if (!ids || !matrix || B <= 0 || Ho <= 0 || Wo <= 0 || h <= 0 || w <= 0 || superpixel_size <= 0 || S <= 0 || ((uintptr_t)matrix & 15))
        return OESS_EINVAL;
    const size_t need = oess_pool_matrix_bytes(S, B, h, w);
    if (matrix_bytes < need) return OESS_ENOMEM;
    hipStream_t st = (hipStream_t)stream;
    OESS_HIP(hipMemsetAsync(matrix, 0, need, st));
    u64_t* M = (u64_t*)matrix;
    int* cnt = (int*)(M + pm_matrix_words(S, B, h, w));
    const Axis ay = make_axis(h, Ho, align_corners), ax = make_axis(w, Wo, align_corners);
    const int64_t items = (int64_t)B * Ho * w;
    int64_t g = (items + THREADS - 1) / THREADS;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(poolmat_scatter_kernel, dim3((unsigned)g), dim3(THREADS), 0, st, ids, B, superpixel_size, S, ay, ax, M, cnt);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_pool_matrix_fwd(const void* matrix, const void* y, long long y_pix_stride, int is_bf16, int B, int h, int w, int C, int S,
                         float* k, float* count, oess_stream_t stream) {
    if (!matrix || !y || !k || !count || B <= 0 || h <= 0 || w <= 0 || C <= 0 || C > 4 * THREADS || S <= 0 || y_pix_stride < C) return OESS_EINVAL;
    const u64_t* M = (const u64_t*)matrix;
    const int* cnt = (const int*)(M + pm_matrix_words(S, B, h, w));
    const int64_t BQ = (int64_t)B * h * w;
    hipStream_t st = (hipStream_t)stream;
    if (is_bf16) hipLaunchKernelGGL(poolmat_fwd_kernel<true>, dim3((unsigned)S), dim3(THREADS), 0, st, M, cnt, y, (int64_t)y_pix_stride, BQ, C, k, count);
    else hipLaunchKernelGGL(poolmat_fwd_kernel<false>, dim3((unsigned)S), dim3(THREADS), 0, st, M, cnt, y, (int64_t)y_pix_stride, BQ, C, k, count);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_pool_matrix_bwd(const void* matrix, const float* grad_k, int B, int h, int w, int C, int S, void* grad_y, long long gy_pix_stride,
                         int is_bf16, oess_stream_t stream) {
    if (!matrix || !grad_k || !grad_y || B <= 0 || h <= 0 || w <= 0 || C <= 0 || C > 4 * THREADS || S <= 0 || gy_pix_stride < C) return OESS_EINVAL;
    const u64_t* M = (const u64_t*)matrix;
    const int* cnt = (const int*)(M + pm_matrix_words(S, B, h, w));
    const int64_t BQ = (int64_t)B * h * w;
    if (BQ > 0x7fffffffLL) return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (is_bf16) hipLaunchKernelGGL(poolmat_bwd_kernel<true>, dim3((unsigned)BQ), dim3(THREADS), 0, st, M, cnt, grad_k, BQ, C, S, grad_y, (int64_t)gy_pix_stride);
    else hipLaunchKernelGGL(poolmat_bwd_kernel<false>, dim3((unsigned)BQ), dim3(THREADS), 0, st, M, cnt, grad_k, BQ, C, S, grad_y, (int64_t)gy_pix_stride);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // extern "C"
