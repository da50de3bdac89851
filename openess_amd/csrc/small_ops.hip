// Small kernels that replace the last library (ATen / MIOpen / hipBLASLt) launches on the DeepLabv3 training path (gfx950):
//   * MaxPool2d(3, stride 2, padding 1) forward / backward on NHWC bf16 (ResNet stem, models/_resnet.py:124 of the reference)
//   * Dropout forward / backward (ASPP projection, models/deeplabv3.py:343 of the reference): counter-based Philox mask, recomputed
//     in the backward pass instead of stored
//   * ASPP image-pooling branch (models/deeplabv3.py:305-316): 1x1 conv on the B x Cin pooled vector + BatchNorm(train) over the
//     B samples + ReLU, forward and backward -- a B-row GEMV, not a GEMM
// All HBM / latency-bound, 16-byte accesses, no atomics (bit-repeatable).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "oess.h"
#include "oess_common.h"

namespace {
using namespace oess;
constexpr int THREADS = 256;

union V8 { uint4 q; uint16_t h[8]; };
union B8 { uint2 q; uint8_t b[8]; };

// ------------------------------------------------------------------------------------------------ MaxPool2d(3, 2, 1)
// Window of output (oy, ox): input rows 2 oy - 1 .. 2 oy + 1, columns 2 ox - 1 .. 2 ox + 1 (out-of-range taps skipped = -inf
// padding).  Tie rule of ATen's kernel: scan in row-major order, take a value if it is GREATER than the running maximum or NaN,
// the running index starts at the first valid tap.  idx (optional): tap number 0..8 of the winner, for the backward pass.
__global__ __launch_bounds__(THREADS) void maxpool3x3s2_fwd_kernel(const uint16_t* __restrict__ in, int64_t ips, int B, int H, int W, int C,
                                                                   int Ho, int Wo, uint16_t* __restrict__ out, int64_t ops,
                                                                   uint8_t* __restrict__ idx) {
    const int c8 = C >> 3;
    const int64_t total = (int64_t)B * Ho * Wo * c8;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
        const int c0 = (int)(i % c8) * 8;
        int64_t p = i / c8;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float mx[8]; uint8_t mi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { mx[k] = -__builtin_inff(); mi[k] = 0; }
        bool first = true;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                V8 v;
                v.q = *reinterpret_cast<const uint4*>(in + (((int64_t)b * H + iy) * W + ix) * ips + c0);
                const uint8_t tap = (uint8_t)(dy * 3 + dx);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float f = bf16_to_f32(v.h[k]);
                    if (first) mi[k] = tap;
                    if (f > mx[k] || f != f) { mx[k] = f; mi[k] = tap; }
                }
                first = false;
            }
        }
        V8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o.h[k] = f32_to_bf16(mx[k]);        // exact: the maximum is one of the bf16 inputs
        const int64_t po = ((int64_t)b * Ho + oy) * Wo + ox;
        *reinterpret_cast<uint4*>(out + po * ops + c0) = o.q;
        if (idx) {
            B8 m;
#pragma unroll
            for (int k = 0; k < 8; ++k) m.b[k] = mi[k];
            *reinterpret_cast<uint2*>(idx + po * C + c0) = m.q;
        }
    }
}

// dx[b, y, x, c] = sum over the (at most four) windows that contain (y, x) of dy[window, c] where that window's winner is
// this pixel: a gather, every input pixel written once, fixed summation order (oy, then ox).
__global__ __launch_bounds__(THREADS) void maxpool3x3s2_bwd_kernel(const uint16_t* __restrict__ dy, int64_t dps, const uint8_t* __restrict__ idx,
                                                                   int B, int H, int W, int C, int Ho, int Wo,
                                                                   uint16_t* __restrict__ dx, int64_t xps) {
    const int c8 = C >> 3;
    const int64_t total = (int64_t)B * H * W * c8;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
        const int c0 = (int)(i % c8) * 8;
        int64_t p = i / c8;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        const int oy_hi = min((y + 1) >> 1, Ho - 1), ox_hi = min((x + 1) >> 1, Wo - 1);
        for (int oy = y >> 1; oy <= oy_hi; ++oy)
            for (int ox = x >> 1; ox <= ox_hi; ++ox) {
                const uint8_t tap = (uint8_t)((y - (2 * oy - 1)) * 3 + (x - (2 * ox - 1)));
                const int64_t po = ((int64_t)b * Ho + oy) * Wo + ox;
                B8 m; V8 g;
                m.q = *reinterpret_cast<const uint2*>(idx + po * C + c0);
                g.q = *reinterpret_cast<const uint4*>(dy + po * dps + c0);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += (m.b[k] == tap) ? bf16_to_f32(g.h[k]) : 0.f;
            }
        *reinterpret_cast<uint4*>(dx + (((int64_t)b * H + y) * W + x) * xps + c0) = pack_bf16x8(acc);
    }
}

// ------------------------------------------------------------------------------------------------ dropout
// Philox-4x32-10 (Salmon et al. 2011): counter = (element group, call offset), key = seed.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}
// y = x * keep / (1 - p) with keep ~ Bernoulli(1 - thr / 65536); the same (seed, offset) reproduces the mask, so the backward
// pass is this kernel applied to the gradient.  8 channels (16 bytes) per thread, one Philox call per thread.
__global__ __launch_bounds__(THREADS) void dropout_kernel(const uint16_t* __restrict__ x, int64_t xps, uint16_t* __restrict__ y, int64_t yps,
                                                          int64_t P, int C, unsigned thr, float scale, unsigned long long seed,
                                                          unsigned long long offset) {
    const int c8 = C >> 3;
    const int64_t total = P * c8;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
        const int64_t p = i / c8;
        const int c0 = (int)(i - p * c8) * 8;
        const uint4 r = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
        V8 v;
        v.q = *reinterpret_cast<const uint4*>(x + p * xps + c0);
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned u = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
            f[k] = (u >= thr) ? bf16_to_f32(v.h[k]) * scale : 0.f;
        }
        *reinterpret_cast<uint4*>(y + p * yps + c0) = pack_bf16x8(f);
    }
}

// ------------------------------------------------------------------------------------------------ ASPP pooling branch
// z[b][c] = relu(BN_train_over_b(sum_k w[c][k] pooled[b][k])): one wave per output channel, lanes split k, B <= 16 samples.
constexpr int AP_MAXB = 16;
__global__ __launch_bounds__(THREADS) void aspp_pool_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ w,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ run_mean, float* __restrict__ run_var, float momentum,
                                                                float eps, float in_scale, int B, int Cin, int Cout,
                                                                float* __restrict__ y_pre, float* __restrict__ stat, float* __restrict__ z,
                                                                uint16_t* __restrict__ z_bf16) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    if (c >= Cout) return;
    float acc[AP_MAXB];
#pragma unroll
    for (int b = 0; b < AP_MAXB; ++b) acc[b] = 0.f;
    for (int k = lane * 4; k < Cin; k += 256) {                     // Cin % 4 == 0 (host-checked)
        const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)c * Cin + k);
#pragma unroll
        for (int b = 0; b < AP_MAXB; ++b)
            if (b < B) {
                const float4 xv = *reinterpret_cast<const float4*>(pooled + (size_t)b * Cin + k);
                acc[b] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
            }
    }
#pragma unroll
    for (int b = 0; b < AP_MAXB; ++b) acc[b] = wave_sum(acc[b]) * in_scale;     // pooled = in_scale * (per-sample channel sums)
    if (lane != 0) return;
    double s = 0.0, ss = 0.0;
    for (int b = 0; b < B; ++b) s += acc[b];
    const double mean = s / B;
    for (int b = 0; b < B; ++b) { const double d = acc[b] - mean; ss += d * d; }
    const double var = ss / B;                                      // biased: what normalises; unbiased: what the running estimate keeps
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma[c], bt = beta[c];
    for (int b = 0; b < B; ++b) {
        y_pre[(size_t)b * Cout + c] = acc[b];
        const float v = (acc[b] - (float)mean) * rstd * g + bt;
        z[(size_t)b * Cout + c] = v > 0.f ? v : 0.f;
        if (z_bf16) z_bf16[(size_t)b * Cout + c] = f32_to_bf16(v > 0.f ? v : 0.f);
    }
    stat[c] = (float)mean; stat[Cout + c] = rstd;
    if (run_mean) {
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(B > 1 ? ss / (B - 1) : var);
    }
}

// per channel: ReLU mask, BatchNorm backward over the B samples -> dy[b][c], dgamma[c], dbeta[c]
__global__ __launch_bounds__(THREADS) void aspp_pool_bn_bwd_kernel(const float* __restrict__ gz, const float* __restrict__ y_pre,
                                                                   const float* __restrict__ stat, const float* __restrict__ z,
                                                                   const float* __restrict__ gamma, int B, int Cout,
                                                                   float* __restrict__ dy, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * THREADS + threadIdx.x;
    if (c >= Cout) return;
    const float mean = stat[c], rstd = stat[Cout + c];
    double db = 0.0, dg = 0.0;
    for (int b = 0; b < B; ++b) {
        const float g = z[(size_t)b * Cout + c] > 0.f ? gz[(size_t)b * Cout + c] : 0.f;
        db += g; dg += (double)g * ((y_pre[(size_t)b * Cout + c] - mean) * rstd);
    }
    dgamma[c] = (float)dg; dbeta[c] = (float)db;
    const float k = gamma[c] * rstd, mdb = (float)(db / B), mdg = (float)(dg / B);
    for (int b = 0; b < B; ++b) {
        const float g = z[(size_t)b * Cout + c] > 0.f ? gz[(size_t)b * Cout + c] : 0.f;
        const float xh = (y_pre[(size_t)b * Cout + c] - mean) * rstd;
        dy[(size_t)b * Cout + c] = k * (g - mdb - xh * mdg);
    }
}
// blocks [0, nW): dW[c][k] = in_scale sum_b dy[b][c] sums[b][k];   blocks [nW, ..): dsums[b][k] = in_scale sum_c dy[b][c] w[c][k] (bf16)
__global__ __launch_bounds__(THREADS) void aspp_pool_gemv_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pooled,
                                                                     const float* __restrict__ w, float in_scale, int B, int Cin, int Cout,
                                                                     int nW, float* __restrict__ dw, uint16_t* __restrict__ dpooled) {
    if ((int)blockIdx.x < nW) {
        const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
        if (i >= (int64_t)Cout * Cin) return;
        const int c = (int)(i / Cin), k = (int)(i - (int64_t)c * Cin);
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += dy[(size_t)b * Cout + c] * pooled[(size_t)b * Cin + k];
        dw[i] = a * in_scale;
    } else if (dpooled) {
        const int64_t i = (int64_t)(blockIdx.x - nW) * THREADS + threadIdx.x;
        if (i >= (int64_t)B * Cin) return;
        const int b = (int)(i / Cin), k = (int)(i - (int64_t)b * Cin);
        float a = 0.f;
        for (int c = 0; c < Cout; ++c) a += dy[(size_t)b * Cout + c] * w[(size_t)c * Cin + k];
        dpooled[i] = f32_to_bf16(a * in_scale);
    }
}

int grid_for(int64_t items) {
    int64_t g = (items + THREADS - 1) / THREADS;
    const int64_t cap = (int64_t)num_cus() * 16;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" {

int oess_maxpool3x3s2_fwd_nhwc_bf16(const void* in, long long in_pix_stride, int B, int H, int W, int C, void* out, long long out_pix_stride,
                                    unsigned char* idx, oess_stream_t stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (in_pix_stride & 7) || (out_pix_stride & 7) || in_pix_stride < C ||
        out_pix_stride < C || ((uintptr_t)in & 15) || ((uintptr_t)out & 15) || ((uintptr_t)idx & 7))
        return OESS_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel, dim3(grid_for((int64_t)B * Ho * Wo * (C >> 3))), dim3(THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)in, (int64_t)in_pix_stride, B, H, W, C, Ho, Wo, (uint16_t*)out, (int64_t)out_pix_stride, idx);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_maxpool3x3s2_bwd_nhwc_bf16(const void* grad_out, long long go_pix_stride, const unsigned char* idx, int B, int H, int W, int C,
                                    void* grad_in, long long gi_pix_stride, oess_stream_t stream) {
    if (!grad_out || !idx || !grad_in || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (go_pix_stride & 7) || (gi_pix_stride & 7) ||
        go_pix_stride < C || gi_pix_stride < C || ((uintptr_t)grad_out & 15) || ((uintptr_t)grad_in & 15) || ((uintptr_t)idx & 7))
        return OESS_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(grid_for((int64_t)B * H * W * (C >> 3))), dim3(THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)grad_out, (int64_t)go_pix_stride, idx, B, H, W, C, Ho, Wo, (uint16_t*)grad_in, (int64_t)gi_pix_stride);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_dropout_nhwc_bf16(const void* x, long long x_pix_stride, void* y, long long y_pix_stride, long long P, int C, float p,
                           unsigned long long seed, unsigned long long offset, oess_stream_t stream) {
    if (!x || !y || P <= 0 || C <= 0 || (C & 7) || (x_pix_stride & 7) || (y_pix_stride & 7) || x_pix_stride < C || y_pix_stride < C ||
        !(p >= 0.f && p < 1.f) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
        return OESS_EINVAL;
    const unsigned thr = (unsigned)(p * 65536.0f + 0.5f);
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(P * (C >> 3))), dim3(THREADS), 0, (hipStream_t)stream, (const uint16_t*)x,
                       (int64_t)x_pix_stride, (uint16_t*)y, (int64_t)y_pix_stride, (int64_t)P, C, thr, 1.0f / (1.0f - p), seed, offset);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_aspp_pool_fwd_f32(const float* pooled, float in_scale, const float* w, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float momentum, float eps, int B, int Cin, int Cout, float* y_pre, float* stat, float* z,
                           void* z_bf16, oess_stream_t stream) {
    if (!pooled || !w || !gamma || !beta || !y_pre || !stat || !z || B < 2 || B > AP_MAXB || Cin <= 0 || (Cin & 3) || Cout <= 0 ||
        (!running_mean) != (!running_var))
        return OESS_EINVAL;
    hipLaunchKernelGGL(aspp_pool_fwd_kernel, dim3((Cout + 3) / 4), dim3(THREADS), 0, (hipStream_t)stream, pooled, w, gamma, beta, running_mean,
                       running_var, momentum, eps, in_scale, B, Cin, Cout, y_pre, stat, z, (uint16_t*)z_bf16);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_aspp_pool_bwd_f32(const float* grad_z, const float* pooled, float in_scale, const float* w, const float* gamma, const float* y_pre,
                           const float* stat, const float* z, int B, int Cin, int Cout, float* dy_scratch, float* grad_w,
                           float* grad_gamma, float* grad_beta, void* grad_pooled_bf16, oess_stream_t stream) {
    if (!grad_z || !pooled || !w || !gamma || !y_pre || !stat || !z || !dy_scratch || !grad_w || !grad_gamma || !grad_beta || B < 2 ||
        B > AP_MAXB || Cin <= 0 || Cout <= 0)
        return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(aspp_pool_bn_bwd_kernel, dim3((Cout + THREADS - 1) / THREADS), dim3(THREADS), 0, st, grad_z, y_pre, stat, z, gamma, B,
                       Cout, dy_scratch, grad_gamma, grad_beta);
    const int nW = (int)(((int64_t)Cout * Cin + THREADS - 1) / THREADS);
    const int nP = grad_pooled_bf16 ? (int)(((int64_t)B * Cin + THREADS - 1) / THREADS) : 0;
    hipLaunchKernelGGL(aspp_pool_gemv_bwd_kernel, dim3(nW + nP), dim3(THREADS), 0, st, (const float*)dy_scratch, pooled, w, in_scale, B, Cin, Cout,
                       nW, grad_w, (uint16_t*)grad_pooled_bf16);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // extern "C"
