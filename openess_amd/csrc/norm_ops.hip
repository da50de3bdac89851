// Normalisation / resampling kernels on NHWC bf16 activations for gfx950 (all HBM-bound streaming
// kernels: 16-byte accesses, fp32 statistics, one atomic per (workgroup, channel)).
//   - per-channel statistics over G groups (G = 1: BatchNorm batch statistics; G = B: InstanceNorm)
//   - finalize (mean / rstd / affine -> scale, shift; BatchNorm running-stat update)
//   - apply  y = act(x*scale + shift [+ residual])
//   - InstanceNorm(+ReLU) backward: statistics pass + apply pass
//   - nearest x2 upsample into a channel slice (fwd) and its adjoint (2x2 sum)
//   - bilinear x4 (align_corners=True) fused with the L2 channel normalisation of the teacher head
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "oess.h"
#include "oess_common.h"

namespace {
using namespace oess;
constexpr int THREADS = 256;

union Pack8 { uint4 q; uint16_t h[8]; };

// ---------------------------------------------------------------------------------------------
// statistics: per-workgroup partial sums part[chunk][2][G*C] over the chunk's pixels of group g.  grid = (chunks, G).
// NO atomics: every workgroup owns its row of the partials buffer and partials_reduce_kernel adds the rows in a fixed order
// in double, so the statistics (and everything downstream of them) are bit-repeatable from run to run.
// thread layout: cl = C/8 channel-lanes per pixel (power of two not required), rows = THREADS/cl pixels in flight.
// MODE 0: plain sums of x.   MODE 1 (InstanceNorm backward): s1 = sum g, s2 = sum g*xhat with
//         xhat = (x-mean)*rstd, g = dy * (relu ? xhat > 0 : 1).
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(THREADS) void stats_kernel(const uint16_t* __restrict__ x, int64_t xps,
                                                        const uint16_t* __restrict__ dy, int64_t dps,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        int relu, int64_t ppg, int C, float* __restrict__ part,
                                                        const uint16_t* __restrict__ yout = nullptr, int64_t yps = 0) {
    extern __shared__ float red[];                 // [rows][C][2]
    const int cl = C >> 3;
    const int rows = THREADS / cl;
    const int lane_c = threadIdx.x % cl, row = threadIdx.x / cl;
    const int g = blockIdx.y;
    const int64_t ppw = (ppg + gridDim.x - 1) / gridDim.x;           // pixels per workgroup
    const int64_t p_beg = (int64_t)blockIdx.x * ppw;
    int64_t p_end = p_beg + ppw;
    if (p_end > ppg) p_end = ppg;
    float a1[8], a2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a1[k] = 0.f; a2[k] = 0.f; }
    float mu[8], rs[8];
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { mu[k] = mean[(int64_t)g * C + lane_c * 8 + k]; rs[k] = rstd[(int64_t)g * C + lane_c * 8 + k]; }
    }
    if (row < rows) {
        int64_t p = p_beg + row;
        if (MODE == 0) {
            // four independent 16-byte loads in flight per lane (one per iteration left the kernel latency bound: 2 TB/s)
            for (; p + 3 * rows < p_end; p += 4 * rows) {
                Pack8 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u].q = *reinterpret_cast<const uint4*>(x + ((int64_t)g * ppg + p + u * rows) * xps + lane_c * 8);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const float f = bf16_to_f32(v[u].h[k]); a1[k] += f; a2[k] += f * f; }
            }
        } else {
            for (; p + rows < p_end; p += 2 * rows) {
                Pack8 v[2], d[2], yo[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int64_t pix = (int64_t)g * ppg + p + u * rows;
                    v[u].q = *reinterpret_cast<const uint4*>(x + pix * xps + lane_c * 8);
                    d[u].q = *reinterpret_cast<const uint4*>(dy + pix * dps + lane_c * 8);
                    if (relu && yout) yo[u].q = *reinterpret_cast<const uint4*>(yout + pix * yps + lane_c * 8);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float xh = (bf16_to_f32(v[u].h[k]) - mu[k]) * rs[k];
                        float gg = bf16_to_f32(d[u].h[k]);
                        if (relu && !(yout ? bf16_to_f32(yo[u].h[k]) > 0.f : xh > 0.f)) gg = 0.f;
                        a1[k] += gg; a2[k] += gg * xh;
                    }
            }
        }
        for (; p < p_end; p += rows) {
            const int64_t pix = (int64_t)g * ppg + p;
            Pack8 v;
            v.q = *reinterpret_cast<const uint4*>(x + pix * xps + lane_c * 8);
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float f = bf16_to_f32(v.h[k]); a1[k] += f; a2[k] += f * f; }
            } else {
                Pack8 d, yo;
                d.q = *reinterpret_cast<const uint4*>(dy + pix * dps + lane_c * 8);
                if (relu && yout) yo.q = *reinterpret_cast<const uint4*>(yout + pix * yps + lane_c * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float xh = (bf16_to_f32(v.h[k]) - mu[k]) * rs[k];
                    float gg = bf16_to_f32(d.h[k]);
                    // ReLU mask: from the stored output when there is an affine / residual, else y > 0 <=> xhat > 0
                    if (relu && !(yout ? bf16_to_f32(yo.h[k]) > 0.f : xh > 0.f)) gg = 0.f;
                    a1[k] += gg; a2[k] += gg * xh;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[(row * C + lane_c * 8 + k) * 2 + 0] = a1[k];
            red[(row * C + lane_c * 8 + k) * 2 + 1] = a2[k];
        }
    }
    __syncthreads();
    const int64_t GC = (int64_t)gridDim.y * C;
    for (int c = threadIdx.x; c < C; c += THREADS) {
        float s1 = 0.f, s2 = 0.f;
        for (int r = 0; r < rows; ++r) { s1 += red[(r * C + c) * 2]; s2 += red[(r * C + c) * 2 + 1]; }
        part[((int64_t)blockIdx.x * 2 + 0) * GC + (int64_t)g * C + c] = s1;
        part[((int64_t)blockIdx.x * 2 + 1) * GC + (int64_t)g * C + c] = s2;
    }
}

// mean / rstd / (scale, shift) of one (group, channel) from its double totals; optional BatchNorm running-stat update.
__device__ __forceinline__ void finalize_one(double S, double Q, int i, int c, bool bn_running, float count, float eps,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                             float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                             float* __restrict__ scale, float* __restrict__ shift) {
    const double m = S / (double)count;
    double var = Q / (double)count - m * m;          // biased variance (normalisation uses it in BN and IN); no fp32 cancellation
    if (var < 0.0) var = 0.0;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    mean_out[i] = (float)m; rstd_out[i] = r;
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    scale[i] = ga * r;
    shift[i] = be - (float)m * ga * r;
    if (bn_running) {                                // nn.BatchNorm2d: running_var uses the UNBIASED estimate
        const float unb = count > 1.f ? (float)(var * (double)count / ((double)count - 1.0)) : (float)var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
}

// Fixed-order reduction of stats_kernel's partials part[chunks][2][N] (N = G*C) in double.  block = 16 columns x 16 chunk lanes;
// lane tl adds chunks tl, tl+16, ... in order, the 16 lane sums are added 0..15: the result does not depend on scheduling.
// FIN = false: o1[N] = sum, o2[N] = second sum (backward: d(beta), d(gamma) / the InstanceNorm sums).
// FIN = true : mean / rstd / scale / shift (+ running statistics when G == 1), E[x^2] - E[x]^2 in double.
template <bool FIN>
__global__ __launch_bounds__(256) void partials_reduce_kernel(const float* __restrict__ part, int chunks, int G, int C,
                                                              float* __restrict__ o1, float* __restrict__ o2, float count, float eps,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              float momentum, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                              float* __restrict__ scale, float* __restrict__ shift) {
    // 16 columns x 16 chunk lanes per block (a 512-chunk table is 32 iterations of two loads per lane, unrolled by 4; with 8
    // chunk lanes x 32 columns it was 64 dependent-latency iterations: 12 us per call, 60 calls per frame2recon step)
    __shared__ double red[16][16][2];
    const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int64_t N = (int64_t)G * C;
    const int64_t i = (int64_t)blockIdx.x * 16 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (i < N) {
#pragma unroll 4
        for (int t = tl; t < chunks; t += 16) {
            s1 += (double)part[((int64_t)t * 2) * N + i];
            s2 += (double)part[((int64_t)t * 2 + 1) * N + i];
        }
    }
    red[tl][cl][0] = s1; red[tl][cl][1] = s2;
    __syncthreads();
    if (tl != 0 || i >= N) return;
#pragma unroll
    for (int k = 1; k < 16; ++k) { s1 += red[k][cl][0]; s2 += red[k][cl][1]; }
    if (!FIN) { o1[i] = (float)s1; o2[i] = (float)s2; return; }
    finalize_one(s1, s2, (int)i, (int)(i % C), running_mean != nullptr && G == 1, count, eps, gamma, beta, running_mean, running_var,
                 momentum, mean_out, rstd_out, scale, shift);
}

// Thread layout of both apply kernels: lane_c = channel chunk (8 channels) is FIXED per thread, rows = THREADS / (C/8)
// pixels per block iteration, grid = (pixel chunks, G): no per-element 64-bit divisions, per-channel constants stay in
// registers for the whole loop.  Needs C/8 <= THREADS (C <= 2048).
__device__ __forceinline__ uint4 nt_load16(const uint16_t* p) {
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u4v t = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p));
    return make_uint4(t[0], t[1], t[2], t[3]);
}
template <int NT>
__global__ __launch_bounds__(THREADS) void apply_kernel(const uint16_t* __restrict__ x, int64_t xps,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const uint16_t* __restrict__ res, int64_t rps, int relu,
                                                        int64_t ppg, int G, int C, uint16_t* __restrict__ out, int64_t ops) {
    const int cl = C >> 3;
    const int rows = THREADS / cl;
    const int lane_c = threadIdx.x % cl, row = threadIdx.x / cl;
    if (row >= rows) return;
    const int g = blockIdx.y, c0 = lane_c * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = scale[(int64_t)g * C + c0 + k]; sh[k] = shift[(int64_t)g * C + c0 + k]; }
    const int64_t base = (int64_t)g * ppg;
    for (int64_t p = (int64_t)blockIdx.x * rows + row; p < ppg; p += (int64_t)gridDim.x * rows) {
        const int64_t pix = base + p;
        Pack8 v, r;
        float f[8];
        // NT & 1: the raw conv result is dead after this pass, NT & 2: so is the residual (the block input) -- streaming loads
        if (NT & 1) v.q = nt_load16(x + pix * xps + c0); else v.q = *reinterpret_cast<const uint4*>(x + pix * xps + c0);
        if (res) { if (NT & 2) r.q = nt_load16(res + pix * rps + c0); else r.q = *reinterpret_cast<const uint4*>(res + pix * rps + c0); }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f[k] = bf16_to_f32(v.h[k]) * sc[k] + sh[k];
            if (res) f[k] += bf16_to_f32(r.h[k]);
            if (relu) f[k] = fmaxf(f[k], 0.f);
        }
        if (NT & 4) {
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            const uint4 o = pack_bf16x8(f);
            __builtin_nontemporal_store(u4v{o.x, o.y, o.z, o.w}, reinterpret_cast<u4v*>(out + pix * ops + c0));
        } else
        *reinterpret_cast<uint4*>(out + pix * ops + c0) = pack_bf16x8(f);
    }
}

// Normalisation backward: g = dy * mask;  dx = gamma * rstd * (g - s1/N - xhat * s2/N);  d(residual) = g
// (gamma == nullptr: affine-free InstanceNorm; yout != nullptr: ReLU mask taken from the stored output)
__global__ __launch_bounds__(THREADS) void in_bwd_apply_kernel(const uint16_t* __restrict__ x, int64_t xps,
                                                               const uint16_t* __restrict__ dy, int64_t dps,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ s1, const float* __restrict__ s2,
                                                               int relu, int64_t ppg, int G, int C,
                                                               uint16_t* __restrict__ dx, int64_t gps,
                                                               const uint16_t* __restrict__ yout = nullptr, int64_t yps = 0,
                                                               const float* __restrict__ gamma = nullptr,
                                                               uint16_t* __restrict__ dres = nullptr, int64_t drps = 0) {
    const int cl = C >> 3;
    const int rows = THREADS / cl;
    const int lane_c = threadIdx.x % cl, row = threadIdx.x / cl;
    if (row >= rows) return;
    const int g = blockIdx.y, c0 = lane_c * 8;
    const float invn = 1.0f / (float)ppg;
    float mu[8], rs[8], a1[8], a2[8], gr[8];               // per channel: mean, rstd, s1/N, s2/N, gamma*rstd
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t gc = (int64_t)g * C + c0 + k;
        mu[k] = mean[gc]; rs[k] = rstd[gc]; a1[k] = s1[gc] * invn; a2[k] = s2[gc] * invn;
        gr[k] = (gamma ? gamma[c0 + k] : 1.0f) * rs[k];
    }
    const int64_t base = (int64_t)g * ppg;
    for (int64_t p = (int64_t)blockIdx.x * rows + row; p < ppg; p += (int64_t)gridDim.x * rows) {
        const int64_t pix = base + p;
        Pack8 v, d, yo;
        float od[8], og[8];
        v.q = *reinterpret_cast<const uint4*>(x + pix * xps + c0);
        d.q = *reinterpret_cast<const uint4*>(dy + pix * dps + c0);
        if (relu && yout) yo.q = *reinterpret_cast<const uint4*>(yout + pix * yps + c0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xh = (bf16_to_f32(v.h[k]) - mu[k]) * rs[k];
            float gg = bf16_to_f32(d.h[k]);
            if (relu && !(yout ? bf16_to_f32(yo.h[k]) > 0.f : xh > 0.f)) gg = 0.f;
            od[k] = gr[k] * (gg - a1[k] - xh * a2[k]);
            og[k] = gg;
        }
        *reinterpret_cast<uint4*>(dx + pix * gps + c0) = pack_bf16x8(od);
        if (dres) *reinterpret_cast<uint4*>(dres + pix * drps + c0) = pack_bf16x8(og);
    }
}

// BatchNorm backward, second half, for C % 64 == 0 and <= 512 partial rows: the fixed-order reduction of stats_kernel<1>'s
// partials AND the apply pass in ONE launch (a frame2recon / fine-tune step has 59 BatchNorm backwards: 59 fewer launches).
// Workgroup (64-channel group x, pixel chunk y) first adds the partial rows of ITS 64 channels -- same order as
// partials_reduce_kernel (chunk lane tl adds rows tl, tl + 16, ... in double, the 16 lane sums are added 0..15), so d(beta),
// d(gamma) and dx are bit-identical to the three-launch path; chunk 0 publishes d(beta) / d(gamma).  Redundant work per
// workgroup: chunks x 512 bytes of L2-resident partials.  Then 8 lanes own a pixel's 64 channels (one 16-byte access per tensor).
__global__ __launch_bounds__(THREADS) void bn_bwd_reduce_apply_kernel(const float* __restrict__ part, int chunks, int C,
                                                                      const uint16_t* __restrict__ x, int64_t xps,
                                                                      const uint16_t* __restrict__ dy, int64_t dps,
                                                                      const uint16_t* __restrict__ yout, int64_t yps,
                                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                      const float* __restrict__ gamma, int relu, int64_t pixels, int64_t ppc,
                                                                      float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                                      uint16_t* __restrict__ dx, int64_t gps,
                                                                      uint16_t* __restrict__ dres, int64_t drps) {
    __shared__ double red[16][64][2];
    __shared__ float s_a1[64], s_a2[64];
    const int cg = blockIdx.x * 64;
    {
        const int c4 = (threadIdx.x & 15) * 4, tl = threadIdx.x >> 4;
        double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
        const float* p0 = part + cg + c4;
#pragma unroll 4
        for (int t = tl; t < chunks; t += 16) {
            const float4 u = *reinterpret_cast<const float4*>(p0 + ((size_t)t * 2) * C);
            const float4 w = *reinterpret_cast<const float4*>(p0 + ((size_t)t * 2 + 1) * C);
            a1[0] += (double)u.x; a1[1] += (double)u.y; a1[2] += (double)u.z; a1[3] += (double)u.w;
            a2[0] += (double)w.x; a2[1] += (double)w.y; a2[2] += (double)w.z; a2[3] += (double)w.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[tl][c4 + k][0] = a1[k]; red[tl][c4 + k][1] = a2[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        double s1 = red[0][threadIdx.x][0], s2 = red[0][threadIdx.x][1];
#pragma unroll
        for (int k = 1; k < 16; ++k) { s1 += red[k][threadIdx.x][0]; s2 += red[k][threadIdx.x][1]; }
        const float f1 = (float)s1, f2 = (float)s2;                  // d(beta), d(gamma): the fp32 values the apply pass uses
        s_a1[threadIdx.x] = f1; s_a2[threadIdx.x] = f2;
        if (blockIdx.y == 0) { dbeta[cg + threadIdx.x] = f1; dgamma[cg + threadIdx.x] = f2; }
    }
    __syncthreads();
    const int sub = threadIdx.x & 7, pl = threadIdx.x >> 3;           // 8 lanes per pixel, 32 pixels per iteration
    const int c0 = cg + sub * 8;
    const float invn = 1.0f / (float)pixels;
    float mu[8], rs[8], a1[8], a2[8], gr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[c0 + k]; rs[k] = rstd[c0 + k];
        a1[k] = s_a1[sub * 8 + k] * invn; a2[k] = s_a2[sub * 8 + k] * invn;
        gr[k] = (gamma ? gamma[c0 + k] : 1.0f) * rs[k];
    }
    const int64_t p_beg = (int64_t)blockIdx.y * ppc;
    int64_t p_end = p_beg + ppc;
    if (p_end > pixels) p_end = pixels;
    for (int64_t pix = p_beg + pl; pix < p_end; pix += 32) {
        Pack8 v, d, yo;
        float od[8], og[8];
        v.q = *reinterpret_cast<const uint4*>(x + pix * xps + c0);
        d.q = *reinterpret_cast<const uint4*>(dy + pix * dps + c0);
        if (relu && yout) yo.q = *reinterpret_cast<const uint4*>(yout + pix * yps + c0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xh = (bf16_to_f32(v.h[k]) - mu[k]) * rs[k];
            float gg = bf16_to_f32(d.h[k]);
            if (relu && !(yout ? bf16_to_f32(yo.h[k]) > 0.f : xh > 0.f)) gg = 0.f;
            od[k] = gr[k] * (gg - a1[k] - xh * a2[k]);
            og[k] = gg;
        }
        *reinterpret_cast<uint4*>(dx + pix * gps + c0) = pack_bf16x8(od);
        if (dres) *reinterpret_cast<uint4*>(dres + pix * drps + c0) = pack_bf16x8(og);
    }
}

// nearest x2: out[b, 2y+dy, 2x+dx, :] = in[b, y, x, :]   (F.interpolate(scale_factor=2, mode='nearest'))
__global__ __launch_bounds__(THREADS) void up2_kernel(const uint16_t* __restrict__ in, int64_t ips, int B, int H, int W,
                                                      int C, uint16_t* __restrict__ out, int64_t ops) {
    const int cl = C >> 3;
    const int64_t total = (int64_t)B * (2 * H) * (2 * W) * cl;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
        const int64_t opix = i / cl;
        const int c0 = (int)(i - opix * cl) * 8;
        const int ox = (int)(opix % (2 * W));
        const int64_t t = opix / (2 * W);
        const int oy = (int)(t % (2 * H));
        const int64_t b = t / (2 * H);
        const int64_t ipix = (b * H + (oy >> 1)) * W + (ox >> 1);
        *reinterpret_cast<uint4*>(out + opix * ops + c0) = *reinterpret_cast<const uint4*>(in + ipix * ips + c0);
    }
}

// zero insertion (the "dilate" half of a strided convolution's data gradient): z[b, s*y, s*x, :] = in[b, y, x, :],
// every other pixel of the Hz x Wz grid is zero.  dX of a stride-s conv = stride-1 conv of z with the rotated weights.
__global__ __launch_bounds__(THREADS) void zero_insert_kernel(const uint16_t* __restrict__ in, int64_t ips, int B, int H, int W,
                                                              int C, int s, int Hz, int Wz, uint16_t* __restrict__ out, int64_t ops) {
    const int cl = C >> 3;
    const int64_t total = (int64_t)B * Hz * Wz * cl;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
        const int64_t opix = i / cl;
        const int c0 = (int)(i - opix * cl) * 8;
        const int ox = (int)(opix % Wz);
        const int64_t t = opix / Wz;
        const int oy = (int)(t % Hz);
        const int64_t b = t / Hz;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        const int iy = oy / s, ix = ox / s;
        if (iy * s == oy && ix * s == ox && iy < H && ix < W) v = *reinterpret_cast<const uint4*>(in + ((b * H + iy) * W + ix) * ips + c0);
        *reinterpret_cast<uint4*>(out + opix * ops + c0) = v;
    }
}

// adjoint: gin[b, y, x, :] = sum of the 2x2 block of gout
__global__ __launch_bounds__(THREADS) void down2_sum_kernel(const uint16_t* __restrict__ gout, int64_t gps, int B, int H, int W,
                                                            int C, uint16_t* __restrict__ gin, int64_t ips) {
    const int cl = C >> 3;
    const int64_t total = (int64_t)B * H * W * cl;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
        const int64_t ipix = i / cl;
        const int c0 = (int)(i - ipix * cl) * 8;
        const int x = (int)(ipix % W);
        const int64_t t = ipix / W;
        const int y = (int)(t % H);
        const int64_t b = t / H;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int64_t opix = (b * 2 * H + 2 * y + dy) * (2 * W) + 2 * x + dx;
                Pack8 v;
                v.q = *reinterpret_cast<const uint4*>(gout + opix * gps + c0);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += bf16_to_f32(v.h[k]);
            }
        Pack8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o.h[k] = f32_to_bf16(acc[k]);
        *reinterpret_cast<uint4*>(gin + ipix * ips + c0) = o.q;
    }
}

// bilinear upsample by `scale` with align_corners=True, then L2-normalise over channels (eps 1e-12):
// nn.Upsample(scale_factor=4, bilinear, align_corners=True) + F.normalize(p=2, dim=1)  (image_model.py:121-143)
// one wave per output pixel, lane handles C/64 channels (C = 256 -> 4).
__global__ __launch_bounds__(THREADS) void bilinear_l2_kernel(const uint16_t* __restrict__ in, int64_t ips, int B, int H,
                                                              int W, int C, int scale, int normalize,
                                                              uint16_t* __restrict__ out, int64_t ops, float* __restrict__ inv_out) {
    const int Ho = H * scale, Wo = W * scale;
    const float ry = (Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float rx = (Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cpl = C >> 6;                        // channels per lane (C % 64 == 0, cpl <= 8)
    const int64_t total = (int64_t)B * Ho * Wo;
    for (int64_t opix = (int64_t)blockIdx.x * (THREADS / 64) + wave; opix < total; opix += (int64_t)gridDim.x * (THREADS / 64)) {
        const int ox = (int)(opix % Wo);
        const int64_t t = opix / Wo;
        const int oy = (int)(t % Ho);
        const int64_t b = t / Ho;
        const float fy = ry * oy, fx = rx * ox;
        int y0 = (int)fy, x0 = (int)fx;
        const int y1 = (y0 < H - 1) ? y0 + 1 : y0, x1 = (x0 < W - 1) ? x0 + 1 : x0;
        const float wy = fy - y0, wx = fx - x0;
        const uint16_t* p00 = in + ((b * H + y0) * W + x0) * ips + lane * cpl;
        const uint16_t* p01 = in + ((b * H + y0) * W + x1) * ips + lane * cpl;
        const uint16_t* p10 = in + ((b * H + y1) * W + x0) * ips + lane * cpl;
        const uint16_t* p11 = in + ((b * H + y1) * W + x1) * ips + lane * cpl;
        float v[8];
        float ss = 0.f;
        for (int k = 0; k < cpl; ++k) {
            const float a = bf16_to_f32(p00[k]), bb = bf16_to_f32(p01[k]), c = bf16_to_f32(p10[k]), d = bf16_to_f32(p11[k]);
            const float top = a + (bb - a) * wx, bot = c + (d - c) * wx;
            v[k] = top + (bot - top) * wy;
            ss += v[k] * v[k];
        }
        float inv = 1.0f;
        if (normalize) {
            ss = wave_sum(ss);
            inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        }
        uint16_t* o = out + opix * ops + lane * cpl;
        for (int k = 0; k < cpl; ++k) o[k] = f32_to_bf16(v[k] * inv);
        if (inv_out && lane == 0) inv_out[opix] = inv;
    }
}

// vectorised form for C/8 in {8, 16, 32, 64}: LPP = C/8 lanes own one output pixel, 16 bytes (8 channels) each;
// the four corner reads and the store are whole 16-byte accesses, the channel norm is an LPP-lane butterfly.
template <int LPP>
__global__ __launch_bounds__(THREADS) void bilinear_l2_vec_kernel(const uint16_t* __restrict__ in, int64_t ips, int B, int H,
                                                                  int W, int scale, int normalize,
                                                                  uint16_t* __restrict__ out, int64_t ops, int run_len,
                                                                  float* __restrict__ inv_out, int nt_out) {
    const int Ho = H * scale, Wo = W * scale;
    const float ry = (Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float rx = (Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    constexpr int PPB = THREADS / LPP;             // output pixels per block iteration
    const int sub = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    // one workgroup per output row (blockIdx.x = b * Ho + oy): row-level index math is scalar
    const int oy = blockIdx.x % Ho;
    const int64_t b = blockIdx.x / Ho;
    const float fy = ry * oy;
    const int y0 = (int)fy;
    const int y1 = (y0 < H - 1) ? y0 + 1 : y0;
    const float wy = fy - y0;
    const int64_t row0 = (b * H + y0) * W, row1 = (b * H + y1) * W, orow = (b * Ho + oy) * (int64_t)Wo;
    // The kernel was VALU bound (~150 instructions per 8 channels: 4 x 8 bf16 conversions and 7 FMAs per channel and
    // pixel, a division, ...), not memory bound.  Pixel slot pl walks a CONTIGUOUS run of output pixels, the source column
    // advances once every `scale` outputs, and everything that depends only on the column is computed once per column:
    // L = left corners blended vertically (wy is a row constant), D = right - left; per pixel v = L + D * wx.
    const int RUN = run_len;
    union U { uint4 q; uint16_t h[8]; };
    float Lc[8], Rc[8], Dc[8];                     // vertically blended left / right columns, and their difference
    int cx0 = -1, cx1 = -1;
    auto column = [&](int x, float (&dst)[8]) {
        U t, u;
        t.q = *reinterpret_cast<const uint4*>(in + (row0 + x) * ips + sub * 8);
        u.q = *reinterpret_cast<const uint4*>(in + (row1 + x) * ips + sub * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float ft = bf16_to_f32(t.h[k]), fu = bf16_to_f32(u.h[k]);
            dst[k] = ft + (fu - ft) * wy;
        }
    };
    for (int base = pl * RUN; base < Wo; base += PPB * RUN)
    for (int ox = base; ox < base + RUN && ox < Wo; ++ox) {
        const float fx = rx * ox;
        const int x0 = (int)fx;
        const int x1 = (x0 < W - 1) ? x0 + 1 : x0;
        const float wx = fx - x0;
        if (x0 != cx0 || x1 != cx1) {
            if (x0 == cx1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) Lc[k] = Rc[k];
            } else column(x0, Lc);
            if (x1 == x0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) Rc[k] = Lc[k];
            } else column(x1, Rc);
#pragma unroll
            for (int k = 0; k < 8; ++k) Dc[k] = Rc[k] - Lc[k];
            cx0 = x0; cx1 = x1;
        }
        float v[8];
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = Lc[k] + Dc[k] * wx; ss += v[k] * v[k]; }
        if (normalize) {
#pragma unroll
            for (int m = LPP / 2; m > 0; m >>= 1) ss += __shfl_xor(ss, m, 64);
            const float inv = rsqrtf(fmaxf(ss, 1e-24f));           // = 1 / max(sqrt(ss), 1e-12)
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= inv;
            if (inv_out && sub == 0) inv_out[orow + ox] = inv;     // the L2 adjoint's 1 / |x| (training form)
        }
        if (nt_out) {           // a result larger than the memory-side cache, read much later (the teacher's 1.15 GB feature map)
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            const uint4 o = pack_bf16x8(v);
            __builtin_nontemporal_store(u4v{o.x, o.y, o.z, o.w}, reinterpret_cast<u4v*>(out + (orow + ox) * ops + sub * 8));
        } else
        *reinterpret_cast<uint4*>(out + (orow + ox) * ops + sub * 8) = pack_bf16x8(v);
    }
}

// Reduce + finalize the conv epilogue's per-tile partials [tiles][2][C] (the frozen teacher and the DeepLab forward issue 53-59
// of these per pass).  Block (channel group x, slice y) sums its slice of the tiles in DOUBLE; with one slice (tiles <= 64) it
// finalizes directly.  With several slices there are two forms:
//   MODE 1 + MODE 2 (default, two launches): the slice sums are parked in scratch[y][2][C] with plain stores; a second launch of
//     one block per channel group adds them IN SLICE ORDER and computes mean / rstd / scale / shift / running statistics in double.
//     The kernel boundary is the only synchronisation: nothing outside the HIP memory model.
//   MODE 0 (opt-in, OESS_BN_ONE_LAUNCH=1, kept for A/B): one launch; slice sums parked with RETURNING device-scope exchanges,
//     a ticket per channel group, the last-ticket block reads the slices back with agent-scope atomic loads and adds them in
//     slice order.  It passes its stress test but relies on the hardware performing relaxed atomics in program order once their
//     results are back -- not a guarantee of the memory model (no release / acquire: an agent-scope release writes back the
//     whole XCD L2, which right after a convolution holds megabytes of dirty output: 25-31 us per call measured).
// Both are bit-repeatable and give identical bits (same partial sums, same order).
template <int MODE>
__global__ __launch_bounds__(256) void reduce_finalize_kernel(const float* __restrict__ part, int tiles, int C, double* __restrict__ scratch,
                                                              unsigned int* __restrict__ counter, int nslices, float count, float eps,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              float momentum, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                              float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double red[8][32][2];
    __shared__ unsigned int ticket;
    const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (MODE != 2) {
        if (c < C)
            for (int t = blockIdx.y * 8 + tl; t < tiles; t += gridDim.y * 8) {
                s1 += (double)part[((size_t)t * 2) * C + c];
                s2 += (double)part[((size_t)t * 2 + 1) * C + c];
            }
        red[tl][cl][0] = s1; red[tl][cl][1] = s2;
        __syncthreads();
        if (tl == 0 && c < C) {
#pragma unroll
            for (int k = 1; k < 8; ++k) { s1 += red[k][cl][0]; s2 += red[k][cl][1]; }
        }
    }
    if (MODE == 1) {                                             // slice sums out; the next launch adds them
        if (tl == 0 && c < C) {
            scratch[((size_t)blockIdx.y * 2) * C + c] = s1;
            scratch[((size_t)blockIdx.y * 2 + 1) * C + c] = s2;
        }
        return;
    }
    if (MODE == 2) {                                             // all 256 threads fetch (slice tl, tl+8, ...), then a fixed-order sum
        double p1 = 0.0, p2 = 0.0;
        if (c < C)
            for (int y = tl; y < nslices; y += 8) {
                p1 += scratch[((size_t)y * 2) * C + c];
                p2 += scratch[((size_t)y * 2 + 1) * C + c];
            }
        red[tl][cl][0] = p1; red[tl][cl][1] = p2;
        __syncthreads();
        if (tl == 0 && c < C) {
            s1 = p1; s2 = p2;
#pragma unroll
            for (int k = 1; k < 8; ++k) { s1 += red[k][cl][0]; s2 += red[k][cl][1]; }
        }
    }
    if (MODE == 0 && gridDim.y > 1) {
        if (tl == 0 && c < C) {
            unsigned long long* slot = reinterpret_cast<unsigned long long*>(scratch) + ((size_t)blockIdx.y * 2) * C + c;
            const unsigned long long r1 = atomicExch(slot, (unsigned long long)__double_as_longlong(s1));
            const unsigned long long r2 = atomicExch(slot + C, (unsigned long long)__double_as_longlong(s2));
            asm volatile("" :: "v"(r1), "v"(r2));                // wait for both exchanges to have been performed
        }
        __syncthreads();
        if (threadIdx.x == 0) ticket = atomicAdd(&counter[blockIdx.x], 1u);
        __syncthreads();
        if (ticket != gridDim.y - 1) return;
        // last block of this channel group: all 256 threads fetch (slice tl, tl+8, ...), then a fixed-order sum
        double p1 = 0.0, p2 = 0.0;
        if (c < C)
            for (int y = tl; y < (int)gridDim.y; y += 8) {
                const double* slot = scratch + ((size_t)y * 2) * C + c;
                p1 += __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                p2 += __hip_atomic_load(slot + C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        __syncthreads();                                         // red[] of the first phase has been consumed by tl == 0 above
        red[tl][cl][0] = p1; red[tl][cl][1] = p2;
        __syncthreads();
        if (tl == 0 && c < C) {
            s1 = p1; s2 = p2;
#pragma unroll
            for (int k = 1; k < 8; ++k) { s1 += red[k][cl][0]; s2 += red[k][cl][1]; }
        }
        if (threadIdx.x == 0) atomicExch(&counter[blockIdx.x], 0u);
    }
    if (tl == 0 && c < C)
        finalize_one(s1, s2, c, c, running_mean != nullptr, count, eps, gamma, beta, running_mean, running_var, momentum, mean_out,
                     rstd_out, scale, shift);
}

// Small maps (M <= ~40 k pixels: every layer of DeepLabv3's output-stride-16 half, 59 BatchNorms per forward): the
// reduce/finalize launch AND the apply launch above in ONE kernel, with no cross-workgroup protocol at all.  Workgroup
// (64-channel group x, pixel chunk y) first reduces the tile partials of ITS 64 channels itself (tiles * 512 B, L2 resident;
// 4 tile lanes in double, fixed order -> every workgroup of the group computes bit-identical scale / shift), keeps scale /
// shift in LDS, and then applies them to its pixel chunk: y = act(x * scale + shift [+ residual]).  Chunk 0 also publishes
// mean / rstd and updates the running statistics.  Redundant reduction work per workgroup <= the bytes of its own chunk.
__global__ __launch_bounds__(256) void tile_stats_apply_kernel(const float* __restrict__ part, int tiles, int C, float count, float eps,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               float momentum, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                               const uint16_t* __restrict__ x, int64_t xps,
                                                               const uint16_t* __restrict__ res, int64_t rps, int relu, int64_t pixels,
                                                               int64_t ppc, uint16_t* __restrict__ out, int64_t ops) {
    __shared__ double red[16][64][2];
    __shared__ float ssc[64], ssh[64];
    {   // 16 tile lanes x 16 float4 column lanes: a 70-tile table is 5 independent iterations of two 16-byte loads per thread
        // (4 tile lanes x 64 scalar columns was 18 dependent-latency iterations: 12 of the kernel's 17 us)
        const int c4 = (threadIdx.x & 15) * 4, tl = threadIdx.x >> 4;
        const float* p0 = part + (size_t)blockIdx.x * 64 + c4;
        double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int t = tl; t < tiles; t += 16) {
            const float4 u = *reinterpret_cast<const float4*>(p0 + ((size_t)t * 2) * C);
            const float4 w = *reinterpret_cast<const float4*>(p0 + ((size_t)t * 2 + 1) * C);
            a1[0] += (double)u.x; a1[1] += (double)u.y; a1[2] += (double)u.z; a1[3] += (double)u.w;
            a2[0] += (double)w.x; a2[1] += (double)w.y; a2[2] += (double)w.z; a2[3] += (double)w.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[tl][c4 + k][0] = a1[k]; red[tl][c4 + k][1] = a2[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cl = threadIdx.x, c = blockIdx.x * 64 + cl;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { s1 += red[k][cl][0]; s2 += red[k][cl][1]; }
        const double m = s1 / (double)count;
        double var = s2 / (double)count - m * m;
        if (var < 0.0) var = 0.0;
        const float r = (float)(1.0 / sqrt(var + (double)eps));
        const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
        ssc[cl] = ga * r;
        ssh[cl] = be - (float)m * ga * r;
        if (blockIdx.y == 0) {
            if (mean_out) mean_out[c] = (float)m;
            if (rstd_out) rstd_out[c] = r;
            if (running_mean) {
                const float unb = count > 1.f ? (float)(var * (double)count / ((double)count - 1.0)) : (float)var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
            }
        }
    }
    __syncthreads();
    const int lane_c = threadIdx.x & 7, prow = threadIdx.x >> 3;           // 8 lanes x 16 B per pixel, 32 pixels per iteration
    const int c0 = blockIdx.x * 64 + lane_c * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = ssc[lane_c * 8 + k]; sh[k] = ssh[lane_c * 8 + k]; }
    const int64_t p_beg = (int64_t)blockIdx.y * ppc;
    int64_t p_end = p_beg + ppc;
    if (p_end > pixels) p_end = pixels;
    for (int64_t p = p_beg + prow; p < p_end; p += 32) {
        Pack8 v, r;
        float f[8];
        v.q = *reinterpret_cast<const uint4*>(x + p * xps + c0);
        if (res) r.q = *reinterpret_cast<const uint4*>(res + p * rps + c0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f[k] = bf16_to_f32(v.h[k]) * sc[k] + sh[k];
            if (res) f[k] += bf16_to_f32(r.h[k]);
            if (relu) f[k] = fmaxf(f[k], 0.f);
        }
        *reinterpret_cast<uint4*>(out + p * ops + c0) = pack_bf16x8(f);
    }
}

// chunks per group.  Every workgroup ends with 2*C float atomics on the same few cache lines, and those serialise in L2:
// measured on a 72 MB tensor, forward statistics 34.6 us with 1024 workgroups, 16.1 us with 256 (four 16-byte loads in
// flight per lane keep the HBM stream busy); the backward sums read two or three tensors and want 512.
unsigned stats_chunks(int64_t ppg, int G, int target = 256) {
    int64_t want = (target + G - 1) / G;
    const int64_t hi = (ppg + 63) / 64;
    if (want > hi) want = hi;
    if (want < 1) want = 1;
    return (unsigned)want;
}

// grid of the apply kernels: x = pixel chunks (enough workgroups to fill the chip a few times over, each thread keeping
// >= 4 pixels in its loop), y = groups
dim3 apply_grid(int64_t ppg, int G, int C) {
    const int rows = THREADS / (C >> 3);
    int64_t gx = (ppg + (int64_t)rows * 4 - 1) / ((int64_t)rows * 4);
    const int64_t cap = (8192 + G - 1) / G;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    return dim3((unsigned)gx, (unsigned)G);
}

int grid_for(int64_t work) {
    int64_t g = (work + THREADS - 1) / THREADS;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    return (int)g;
}

}  // namespace

// launch stats_kernel<MODE> into the partials buffer; returns the number of chunk rows written (0 on a bad workspace)
template <int MODE>
static int launch_stats(hipStream_t st, const void* x, long long xps, const void* dy, long long dps, const float* mean,
                        const float* rstd, int relu, int G, long long ppg, int C, float* partials, size_t partials_bytes,
                        const void* yout = nullptr, long long yps = 0) {
    const int cl = C >> 3;
    if (cl > THREADS) return 0;
    const int rows = THREADS / cl;
    const size_t lds = (size_t)rows * C * 2 * sizeof(float);
    const unsigned chunks = stats_chunks(ppg, G, MODE == 0 ? 256 : 512);
    if (!partials || partials_bytes < (size_t)chunks * 2 * G * C * sizeof(float)) return 0;
    hipLaunchKernelGGL(stats_kernel<MODE>, dim3(chunks, (unsigned)G), dim3(THREADS), lds, st, (const uint16_t*)x, (int64_t)xps,
                       (const uint16_t*)dy, (int64_t)dps, mean, rstd, relu, (int64_t)ppg, C, partials, (const uint16_t*)yout,
                       (int64_t)yps);
    return (int)chunks;
}

extern "C" {

size_t oess_norm_partials_bytes(int G, long long pixels_per_group, int C, int backward) {
    if (G <= 0 || pixels_per_group <= 0 || C <= 0) return 0;
    return (size_t)stats_chunks(pixels_per_group, G, backward ? 512 : 256) * 2 * (size_t)G * C * sizeof(float);
}

int oess_norm_stats_nhwc_bf16(const void* x, long long x_pix_stride, int G, long long pixels_per_group, int C,
                              float* sum, float* sumsq, float* partials, size_t partials_bytes, oess_stream_t stream) {
    if (!x || !sum || !sumsq || G <= 0 || pixels_per_group <= 0 || C <= 0 || (C & 7) || C > 2048 || (x_pix_stride & 7))
        return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int chunks = launch_stats<0>(st, x, x_pix_stride, nullptr, 0, nullptr, nullptr, 0, G, pixels_per_group, C, partials,
                                       partials_bytes);
    if (!chunks) return OESS_EINVAL;
    hipLaunchKernelGGL(partials_reduce_kernel<false>, dim3((unsigned)(((int64_t)G * C + 15) / 16)), dim3(256), 0, st, partials, chunks, G,
                       C, sum, sumsq, 1.f, 0.f, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_norm_stats_finalize_nhwc_bf16(const void* x, long long x_pix_stride, int G, long long pixels_per_group, int C, float eps,
                                       const float* gamma, const float* beta, float* running_mean, float* running_var,
                                       float momentum, float* mean, float* rstd, float* scale, float* shift, float* partials,
                                       size_t partials_bytes, oess_stream_t stream) {
    if (!x || !mean || !rstd || !scale || !shift || G <= 0 || pixels_per_group <= 0 || C <= 0 || (C & 7) || C > 2048 ||
        (x_pix_stride & 7))
        return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int chunks = launch_stats<0>(st, x, x_pix_stride, nullptr, 0, nullptr, nullptr, 0, G, pixels_per_group, C, partials,
                                       partials_bytes);
    if (!chunks) return OESS_EINVAL;
    hipLaunchKernelGGL(partials_reduce_kernel<true>, dim3((unsigned)(((int64_t)G * C + 15) / 16)), dim3(256), 0, st, partials, chunks, G,
                       C, nullptr, nullptr, (float)pixels_per_group, eps, gamma, beta, running_mean, running_var, momentum, mean, rstd,
                       scale, shift);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_norm_reduce_finalize_tile_stats(const float* tile_stats, int tiles, int C, double* scratch,
                                         unsigned int* counters, float count, float eps, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, float momentum, float* mean, float* rstd,
                                         float* scale, float* shift, oess_stream_t stream) {
    if (!tile_stats || !scratch || !counters || !mean || !rstd || !scale || !shift || tiles <= 0 || C <= 0 || count <= 0.f)
        return OESS_EINVAL;
    int gy = (tiles + 63) / 64;                    // >= 8 tiles per tile lane
    if (gy > 32) gy = 32;
    if (gy < 1) gy = 1;
    static const bool one_launch = [] { const char* e = getenv("OESS_BN_ONE_LAUNCH"); return e && e[0] == '1'; }();
    const dim3 g1((C + 31) / 32, gy), g2((C + 31) / 32, 1);
    hipStream_t st = (hipStream_t)stream;
    if (gy == 1 || one_launch) {
        hipLaunchKernelGGL(reduce_finalize_kernel<0>, g1, dim3(256), 0, st, tile_stats, tiles, C, scratch, counters, gy, count, eps, gamma, beta,
                           running_mean, running_var, momentum, mean, rstd, scale, shift);
    } else {
        hipLaunchKernelGGL(reduce_finalize_kernel<1>, g1, dim3(256), 0, st, tile_stats, tiles, C, scratch, counters, gy, count, eps, gamma, beta,
                           running_mean, running_var, momentum, mean, rstd, scale, shift);
        hipLaunchKernelGGL(reduce_finalize_kernel<2>, g2, dim3(256), 0, st, tile_stats, tiles, C, scratch, counters, gy, count, eps, gamma, beta,
                           running_mean, running_var, momentum, mean, rstd, scale, shift);
    }
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_norm_tile_stats_apply_nhwc_bf16(const float* tile_stats, int tiles, int C, float count, float eps, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var, float momentum, float* mean,
                                         float* rstd, const void* x, long long x_pix_stride, const void* residual,
                                         long long res_pix_stride, int relu, long long pixels, void* out, long long out_pix_stride,
                                         oess_stream_t stream) {
    if (!tile_stats || !x || !out || tiles <= 0 || tiles > 512 || C <= 0 || (C & 63) || count <= 0.f || pixels <= 0 ||
        (x_pix_stride & 7) || (out_pix_stride & 7) || (residual && (res_pix_stride & 7)))
        return OESS_EINVAL;
    const int groups = C / 64;
    int64_t chunks = (1024 + groups - 1) / groups;                   // ~1024 workgroups, >= 128 pixels each
    const int64_t maxc = (pixels + 127) / 128;
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    int64_t ppc = (pixels + chunks - 1) / chunks;
    ppc = (ppc + 31) / 32 * 32;
    chunks = (pixels + ppc - 1) / ppc;
    hipLaunchKernelGGL(tile_stats_apply_kernel, dim3((unsigned)groups, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, tile_stats,
                       tiles, C, count, eps, gamma, beta, running_mean, running_var, momentum, mean, rstd, (const uint16_t*)x,
                       (int64_t)x_pix_stride, (const uint16_t*)residual, (int64_t)res_pix_stride, relu, (int64_t)pixels, ppc,
                       (uint16_t*)out, (int64_t)out_pix_stride);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_norm_apply_nhwc_bf16(const void* x, long long x_pix_stride, const float* scale, const float* shift,
                              const void* residual, long long res_pix_stride, int relu, int G, long long pixels_per_group,
                              int C, void* out, long long out_pix_stride, oess_stream_t stream) {
    if (!x || !scale || !shift || !out || G <= 0 || pixels_per_group <= 0 || C <= 0 || (C & 7) || (x_pix_stride & 7) ||
        (out_pix_stride & 7) || (residual && (res_pix_stride & 7)))
        return OESS_EINVAL;
    if ((C >> 3) > THREADS) return OESS_EINVAL;
    static const int nt = [] { const char* e = getenv("OESS_APPLY_NT"); return e ? atoi(e) : 2; }();
#define OESS_APPLY_LAUNCH(NT) hipLaunchKernelGGL(apply_kernel<NT>, apply_grid(pixels_per_group, G, C), dim3(THREADS), 0, \
                       (hipStream_t)stream, (const uint16_t*)x, (int64_t)x_pix_stride, scale, shift, (const uint16_t*)residual, \
                       (int64_t)res_pix_stride, relu, (int64_t)pixels_per_group, G, C, (uint16_t*)out, (int64_t)out_pix_stride)
    if (nt == 1) OESS_APPLY_LAUNCH(1); else if (nt == 2) OESS_APPLY_LAUNCH(2); else if (nt == 3) OESS_APPLY_LAUNCH(3);
    else if (nt == 6) OESS_APPLY_LAUNCH(6); else if (nt == 4) OESS_APPLY_LAUNCH(4); else OESS_APPLY_LAUNCH(0);
#undef OESS_APPLY_LAUNCH
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_instnorm_bwd_nhwc_bf16(const void* x, long long x_pix_stride, const void* dy, long long dy_pix_stride,
                                const float* mean, const float* rstd, int relu, int G, long long pixels_per_group, int C,
                                float* s1, float* s2, void* dx, long long dx_pix_stride, float* partials, size_t partials_bytes,
                                oess_stream_t stream) {
    if (!x || !dy || !mean || !rstd || !s1 || !s2 || !dx || G <= 0 || pixels_per_group <= 0 || C <= 0 || (C & 7) ||
        C > 2048 || (x_pix_stride & 7) || (dy_pix_stride & 7) || (dx_pix_stride & 7))
        return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int chunks = launch_stats<1>(st, x, x_pix_stride, dy, dy_pix_stride, mean, rstd, relu, G, pixels_per_group, C, partials,
                                       partials_bytes);
    if (!chunks) return OESS_EINVAL;
    hipLaunchKernelGGL(partials_reduce_kernel<false>, dim3((unsigned)(((int64_t)G * C + 15) / 16)), dim3(256), 0, st, partials, chunks, G,
                       C, s1, s2, 1.f, 0.f, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(in_bwd_apply_kernel, apply_grid(pixels_per_group, G, C), dim3(THREADS), 0, st,
                       (const uint16_t*)x, (int64_t)x_pix_stride, (const uint16_t*)dy, (int64_t)dy_pix_stride, mean, rstd, s1,
                       s2, relu, (int64_t)pixels_per_group, G, C, (uint16_t*)dx, (int64_t)dx_pix_stride);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_batchnorm_bwd_nhwc_bf16(const void* x, long long x_pix_stride, const void* dy, long long dy_pix_stride, const void* y_out,
                                 long long y_pix_stride, const float* mean, const float* rstd, const float* gamma, int relu,
                                 long long pixels, int C, float* dbeta, float* dgamma, void* dx, long long dx_pix_stride,
                                 void* dresidual, long long dres_pix_stride, float* partials, size_t partials_bytes,
                                 oess_stream_t stream) {
    if (!x || !dy || !mean || !rstd || !dbeta || !dgamma || !dx || pixels <= 0 || C <= 0 || (C & 7) || C > 2048 ||
        (x_pix_stride & 7) || (dy_pix_stride & 7) || (dx_pix_stride & 7) || (relu && !y_out) || (y_out && (y_pix_stride & 7)) ||
        (dresidual && (dres_pix_stride & 7)))
        return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // dbeta = sum g, dgamma = sum g * xhat (exactly the two sums dx needs)
    const int chunks = launch_stats<1>(st, x, x_pix_stride, dy, dy_pix_stride, mean, rstd, relu, 1, pixels, C, partials, partials_bytes,
                                       y_out, y_pix_stride);
    if (!chunks) return OESS_EINVAL;
    static const bool three_launches = [] { const char* e = getenv("OESS_BN_BWD_THREE_LAUNCHES"); return e && e[0] == '1'; }();   // A/B only
    if ((C & 63) == 0 && chunks <= 512 && !three_launches) {
        // reduce + apply in one launch (bit-identical to the path below)
        const int groups = C / 64;
        int64_t pch = (2048 + groups - 1) / groups;                     // ~2048 workgroups, >= 128 pixels each
        const int64_t maxc = (pixels + 127) / 128;
        if (pch > maxc) pch = maxc;
        if (pch < 1) pch = 1;
        int64_t ppc = (pixels + pch - 1) / pch;
        ppc = (ppc + 31) / 32 * 32;
        pch = (pixels + ppc - 1) / ppc;
        hipLaunchKernelGGL(bn_bwd_reduce_apply_kernel, dim3((unsigned)groups, (unsigned)pch), dim3(THREADS), 0, st, partials, chunks, C,
                           (const uint16_t*)x, (int64_t)x_pix_stride, (const uint16_t*)dy, (int64_t)dy_pix_stride,
                           (const uint16_t*)y_out, (int64_t)y_pix_stride, mean, rstd, gamma, relu, (int64_t)pixels, ppc, dbeta, dgamma,
                           (uint16_t*)dx, (int64_t)dx_pix_stride, (uint16_t*)dresidual, (int64_t)dres_pix_stride);
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
    hipLaunchKernelGGL(partials_reduce_kernel<false>, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, st, partials, chunks, 1, C, dbeta,
                       dgamma, 1.f, 0.f, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(in_bwd_apply_kernel, apply_grid(pixels, 1, C), dim3(THREADS), 0, st, (const uint16_t*)x,
                       (int64_t)x_pix_stride, (const uint16_t*)dy, (int64_t)dy_pix_stride, mean, rstd, dbeta, dgamma, relu,
                       (int64_t)pixels, 1, C, (uint16_t*)dx, (int64_t)dx_pix_stride, (const uint16_t*)y_out, (int64_t)y_pix_stride,
                       gamma, (uint16_t*)dresidual, (int64_t)dres_pix_stride);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_upsample_nearest2x_nhwc_bf16(const void* in, long long in_pix_stride, int B, int H, int W, int C, void* out,
                                      long long out_pix_stride, oess_stream_t stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (in_pix_stride & 7) || (out_pix_stride & 7))
        return OESS_EINVAL;
    hipLaunchKernelGGL(up2_kernel, dim3(grid_for((int64_t)B * 4 * H * W * (C >> 3))), dim3(THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)in, (int64_t)in_pix_stride, B, H, W, C, (uint16_t*)out, (int64_t)out_pix_stride);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_zero_insert_nhwc_bf16(const void* in, long long in_pix_stride, int B, int H, int W, int C, int stride, int Hz, int Wz,
                               void* out, long long out_pix_stride, oess_stream_t stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (in_pix_stride & 7) || (out_pix_stride & 7) || stride <= 0 ||
        Hz < (H - 1) * stride + 1 || Wz < (W - 1) * stride + 1)
        return OESS_EINVAL;
    hipLaunchKernelGGL(zero_insert_kernel, dim3(grid_for((int64_t)B * Hz * Wz * (C >> 3))), dim3(THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)in, (int64_t)in_pix_stride, B, H, W, C, stride, Hz, Wz, (uint16_t*)out, (int64_t)out_pix_stride);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_downsample_sum2x_nhwc_bf16(const void* gout, long long gout_pix_stride, int B, int H, int W, int C, void* gin,
                                    long long gin_pix_stride, oess_stream_t stream) {
    if (!gout || !gin || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (gout_pix_stride & 7) || (gin_pix_stride & 7))
        return OESS_EINVAL;
    hipLaunchKernelGGL(down2_sum_kernel, dim3(grid_for((int64_t)B * H * W * (C >> 3))), dim3(THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)gout, (int64_t)gout_pix_stride, B, H, W, C, (uint16_t*)gin, (int64_t)gin_pix_stride);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_bilinear_l2norm_nhwc_bf16(const void* in, long long in_pix_stride, int B, int H, int W, int C, int scale,
                                   int normalize, void* out, long long out_pix_stride, float* inv_norm, oess_stream_t stream) {
    if (inv_norm && !normalize) return OESS_EINVAL;
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 63) || C > 512 || scale <= 0) return OESS_EINVAL;
    const int64_t total = (int64_t)B * H * scale * W * scale;
    const int lpp = C / 8;
    if ((lpp == 8 || lpp == 16 || lpp == 32 || lpp == 64) && (in_pix_stride & 7) == 0 && (out_pix_stride & 7) == 0 &&
        ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0) {
        const int64_t gv = (int64_t)B * H * scale;            // one workgroup per output row
        // run length per pixel slot: the whole row split in PPB runs (measured 640-wide: runs of 1 / 4 / 8 / 16 / 80 pixels
        // = 486 / 371 / 356 / 329 / 334 us; the rest is instruction issue, ~55 VALU ops per 8 channels)
        const int ppb = THREADS / lpp;
        const int run_len = (W * scale + ppb - 1) / ppb;
        static const int nt_env = [] { const char* e = getenv("OESS_BILINEAR_NT"); return e ? atoi(e) : -1; }();
        const int nt_out = nt_env >= 0 ? nt_env : (total * C * 2 > (256ll << 20));
#define OESS_BL(LPP_)                                                                                                  \
        hipLaunchKernelGGL(bilinear_l2_vec_kernel<LPP_>, dim3((unsigned)gv), dim3(THREADS), 0, (hipStream_t)stream,    \
                           (const uint16_t*)in, (int64_t)in_pix_stride, B, H, W, scale, normalize, (uint16_t*)out,      \
                           (int64_t)out_pix_stride, run_len, inv_norm, nt_out)
        if (lpp == 8) OESS_BL(8); else if (lpp == 16) OESS_BL(16); else if (lpp == 32) OESS_BL(32); else OESS_BL(64);
#undef OESS_BL
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
    int64_t g = (total + 3) / 4;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(bilinear_l2_kernel, dim3((unsigned)g), dim3(THREADS), 0, (hipStream_t)stream, (const uint16_t*)in,
                       (int64_t)in_pix_stride, B, H, W, C, scale, normalize, (uint16_t*)out, (int64_t)out_pix_stride, inv_norm);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // extern "C"
