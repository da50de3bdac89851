// Bilinear resampling and channel L2 normalisation with their adjoints (NHWC, explicit pixel strides, bf16 or fp32).
//   deeplabv3_resnet50.forward: F.interpolate(size=input, bilinear, align_corners=False) on logits (fp32, K channels)
//     and features (256 ch)                                                        models/deeplabv3.py:179-189
//   DilationFeatureExtractor: nn.Upsample(x4, bilinear, align_corners=True) + F.normalize(p=2, dim=1)
//                                                                                  models/image_model.py:121-143
// Index rule = ATen's area_pixel_compute_source_index in fp32: align_corners ? dst * (in-1)/(out-1)
// : max((dst + 0.5) * in/out - 0.5, 0);  i0 = (int)src, i1 = min(i0 + 1, in - 1), lambda = src - i0.
// The backward is a deterministic GATHER in two separable passes (x then y): every input pixel sums the output
// pixels whose footprint contains it, found by re-evaluating the forward rule over a small candidate range - no
// atomics (the library's scatter-atomic kernel took 66 ms for one 8 x 256 x 440 x 640 bf16 gradient here).
// HBM-bound: the forward reads 4 corners (L2 hits) and writes each output once; the backward reads the output gradient
// ~2x (two neighbouring input columns share it through L2) and writes a (B, Ho, W, C) fp32 intermediate.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "oess.h"
#include "oess_common.h"
#include "bilinear_axis.h"

namespace {
using namespace oess;
constexpr int THREADS = 256;

// ---- VEC-wide typed access: bf16 x8 / fp32 x4 as one 16-byte access, or scalars
template <bool BF16, int VEC>
__device__ __forceinline__ void loadv(const void* base, int64_t off, float (&v)[VEC]) {
    if constexpr (BF16 && VEC == 8) {
        union { uint4 q; uint16_t h[8]; } u;
        u.q = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + off);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = bf16_to_f32(u.h[k]);
    } else if constexpr (!BF16 && VEC == 4) {
        const float4 q = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else if constexpr (!BF16 && VEC == 8) {
        const float4 q = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
        const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off + 4);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; v[4] = r.x; v[5] = r.y; v[6] = r.z; v[7] = r.w;
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            v[k] = BF16 ? bf16_to_f32(reinterpret_cast<const uint16_t*>(base)[off + k]) : reinterpret_cast<const float*>(base)[off + k];
    }
}
template <bool BF16, int VEC>
__device__ __forceinline__ void storev(void* base, int64_t off, const float (&v)[VEC]) {
    if constexpr (BF16 && VEC == 8) {
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + off) = pack_bf16x8(v);
    } else if constexpr (!BF16 && VEC == 4) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (!BF16 && VEC == 8) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if constexpr (BF16) reinterpret_cast<uint16_t*>(base)[off + k] = f32_to_bf16(v[k]);
            else reinterpret_cast<float*>(base)[off + k] = v[k];
        }
    }
}

// All three resampling kernels use one workgroup per image ROW (blockIdx.x = b * rows + y): the row-level index
// arithmetic is scalar and the per-element part is one 32-bit division by the channel-vector count (per-element 64-bit
// divisions made the first versions ALU-bound: 0.53 ms for a 1.15 GB output that streams in 0.2 ms).
template <bool BF16, int VEC>
__global__ __launch_bounds__(THREADS) void resize_fwd_kernel(const void* __restrict__ in, int64_t ips, int B, int C, Axis ay, Axis ax,
                                                             void* __restrict__ out, int64_t ops) {
    const int cv = C / VEC;
    const int oy = blockIdx.x % ay.out;
    const int64_t b = blockIdx.x / ay.out;
    int y0, y1; float wy;
    src_index(ay, oy, y0, y1, wy);
    const float hy = 1.f - wy;
    const int64_t row0 = (b * ay.in + y0) * ax.in, row1 = (b * ay.in + y1) * ax.in, orow = (b * ay.out + oy) * (int64_t)ax.out;
    const int n = ax.out * cv;
    for (int j = threadIdx.x; j < n; j += THREADS) {
        const int ox = j / cv, c = (j - ox * cv) * VEC;
        int x0, x1; float wx;
        src_index(ax, ox, x0, x1, wx);
        float p00[VEC], p01[VEC], p10[VEC], p11[VEC], r[VEC];
        loadv<BF16, VEC>(in, (row0 + x0) * ips + c, p00);
        loadv<BF16, VEC>(in, (row0 + x1) * ips + c, p01);
        loadv<BF16, VEC>(in, (row1 + x0) * ips + c, p10);
        loadv<BF16, VEC>(in, (row1 + x1) * ips + c, p11);
        const float hx = 1.f - wx;
#pragma unroll
        for (int k = 0; k < VEC; ++k)       // ATen's association (UpSampleBilinear2d.cu): rows first, then the two rows
            r[k] = hy * (hx * p00[k] + wx * p01[k]) + wy * (hx * p10[k] + wx * p11[k]);
        storev<BF16, VEC>(out, (orow + ox) * ops + c, r);
    }
}

// Upsampling form of the bf16 forward (DeepLabv3 returns its 256-channel OS16 feature map at the input size: 28 x 40 -> 440 x 640,
// a 1.15 GB write per batch of 8).  One workgroup per OUTPUT ROW: the two source rows are blended vertically ONCE into an
// fp32 LDS row V[x][c] = hy * in[y0][x][c] + wy * in[y1][x][c] (W_in * C * 4 bytes), every output pixel is then
// hx * V[x0] + wx * V[x1]: 2 LDS reads and 2 FMAs per channel instead of 4 global loads, 4 conversions and 7 FLOPs, and the
// output leaves through non-temporal 16-byte stores.  (Association: columns first, then x -- ATen blends x first; the
// difference is fp32 rounding, 2^-16 of the bf16 result's ulp.  The general kernel above keeps ATen's order for fp32.)
__global__ __launch_bounds__(THREADS) void resize_up_rows_bf16_kernel(const uint16_t* __restrict__ in, int64_t ips, int C, Axis ay, Axis ax,
                                                                      uint16_t* __restrict__ out, int64_t ops) {
    extern __shared__ __attribute__((aligned(16))) float vrow[];            // [ax.in][C]
    const int cv = C >> 3;
    const int oy = blockIdx.x % ay.out;
    const int64_t b = blockIdx.x / ay.out;
    int y0, y1; float wy;
    src_index(ay, oy, y0, y1, wy);
    const float hy = 1.f - wy;
    const int64_t row0 = (b * ay.in + y0) * ax.in, row1 = (b * ay.in + y1) * ax.in, orow = (b * ay.out + oy) * (int64_t)ax.out;
    for (int j = threadIdx.x; j < ax.in * cv; j += THREADS) {
        const int x = j / cv, c = (j - x * cv) * 8;
        float a[8], d[8];
        loadv<true, 8>(in, (row0 + x) * ips + c, a);
        loadv<true, 8>(in, (row1 + x) * ips + c, d);
        float4* dst = reinterpret_cast<float4*>(vrow + (size_t)x * C + c);
        dst[0] = make_float4(hy * a[0] + wy * d[0], hy * a[1] + wy * d[1], hy * a[2] + wy * d[2], hy * a[3] + wy * d[3]);
        dst[1] = make_float4(hy * a[4] + wy * d[4], hy * a[5] + wy * d[5], hy * a[6] + wy * d[6], hy * a[7] + wy * d[7]);
    }
    __syncthreads();
    // thread = (channel chunk fixed, pixel lane): the column weights of a pixel are computed once per pixel lane iteration
    const int cl = threadIdx.x % cv, pl = threadIdx.x / cv, ppi = THREADS / cv;
    if (pl >= ppi) return;
    const int c = cl * 8;
    for (int ox = pl; ox < ax.out; ox += ppi) {
        int x0, x1; float wx;
        src_index(ax, ox, x0, x1, wx);
        const float hx = 1.f - wx;
        const float4* p0 = reinterpret_cast<const float4*>(vrow + (size_t)x0 * C + c);
        const float4* p1 = reinterpret_cast<const float4*>(vrow + (size_t)x1 * C + c);
        const float4 a0 = p0[0], a1 = p0[1], b0 = p1[0], b1 = p1[1];
        float r[8] = {hx * a0.x + wx * b0.x, hx * a0.y + wx * b0.y, hx * a0.z + wx * b0.z, hx * a0.w + wx * b0.w,
                      hx * a1.x + wx * b1.x, hx * a1.y + wx * b1.y, hx * a1.z + wx * b1.z, hx * a1.w + wx * b1.w};
        const uint4 q = pack_bf16x8(r);
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_nt;
        __builtin_nontemporal_store(u32x4_nt{q.x, q.y, q.z, q.w}, reinterpret_cast<u32x4_nt*>(out + (orow + ox) * ops + c));
    }
}

// backward pass 1 (x): tmp[b, oy, ix, c] = sum_ox w(ox -> ix) * gout[b, oy, ox, c]        (tmp fp32, dense)
template <bool BF16, int VEC>
__global__ __launch_bounds__(THREADS) void resize_bwd_x_kernel(const void* __restrict__ gout, int64_t gps, int B, int C, Axis ay, Axis ax,
                                                               float* __restrict__ tmp) {
    const int cv = C / VEC;
    const int64_t t = blockIdx.x;                             // t = b * Ho + oy
    const int n = ax.in * cv;
    for (int j = threadIdx.x; j < n; j += THREADS) {
        const int ix = j / cv, c = (j - ix * cv) * VEC;
        int lo, hi;
        candidates(ax, ix, lo, hi);
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        for (int ox = lo; ox <= hi; ++ox) {
            const float w = weight_for(ax, ox, ix);
            if (w != 0.f) {
                float g[VEC];
                loadv<BF16, VEC>(gout, (t * ax.out + ox) * gps + c, g);
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] += w * g[k];
            }
        }
        storev<false, (VEC == 8 ? 8 : VEC)>(tmp, (t * ax.in + ix) * (int64_t)C + c, acc);
    }
}

// backward pass 2 (y): gin[b, iy, ix, c] = sum_oy w(oy -> iy) * tmp[b, oy, ix, c]
template <bool BF16, int VEC>
__global__ __launch_bounds__(THREADS) void resize_bwd_y_kernel(const float* __restrict__ tmp, int B, int C, Axis ay, Axis ax,
                                                               void* __restrict__ gin, int64_t gps) {
    const int cv = C / VEC;
    const int iy = blockIdx.x % ay.in;
    const int64_t b = blockIdx.x / ay.in;
    int lo, hi;
    candidates(ay, iy, lo, hi);
    const int n = ax.in * cv;
    for (int j = threadIdx.x; j < n; j += THREADS) {
        const int ix = j / cv, c = (j - ix * cv) * VEC;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        for (int oy = lo; oy <= hi; ++oy) {
            const float w = weight_for(ay, oy, iy);
            if (w != 0.f) {
                float g[VEC];
                loadv<false, (VEC == 8 ? 8 : VEC)>(tmp, ((b * ay.out + oy) * ax.in + ix) * (int64_t)C + c, g);
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] += w * g[k];
            }
        }
        storev<BF16, VEC>(gin, ((b * ay.in + iy) * ax.in + ix) * gps + c, acc);
    }
}

// ---- superpixel mean of L2-normalised, bilinearly upsampled features: the backward of
//      k = scatter_mean(normalize(upsample(x)))            models/image_model.py:121-143 + training/pretrain_trainer.py:445-465
// in ONE pass over the saved normalised map instead of three full-resolution round trips (row gather of gk / (n + 1e-6) -> L2
// adjoint -> x pass of the bilinear adjoint: 1.15 GB written, read, written and read again at 8 x 256 x 440 x 640).
// Every output pixel's gradient row is a row of the S x C quotient table (L2-resident), so the x pass of the bilinear adjoint
// forms the L2 adjoint  inv * (g - y * <y, g>)  on the fly: the LPP = C / 8 lanes that own one input column hold all C
// channels of an output pixel, the dot product is a butterfly over those lanes.  Deterministic (no atomics); the y pass is
// resize_bwd_y_kernel unchanged.
__global__ __launch_bounds__(THREADS) void pool_table_kernel(const float* __restrict__ gk, const float* __restrict__ count, int S, int C,
                                                             float* __restrict__ table) {
    const int64_t n = (int64_t)S * C;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS)
        table[i] = gk[i] / __fadd_rn(count[i / C], 1e-6f);
}

// Each output pixel is visited ONCE, by the LPP lanes of its LEFT input column x0: they form its adjoint row and add (1 - lam) of
// it to their own column's sum A and lam of it to the right neighbour's share B; B reaches the neighbour through an LDS slot
// (8 columns per iteration + the carry of the previous iteration), tmp[ix] = A(ix) + B(ix - 1): fixed order, no atomics.
template <int LPP>
__global__ __launch_bounds__(THREADS) void l2pool_bwd_x_kernel(const uint16_t* __restrict__ feat, int64_t fps, const float* __restrict__ inv,
                                                               const int64_t* __restrict__ ids, const float* __restrict__ table, int sps,
                                                               int S, float eps, Axis ay, Axis ax, float* __restrict__ tmp) {
    constexpr int C = LPP * 8, G = THREADS / LPP;             // G input columns per iteration
    __shared__ float share[(G + 1) * C];                      // slot g + 1 = column (base + g)'s B; slot 0 = carry
    const int64_t t = blockIdx.x;                             // t = b * Ho + oy
    const int64_t id_off = (t / ay.out) * (int64_t)sps;
    const int sub = threadIdx.x % LPP, grp = threadIdx.x / LPP, c = sub * 8;
    const float clamp_at = 1.0f / eps;
    int64_t held = -1;                                        // table row in g[]: neighbouring pixels mostly share their superpixel,
    float g[8];                                               // so the 1 KB row is fetched once per run, not once per pixel
#pragma unroll
    for (int k = 0; k < 8; ++k) { g[k] = 0.f; if (grp == 0) share[c + k] = 0.f; }
    for (int base = 0; base < ax.in; base += G) {
        const int ix = base + grp;
        float A[8], Bq[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { A[k] = 0.f; Bq[k] = 0.f; }
        if (ix < ax.in) {
            int lo, hi;
            candidates(ax, ix, lo, hi);
            for (int ox = lo; ox <= hi; ++ox) {               // (everything below is uniform over the LPP lanes of this column)
                int x0, x1; float lam;
                src_index(ax, ox, x0, x1, lam);
                if (x0 != ix) continue;
                const int64_t p = t * ax.out + ox;
                const int64_t gid = ids[p] + id_off;
                if (gid < 0 || gid >= S) continue;            // rows outside the table received no gradient
                float f[8];
                loadv<true, 8>(feat, p * fps + c, f);
                if (gid != held) { loadv<false, 8>(table, gid * C + c, g); held = gid; }
                float dot = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) dot += f[k] * g[k];
#pragma unroll
                for (int m = LPP / 2; m > 0; m >>= 1) dot += __shfl_xor(dot, m, 64);
                const float iv = inv[p];
                if (iv >= clamp_at) dot = 0.f;                // |x| <= eps: y = x / eps, no projection term
                const float wa = (x1 == x0) ? iv : iv * (1.0f - lam), wb = (x1 == x0) ? 0.f : iv * lam;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float r = g[k] - f[k] * dot;
                    A[k] += wa * r;
                    Bq[k] += wb * r;
                }
            }
        }
        float4* sl = reinterpret_cast<float4*>(share + (grp + 1) * C + c);
        sl[0] = make_float4(Bq[0], Bq[1], Bq[2], Bq[3]);
        sl[1] = make_float4(Bq[4], Bq[5], Bq[6], Bq[7]);
        __syncthreads();
        const float4* sr = reinterpret_cast<const float4*>(share + grp * C + c);
        const float4 r0 = sr[0], r1 = sr[1];
        if (ix < ax.in) {
            float o[8] = {A[0] + r0.x, A[1] + r0.y, A[2] + r0.z, A[3] + r0.w, A[4] + r1.x, A[5] + r1.y, A[6] + r1.z, A[7] + r1.w};
            storev<false, 8>(tmp, (t * ax.in + ix) * (int64_t)C + c, o);
        }
        __syncthreads();
        if (grp == G - 1) {                                   // carry into the next iteration's first column
            float4* s0 = reinterpret_cast<float4*>(share + c);
            s0[0] = make_float4(Bq[0], Bq[1], Bq[2], Bq[3]);
            s0[1] = make_float4(Bq[4], Bq[5], Bq[6], Bq[7]);
        }
    }
}

// ---- F.normalize(p=2, dim=channels, eps): y = x / max(|x|, eps); inv = 1 / max(|x|, eps) kept for the backward.
// one wave per pixel, lanes stride over channels
template <bool BF16>
__global__ __launch_bounds__(THREADS) void l2norm_fwd_kernel(const void* __restrict__ x, int64_t xs, int64_t P, int C, float eps,
                                                             void* __restrict__ y, int64_t ys, float* __restrict__ inv_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t p = (int64_t)blockIdx.x * (THREADS / 64) + wave; p < P; p += (int64_t)gridDim.x * (THREADS / 64)) {
        float ss = 0.f;
        for (int c = lane; c < C; c += 64) {
            float v[1]; loadv<BF16, 1>(x, p * xs + c, v);
            ss += v[0] * v[0];
        }
        ss = wave_sum(ss);
        const float inv = 1.0f / fmaxf(sqrtf(ss), eps);
        for (int c = lane; c < C; c += 64) {
            float v[1]; loadv<BF16, 1>(x, p * xs + c, v);
            v[0] *= inv;
            storev<BF16, 1>(y, p * ys + c, v);
        }
        if (lane == 0 && inv_out) inv_out[p] = inv;
    }
}
// gx = inv * (g - y * sum_c(g * y))   (pixels whose norm was clamped by eps: gx = inv * g, |x| < eps never happens for
// normalised features; the clamp's zero-gradient branch is reproduced through `clamped`)
template <bool BF16>
__global__ __launch_bounds__(THREADS) void l2norm_bwd_kernel(const void* __restrict__ y, int64_t ys, const void* __restrict__ g, int64_t gs,
                                                             const float* __restrict__ inv_in, int64_t P, int C, float eps,
                                                             void* __restrict__ gx, int64_t gxs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t p = (int64_t)blockIdx.x * (THREADS / 64) + wave; p < P; p += (int64_t)gridDim.x * (THREADS / 64)) {
        float dot = 0.f;
        for (int c = lane; c < C; c += 64) {
            float a[1], b[1]; loadv<BF16, 1>(y, p * ys + c, a); loadv<BF16, 1>(g, p * gs + c, b);
            dot += a[0] * b[0];
        }
        dot = wave_sum(dot);
        const float inv = inv_in[p];
        const bool clamped = inv >= 1.0f / eps;              // |x| <= eps: y = x / eps, d/dx = g / eps
        for (int c = lane; c < C; c += 64) {
            float a[1], b[1]; loadv<BF16, 1>(y, p * ys + c, a); loadv<BF16, 1>(g, p * gs + c, b);
            float r[1] = {clamped ? inv * b[0] : inv * (b[0] - a[0] * dot)};
            storev<BF16, 1>(gx, p * gxs + c, r);
        }
    }
}

// vectorised forms: LPP = C / VEC lanes own one pixel (LPP in {8, 16, 32, 64}), one 16-byte access per lane
template <bool BF16, int LPP>
__global__ __launch_bounds__(THREADS) void l2norm_fwd_vec_kernel(const void* __restrict__ x, int64_t xs, int64_t P, float eps,
                                                                 void* __restrict__ y, int64_t ys, float* __restrict__ inv_out) {
    constexpr int VEC = BF16 ? 8 : 4;
    constexpr int PPB = THREADS / LPP;
    const int sub = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    for (int64_t p = (int64_t)blockIdx.x * PPB + pl; p < P; p += (int64_t)gridDim.x * PPB) {
        float v[VEC];
        loadv<BF16, VEC>(x, p * xs + sub * VEC, v);
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) ss += v[k] * v[k];
#pragma unroll
        for (int m = LPP / 2; m > 0; m >>= 1) ss += __shfl_xor(ss, m, 64);
        const float inv = 1.0f / fmaxf(sqrtf(ss), eps);
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] *= inv;
        storev<BF16, VEC>(y, p * ys + sub * VEC, v);
        if (sub == 0 && inv_out) inv_out[p] = inv;
    }
}
template <bool BF16, int LPP>
__global__ __launch_bounds__(THREADS) void l2norm_bwd_vec_kernel(const void* __restrict__ y, int64_t ys, const void* __restrict__ g,
                                                                 int64_t gs, const float* __restrict__ inv_in, int64_t P, float eps,
                                                                 void* __restrict__ gx, int64_t gxs) {
    constexpr int VEC = BF16 ? 8 : 4;
    constexpr int PPB = THREADS / LPP;
    const int sub = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    for (int64_t p = (int64_t)blockIdx.x * PPB + pl; p < P; p += (int64_t)gridDim.x * PPB) {
        float a[VEC], b[VEC], r[VEC];
        loadv<BF16, VEC>(y, p * ys + sub * VEC, a);
        loadv<BF16, VEC>(g, p * gs + sub * VEC, b);
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) dot += a[k] * b[k];
#pragma unroll
        for (int m = LPP / 2; m > 0; m >>= 1) dot += __shfl_xor(dot, m, 64);
        const float inv = inv_in[p];
        const bool clamped = inv >= 1.0f / eps;
#pragma unroll
        for (int k = 0; k < VEC; ++k) r[k] = clamped ? inv * b[k] : inv * (b[k] - a[k] * dot);
        storev<BF16, VEC>(gx, p * gxs + sub * VEC, r);
    }
}

unsigned grid_for(int64_t work) {
    int64_t g = (work + THREADS - 1) / THREADS;
    if (g < 1) g = 1;
    if (g > 262144) g = 262144;
    return (unsigned)g;
}
// 16-byte vector path only when every access is aligned
bool vec_ok(const void* p, long long ps, int C, int is_bf16) {
    const int v = is_bf16 ? 8 : 4;
    return (C % v) == 0 && (ps % v) == 0 && ((uintptr_t)p & 15) == 0;
}
}  // namespace

extern "C" {

int oess_resize_bilinear_nhwc_fwd(const void* in, long long in_pix_stride, int B, int H, int W, int C, int is_bf16, int Ho, int Wo,
                                  int align_corners, void* out, long long out_pix_stride, oess_stream_t stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || in_pix_stride < C || out_pix_stride < C) return OESS_EINVAL;
    const Axis ay = make_axis(H, Ho, align_corners), ax = make_axis(W, Wo, align_corners);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = vec_ok(in, in_pix_stride, C, is_bf16) && vec_ok(out, out_pix_stride, C, is_bf16);
    if (is_bf16 && vec && Wo >= 4 * W && Ho >= 2 * H && (C >> 3) <= THREADS && (size_t)W * C * 4 <= 64 * 1024) {
        hipLaunchKernelGGL(resize_up_rows_bf16_kernel, dim3((unsigned)((int64_t)B * Ho)), dim3(THREADS), (size_t)W * C * 4, st,
                           (const uint16_t*)in, (int64_t)in_pix_stride, C, ay, ax, (uint16_t*)out, (int64_t)out_pix_stride);
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
#define OESS_RS(BF, V) hipLaunchKernelGGL((resize_fwd_kernel<BF, V>), dim3((unsigned)((int64_t)B * Ho)), dim3(THREADS), 0, st, in, (int64_t)in_pix_stride, B, C, ay, ax, out, (int64_t)out_pix_stride)
    if (is_bf16) { if (vec) OESS_RS(true, 8); else OESS_RS(true, 1); }
    else { if (vec) OESS_RS(false, 4); else OESS_RS(false, 1); }
#undef OESS_RS
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

size_t oess_resize_bilinear_bwd_workspace_bytes(int B, int W, int C, int Ho) {
    if (B <= 0 || W <= 0 || C <= 0 || Ho <= 0) return 0;
    return (size_t)B * Ho * W * C * sizeof(float);
}

int oess_resize_bilinear_nhwc_bwd(const void* grad_out, long long gout_pix_stride, int B, int H, int W, int C, int is_bf16, int Ho,
                                  int Wo, int align_corners, void* workspace, size_t workspace_bytes, void* grad_in,
                                  long long gin_pix_stride, oess_stream_t stream) {
    if (!grad_out || !grad_in || !workspace || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || gout_pix_stride < C ||
        gin_pix_stride < C)
        return OESS_EINVAL;
    if (workspace_bytes < oess_resize_bilinear_bwd_workspace_bytes(B, W, C, Ho) || ((uintptr_t)workspace & 15)) return OESS_ENOMEM;
    const Axis ay = make_axis(H, Ho, align_corners), ax = make_axis(W, Wo, align_corners);
    hipStream_t st = (hipStream_t)stream;
    float* tmp = (float*)workspace;
    const bool vec = vec_ok(grad_out, gout_pix_stride, C, is_bf16) && vec_ok(grad_in, gin_pix_stride, C, is_bf16);
#define OESS_RB(BF, V)                                                                                                             \
    {                                                                                                                              \
        hipLaunchKernelGGL((resize_bwd_x_kernel<BF, V>), dim3((unsigned)((int64_t)B * Ho)), dim3(THREADS), 0, st, grad_out,              \
                           (int64_t)gout_pix_stride, B, C, ay, ax, tmp);                                                           \
        hipLaunchKernelGGL((resize_bwd_y_kernel<BF, V>), dim3((unsigned)((int64_t)B * H)), dim3(THREADS), 0, st, (const float*)tmp, B,  \
                           C, ay, ax, grad_in, (int64_t)gin_pix_stride);                                                           \
    }
    if (is_bf16) { if (vec) OESS_RB(true, 8) else OESS_RB(true, 1) }
    else { if (vec) OESS_RB(false, 4) else OESS_RB(false, 1) }
#undef OESS_RB
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_l2norm_nhwc_fwd(const void* x, long long x_pix_stride, int64_t P, int C, int is_bf16, float eps, void* y,
                         long long y_pix_stride, float* inv_norm, oess_stream_t stream) {
    if (!x || !y || P <= 0 || C <= 0 || x_pix_stride < C || y_pix_stride < C || eps <= 0.f) return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    {
        const int vecw = is_bf16 ? 8 : 4, lpp = (C % vecw == 0) ? C / vecw : 0;
        if ((lpp == 8 || lpp == 16 || lpp == 32 || lpp == 64) && vec_ok(x, x_pix_stride, C, is_bf16) && vec_ok(y, y_pix_stride, C, is_bf16)) {
            const unsigned gv = grid_for(P * lpp);
#define OESS_LN(BF, L) hipLaunchKernelGGL((l2norm_fwd_vec_kernel<BF, L>), dim3(gv), dim3(THREADS), 0, st, x, (int64_t)x_pix_stride, P, eps, y, (int64_t)y_pix_stride, inv_norm)
            if (is_bf16) { if (lpp == 8) OESS_LN(true, 8); else if (lpp == 16) OESS_LN(true, 16); else if (lpp == 32) OESS_LN(true, 32); else OESS_LN(true, 64); }
            else { if (lpp == 8) OESS_LN(false, 8); else if (lpp == 16) OESS_LN(false, 16); else if (lpp == 32) OESS_LN(false, 32); else OESS_LN(false, 64); }
#undef OESS_LN
            OESS_HIP(hipGetLastError());
            return OESS_OK;
        }
    }
    const unsigned g = grid_for(P * 64);
    if (is_bf16) hipLaunchKernelGGL(l2norm_fwd_kernel<true>, dim3(g), dim3(THREADS), 0, st, x, (int64_t)x_pix_stride, P, C, eps, y, (int64_t)y_pix_stride, inv_norm);
    else hipLaunchKernelGGL(l2norm_fwd_kernel<false>, dim3(g), dim3(THREADS), 0, st, x, (int64_t)x_pix_stride, P, C, eps, y, (int64_t)y_pix_stride, inv_norm);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_l2norm_nhwc_bwd(const void* y, long long y_pix_stride, const void* grad_y, long long gy_pix_stride, const float* inv_norm,
                         int64_t P, int C, int is_bf16, float eps, void* grad_x, long long gx_pix_stride, oess_stream_t stream) {
    if (!y || !grad_y || !inv_norm || !grad_x || P <= 0 || C <= 0 || y_pix_stride < C || gy_pix_stride < C || gx_pix_stride < C || eps <= 0.f)
        return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    {
        const int vecw = is_bf16 ? 8 : 4, lpp = (C % vecw == 0) ? C / vecw : 0;
        if ((lpp == 8 || lpp == 16 || lpp == 32 || lpp == 64) && vec_ok(y, y_pix_stride, C, is_bf16) && vec_ok(grad_y, gy_pix_stride, C, is_bf16) &&
            vec_ok(grad_x, gx_pix_stride, C, is_bf16)) {
            const unsigned gv = grid_for(P * lpp);
#define OESS_LN(BF, L) hipLaunchKernelGGL((l2norm_bwd_vec_kernel<BF, L>), dim3(gv), dim3(THREADS), 0, st, y, (int64_t)y_pix_stride, grad_y, (int64_t)gy_pix_stride, inv_norm, P, eps, grad_x, (int64_t)gx_pix_stride)
            if (is_bf16) { if (lpp == 8) OESS_LN(true, 8); else if (lpp == 16) OESS_LN(true, 16); else if (lpp == 32) OESS_LN(true, 32); else OESS_LN(true, 64); }
            else { if (lpp == 8) OESS_LN(false, 8); else if (lpp == 16) OESS_LN(false, 16); else if (lpp == 32) OESS_LN(false, 32); else OESS_LN(false, 64); }
#undef OESS_LN
            OESS_HIP(hipGetLastError());
            return OESS_OK;
        }
    }
    const unsigned g = grid_for(P * 64);
    if (is_bf16) hipLaunchKernelGGL(l2norm_bwd_kernel<true>, dim3(g), dim3(THREADS), 0, st, y, (int64_t)y_pix_stride, grad_y, (int64_t)gy_pix_stride, inv_norm, P, C, eps, grad_x, (int64_t)gx_pix_stride);
    else hipLaunchKernelGGL(l2norm_bwd_kernel<false>, dim3(g), dim3(THREADS), 0, st, y, (int64_t)y_pix_stride, grad_y, (int64_t)gy_pix_stride, inv_norm, P, C, eps, grad_x, (int64_t)gx_pix_stride);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

size_t oess_bilinear_l2norm_pool_bwd_workspace_bytes(int B, int W, int C, int Ho, int S) {
    if (B <= 0 || W <= 0 || C <= 0 || Ho <= 0 || S <= 0) return 0;
    return (size_t)B * Ho * W * C * sizeof(float) + (((size_t)S * C * sizeof(float) + 255) & ~(size_t)255);
}

int oess_bilinear_l2norm_pool_bwd_bf16(const void* feat, long long feat_pix_stride, const float* inv_norm, const int64_t* ids,
                                       const float* grad_k, const float* count, int superpixel_size, int S, int B, int H, int W, int C,
                                       int Ho, int Wo, int align_corners, float eps, void* workspace, size_t workspace_bytes,
                                       void* grad_in, long long gin_pix_stride, oess_stream_t stream) {
    if (!feat || !inv_norm || !ids || !grad_k || !count || !workspace || !grad_in || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 ||
        S <= 0 || superpixel_size <= 0 || feat_pix_stride < C || gin_pix_stride < C || eps <= 0.f)
        return OESS_EINVAL;
    const int lpp = (C % 8 == 0) ? C / 8 : 0;
    if (!(lpp == 8 || lpp == 16 || lpp == 32 || lpp == 64) || !vec_ok(feat, feat_pix_stride, C, 1) || !vec_ok(grad_in, gin_pix_stride, C, 1))
        return OESS_EINVAL;
    if (workspace_bytes < oess_bilinear_l2norm_pool_bwd_workspace_bytes(B, W, C, Ho, S) || ((uintptr_t)workspace & 15)) return OESS_ENOMEM;
    const Axis ay = make_axis(H, Ho, align_corners), ax = make_axis(W, Wo, align_corners);
    hipStream_t st = (hipStream_t)stream;
    float* tmp = (float*)workspace;
    float* table = tmp + (size_t)B * Ho * W * C;
    hipLaunchKernelGGL(pool_table_kernel, dim3(grid_for((int64_t)S * C)), dim3(THREADS), 0, st, grad_k, count, S, C, table);
    const dim3 gx((unsigned)((int64_t)B * Ho));
#define OESS_LP(L) hipLaunchKernelGGL((l2pool_bwd_x_kernel<L>), gx, dim3(THREADS), 0, st, (const uint16_t*)feat, (int64_t)feat_pix_stride, \
                                      inv_norm, ids, (const float*)table, superpixel_size, S, eps, ay, ax, tmp)
    if (lpp == 8) OESS_LP(8); else if (lpp == 16) OESS_LP(16); else if (lpp == 32) OESS_LP(32); else OESS_LP(64);
#undef OESS_LP
    hipLaunchKernelGGL((resize_bwd_y_kernel<true, 8>), dim3((unsigned)((int64_t)B * H)), dim3(THREADS), 0, st, (const float*)tmp, B, C, ay, ax,
                       grad_in, (int64_t)gin_pix_stride);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}


}  // extern "C"
