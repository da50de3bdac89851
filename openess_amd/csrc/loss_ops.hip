// a16 consistency losses (training/openess_trainer.py:497-503), HBM-bound, deterministic two-stage reductions:
//   cons_feat_loss = L1Loss(feat_a, feat_b)                       = mean |a - b|
//   cons_pred_loss = mean(1 - cosine_similarity(la, lb, dim=1))   , cos = sum_c (a / max(|a|, eps)) (b / max(|b|, eps))
// (the per-tensor norm clamp is torch >= 1.12's formula, the one the reference's pinned torch 2.1 runs).
// Operands are NHWC (channels contiguous, explicit pixel stride), bf16 or fp32; gradients keep the operand dtype.
// Algorithmic bytes: forward 2 reads, backward 2 reads + 2 writes of the operands.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "oess.h"
#include "oess_common.h"

namespace {
using namespace oess;
constexpr int THREADS = 256;
constexpr int MAX_PARTIALS = 1024;

template <bool BF16>
__device__ __forceinline__ float ld(const void* p, int64_t i) {
    if constexpr (BF16) return bf16_to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
    else return reinterpret_cast<const float*>(p)[i];
}
template <bool BF16>
__device__ __forceinline__ void st(void* p, int64_t i, float v) {
    if constexpr (BF16) reinterpret_cast<uint16_t*>(p)[i] = f32_to_bf16(v);
    else reinterpret_cast<float*>(p)[i] = v;
}

__device__ __forceinline__ void block_partial(double v, double* partials) {
    __shared__ double red[THREADS / 64];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < THREADS / 64; ++w) s += red[w];
        partials[blockIdx.x] = s;
    }
}

// second stage: one block sums the partials in a fixed order -> loss = offset + scale * sum
__global__ __launch_bounds__(THREADS) void finish_kernel(const double* __restrict__ partials, int n, double offset, double scale,
                                                         float* __restrict__ loss) {
    __shared__ double red[THREADS];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += THREADS) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(offset + scale * red[0]);
}

template <bool BF16>
__global__ __launch_bounds__(THREADS) void l1_fwd_kernel(const void* __restrict__ a, const void* __restrict__ b, int64_t n,
                                                         double* __restrict__ partials) {
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS)
        s += (double)fabsf(ld<BF16>(a, i) - (b ? ld<BF16>(b, i) : 0.f));
    block_partial(s, partials);
}

template <bool BF16>
__global__ __launch_bounds__(THREADS) void l1_bwd_kernel(const void* __restrict__ a, const void* __restrict__ b, int64_t n,
                                                         const float* __restrict__ gout, float inv_n, void* __restrict__ ga,
                                                         void* __restrict__ gb) {
    const float g = gout[0] * inv_n;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const float d = ld<BF16>(a, i) - (b ? ld<BF16>(b, i) : 0.f);
        const float s = d > 0.f ? g : (d < 0.f ? -g : 0.f);          // sign(0) = 0 like torch
        if (ga) st<BF16>(ga, i, s);
        if (gb) st<BF16>(gb, i, -s);
    }
}

// 16-byte forms (n % VEC elements of tail go through the scalar kernels' arithmetic in the same launch); b == nullptr: mean |a|
template <bool BF16>
__device__ __forceinline__ void ldv(const void* p, int64_t i, float (&v)[BF16 ? 8 : 4]) {
    if constexpr (BF16) {
        const uint4 q = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p) + i);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
    } else {
        const float4 q = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
}
template <bool BF16>
__device__ __forceinline__ void stv(void* p, int64_t i, const float (&v)[BF16 ? 8 : 4]) {
    if constexpr (BF16) *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p) + i) = pack_bf16x8(v);
    else *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i) = make_float4(v[0], v[1], v[2], v[3]);
}

template <bool BF16>
__global__ __launch_bounds__(THREADS) void l1_fwd_vec_kernel(const void* __restrict__ a, const void* __restrict__ b, int64_t n,
                                                             double* __restrict__ partials) {
    constexpr int VEC = BF16 ? 8 : 4;
    const int64_t nv = n / VEC;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < nv; i += (int64_t)gridDim.x * THREADS) {
        float x[VEC], y[VEC];
        ldv<BF16>(a, i * VEC, x);
        if (b) ldv<BF16>(b, i * VEC, y);
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) t += fabsf(b ? x[k] - y[k] : x[k]);
        s += (double)t;
    }
    if (blockIdx.x == 0)
        for (int64_t i = nv * VEC + threadIdx.x; i < n; i += THREADS) s += (double)fabsf(ld<BF16>(a, i) - (b ? ld<BF16>(b, i) : 0.f));
    block_partial(s, partials);
}

template <bool BF16>
__global__ __launch_bounds__(THREADS) void l1_bwd_vec_kernel(const void* __restrict__ a, const void* __restrict__ b, int64_t n,
                                                             const float* __restrict__ gout, float inv_n, void* __restrict__ ga,
                                                             void* __restrict__ gb) {
    constexpr int VEC = BF16 ? 8 : 4;
    const int64_t nv = n / VEC;
    const float g = gout[0] * inv_n;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < nv; i += (int64_t)gridDim.x * THREADS) {
        float x[VEC], y[VEC], sa[VEC], sb[VEC];
        ldv<BF16>(a, i * VEC, x);
        if (b) ldv<BF16>(b, i * VEC, y);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float d = b ? x[k] - y[k] : x[k];
            sa[k] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
            sb[k] = -sa[k];
        }
        if (ga) stv<BF16>(ga, i * VEC, sa);
        if (gb) stv<BF16>(gb, i * VEC, sb);
    }
    if (blockIdx.x == 0)
        for (int64_t i = nv * VEC + threadIdx.x; i < n; i += THREADS) {
            const float d = ld<BF16>(a, i) - (b ? ld<BF16>(b, i) : 0.f);
            const float sgn = d > 0.f ? g : (d < 0.f ? -g : 0.f);
            if (ga) st<BF16>(ga, i, sgn);
            if (gb) st<BF16>(gb, i, -sgn);
        }
}

// one wave per pixel, lanes stride over channels; accumulates sum over pixels of cos
template <bool BF16>
__global__ __launch_bounds__(THREADS) void cos_fwd_kernel(const void* __restrict__ a, int64_t as, const void* __restrict__ b,
                                                          int64_t bs, int64_t P, int C, float eps, double* __restrict__ partials) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * (THREADS / 64) + wave; p < P; p += (int64_t)gridDim.x * (THREADS / 64)) {
        float ab = 0.f, aa = 0.f, bb = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float x = ld<BF16>(a, p * as + c), y = ld<BF16>(b, p * bs + c);
            ab += x * y; aa += x * x; bb += y * y;
        }
        ab = wave_sum(ab); aa = wave_sum(aa); bb = wave_sum(bb);
        if (lane == 0) acc += (double)(ab / (fmaxf(sqrtf(aa), eps) * fmaxf(sqrtf(bb), eps)));
    }
    block_partial(acc, partials);        // lanes other than 0 contribute 0
}

template <bool BF16>
__global__ __launch_bounds__(THREADS) void cos_bwd_kernel(const void* __restrict__ a, int64_t as, const void* __restrict__ b,
                                                          int64_t bs, int64_t P, int C, float eps, const float* __restrict__ gout,
                                                          float inv_p, void* __restrict__ ga, int64_t gas, void* __restrict__ gb,
                                                          int64_t gbs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float g = -gout[0] * inv_p;                    // d loss / d cos
    for (int64_t p = (int64_t)blockIdx.x * (THREADS / 64) + wave; p < P; p += (int64_t)gridDim.x * (THREADS / 64)) {
        float ab = 0.f, aa = 0.f, bb = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float x = ld<BF16>(a, p * as + c), y = ld<BF16>(b, p * bs + c);
            ab += x * y; aa += x * x; bb += y * y;
        }
        ab = wave_sum(ab); aa = wave_sum(aa); bb = wave_sum(bb);
        const float na = sqrtf(aa), nb = sqrtf(bb);
        const float ca = fmaxf(na, eps), cb = fmaxf(nb, eps);
        const float inv = 1.0f / (ca * cb), cosv = ab * inv;
        // d cos / dx = y / (ca cb) - cos * x / (ca * |x|) while the norm is above eps (the clamp has zero gradient below)
        const float ka = na > eps ? cosv / (ca * na) : 0.f, kb = nb > eps ? cosv / (cb * nb) : 0.f;
        for (int c = lane; c < C; c += 64) {
            const float x = ld<BF16>(a, p * as + c), y = ld<BF16>(b, p * bs + c);
            if (ga) st<BF16>(ga, p * gas + c, g * (y * inv - ka * x));
            if (gb) st<BF16>(gb, p * gbs + c, g * (x * inv - kb * y));
        }
    }
}

// Dense small-C form (the logits: fp32 [P][K], K = 11 -- a wave per pixel kept 53 of 64 lanes idle and took 0.53 + 0.59 ms at
// 8 x 440 x 640): a workgroup moves 256 pixels x C floats of both operands through LDS with 16-byte accesses, one thread per pixel
// (channels added in order), the gradients leave through the same tiles.
constexpr int COS_CMAX = 24;             // 2 x 256 x 24 floats = 48 KB of LDS
__device__ __forceinline__ void cos_tile_load(const float* __restrict__ src, int64_t total, float* dst) {
    const int64_t n4 = total >> 2;
    for (int64_t j = threadIdx.x; j < n4; j += THREADS) *reinterpret_cast<float4*>(dst + 4 * j) = *reinterpret_cast<const float4*>(src + 4 * j);
    for (int64_t j = 4 * n4 + threadIdx.x; j < total; j += THREADS) dst[j] = src[j];
}
__device__ __forceinline__ void cos_tile_store(float* __restrict__ dst, int64_t total, const float* src) {
    const int64_t n4 = total >> 2;
    for (int64_t j = threadIdx.x; j < n4; j += THREADS) *reinterpret_cast<float4*>(dst + 4 * j) = *reinterpret_cast<const float4*>(src + 4 * j);
    for (int64_t j = 4 * n4 + threadIdx.x; j < total; j += THREADS) dst[j] = src[j];
}

__global__ __launch_bounds__(THREADS) void cos_fwd_dense_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t P, int C,
                                                                float eps, double* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float cos_s[];
    float* sa = cos_s; float* sb = cos_s + THREADS * C;
    double acc = 0.0;
    for (int64_t p0 = (int64_t)blockIdx.x * THREADS; p0 < P; p0 += (int64_t)gridDim.x * THREADS) {
        const int64_t n = (P - p0 < THREADS) ? P - p0 : THREADS;
        __syncthreads();
        cos_tile_load(a + p0 * C, n * C, sa);
        cos_tile_load(b + p0 * C, n * C, sb);
        __syncthreads();
        if ((int64_t)threadIdx.x < n) {
            float ab = 0.f, aa = 0.f, bb = 0.f;
            for (int c = 0; c < C; ++c) {
                const float x = sa[threadIdx.x * C + c], y = sb[threadIdx.x * C + c];
                ab += x * y; aa += x * x; bb += y * y;
            }
            acc += (double)(ab / (fmaxf(sqrtf(aa), eps) * fmaxf(sqrtf(bb), eps)));
        }
    }
    block_partial(acc, partials);
}

__global__ __launch_bounds__(THREADS) void cos_bwd_dense_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t P, int C,
                                                                float eps, const float* __restrict__ gout, float inv_p,
                                                                float* __restrict__ ga, float* __restrict__ gb) {
    extern __shared__ __attribute__((aligned(16))) float cos_s[];
    float* sa = cos_s; float* sb = cos_s + THREADS * C;
    const float g = -gout[0] * inv_p;
    for (int64_t p0 = (int64_t)blockIdx.x * THREADS; p0 < P; p0 += (int64_t)gridDim.x * THREADS) {
        const int64_t n = (P - p0 < THREADS) ? P - p0 : THREADS;
        __syncthreads();
        cos_tile_load(a + p0 * C, n * C, sa);
        cos_tile_load(b + p0 * C, n * C, sb);
        __syncthreads();
        if ((int64_t)threadIdx.x < n) {
            float* ra = sa + threadIdx.x * C; float* rb = sb + threadIdx.x * C;
            float ab = 0.f, aa = 0.f, bb = 0.f;
            for (int c = 0; c < C; ++c) { ab += ra[c] * rb[c]; aa += ra[c] * ra[c]; bb += rb[c] * rb[c]; }
            const float na = sqrtf(aa), nb = sqrtf(bb);
            const float ca = fmaxf(na, eps), cb = fmaxf(nb, eps);
            const float inv = 1.0f / (ca * cb), cosv = ab * inv;
            const float ka = na > eps ? cosv / (ca * na) : 0.f, kb = nb > eps ? cosv / (cb * nb) : 0.f;
            for (int c = 0; c < C; ++c) {
                const float x = ra[c], y = rb[c];
                ra[c] = g * (y * inv - ka * x);
                rb[c] = g * (x * inv - kb * y);
            }
        }
        __syncthreads();
        if (ga) cos_tile_store(ga + p0 * C, n * C, sa);
        if (gb) cos_tile_store(gb + p0 * C, n * C, sb);
    }
}

// ---------------------------------------------------------------------------------------------
// a14 PointInfoNCE (utils/loss_functions.py:147-154): loss = CE(k q^T / T, arange(S)), mean over the S rows.
// S <= ~100 * B superpixel rows, C = 256: everything is L2 resident; plain fp32 FMA kernels (the logits are divided
// by T = 0.07, so bf16 operands would cost 14x their rounding error).
//   nce_logits_kernel : G[i][j] = k_i . q_j / T                     (tile 16 x 16 outputs per workgroup, LDS staged)
//   nce_rows_kernel   : row log-sum-exp -> loss partials; G[i][j] <- (softmax_ij - [i == j]) / (S T)   (= dL/d(k q^T))
//   nce_grad_kernel   : dk = g * G q ,  dq = g * G^T k
// ---------------------------------------------------------------------------------------------
constexpr int NT = 16;
__global__ __launch_bounds__(NT * NT) void nce_logits_kernel(const float* __restrict__ k, const float* __restrict__ q, int S, int C,
                                                             float inv_t, float* __restrict__ G) {
    __shared__ float lk[NT][NT + 1], lq[NT][NT + 1];
    const int tx = threadIdx.x % NT, ty = threadIdx.x / NT;
    const int i = blockIdx.y * NT + ty, j = blockIdx.x * NT + tx;
    float acc = 0.f;
    for (int c0 = 0; c0 < C; c0 += NT) {
        const int ik = blockIdx.y * NT + ty, jq = blockIdx.x * NT + ty;
        lk[ty][tx] = (ik < S && c0 + tx < C) ? k[(int64_t)ik * C + c0 + tx] : 0.f;
        lq[ty][tx] = (jq < S && c0 + tx < C) ? q[(int64_t)jq * C + c0 + tx] : 0.f;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NT; ++c) acc += lk[ty][c] * lq[tx][c];
        __syncthreads();
    }
    if (i < S && j < S) G[(int64_t)i * S + j] = acc * inv_t;
}

__global__ __launch_bounds__(THREADS) void nce_rows_kernel(float* __restrict__ G, int S, float inv_st, double* __restrict__ partials) {
    __shared__ float red[THREADS / 64];
    const int i = blockIdx.x;
    float* row = G + (int64_t)i * S;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < S; j += THREADS) m = fmaxf(m, row[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = red[0];
    for (int w = 1; w < THREADS / 64; ++w) m = fmaxf(m, red[w]);
    __syncthreads();
    float s = 0.f;
    for (int j = threadIdx.x; j < S; j += THREADS) s += expf(row[j] - m);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    s = 0.f;
    for (int w = 0; w < THREADS / 64; ++w) s += red[w];
    const float lse = m + logf(s);
    if (threadIdx.x == 0) partials[i] = (double)(lse - row[i]);          // -log softmax_ii
    __syncthreads();
    for (int j = threadIdx.x; j < S; j += THREADS) {
        const float p = expf(row[j] - lse);
        row[j] = (p - (j == i ? 1.0f : 0.0f)) * inv_st;
    }
}

// out[i][c] = g * sum_j G[i][j] * v[j][c]     (TRANS: sum_j G[j][i] * v[j][c])
// generic form: one row i per workgroup, one channel per thread, j sequential (any C)
template <bool TRANS>
__global__ __launch_bounds__(THREADS) void nce_grad_kernel(const float* __restrict__ G, const float* __restrict__ v, int S, int C,
                                                           const float* __restrict__ gout, float* __restrict__ out) {
    const int i = blockIdx.x;
    const float g = gout[0];
    for (int c = threadIdx.x; c < C; c += THREADS) {
        float acc = 0.f;
        for (int j = 0; j < S; ++j) acc += (TRANS ? G[(int64_t)j * S + i] : G[(int64_t)i * S + j]) * v[(int64_t)j * C + c];
        out[(int64_t)i * C + c] = g * acc;
    }
}
// C % 4 == 0 (the 256-channel features of the step): the generic form is a chain of S dependent load -> FMA steps per thread
// (178 / 131 us for S = 800 in the frame2voxel_full trace: pure latency).  Here a workgroup owns NCE_IT rows, a lane owns four
// channels (16-byte loads of v), the four waves take j = w, w + 4, ... with four loads in flight each, and the wave partials are
// added in wave order through LDS (fixed order: bit-repeatable).
constexpr int NCE_IT = 2;
template <bool TRANS>
__global__ __launch_bounds__(THREADS) void nce_grad_vec_kernel(const float* __restrict__ G, const float* __restrict__ v, int S, int C,
                                                               const float* __restrict__ gout, float* __restrict__ out) {
    static_assert(THREADS == 256, "four waves");
    __shared__ float red[4][NCE_IT][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.x * NCE_IT;
    const float g = gout[0];
    for (int cb = 0; cb < C; cb += 256) {
        const int c = cb + lane * 4;
        float4 acc[NCE_IT];
#pragma unroll
        for (int t = 0; t < NCE_IT; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) {
#pragma unroll 4
            for (int j = wave; j < S; j += 4) {
                const float4 vv = *reinterpret_cast<const float4*>(v + (int64_t)j * C + c);
#pragma unroll
                for (int t = 0; t < NCE_IT; ++t) {
                    const int i = i0 + t;
                    const float w = (i < S) ? (TRANS ? G[(int64_t)j * S + i] : G[(int64_t)i * S + j]) : 0.f;
                    acc[t].x += w * vv.x; acc[t].y += w * vv.y; acc[t].z += w * vv.z; acc[t].w += w * vv.w;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NCE_IT; ++t) *reinterpret_cast<float4*>(&red[wave][t][lane * 4]) = acc[t];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NCE_IT; ++t) {
            const int i = i0 + t, cc = cb + (int)threadIdx.x;
            if (i < S && cc < C)
                out[(int64_t)i * C + cc] = g * (((red[0][t][threadIdx.x] + red[1][t][threadIdx.x]) + red[2][t][threadIdx.x]) + red[3][t][threadIdx.x]);
        }
        __syncthreads();
    }
}

int grid_of(int64_t work_items, int per_block) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > MAX_PARTIALS) g = MAX_PARTIALS;
    return (int)g;
}
}  // namespace

extern "C" {

size_t oess_loss_partials_bytes(void) { return MAX_PARTIALS * sizeof(double); }

int oess_l1_mean_fwd(const void* a, const void* b, int64_t n, int is_bf16, void* partials, float* loss, oess_stream_t stream) {
    if (!a || !partials || !loss || n <= 0) return OESS_EINVAL;             // b == NULL: mean |a|
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
    const int grid = vec ? grid_of(n, THREADS * 32) : grid_of(n, THREADS * 8);
    if (vec) {
        if (is_bf16) hipLaunchKernelGGL(l1_fwd_vec_kernel<true>, dim3(grid), dim3(THREADS), 0, st, a, b, n, (double*)partials);
        else hipLaunchKernelGGL(l1_fwd_vec_kernel<false>, dim3(grid), dim3(THREADS), 0, st, a, b, n, (double*)partials);
    } else if (is_bf16) hipLaunchKernelGGL(l1_fwd_kernel<true>, dim3(grid), dim3(THREADS), 0, st, a, b, n, (double*)partials);
    else hipLaunchKernelGGL(l1_fwd_kernel<false>, dim3(grid), dim3(THREADS), 0, st, a, b, n, (double*)partials);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(THREADS), 0, st, (const double*)partials, grid, 0.0, 1.0 / (double)n, loss);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_l1_mean_bwd(const void* a, const void* b, int64_t n, int is_bf16, const float* grad_out, void* grad_a, void* grad_b,
                     oess_stream_t stream) {
    if (!a || !grad_out || (!grad_a && !grad_b) || n <= 0) return OESS_EINVAL;
    int64_t g = (n + THREADS * 4 - 1) / (THREADS * 4);
    if (g > 65536) g = 65536;
    hipStream_t st = (hipStream_t)stream;
    const float inv_n = (float)(1.0 / (double)n);
    if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)grad_a | (uintptr_t)grad_b) & 15) == 0) {
        int64_t gv = (n + THREADS * 16 - 1) / (THREADS * 16);
        if (gv > 65536) gv = 65536;
        if (is_bf16) hipLaunchKernelGGL(l1_bwd_vec_kernel<true>, dim3((unsigned)gv), dim3(THREADS), 0, st, a, b, n, grad_out, inv_n, grad_a, grad_b);
        else hipLaunchKernelGGL(l1_bwd_vec_kernel<false>, dim3((unsigned)gv), dim3(THREADS), 0, st, a, b, n, grad_out, inv_n, grad_a, grad_b);
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
    if (is_bf16) hipLaunchKernelGGL(l1_bwd_kernel<true>, dim3((unsigned)g), dim3(THREADS), 0, st, a, b, n, grad_out, inv_n, grad_a, grad_b);
    else hipLaunchKernelGGL(l1_bwd_kernel<false>, dim3((unsigned)g), dim3(THREADS), 0, st, a, b, n, grad_out, inv_n, grad_a, grad_b);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_cosine_mean_fwd(const void* a, long long a_pix_stride, const void* b, long long b_pix_stride, int64_t P, int C,
                         int is_bf16, float eps, void* partials, float* loss, oess_stream_t stream) {
    if (!a || !b || !partials || !loss || P <= 0 || C <= 0 || a_pix_stride < C || b_pix_stride < C) return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!is_bf16 && C <= COS_CMAX && a_pix_stride == C && b_pix_stride == C && (((uintptr_t)a | (uintptr_t)b) & 15) == 0) {
        const int gd = grid_of(P, THREADS * 4);
        hipLaunchKernelGGL(cos_fwd_dense_kernel, dim3(gd), dim3(THREADS), (size_t)2 * THREADS * C * sizeof(float), st, (const float*)a,
                           (const float*)b, P, C, eps, (double*)partials);
        hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(THREADS), 0, st, (const double*)partials, gd, 1.0, -1.0 / (double)P, loss);
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
    const int grid = grid_of(P, (THREADS / 64) * 8);
    if (is_bf16) hipLaunchKernelGGL(cos_fwd_kernel<true>, dim3(grid), dim3(THREADS), 0, st, a, (int64_t)a_pix_stride, b, (int64_t)b_pix_stride, P, C, eps, (double*)partials);
    else hipLaunchKernelGGL(cos_fwd_kernel<false>, dim3(grid), dim3(THREADS), 0, st, a, (int64_t)a_pix_stride, b, (int64_t)b_pix_stride, P, C, eps, (double*)partials);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(THREADS), 0, st, (const double*)partials, grid, 1.0, -1.0 / (double)P, loss);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_cosine_mean_bwd(const void* a, long long a_pix_stride, const void* b, long long b_pix_stride, int64_t P, int C,
                         int is_bf16, float eps, const float* grad_out, void* grad_a, long long ga_pix_stride, void* grad_b,
                         long long gb_pix_stride, oess_stream_t stream) {
    if (!a || !b || !grad_out || (!grad_a && !grad_b) || P <= 0 || C <= 0 || a_pix_stride < C || b_pix_stride < C) return OESS_EINVAL;
    int64_t g = (P + (THREADS / 64) * 4 - 1) / ((THREADS / 64) * 4);
    if (g > 65536) g = 65536;
    hipStream_t st = (hipStream_t)stream;
    const float inv_p = (float)(1.0 / (double)P);
    if (!is_bf16 && C <= COS_CMAX && a_pix_stride == C && b_pix_stride == C && (!grad_a || ga_pix_stride == C) && (!grad_b || gb_pix_stride == C) &&
        (((uintptr_t)a | (uintptr_t)b | (uintptr_t)grad_a | (uintptr_t)grad_b) & 15) == 0) {
        int64_t gd = (P + THREADS * 2 - 1) / (THREADS * 2);
        if (gd > 65536) gd = 65536;
        hipLaunchKernelGGL(cos_bwd_dense_kernel, dim3((unsigned)gd), dim3(THREADS), (size_t)2 * THREADS * C * sizeof(float), st, (const float*)a,
                           (const float*)b, P, C, eps, grad_out, inv_p, (float*)grad_a, (float*)grad_b);
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
    if (is_bf16) hipLaunchKernelGGL(cos_bwd_kernel<true>, dim3((unsigned)g), dim3(THREADS), 0, st, a, (int64_t)a_pix_stride, b, (int64_t)b_pix_stride, P, C, eps, grad_out, inv_p, grad_a, (int64_t)ga_pix_stride, grad_b, (int64_t)gb_pix_stride);
    else hipLaunchKernelGGL(cos_bwd_kernel<false>, dim3((unsigned)g), dim3(THREADS), 0, st, a, (int64_t)a_pix_stride, b, (int64_t)b_pix_stride, P, C, eps, grad_out, inv_p, grad_a, (int64_t)ga_pix_stride, grad_b, (int64_t)gb_pix_stride);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_nce_loss_fwd(const float* k, const float* q, int S, int C, float temperature, float* grad_logits /*[S x S]*/, void* partials,
                      size_t partials_bytes, float* loss, oess_stream_t stream) {
    if (!k || !q || !grad_logits || !partials || !loss || S <= 0 || C <= 0 || temperature <= 0.f || partials_bytes < (size_t)S * sizeof(double))
        return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 g2((S + NT - 1) / NT, (S + NT - 1) / NT);
    hipLaunchKernelGGL(nce_logits_kernel, g2, dim3(NT * NT), 0, st, k, q, S, C, 1.0f / temperature, grad_logits);
    hipLaunchKernelGGL(nce_rows_kernel, dim3(S), dim3(THREADS), 0, st, grad_logits, S, 1.0f / ((float)S * temperature), (double*)partials);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(THREADS), 0, st, (const double*)partials, S, 0.0, 1.0 / (double)S, loss);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_nce_loss_bwd(const float* grad_logits, const float* k, const float* q, int S, int C, const float* grad_out, float* grad_k,
                      float* grad_q, oess_stream_t stream) {
    if (!grad_logits || !k || !q || !grad_out || (!grad_k && !grad_q) || S <= 0 || C <= 0) return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (C & 3) == 0 && (((uintptr_t)k | (uintptr_t)q) & 15) == 0;
    const dim3 gv((S + NCE_IT - 1) / NCE_IT);
    if (grad_k) {
        if (vec) hipLaunchKernelGGL(nce_grad_vec_kernel<false>, gv, dim3(THREADS), 0, st, grad_logits, q, S, C, grad_out, grad_k);
        else hipLaunchKernelGGL(nce_grad_kernel<false>, dim3(S), dim3(THREADS), 0, st, grad_logits, q, S, C, grad_out, grad_k);
    }
    if (grad_q) {
        if (vec) hipLaunchKernelGGL(nce_grad_vec_kernel<true>, gv, dim3(THREADS), 0, st, grad_logits, k, S, C, grad_out, grad_q);
        else hipLaunchKernelGGL(nce_grad_kernel<true>, dim3(S), dim3(THREADS), 0, st, grad_logits, k, S, C, grad_out, grad_q);
    }
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // extern "C"
