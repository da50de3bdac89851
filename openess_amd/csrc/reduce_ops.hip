// HBM-bound reductions of the OpenESS hot path for gfx950:
//   K2  masked (non-zero) normalisation        K7  superpixel scatter-mean fwd / bwd
//   K9  Dice + cross-entropy fwd / bwd         K11 confusion matrix
// All are single-pass-over-HBM streaming kernels: 16-byte loads per lane, wave-shuffle + LDS block
// reductions, one global atomic per (workgroup, accumulator).  No MFMA (integer / byte / fp32 work).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>
#include "oess.h"
#include "oess_common.h"

namespace {
using namespace oess;
constexpr int THREADS = 256;

__host__ int stream_grid(int64_t work_items, int per_block) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;          // 256 CUs x 8 resident workgroups; grid-stride the rest
    return (int)g;
}

// block-wide sum of NV doubles per thread -> atomicAdd into dst[0..NV)
template <int NV>
__device__ __forceinline__ void block_atomic_add(double (&v)[NV], double* dst) {
    __shared__ double red[THREADS / 64][NV];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = wave_sum(v[i]);
        if (lane == 0) red[w][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0;
        for (int k = 0; k < THREADS / 64; ++k) s += red[k][threadIdx.x];
        if (s != 0.0) atomicAdd(&dst[threadIdx.x], s);
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------ K2
// Tensor viewed as `nchunk` contiguous chunks of L floats; chunk j starts at in + in_off + j*in_stride,
// out is dense.  (Dense tensor: nchunk = 1.)
// grid = (blocks per chunk, nchunk): no per-element index division (a 64-bit divide per float4 made this ALU-bound).
__global__ __launch_bounds__(THREADS) void norm_stats_kernel(const float* __restrict__ in, int64_t L, int64_t nchunk,
                                                             int64_t in_stride, int64_t in_off, int vec,
                                                             double* __restrict__ stats) {
    double acc[3] = {0.0, 0.0, 0.0};
    // blockIdx.z = slice of a multi-slice launch (oess_masked_stats_slices_f32): slice z starts z * L floats further and
    // accumulates into stats[4 z ..]
    const float* src = in + in_off + (int64_t)blockIdx.z * L + (int64_t)blockIdx.y * in_stride;
    stats += 4 * blockIdx.z;
    const int64_t tid = (int64_t)blockIdx.x * THREADS + threadIdx.x, nthr = (int64_t)gridDim.x * THREADS;
    if (vec) {
        // four 16-byte loads in flight per lane; accumulation stays in double (the reference sums in float64-exact order
        // only up to its final float32 rounding, tests pin 2e-5 / 2e-6)
        const int64_t L4 = L >> 2;
        int64_t r = tid;
        for (; r + 3 * nthr < L4; r += 4 * nthr) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(src + (r + u * nthr) * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double d = (double)a[k];
                    acc[0] += d; acc[1] += d * d; acc[2] += (a[k] != 0.0f) ? 1.0 : 0.0;
                }
            }
        }
        for (; r < L4; r += nthr) {
            const float4 v = *reinterpret_cast<const float4*>(src + r * 4);
            const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double d = (double)a[k];
                acc[0] += d; acc[1] += d * d; acc[2] += (a[k] != 0.0f) ? 1.0 : 0.0;
            }
        }
    } else {
        for (int64_t r = tid; r < L; r += nthr) {
            const float a = src[r];
            const double d = (double)a;
            acc[0] += d; acc[1] += d * d; acc[2] += (a != 0.0f) ? 1.0 : 0.0;
        }
    }
    block_atomic_add<3>(acc, stats);
}

__global__ __launch_bounds__(THREADS) void norm_apply_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             int64_t L, int64_t nchunk, int64_t in_stride,
                                                             int64_t in_off, int vec, const double* __restrict__ stats) {
    const double nnz = stats[2];
    const bool active = nnz > 0.0;                    // inference_utils.py:80 `if num_nonzeros > 0`
    // inference_utils.py:81-82 in float32: mean = sum/n ; std = sqrt(sum(x^2)/n - mean^2)
    const float nf = (float)nnz;
    const float mean = (float)stats[0] / nf;
    const float var = __fsub_rn((float)stats[1] / nf, __fmul_rn(mean, mean));
    const float stdv = sqrtf(var);
    const float* src = in + in_off + (int64_t)blockIdx.y * in_stride;
    float* dst = out + (int64_t)blockIdx.y * L;
    const int64_t tid = (int64_t)blockIdx.x * THREADS + threadIdx.x, nthr = (int64_t)gridDim.x * THREADS;
    if (vec) {
        const int64_t L4 = L >> 2;
        for (int64_t r = tid; r < L4; r += nthr) {
            float4 v = *reinterpret_cast<const float4*>(src + r * 4);
            if (active) {
                float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    a[k] = __fmul_rn((a[k] != 0.0f) ? 1.0f : 0.0f, __fsub_rn(a[k], mean)) / stdv;
                v = make_float4(a[0], a[1], a[2], a[3]);
            }
            *reinterpret_cast<float4*>(dst + r * 4) = v;
        }
    } else {
        for (int64_t r = tid; r < L; r += nthr) {
            float a = src[r];
            if (active) a = __fmul_rn((a != 0.0f) ? 1.0f : 0.0f, __fsub_rn(a, mean)) / stdv;
            dst[r] = a;
        }
    }
}

// blocks per chunk: >= 8 items per thread, the whole launch <= 256 workgroups (one block reduction + 3 same-address atomics each)
__host__ dim3 norm_grid(int64_t L, int64_t nchunk, int vec) {
    int64_t gx = (L / (vec ? 4 : 1) + THREADS * 8 - 1) / (THREADS * 8);
    int64_t cap = 256 / (nchunk < 1 ? 1 : nchunk);      // measured: 2048 workgroups 23.9 us, 256 14.5 us (same-address atomics serialise in L2)
    if (cap < 1) cap = 1;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    return dim3((unsigned)gx, (unsigned)nchunk);
}

int run_normalize(const float* in, float* out, int64_t L, int64_t nchunk, int64_t in_stride, int64_t in_off,
                  double* stats, hipStream_t st) {
    if (!in || !out || !stats || L < 0 || nchunk < 0) return OESS_EINVAL;
    OESS_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(double), st));
    if (L * nchunk == 0) return OESS_OK;
    const int vec = ((L & 3) == 0) && ((in_stride & 3) == 0) && ((in_off & 3) == 0) &&
                    (((uintptr_t)in & 15) == 0) && (((uintptr_t)out & 15) == 0);
    if (nchunk > 65535) return OESS_EINVAL;
    const dim3 grid = norm_grid(L, nchunk, vec);
    hipLaunchKernelGGL(norm_stats_kernel, grid, dim3(THREADS), 0, st, in, L, nchunk, in_stride, in_off, vec, stats);
    hipLaunchKernelGGL(norm_apply_kernel, grid, dim3(THREADS), 0, st, in, out, L, nchunk, in_stride, in_off, vec,
                       stats);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

// ------------------------------------------------------------------------------------------ K7
// Wavefront-segmented scatter-mean.  Feature rows are pixel-major [P][Cf].  A workgroup owns a run
// of consecutive pixels and a 64-channel slice; lane = channel, the 4 waves take interleaved
// sub-runs.  Superpixels are spatially coherent, so each lane keeps a RUN accumulator in a register
// and only touches LDS (ds_add_f32) when the id changes; LDS holds [256 local ids][64 ch].  Touched
// rows are flushed with one global atomic per (workgroup, id, channel).  Raw ids outside [0,256)
// (the reference reads uint8 PNGs, so they do not occur there) take a direct global-atomic path.
constexpr int SEG_CH = 64;
constexpr int SEG_LOCAL = 256;
constexpr int SEG_PIX_PER_WG = 4096;

template <bool BF16>
__device__ __forceinline__ float load_feat(const void* feat, int64_t idx) {
    if (BF16) return bf16_to_f32(((const uint16_t*)feat)[idx]);
    return ((const float*)feat)[idx];
}

template <bool BF16>
__global__ __launch_bounds__(THREADS) void segmean_fwd_kernel(const void* __restrict__ feat,
                                                              const int64_t* __restrict__ ids, int64_t P, int64_t pps,
                                                              int sps, int Cf, int S, float* __restrict__ k,
                                                              float* __restrict__ count) {
    __shared__ float acc[SEG_LOCAL][SEG_CH];
    __shared__ int cnt[SEG_LOCAL];
    __shared__ int touched[SEG_LOCAL];
    const int slice = blockIdx.y;                        // 64-channel slice
    const int c = slice * SEG_CH + (threadIdx.x & 63);
    const bool c_ok = c < Cf;
    const int wave = threadIdx.x >> 6;
    // chunk never crosses a sample boundary: chunks are laid out per sample
    const int64_t chunks_per_sample = (pps + SEG_PIX_PER_WG - 1) / SEG_PIX_PER_WG;
    const int64_t b = blockIdx.x / chunks_per_sample, ch = blockIdx.x - b * chunks_per_sample;
    const int64_t p_beg = b * pps + ch * SEG_PIX_PER_WG;
    int64_t p_end = p_beg + SEG_PIX_PER_WG;
    if (p_end > (b + 1) * pps) p_end = (b + 1) * pps;
    if (p_end > P) p_end = P;
    for (int i = threadIdx.x; i < SEG_LOCAL * SEG_CH; i += THREADS) (&acc[0][0])[i] = 0.0f;
    for (int i = threadIdx.x; i < SEG_LOCAL; i += THREADS) { cnt[i] = 0; touched[i] = 0; }
    __syncthreads();
    const int64_t id_off = b * (int64_t)sps;
    // each wave walks a contiguous quarter of the chunk so that runs stay long
    const int64_t len = p_end - p_beg;
    const int64_t q = (len + 3) / 4;
    int64_t w_beg = p_beg + wave * q, w_end = w_beg + q;
    if (w_end > p_end) w_end = p_end;
    int64_t cur = -1;                                    // current raw id of the run (wave-uniform)
    float run = 0.0f;
    int run_n = 0;
    const int lane = threadIdx.x & 63;
    auto flush = [&]() {
        if (run_n == 0) return;
        if (cur >= 0 && cur < SEG_LOCAL) {
            if (c_ok) atomicAdd(&acc[cur][lane], run);
            if (lane == 0) { atomicAdd(&cnt[cur], run_n); touched[cur] = 1; }
        } else {
            const int64_t gid = cur + id_off;
            if (gid >= 0 && gid < S) {
                if (c_ok) atomicAdd(&k[gid * Cf + c], run);
                if (lane == 0 && slice == 0) atomicAdd(&count[gid], (float)run_n);
            }
        }
    };
    constexpr int U = 16;                                // independent loads in flight per wave
    for (int64_t p = w_beg; p < w_end; p += U) {
        float v[U];
        int64_t id[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t pu = p + u;
            const bool ok = pu < w_end;
            id[u] = ok ? ids[pu] : cur;                  // wave-uniform address -> scalar load
            v[u] = (ok && c_ok) ? load_feat<BF16>(feat, pu * Cf + c) : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + u < w_end) {
                if (id[u] != cur) { flush(); cur = id[u]; run = 0.0f; run_n = 0; }
                run += v[u];
                run_n += 1;
            }
        }
    }
    flush();
    __syncthreads();
    for (int i = wave; i < SEG_LOCAL; i += THREADS / 64) {
        if (!touched[i]) continue;
        const int64_t gid = i + id_off;
        if (gid < 0 || gid >= S) continue;
        if (c_ok) atomicAdd(&k[gid * Cf + c], acc[i][threadIdx.x & 63]);
        if ((threadIdx.x & 63) == 0 && slice == 0) atomicAdd(&count[gid], (float)cnt[i]);
    }
}

// Vectorised form (C/CPL lanes per pixel, CPL = 8 bf16 / 4 fp32 channels = one 16-byte load per lane): a wave covers
// 64/LPP whole pixel rows per load instruction and keeps U of them in flight; each LPP-lane group walks its own
// contiguous pixel run with a register run-accumulator and only touches the LDS table [VLOCAL ids][C] when the id
// changes.  One workgroup per CU (the table is up to 128 KB); 8 groups x 8 loads x 512 B = 32 KB in flight per CU.

constexpr int SEGV_THREADS = 512;        // 8 waves on the one workgroup a CU can hold (the LDS table is up to 128 KB)
template <bool BF16, int LPP>
__global__ __launch_bounds__(SEGV_THREADS) void segmean_fwd_vec_kernel(const void* __restrict__ feat, const int64_t* __restrict__ ids,
                                                                  int64_t P, int64_t pps, int sps, int S, int vlocal,
                                                                  float* __restrict__ k, float* __restrict__ count, int pix_per_wg) {
    constexpr int CPL = BF16 ? 8 : 4;
    constexpr int Cf = LPP * CPL;
    constexpr int GROUPS = SEGV_THREADS / LPP;
    extern __shared__ __attribute__((aligned(16))) unsigned char seg_smem[];
    float* acc = reinterpret_cast<float*>(seg_smem);                    // [vlocal][Cf]
    int* cnt = reinterpret_cast<int*>(acc + (size_t)vlocal * Cf);       // [vlocal]
    const int sub = threadIdx.x % LPP, grp = threadIdx.x / LPP;
    const int64_t chunks_per_sample = (pps + pix_per_wg - 1) / pix_per_wg;
    const int64_t b = blockIdx.x / chunks_per_sample, ch = blockIdx.x - b * chunks_per_sample;
    const int64_t p_beg = b * pps + ch * pix_per_wg;
    int64_t p_end = p_beg + pix_per_wg;
    if (p_end > (b + 1) * pps) p_end = (b + 1) * pps;
    if (p_end > P) p_end = P;
    for (int i = threadIdx.x; i < vlocal * Cf; i += SEGV_THREADS) acc[i] = 0.0f;
    for (int i = threadIdx.x; i < vlocal; i += SEGV_THREADS) cnt[i] = 0;
    __syncthreads();
    const int64_t id_off = b * (int64_t)sps;
    const int64_t len = p_end - p_beg;
    const int64_t q = (len + GROUPS - 1) / GROUPS;
    int64_t g_beg = p_beg + grp * q, g_end = g_beg + q;
    if (g_end > p_end) g_end = p_end;
    int64_t cur = -1;
    float run[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) run[c] = 0.f;
    int run_n = 0;
    auto flush = [&]() {
        if (run_n == 0) return;
        if (cur >= 0 && cur < vlocal) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) atomicAdd(&acc[cur * Cf + sub * CPL + c], run[c]);
            if (sub == 0) atomicAdd(&cnt[cur], run_n);
        } else {
            const int64_t gid = cur + id_off;
            if (gid >= 0 && gid < S) {
#pragma unroll
                for (int c = 0; c < CPL; ++c) atomicAdd(&k[gid * Cf + sub * CPL + c], run[c]);
                if (sub == 0) atomicAdd(&count[gid], (float)run_n);
            }
        }
    };
    constexpr int U = 8;
    // Software-pipelined: the U row loads (and ids) of iteration i+1 are issued BEFORE iteration i is walked.  Without it an
    // iteration was "issue U loads, wait for all of them, walk U pixels" -- one memory latency per 8 pixels of every lane
    // group, which is why the bf16 features (half the bytes per pixel) took exactly as long as the fp32 ones.
    using RawRow = typename std::conditional<BF16, uint4, float4>::type;
    RawRow cur_r[U], nxt_r[U];
    int64_t cur_id[U], nxt_id[U];
    auto issue = [&](int64_t p, RawRow (&r)[U], int64_t (&id)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t pu = p + u;
            const bool ok = pu < g_end;
            id[u] = ok ? ids[pu] : (int64_t)-1;
            const int64_t pc = ok ? pu : (g_end > g_beg ? g_end - 1 : p_beg);          // clamped: always a valid row of this chunk
            if constexpr (BF16) r[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(feat) + pc * Cf + sub * 8);
            else r[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(feat) + pc * Cf + sub * 4);
        }
    };
    if (g_beg < g_end) issue(g_beg, cur_r, cur_id);
    for (int64_t p = g_beg; p < g_end; p += U) {
        if (p + U < g_end) issue(p + U, nxt_r, nxt_id);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + u < g_end) {
                float v[CPL];
                if constexpr (BF16) {
                    union { uint4 q4; uint16_t h[8]; } r;
                    r.q4 = cur_r[u];
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = bf16_to_f32(r.h[c]);
                } else {
                    v[0] = cur_r[u].x; v[1] = cur_r[u].y; v[2] = cur_r[u].z; v[3] = cur_r[u].w;
                }
                if (cur_id[u] != cur) {
                    flush();
                    cur = cur_id[u];
#pragma unroll
                    for (int c = 0; c < CPL; ++c) run[c] = 0.f;
                    run_n = 0;
                }
#pragma unroll
                for (int c = 0; c < CPL; ++c) run[c] += v[c];
                run_n += 1;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { cur_r[u] = nxt_r[u]; cur_id[u] = nxt_id[u]; }
    }
    flush();
    __syncthreads();
    for (int i = grp; i < vlocal; i += GROUPS) {
        const int n = cnt[i];
        if (n == 0) continue;
        const int64_t gid = i + id_off;
        if (gid < 0 || gid >= S) continue;
#pragma unroll
        for (int c = 0; c < CPL; ++c) atomicAdd(&k[gid * Cf + sub * CPL + c], acc[i * Cf + sub * CPL + c]);
        if (sub == 0) atomicAdd(&count[gid], (float)n);
    }
}

__global__ __launch_bounds__(THREADS) void segmean_finalize_kernel(float* __restrict__ k, const float* __restrict__ count,
                                                                   int S, int Cf) {
    const int64_t n = (int64_t)S * Cf;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS)
        k[i] = k[i] / __fadd_rn(count[i / Cf], 1e-6f);     // pretrain_trainer.py:462
}

template <bool BF16>
__global__ __launch_bounds__(THREADS) void segmean_bwd_kernel(const float* __restrict__ gk, const float* __restrict__ count,
                                                              const int64_t* __restrict__ ids, int64_t P, int64_t pps,
                                                              int sps, int Cf, int S, void* __restrict__ gfeat) {
    // lane_c = 4-channel chunk fixed per thread, rows = THREADS / (Cf/4) pixels per iteration, blockIdx.y = sample:
    // no per-element 64-bit divisions (they made the first version 3x slower than the stream rate)
    const int cq = Cf >> 2;
    const int rows = THREADS / cq;
    const int lane_c = threadIdx.x % cq, row = threadIdx.x / cq;
    if (row >= rows) return;
    const int c4 = lane_c * 4;
    const int64_t b = blockIdx.y;
    const int64_t id_off = b * (int64_t)sps;
    for (int64_t q = (int64_t)blockIdx.x * rows + row; q < pps; q += (int64_t)gridDim.x * rows) {
        const int64_t p = b * pps + q;
        const int64_t gid = ids[p] + id_off;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gid >= 0 && gid < S) {
            const float d = __fadd_rn(count[gid], 1e-6f);
            const float4 s = *reinterpret_cast<const float4*>(gk + gid * Cf + c4);
            g = make_float4(s.x / d, s.y / d, s.z / d, s.w / d);
        }
        if (BF16) {
            *reinterpret_cast<uint2*>((uint16_t*)gfeat + p * Cf + c4) = make_uint2(pack_bf16x2(g.x, g.y), pack_bf16x2(g.z, g.w));
        } else {
            *reinterpret_cast<float4*>((float*)gfeat + p * Cf + c4) = g;
        }
    }
}

// Backward with the divisions hoisted: every pixel of a superpixel receives the SAME row gk[id] / (count[id] + 1e-6), so the
// S x Cf quotients (IEEE divides, ~25 instructions each with correctly-rounded division) are formed once per call -- already
// rounded to the output dtype -- and the per-pixel pass is a pure row gather: one 16-byte load from the (L2-resident) table and
// one non-temporal 16-byte store per lane.  The per-pixel form above spent ~100 VALU instructions per 8 bytes written.
template <bool BF16>
__global__ __launch_bounds__(THREADS) void segmean_bwd_table_kernel(const float* __restrict__ gk, const float* __restrict__ count, int S,
                                                                    int Cf, void* __restrict__ table) {
    const int64_t n = (int64_t)S * Cf;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const float q = gk[i] / __fadd_rn(count[i / Cf], 1e-6f);
        if (BF16) reinterpret_cast<uint16_t*>(table)[i] = f32_to_bf16(q); else reinterpret_cast<float*>(table)[i] = q;
    }
}

// lanes per pixel = row bytes / 16; blockIdx.y = sample; out-of-table ids give zero rows
__global__ __launch_bounds__(THREADS) void segmean_bwd_gather_kernel(const uint4* __restrict__ table, const int64_t* __restrict__ ids,
                                                                     int64_t pps, int sps, int lpp, int S, uint4* __restrict__ gfeat) {
    typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
    const int rows = THREADS / lpp;
    const int lane_c = threadIdx.x % lpp, row = threadIdx.x / lpp;
    if (row >= rows) return;
    const int64_t b = blockIdx.y;
    const int64_t id_off = b * (int64_t)sps;
    for (int64_t q = (int64_t)blockIdx.x * rows + row; q < pps; q += (int64_t)gridDim.x * rows) {
        const int64_t p = b * pps + q;
        const int64_t gid = ids[p] + id_off;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (gid >= 0 && gid < S) v = table[gid * lpp + lane_c];
        __builtin_nontemporal_store(u32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_nt*>(gfeat + p * lpp + lane_c));
    }
}

// ------------------------------------------------------------------------------------------ K9
template <bool BF16>
__device__ __forceinline__ float load_logit(const void* base, int64_t idx) {
    if (BF16) return bf16_to_f32(((const uint16_t*)base)[idx]);
    return ((const float*)base)[idx];
}

struct LossGeom { int64_t P, pps, sb, sp, sc; int K, ignore; };
constexpr int TASK_LOSS_MAX_ROWS = 1024;           // forward workgroups = rows of partial sums

// sums layout: inter[K] psq[K] ysum[K] ce_sum n_valid
template <int KMAX, bool BF16>
__global__ __launch_bounds__(THREADS) void task_loss_fwd_kernel(const void* __restrict__ logits,
                                                                const int64_t* __restrict__ target, LossGeom g,
                                                                double* __restrict__ sums) {
    float inter[KMAX], psq[KMAX], ysum[KMAX];
#pragma unroll
    for (int c = 0; c < KMAX; ++c) { inter[c] = 0.f; psq[c] = 0.f; ysum[c] = 0.f; }
    float ce = 0.f, nvalid = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * THREADS + threadIdx.x; p < g.P; p += (int64_t)gridDim.x * THREADS) {
        const int64_t t = target[p];
        if (t == g.ignore) continue;                       // mask = target != ignore (loss_functions.py:115)
        const int64_t b = p / g.pps, r = p - b * g.pps;
        const int64_t base = b * g.sb + r * g.sp;
        float z[KMAX];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < KMAX; ++c)
            if (c < g.K) { z[c] = load_logit<BF16>(logits, base + c * g.sc); m = fmaxf(m, z[c]); }
        float den = 0.f, zt = 0.f;
#pragma unroll
        for (int c = 0; c < KMAX; ++c)
            if (c < g.K) { const float sh = z[c] - m; if (c == t) zt = sh; z[c] = expf(sh); den += z[c]; }
        const float inv = 1.0f / den;
#pragma unroll
        for (int c = 0; c < KMAX; ++c)
            if (c < g.K) {
                const float pr = z[c] * inv;
                psq[c] += pr * pr;
                if (c == t) { inter[c] += pr; ysum[c] += 1.0f; }
            }
        ce += logf(den) - zt;                              // -log_softmax(z)[t]
        nvalid += 1.0f;
    }
    // block reduce in double, (3K+2) values
    __shared__ double red[THREADS / 64][3 * KMAX + 2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        double a = wave_sum((double)inter[c]), b2 = wave_sum((double)psq[c]), y = wave_sum((double)ysum[c]);
        if (lane == 0) { red[w][c] = a; red[w][KMAX + c] = b2; red[w][2 * KMAX + c] = y; }
    }
    {
        double a = wave_sum((double)ce), b2 = wave_sum((double)nvalid);
        if (lane == 0) { red[w][3 * KMAX] = a; red[w][3 * KMAX + 1] = b2; }
    }
    __syncthreads();
    // this workgroup's row of partial sums (no atomics: task_loss_finalize_kernel adds the rows in a fixed order)
    double* row = sums + (size_t)(3 * g.K + 2) * (1 + blockIdx.x);
    for (int i = threadIdx.x; i < 3 * KMAX + 2; i += THREADS) {
        double s = 0;
        for (int k2 = 0; k2 < THREADS / 64; ++k2) s += red[k2][i];
        int dst;
        if (i < 3 * KMAX) { const int grp = i / KMAX, c = i % KMAX; if (c >= g.K) continue; dst = grp * g.K + c; }
        else dst = 3 * g.K + (i - 3 * KMAX);
        row[dst] = s;
    }
}

// Dense NHWC fp32 logits ([P][K], what the decoder's classifier conv writes): a lane's K logits are K consecutive floats, so K
// scalar loads per pixel touch every 128-byte line of the wave's span K times at 4 / (4 K) efficiency (76 us forward, 126 us
// backward for 2.25 M pixels x 11 classes = 1.2-1.6 TB/s).  Here a workgroup moves its 256 pixels x K floats through LDS with
// 16-byte accesses (row stride K words: conflict-free reads for odd K) -- same arithmetic as the generic kernels above.
template <int KMAX>
__device__ __forceinline__ void dense_tile_load(const float* __restrict__ src, int64_t total, float* sbuf) {
    const int64_t n4 = total >> 2;
    for (int64_t j = threadIdx.x; j < n4; j += THREADS)
        *reinterpret_cast<float4*>(sbuf + 4 * j) = *reinterpret_cast<const float4*>(src + 4 * j);
    for (int64_t j = 4 * n4 + threadIdx.x; j < total; j += THREADS) sbuf[j] = src[j];
}

template <int KMAX>
__global__ __launch_bounds__(THREADS) void task_loss_fwd_dense_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                                      LossGeom g, double* __restrict__ sums) {
    __shared__ __attribute__((aligned(16))) float sbuf[THREADS * KMAX];
    float inter[KMAX], psq[KMAX], ysum[KMAX];
#pragma unroll
    for (int c = 0; c < KMAX; ++c) { inter[c] = 0.f; psq[c] = 0.f; ysum[c] = 0.f; }
    float ce = 0.f, nvalid = 0.f;
    for (int64_t p0 = (int64_t)blockIdx.x * THREADS; p0 < g.P; p0 += (int64_t)gridDim.x * THREADS) {
        const int64_t n = (g.P - p0 < THREADS) ? g.P - p0 : THREADS;
        __syncthreads();                                   // the previous tile has been read
        dense_tile_load<KMAX>(logits + p0 * g.K, n * g.K, sbuf);
        __syncthreads();
        const int64_t p = p0 + threadIdx.x;
        const int64_t t = (threadIdx.x < n) ? target[p] : (int64_t)g.ignore;
        if (t == g.ignore) continue;                       // (no barrier below this point in the iteration)
        float z[KMAX];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < KMAX; ++c)
            if (c < g.K) { z[c] = sbuf[threadIdx.x * g.K + c]; m = fmaxf(m, z[c]); }
        float den = 0.f, zt = 0.f;
#pragma unroll
        for (int c = 0; c < KMAX; ++c)
            if (c < g.K) { const float sh = z[c] - m; if (c == t) zt = sh; z[c] = expf(sh); den += z[c]; }
        const float inv = 1.0f / den;
#pragma unroll
        for (int c = 0; c < KMAX; ++c)
            if (c < g.K) {
                const float pr = z[c] * inv;
                psq[c] += pr * pr;
                if (c == t) { inter[c] += pr; ysum[c] += 1.0f; }
            }
        ce += logf(den) - zt;
        nvalid += 1.0f;
    }
    __shared__ double red[THREADS / 64][3 * KMAX + 2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        double a = wave_sum((double)inter[c]), b2 = wave_sum((double)psq[c]), y = wave_sum((double)ysum[c]);
        if (lane == 0) { red[w][c] = a; red[w][KMAX + c] = b2; red[w][2 * KMAX + c] = y; }
    }
    {
        double a = wave_sum((double)ce), b2 = wave_sum((double)nvalid);
        if (lane == 0) { red[w][3 * KMAX] = a; red[w][3 * KMAX + 1] = b2; }
    }
    __syncthreads();
    // this workgroup's row of partial sums (no atomics: task_loss_finalize_kernel adds the rows in a fixed order)
    double* row = sums + (size_t)(3 * g.K + 2) * (1 + blockIdx.x);
    for (int i = threadIdx.x; i < 3 * KMAX + 2; i += THREADS) {
        double s = 0;
        for (int k2 = 0; k2 < THREADS / 64; ++k2) s += red[k2][i];
        int dst;
        if (i < 3 * KMAX) { const int grp = i / KMAX, c = i % KMAX; if (c >= g.K) continue; dst = grp * g.K + c; }
        else dst = 3 * g.K + (i - 3 * KMAX);
        row[dst] = s;
    }
}

template <int KMAX>
__global__ __launch_bounds__(THREADS) void task_loss_bwd_dense_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                                      LossGeom g, const double* __restrict__ sums, int flags,
                                                                      float gscale_in, const float* __restrict__ gscale_dev,
                                                                      float* __restrict__ grad) {
    __shared__ __attribute__((aligned(16))) float sbuf[THREADS * KMAX];
    const float gscale = gscale_dev ? gscale_in * gscale_dev[0] : gscale_in;
    __shared__ float sN[KMAX], sD[KMAX];
    if (threadIdx.x < KMAX) {
        const int c = threadIdx.x;
        if (c < g.K) {
            sN[c] = (float)sums[c] * 2.0f + 1.0f;
            sD[c] = (float)sums[g.K + c] + (float)sums[2 * g.K + c] + 1.0f;
        }
    }
    const float inv_nvalid = (float)(1.0 / sums[3 * g.K + 1]);
    const float invK = 1.0f / (float)g.K;
    for (int64_t p0 = (int64_t)blockIdx.x * THREADS; p0 < g.P; p0 += (int64_t)gridDim.x * THREADS) {
        const int64_t n = (g.P - p0 < THREADS) ? g.P - p0 : THREADS;
        const int64_t total = n * g.K;
        __syncthreads();                                   // sN / sD ready; the previous tile's gradient has left LDS
        dense_tile_load<KMAX>(logits + p0 * g.K, total, sbuf);
        __syncthreads();
        if (threadIdx.x < n) {
            const int64_t t = target[p0 + threadIdx.x];
            float dz[KMAX];
            if (t == g.ignore) {
#pragma unroll
                for (int c = 0; c < KMAX; ++c) dz[c] = 0.f;
            } else {
                float z[KMAX];
                float m = -INFINITY;
#pragma unroll
                for (int c = 0; c < KMAX; ++c)
                    if (c < g.K) { z[c] = sbuf[threadIdx.x * g.K + c]; m = fmaxf(m, z[c]); }
                float den = 0.f;
#pragma unroll
                for (int c = 0; c < KMAX; ++c)
                    if (c < g.K) { z[c] = expf(z[c] - m); den += z[c]; }
                const float inv = 1.0f / den;
                float gp[KMAX];
                float dot = 0.f;
#pragma unroll
                for (int c = 0; c < KMAX; ++c)
                    if (c < g.K) {
                        const float pr = z[c] * inv;
                        z[c] = pr;
                        float gg = 0.f;
                        if ((flags & 1) && c != g.ignore) {
                            const float y = (c == t) ? 1.0f : 0.0f;
                            gg = invK * (2.0f * pr * sN[c] - 2.0f * y * sD[c]) / (sD[c] * sD[c]);
                        }
                        gp[c] = gg;
                        dot += gg * pr;
                    }
#pragma unroll
                for (int c = 0; c < KMAX; ++c)
                    if (c < g.K) {
                        float d = z[c] * (gp[c] - dot);
                        if (flags & 2) d += (z[c] - ((c == t) ? 1.0f : 0.0f)) * inv_nvalid;
                        dz[c] = d * gscale;
                    }
            }
            // a thread only ever touches its own K words of the tile: the gradient may overwrite the logits in place
#pragma unroll
            for (int c = 0; c < KMAX; ++c)
                if (c < g.K) sbuf[threadIdx.x * g.K + c] = dz[c];
        }
        __syncthreads();
        float* dst = grad + p0 * g.K;
        const int64_t n4 = total >> 2;
        for (int64_t j = threadIdx.x; j < n4; j += THREADS)
            *reinterpret_cast<float4*>(dst + 4 * j) = *reinterpret_cast<const float4*>(sbuf + 4 * j);
        for (int64_t j = 4 * n4 + threadIdx.x; j < total; j += THREADS) dst[j] = sbuf[j];
    }
}

// sums[0 .. 3K+2) = the rows of the forward workgroups added in a fixed order (bit-repeatable), then the loss.
// 1024 threads = 128 columns x 8 row lanes: lane l adds rows l, l + 8, ... in order, the eight lane sums are added 0..7
// (one thread per column walking all 1 024 rows took 42 us).
__global__ __launch_bounds__(1024) void task_loss_finalize_kernel(double* __restrict__ sums, int rows, int K, int ignore, int flags,
                                                                  float* __restrict__ loss_out) {
    __shared__ double part[8][128];
    __shared__ double tot[3 * 32 + 2];
    const int nv = 3 * K + 2;
    const int col = threadIdx.x & 127, rl = threadIdx.x >> 7;
    {
        double s = 0.0;
        if (col < nv) {
            const double* p = sums + nv + col;
#pragma unroll 4
            for (int r = rl; r < rows; r += 8) s += p[(size_t)r * nv];
        }
        part[rl][col] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < nv) {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < 8; ++l) s += part[l][threadIdx.x];
        tot[threadIdx.x] = s;
        sums[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float dice = 0.f;
    for (int c = 0; c < K; ++c) {
        if (c == ignore) continue;                          // loss_functions.py:128
        const float num = (float)tot[c] * 2.0f + 1.0f;      // BinaryDiceLoss: smooth = 1, p = 2
        const float den = (float)tot[K + c] + (float)tot[2 * K + c] + 1.0f;
        dice += 1.0f - num / den;
    }
    dice /= (float)K;                                       // total_loss / target.shape[1]
    const float ce = (float)(tot[3 * K] / tot[3 * K + 1]);
    float total = 0.f;
    if (flags & 1) total += dice;
    if (flags & 2) total += ce;
    loss_out[0] = total; loss_out[1] = dice; loss_out[2] = ce;
}

template <int KMAX, bool BF16, bool GBF16>
__global__ __launch_bounds__(THREADS) void task_loss_bwd_kernel(const void* __restrict__ logits,
                                                                const int64_t* __restrict__ target, LossGeom g,
                                                                const double* __restrict__ sums, int flags,
                                                                float gscale_in, const float* __restrict__ gscale_dev,
                                                                void* __restrict__ grad) {
    const float gscale = gscale_dev ? gscale_in * gscale_dev[0] : gscale_in;
    __shared__ float sN[KMAX], sD[KMAX];
    if (threadIdx.x < KMAX) {
        const int c = threadIdx.x;
        if (c < g.K) {
            sN[c] = (float)sums[c] * 2.0f + 1.0f;
            sD[c] = (float)sums[g.K + c] + (float)sums[2 * g.K + c] + 1.0f;
        }
    }
    __syncthreads();
    const float inv_nvalid = (float)(1.0 / sums[3 * g.K + 1]);
    const float invK = 1.0f / (float)g.K;
    for (int64_t p = (int64_t)blockIdx.x * THREADS + threadIdx.x; p < g.P; p += (int64_t)gridDim.x * THREADS) {
        const int64_t t = target[p];
        const int64_t b = p / g.pps, r = p - b * g.pps;
        const int64_t base = b * g.sb + r * g.sp;
        float dz[KMAX];
        if (t == g.ignore) {
#pragma unroll
            for (int c = 0; c < KMAX; ++c) dz[c] = 0.f;
        } else {
            float z[KMAX];
            float m = -INFINITY;
#pragma unroll
            for (int c = 0; c < KMAX; ++c)
                if (c < g.K) { z[c] = load_logit<BF16>(logits, base + c * g.sc); m = fmaxf(m, z[c]); }
            float den = 0.f;
#pragma unroll
            for (int c = 0; c < KMAX; ++c)
                if (c < g.K) { z[c] = expf(z[c] - m); den += z[c]; }
            const float inv = 1.0f / den;
            float gp[KMAX];
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < KMAX; ++c)
                if (c < g.K) {
                    const float pr = z[c] * inv;
                    z[c] = pr;
                    float gg = 0.f;
                    if ((flags & 1) && c != g.ignore) {
                        const float y = (c == t) ? 1.0f : 0.0f;
                        gg = invK * (2.0f * pr * sN[c] - 2.0f * y * sD[c]) / (sD[c] * sD[c]);
                    }
                    gp[c] = gg;
                    dot += gg * pr;
                }
#pragma unroll
            for (int c = 0; c < KMAX; ++c)
                if (c < g.K) {
                    float d = z[c] * (gp[c] - dot);
                    if (flags & 2) d += (z[c] - ((c == t) ? 1.0f : 0.0f)) * inv_nvalid;
                    dz[c] = d * gscale;
                }
        }
#pragma unroll
        for (int c = 0; c < KMAX; ++c)
            if (c < g.K) {
                if (GBF16) ((uint16_t*)grad)[base + c * g.sc] = f32_to_bf16(dz[c]);
                else ((float*)grad)[base + c * g.sc] = dz[c];
            }
    }
}

// ------------------------------------------------------------------------------------------ K11
__global__ __launch_bounds__(THREADS) void confusion_kernel(const int64_t* __restrict__ pred,
                                                            const int64_t* __restrict__ label, int64_t n, int K,
                                                            int ignore, unsigned long long* __restrict__ conf) {
    extern __shared__ unsigned int hist[];     // K*K
    for (int i = threadIdx.x; i < K * K; i += THREADS) hist[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const int64_t l = label[i];
        if (l == ignore) continue;
        const int64_t x = pred[i] + (int64_t)K * l;          // metrics.py:19
        if (x >= 0 && x < (int64_t)K * K) atomicAdd(&hist[x], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * K; i += THREADS)
        if (hist[i]) atomicAdd(&conf[i], (unsigned long long)hist[i]);
}

}  // namespace

extern "C" {

int oess_masked_normalize_f32(const float* in, float* out, int64_t n, double* stats, oess_stream_t stream) {
    return run_normalize(in, out, n, 1, 0, 0, stats, (hipStream_t)stream);
}

int oess_masked_normalize_slice_f32(const float* in, float* out, int B, int Ctot, int c0, int Cs, int64_t HW,
                                    double* stats, oess_stream_t stream) {
    if (B <= 0 || Ctot <= 0 || Cs <= 0 || c0 < 0 || c0 + Cs > Ctot || HW <= 0) return OESS_EINVAL;
    return run_normalize(in, out, (int64_t)Cs * HW, B, (int64_t)Ctot * HW, (int64_t)c0 * HW, stats, (hipStream_t)stream);
}

int oess_masked_stats_slice_f32(const float* in, int B, int Ctot, int c0, int Cs, int64_t HW, double* stats,
                                oess_stream_t stream) {
    if (!in || !stats || B <= 0 || Ctot <= 0 || Cs <= 0 || c0 < 0 || c0 + Cs > Ctot || HW <= 0) return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    OESS_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(double), st));
    const int64_t L = (int64_t)Cs * HW, in_stride = (int64_t)Ctot * HW, in_off = (int64_t)c0 * HW;
    const int vec = ((L & 3) == 0) && ((in_stride & 3) == 0) && ((in_off & 3) == 0) && (((uintptr_t)in & 15) == 0);
    if (B > 65535) return OESS_EINVAL;
    hipLaunchKernelGGL(norm_stats_kernel, norm_grid(L, B, vec), dim3(THREADS), 0, st, in, L, (int64_t)B, in_stride, in_off, vec, stats);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_masked_stats_slices_f32(const float* in, int B, int Ctot, int Cs, int n_slices, int64_t HW, double* stats,
                                 oess_stream_t stream) {
    if (!in || !stats || B <= 0 || Ctot <= 0 || Cs <= 0 || n_slices <= 0 || (int64_t)n_slices * Cs > Ctot || HW <= 0 || B > 65535 ||
        n_slices > 65535)
        return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    OESS_HIP(hipMemsetAsync(stats, 0, (size_t)n_slices * 4 * sizeof(double), st));
    const int64_t L = (int64_t)Cs * HW, in_stride = (int64_t)Ctot * HW;
    const int vec = ((L & 3) == 0) && ((in_stride & 3) == 0) && (((uintptr_t)in & 15) == 0);
    dim3 grid = norm_grid(L, B, vec);
    {   // norm_grid caps a launch at 256 workgroups because all of them add into ONE triple of doubles; here every slice has its
        // own triple, so the cap applies per slice: up to 4x more workgroups per slice keep the 720 MB pass on the HBM roofline
        int64_t gx = (L / (vec ? 4 : 1) + THREADS * 8 - 1) / (THREADS * 8);
        int64_t cap = 1024 / B;
        if (cap < 1) cap = 1;
        if (gx > cap) gx = cap;
        if (gx > (int64_t)grid.x) grid.x = (unsigned)gx;
    }
    grid.z = (unsigned)n_slices;
    hipLaunchKernelGGL(norm_stats_kernel, grid, dim3(THREADS), 0, st, in, L, (int64_t)B, in_stride, (int64_t)0, vec, stats);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_segment_mean_fwd(const void* feat, int is_bf16, const int64_t* ids, int64_t P, int64_t pixels_per_sample,
                          int superpixel_size, int Cf, int S, float* k, float* count, oess_stream_t stream) {
    if (!feat || !ids || !k || !count || P <= 0 || pixels_per_sample <= 0 || Cf <= 0 || S <= 0) return OESS_EINVAL;
    if (P % pixels_per_sample != 0) return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    OESS_HIP(hipMemsetAsync(k, 0, (size_t)S * Cf * sizeof(float), st));
    OESS_HIP(hipMemsetAsync(count, 0, (size_t)S * sizeof(float), st));
    const int64_t B = P / pixels_per_sample;
    {   // vectorised kernel: Cf = LPP * (8 bf16 | 4 fp32 channels per lane), LPP in {8, 16, 32, 64}, 16-byte aligned rows
        const int cpl = is_bf16 ? 8 : 4;
        const int lpp = (Cf % cpl == 0) ? Cf / cpl : 0;
        if ((lpp == 8 || lpp == 16 || lpp == 32 || lpp == 64) && ((uintptr_t)feat & 15) == 0) {
            int vlocal = (int)((128 * 1024) / ((size_t)Cf * 4 + 4));      // ids held in the LDS table (<= 128 KB)
            if (vlocal > 256) vlocal = 256;
            const size_t lds = (size_t)vlocal * ((size_t)Cf * 4 + 4);
            // pixels per workgroup: aim at three full rounds of one workgroup per CU (measured at 8 x 440 x 640: 2048 px ->
            // 0.436 ms, 3072 px -> 0.382 / 0.399 ms bf16 / fp32 = 38 % / 73 % of 8 TB/s)
            int ppw = 0;
            {
                long long chunks = (3 * 256 + B / 2) / B;
                if (chunks < 1) chunks = 1;
                long long q = (pixels_per_sample + chunks - 1) / chunks;
                q = (q + 63) / 64 * 64;
                if (q < 512) q = 512;
                if (q > 16384) q = 16384;
                ppw = (int)q;
            }
            const int64_t vchunks = (pixels_per_sample + ppw - 1) / ppw;
            const dim3 vgrid((unsigned)(B * vchunks));
#define OESS_SEGV(BF, L)                                                                                                     \
            {                                                                                                                \
                (void)hipFuncSetAttribute((const void*)&segmean_fwd_vec_kernel<BF, L>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                hipLaunchKernelGGL((segmean_fwd_vec_kernel<BF, L>), vgrid, dim3(SEGV_THREADS), lds, st, feat, ids, P, pixels_per_sample, \
                                   superpixel_size, S, vlocal, k, count, ppw);                                              \
            }
            if (is_bf16) { if (lpp == 8) OESS_SEGV(true, 8) else if (lpp == 16) OESS_SEGV(true, 16) else if (lpp == 32) OESS_SEGV(true, 32) else OESS_SEGV(true, 64) }
            else { if (lpp == 8) OESS_SEGV(false, 8) else if (lpp == 16) OESS_SEGV(false, 16) else if (lpp == 32) OESS_SEGV(false, 32) else OESS_SEGV(false, 64) }
#undef OESS_SEGV
            hipLaunchKernelGGL(segmean_finalize_kernel, dim3(stream_grid((int64_t)S * Cf, THREADS)), dim3(THREADS), 0, st, k, count, S, Cf);
            OESS_HIP(hipGetLastError());
            return OESS_OK;
        }
    }
    const int64_t chunks = (pixels_per_sample + SEG_PIX_PER_WG - 1) / SEG_PIX_PER_WG;
    dim3 grid((unsigned)(B * chunks), (unsigned)((Cf + SEG_CH - 1) / SEG_CH));
    if (is_bf16)
        hipLaunchKernelGGL(segmean_fwd_kernel<true>, grid, dim3(THREADS), 0, st, feat, ids, P, pixels_per_sample,
                           superpixel_size, Cf, S, k, count);
    else
        hipLaunchKernelGGL(segmean_fwd_kernel<false>, grid, dim3(THREADS), 0, st, feat, ids, P, pixels_per_sample,
                           superpixel_size, Cf, S, k, count);
    hipLaunchKernelGGL(segmean_finalize_kernel, dim3(stream_grid((int64_t)S * Cf, THREADS)), dim3(THREADS), 0, st, k,
                       count, S, Cf);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_segment_mean_bwd(const float* grad_k, const float* count, const int64_t* ids, int64_t P,
                          int64_t pixels_per_sample, int superpixel_size, int Cf, int S, void* grad_feat, int is_bf16,
                          void* workspace, size_t workspace_bytes, oess_stream_t stream) {
    if (!grad_k || !count || !ids || !grad_feat || P <= 0 || pixels_per_sample <= 0 || Cf <= 0 || (Cf & 3) || S <= 0)
        return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if ((Cf >> 2) > THREADS || P % pixels_per_sample != 0) return OESS_EINVAL;
    const int64_t nb = P / pixels_per_sample;
    {   // quotient table + row gather when the rows are whole 16-byte lanes and the caller passed the table's scratch
        const size_t row_bytes = (size_t)Cf * (is_bf16 ? 2 : 4);
        const int lpp = (int)(row_bytes / 16);
        if (workspace && workspace_bytes >= (size_t)S * row_bytes && (row_bytes & 15) == 0 && lpp >= 1 && lpp <= THREADS &&
            ((uintptr_t)workspace & 15) == 0 && ((uintptr_t)grad_feat & 15) == 0) {
            const unsigned tg = stream_grid((int64_t)S * Cf, THREADS);
            if (is_bf16) hipLaunchKernelGGL(segmean_bwd_table_kernel<true>, dim3(tg), dim3(THREADS), 0, st, grad_k, count, S, Cf, workspace);
            else hipLaunchKernelGGL(segmean_bwd_table_kernel<false>, dim3(tg), dim3(THREADS), 0, st, grad_k, count, S, Cf, workspace);
            const int rows = THREADS / lpp;
            int64_t gx = (pixels_per_sample + (int64_t)rows * 8 - 1) / ((int64_t)rows * 8);
            const int64_t capx = (8192 + nb - 1) / nb;
            if (gx > capx) gx = capx;
            if (gx < 1) gx = 1;
            hipLaunchKernelGGL(segmean_bwd_gather_kernel, dim3((unsigned)gx, (unsigned)nb), dim3(THREADS), 0, st, (const uint4*)workspace, ids,
                               pixels_per_sample, superpixel_size, lpp, S, (uint4*)grad_feat);
            OESS_HIP(hipGetLastError());
            return OESS_OK;
        }
    }
    const int rows = THREADS / (Cf >> 2);
    int64_t gx = (pixels_per_sample + (int64_t)rows * 4 - 1) / ((int64_t)rows * 4);
    const int64_t capx = (16384 + nb - 1) / nb;
    if (gx > capx) gx = capx;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)nb);
    if (is_bf16)
        hipLaunchKernelGGL(segmean_bwd_kernel<true>, grid, dim3(THREADS), 0, st, grad_k, count, ids, P,
                           pixels_per_sample, superpixel_size, Cf, S, grad_feat);
    else
        hipLaunchKernelGGL(segmean_bwd_kernel<false>, grid, dim3(THREADS), 0, st, grad_k, count, ids, P,
                           pixels_per_sample, superpixel_size, Cf, S, grad_feat);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

size_t oess_task_loss_sums_doubles(int K) { return K > 0 ? (size_t)(3 * K + 2) * (1 + TASK_LOSS_MAX_ROWS) : 0; }

int oess_task_loss_fwd(const void* logits, int is_bf16, const int64_t* target, int64_t P, int64_t pixels_per_sample,
                       int64_t stride_b, int64_t stride_p, int64_t stride_c, int K, int ignore_index, int flags,
                       double* sums, float* loss_out, oess_stream_t stream) {
    if (!logits || !target || !sums || !loss_out || P <= 0 || pixels_per_sample <= 0 || K <= 0 || K > 32)
        return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    LossGeom g{P, pixels_per_sample, stride_b, stride_p, stride_c, K, ignore_index};
    // every workgroup leaves one row of 3K+2 double partial sums behind sums[0 .. 3K+2) (oess_task_loss_sums_doubles); the
    // finalize kernel adds the rows in order.  (Double atomics on the totals instead: serialised in L2 -- 2 048 workgroups spent
    // ~100 us there, a cap of 512 workgroups left the pixel loop latency-bound -- and order-dependent.)
    int grid = stream_grid(P, THREADS * 4);
    if (grid > TASK_LOSS_MAX_ROWS) grid = TASK_LOSS_MAX_ROWS;
    // dense NHWC fp32 ([P][K]): tiles through LDS
    const bool dense = !is_bf16 && stride_c == 1 && stride_p == K && stride_b == pixels_per_sample * K && K <= 16 &&
                       (((uintptr_t)logits) & 15) == 0;
    if (dense) {
        if (K <= 8) hipLaunchKernelGGL((task_loss_fwd_dense_kernel<8>), dim3(grid), dim3(THREADS), 0, st, (const float*)logits, target, g, sums);
        else hipLaunchKernelGGL((task_loss_fwd_dense_kernel<16>), dim3(grid), dim3(THREADS), 0, st, (const float*)logits, target, g, sums);
    } else {
#define LAUNCH_FWD(KM)                                                                                              \
    do {                                                                                                            \
        if (is_bf16)                                                                                                \
            hipLaunchKernelGGL((task_loss_fwd_kernel<KM, true>), dim3(grid), dim3(THREADS), 0, st, logits, target, g, sums); \
        else                                                                                                        \
            hipLaunchKernelGGL((task_loss_fwd_kernel<KM, false>), dim3(grid), dim3(THREADS), 0, st, logits, target, g, sums); \
    } while (0)
    if (K <= 8) LAUNCH_FWD(8); else if (K <= 16) LAUNCH_FWD(16); else LAUNCH_FWD(32);
#undef LAUNCH_FWD
    }
    hipLaunchKernelGGL(task_loss_finalize_kernel, dim3(1), dim3(1024), 0, st, sums, grid, K, ignore_index, flags, loss_out);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_task_loss_bwd(const void* logits, int is_bf16, const int64_t* target, int64_t P, int64_t pixels_per_sample,
                       int64_t stride_b, int64_t stride_p, int64_t stride_c, int K, int ignore_index, int flags,
                       const double* sums, float grad_scale, const float* grad_scale_dev, void* grad_logits,
                       int grad_is_bf16, oess_stream_t stream) {
    if (!logits || !target || !sums || !grad_logits || P <= 0 || pixels_per_sample <= 0 || K <= 0 || K > 32)
        return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    LossGeom g{P, pixels_per_sample, stride_b, stride_p, stride_c, K, ignore_index};
    const int grid = stream_grid(P, THREADS * 4);
    if (!is_bf16 && !grad_is_bf16 && stride_c == 1 && stride_p == K && stride_b == pixels_per_sample * K && K <= 16 &&
        (((uintptr_t)logits) & 15) == 0 && (((uintptr_t)grad_logits) & 15) == 0) {        // dense NHWC fp32: tiles through LDS
        if (K <= 8) hipLaunchKernelGGL((task_loss_bwd_dense_kernel<8>), dim3(grid), dim3(THREADS), 0, st, (const float*)logits, target, g, sums,
                                       flags, grad_scale, grad_scale_dev, (float*)grad_logits);
        else hipLaunchKernelGGL((task_loss_bwd_dense_kernel<16>), dim3(grid), dim3(THREADS), 0, st, (const float*)logits, target, g, sums,
                                flags, grad_scale, grad_scale_dev, (float*)grad_logits);
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
#define LAUNCH_BWD(KM, A, B)                                                                                   \
    hipLaunchKernelGGL((task_loss_bwd_kernel<KM, A, B>), dim3(grid), dim3(THREADS), 0, st, logits, target, g, sums, \
                       flags, grad_scale, grad_scale_dev, grad_logits)
#define DISPATCH_BWD(KM)                                                       \
    do {                                                                       \
        if (is_bf16 && grad_is_bf16) LAUNCH_BWD(KM, true, true);               \
        else if (is_bf16) LAUNCH_BWD(KM, true, false);                         \
        else if (grad_is_bf16) LAUNCH_BWD(KM, false, true);                    \
        else LAUNCH_BWD(KM, false, false);                                     \
    } while (0)
    if (K <= 8) DISPATCH_BWD(8); else if (K <= 16) DISPATCH_BWD(16); else DISPATCH_BWD(32);
#undef DISPATCH_BWD
#undef LAUNCH_BWD
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_confusion_accumulate(const int64_t* pred, const int64_t* label, int64_t n, int K, int ignore_label,
                              int64_t* conf, oess_stream_t stream) {
    if (!pred || !label || !conf || n < 0 || K <= 0 || K > 256) return OESS_EINVAL;
    if (n == 0) return OESS_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = stream_grid(n, THREADS * 8);
    hipLaunchKernelGGL(confusion_kernel, dim3(grid), dim3(THREADS), (size_t)K * K * sizeof(unsigned int), st, pred,
                       label, n, K, ignore_label, (unsigned long long*)conf);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // extern "C"
