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

// block-wide sum of NV doubles per thread -> dst[0..NV) (plain stores: one row of partial sums per workgroup; fixed order:
// xor-shuffle tree inside a wave, then the waves in index order)
template <int NV>
__device__ __forceinline__ void block_partial_store(double (&v)[NV], double* dst) {
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
        dst[threadIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------ K2
// Tensor viewed as `nchunk` contiguous chunks of L floats; chunk j starts at in + in_off + j*in_stride,
// out is dense.  (Dense tensor: nchunk = 1.)
// grid = (blocks per chunk, nchunk): no per-element index division (a 64-bit divide per float4 made this ALU-bound).
// DETERMINISTIC: every workgroup leaves one row {sum, sum of squares, non-zero count} of double partial sums; a finalize kernel
// adds the rows of a slice in a fixed order (double atomics on the totals, as before round 4, made the statistics -- and through
// them every latent of the recurrent encoder -- depend on the order in which workgroups retire).
// stats layout: totals [n_slices][4] followed by the partial rows [n_slices][K2_MAX_ROWS][4] (oess_masked_stats_doubles).
constexpr int K2_MAX_ROWS = 1024;
__global__ __launch_bounds__(THREADS) void norm_stats_kernel(const float* __restrict__ in, int64_t L, int64_t nchunk,
                                                             int64_t in_stride, int64_t in_off, int vec,
                                                             double* __restrict__ stats) {
    double acc[3] = {0.0, 0.0, 0.0};
    // blockIdx.z = slice of a multi-slice launch (oess_masked_stats_slices_f32): slice z starts z * L floats further and
    // accumulates into stats[4 z ..]
    const float* src = in + in_off + (int64_t)blockIdx.z * L + (int64_t)blockIdx.y * in_stride;
    double* row = stats + 4 * (int64_t)gridDim.z + 4 * ((int64_t)blockIdx.z * K2_MAX_ROWS + (int64_t)blockIdx.y * gridDim.x + blockIdx.x);
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
    block_partial_store<3>(acc, row);
}

// one workgroup per slice: rows t, t+256, ... added in sequence by thread t, then a fixed LDS tree
__global__ __launch_bounds__(THREADS) void norm_stats_finalize_kernel(double* __restrict__ stats, int rows) {
    __shared__ double red[3][THREADS];
    const double* part = stats + 4 * (int64_t)gridDim.x + 4 * (int64_t)blockIdx.x * K2_MAX_ROWS;
    double a[3] = {0.0, 0.0, 0.0};
    for (int r = threadIdx.x; r < rows; r += THREADS)
        for (int i = 0; i < 3; ++i) a[i] += part[4 * (int64_t)r + i];
    for (int i = 0; i < 3; ++i) red[i][threadIdx.x] = a[i];
    __syncthreads();
    for (int o = THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int i = 0; i < 3; ++i) red[i][threadIdx.x] += red[i][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 3) stats[4 * (int64_t)blockIdx.x + threadIdx.x] = red[threadIdx.x][0];
    if (threadIdx.x == 3) stats[4 * (int64_t)blockIdx.x + 3] = 0.0;
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

// blocks per chunk: >= 8 items per thread, at most K2_MAX_ROWS workgroups (= partial rows) per slice
__host__ dim3 norm_grid(int64_t L, int64_t nchunk, int vec) {
    int64_t gx = (L / (vec ? 4 : 1) + THREADS * 8 - 1) / (THREADS * 8);
    int64_t cap = K2_MAX_ROWS / (nchunk < 1 ? 1 : nchunk);
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
    if (nchunk > K2_MAX_ROWS) return OESS_EINVAL;
    const dim3 grid = norm_grid(L, nchunk, vec);
    hipLaunchKernelGGL(norm_stats_kernel, grid, dim3(THREADS), 0, st, in, L, nchunk, in_stride, in_off, vec, stats);
    hipLaunchKernelGGL(norm_stats_finalize_kernel, dim3(1), dim3(THREADS), 0, st, stats, (int)(grid.x * grid.y));
    hipLaunchKernelGGL(norm_apply_kernel, grid, dim3(THREADS), 0, st, in, out, L, nchunk, in_stride, in_off, vec,
                       stats);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

// ------------------------------------------------------------------------------------------ K7
// Superpixel scatter-mean, forward.  DETERMINISTIC: every partial sum that meets another one does so as a 64-bit fixed-point
// integer (2^-32 units), so neither the order in which lane groups reach the LDS table nor the order in which workgroups
// reach the global accumulators can change a bit of the result (integer addition is associative; float addition is not --
// the fp32 atomics this replaces made the contrastive step repeat only to ~1e-6).
//   * feature rows are pixel-major [P][Cf]; a workgroup owns a run of consecutive pixels of ONE sample and a 64-channel slice;
//   * a lane group (64 channels / 16-byte lanes) walks its own contiguous sub-run with a fp32 register run-accumulator (fixed
//     order: the pixels of the sub-run in sequence) and converts the run sum to fixed point when the superpixel id changes:
//     ds_add_u64 into the LDS table [256 raw ids][64 channels];
//   * touched rows leave with one pair of global integer atomics per (workgroup, id, channel): the value is split into its low
//     32 bits (added to an unsigned accumulator) and its high part (signed accumulator), i.e. a 96-bit global sum: no range
//     limit on a segment's total;
//   * the finalize kernel recombines (hi * 2^32 + lo) * 2^-32 in double, rounds once to fp32 and divides by (count + 1e-6) as
//     pretrain_trainer.py:462 does.
// Range contract: |feature| < 32768 (a workgroup chunk has at most 65536 pixels, so its per-id sum stays below 2^31 in
// fixed-point units of 2^-32; intermediate wrap-around is harmless in two's complement).  A non-finite or larger input sets
// the error word and the WHOLE output becomes NaN (loud), instead of a silently wrong mean.  Resolution: the mean of a segment
// is exact to 2^-33 absolute before the final fp32 rounding -- tighter than sequential fp32 accumulation for |mean| > 1e-3.
// Raw ids outside [0,256) (the reference reads uint8 PNGs, so they do not occur there) go straight to the global accumulators.
constexpr int SEGF_THREADS = 512;                 // 8 waves on the one workgroup a CU can hold (the table is 130 KB)
constexpr int SEGF_CH = 64;                       // channels per workgroup
constexpr int SEGF_IDS = 256;                     // raw ids held in LDS
constexpr int SEGF_ROW = SEGF_CH + 1;             // u64 per table row: one pad entry rotates the banks from row to row
constexpr int SEGF_MAX_PPW = 65536;
constexpr size_t SEGF_LDS = (size_t)SEGF_IDS * SEGF_ROW * 8 + (size_t)SEGF_IDS * 4;
typedef unsigned long long u64_t;

__device__ __forceinline__ long long seg_to_fixed(float v) { return __float2ll_rn(v * 4294967296.0f); }

__device__ __forceinline__ void seg_global_add(u64_t* __restrict__ acc_lo, u64_t* __restrict__ acc_hi, int64_t idx, long long v) {
    if (v == 0) return;
    atomicAdd(&acc_lo[idx], (u64_t)v & 0xffffffffull);
    const long long hi = v >> 32;                                        // arithmetic shift: v = hi * 2^32 + lo, lo in [0, 2^32)
    if (hi != 0) atomicAdd(&acc_hi[idx], (u64_t)hi);
}

struct SegGeom { int64_t P, pps; int sps, Cf, S, nslice, pix_per_wg; };

// workgroup -> (sample, pixel chunk, channel slice); the slices of one chunk are neighbours in blockIdx
__device__ __forceinline__ void seg_chunk(const SegGeom& g, int& slice, int64_t& b, int64_t& p_beg, int64_t& p_end) {
    slice = blockIdx.x % g.nslice;
    const int64_t lin = blockIdx.x / g.nslice;
    const int64_t chunks_per_sample = (g.pps + g.pix_per_wg - 1) / g.pix_per_wg;
    b = lin / chunks_per_sample;
    const int64_t ch = lin - b * chunks_per_sample;
    p_beg = b * g.pps + ch * g.pix_per_wg;
    p_end = p_beg + g.pix_per_wg;
    if (p_end > (b + 1) * g.pps) p_end = (b + 1) * g.pps;
    if (p_end > g.P) p_end = g.P;
}

// table rows with a non-zero pixel count -> global accumulators.  `pos(ch)` = where channel ch of the slice sits in a row.
template <typename PosFn>
__device__ __forceinline__ void seg_flush_table(const SegGeom& g, const u64_t* tab, const int* cnt, int slice, int64_t id_off,
                                                u64_t* __restrict__ acc_lo, u64_t* __restrict__ acc_hi, int* __restrict__ gcnt, PosFn pos) {
    const int ch = threadIdx.x & 63;
    const int c = slice * SEGF_CH + ch;
    const int ps = pos(ch);
    for (int i = threadIdx.x >> 6; i < SEGF_IDS; i += SEGF_THREADS / 64) {
        const int n = cnt[i];
        if (n == 0) continue;
        const int64_t gid = i + id_off;
        if (gid < 0 || gid >= g.S) continue;
        if (c < g.Cf) seg_global_add(acc_lo, acc_hi, gid * g.Cf + c, (long long)tab[i * SEGF_ROW + ps]);
        if (ch == 0 && slice == 0) atomicAdd(&gcnt[gid], n);
    }
}

// Vectorised form: CPL = 8 bf16 / 4 fp32 channels = one 16-byte load per lane, LPP = 64 / CPL lanes per pixel row, a wave covers
// 64 / LPP whole pixel rows per load instruction and keeps U of them in flight (software-pipelined).
template <bool BF16>
__global__ __launch_bounds__(SEGF_THREADS) void segmean_fwd_fx_kernel(const void* __restrict__ feat, const int64_t* __restrict__ ids,
                                                                 SegGeom g, u64_t* __restrict__ acc_lo, u64_t* __restrict__ acc_hi,
                                                                 int* __restrict__ gcnt, int* __restrict__ err) {
    constexpr int CPL = BF16 ? 8 : 4;
    constexpr int LPP = SEGF_CH / CPL;
    constexpr int GROUPS = SEGF_THREADS / LPP;
    extern __shared__ __attribute__((aligned(16))) unsigned char seg_smem[];
    u64_t* tab = reinterpret_cast<u64_t*>(seg_smem);                               // [SEGF_IDS][SEGF_ROW]; channel sub*CPL+c at c*LPP+sub
    int* cnt = reinterpret_cast<int*>(tab + (size_t)SEGF_IDS * SEGF_ROW);          // [SEGF_IDS]
    const int sub = threadIdx.x % LPP, grp = threadIdx.x / LPP;
    int slice;
    int64_t b, p_beg, p_end;
    seg_chunk(g, slice, b, p_beg, p_end);
    const int Cf = g.Cf;
    const int c0 = slice * SEGF_CH + sub * CPL;                                    // first channel of this lane
    const bool c_ok = c0 < Cf;                                                      // Cf % CPL == 0: a lane is all in or all out
    for (int i = threadIdx.x; i < SEGF_IDS * SEGF_ROW; i += SEGF_THREADS) tab[i] = 0ull;
    for (int i = threadIdx.x; i < SEGF_IDS; i += SEGF_THREADS) cnt[i] = 0;
    __syncthreads();
    const int64_t id_off = b * (int64_t)g.sps;
    const int64_t len = p_end - p_beg;
    const int64_t q = (len + GROUPS - 1) / GROUPS;
    int64_t g_beg = p_beg + grp * q, g_end = g_beg + q;
    if (g_end > p_end) g_end = p_end;
    int64_t cur = -1;
    float run[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) run[c] = 0.f;
    int run_n = 0;
    uint32_t amax = 0;                                                              // max |bit pattern| seen (range / finiteness check)
    auto flush = [&]() {
        if (run_n == 0) return;
        if (cur >= 0 && cur < SEGF_IDS) {
            if (c_ok) {
                u64_t* row = tab + cur * SEGF_ROW + sub;
#pragma unroll
                for (int c = 0; c < CPL; ++c) atomicAdd(&row[c * LPP], (u64_t)seg_to_fixed(run[c]));
            }
            if (sub == 0) atomicAdd(&cnt[cur], run_n);
        } else {
            const int64_t gid = cur + id_off;
            if (gid >= 0 && gid < g.S) {
                if (c_ok) {
#pragma unroll
                    for (int c = 0; c < CPL; ++c) seg_global_add(acc_lo, acc_hi, gid * Cf + c0 + c, seg_to_fixed(run[c]));
                }
                if (sub == 0 && slice == 0) atomicAdd(&gcnt[gid], run_n);
            }
        }
    };
    constexpr int U = 8;
    using RawRow = typename std::conditional<BF16, uint4, float4>::type;
    RawRow cur_r[U], nxt_r[U];
    int64_t cur_id[U], nxt_id[U];
    const int c0_ld = c_ok ? c0 : 0;                                                // masked lanes load a valid address and drop the value
    auto issue = [&](int64_t p, RawRow (&r)[U], int64_t (&id)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t pu = p + u;
            const bool ok = pu < g_end;
            id[u] = ok ? ids[pu] : (int64_t)-1;
            const int64_t pc = ok ? pu : (g_end > g_beg ? g_end - 1 : p_beg);      // clamped: always a valid row of this chunk
            if constexpr (BF16) r[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(feat) + pc * Cf + c0_ld);
            else r[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(feat) + pc * Cf + c0_ld);
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
                    const uint32_t w[4] = {cur_r[u].x, cur_r[u].y, cur_r[u].z, cur_r[u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[2 * j] = __uint_as_float(w[j] << 16);
                        v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
                        const uint32_t lo = (w[j] << 16) & 0x7fffffffu, hi = w[j] & 0x7fff0000u;
                        amax = max(amax, max(lo, hi));
                    }
                } else {
                    v[0] = cur_r[u].x; v[1] = cur_r[u].y; v[2] = cur_r[u].z; v[3] = cur_r[u].w;
#pragma unroll
                    for (int c = 0; c < 4; ++c) amax = max(amax, __float_as_uint(v[c]) & 0x7fffffffu);
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
    if (c_ok && amax >= 0x47000000u) atomicOr(err, 1);                             // |x| >= 32768, inf or NaN
    __syncthreads();
    seg_flush_table(g, tab, cnt, slice, id_off, acc_lo, acc_hi, gcnt, [](int ch) { return (ch % CPL) * LPP + ch / CPL; });
}

// Generic form (any Cf, any alignment): lane = channel of the 64-channel slice, the 8 waves take contiguous eighths of the chunk.
template <bool BF16>
__global__ __launch_bounds__(SEGF_THREADS) void segmean_fwd_fx_scalar_kernel(const void* __restrict__ feat, const int64_t* __restrict__ ids,
                                                                        SegGeom g, u64_t* __restrict__ acc_lo, u64_t* __restrict__ acc_hi,
                                                                        int* __restrict__ gcnt, int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char seg_smem[];
    u64_t* tab = reinterpret_cast<u64_t*>(seg_smem);
    int* cnt = reinterpret_cast<int*>(tab + (size_t)SEGF_IDS * SEGF_ROW);
    int slice;
    int64_t b, p_beg, p_end;
    seg_chunk(g, slice, b, p_beg, p_end);
    const int Cf = g.Cf;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = slice * SEGF_CH + lane;
    const bool c_ok = c < Cf;
    for (int i = threadIdx.x; i < SEGF_IDS * SEGF_ROW; i += SEGF_THREADS) tab[i] = 0ull;
    for (int i = threadIdx.x; i < SEGF_IDS; i += SEGF_THREADS) cnt[i] = 0;
    __syncthreads();
    const int64_t id_off = b * (int64_t)g.sps;
    constexpr int WAVES = SEGF_THREADS / 64;
    const int64_t len = p_end - p_beg;
    const int64_t q = (len + WAVES - 1) / WAVES;
    int64_t w_beg = p_beg + wave * q, w_end = w_beg + q;
    if (w_end > p_end) w_end = p_end;
    int64_t cur = -1;                                    // current raw id of the run (wave-uniform)
    float run = 0.0f;
    int run_n = 0;
    uint32_t amax = 0;
    auto flush = [&]() {
        if (run_n == 0) return;
        if (cur >= 0 && cur < SEGF_IDS) {
            if (c_ok) atomicAdd(&tab[cur * SEGF_ROW + lane], (u64_t)seg_to_fixed(run));
            if (lane == 0) atomicAdd(&cnt[cur], run_n);
        } else {
            const int64_t gid = cur + id_off;
            if (gid >= 0 && gid < g.S) {
                if (c_ok) seg_global_add(acc_lo, acc_hi, gid * Cf + c, seg_to_fixed(run));
                if (lane == 0 && slice == 0) atomicAdd(&gcnt[gid], run_n);
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
            v[u] = 0.0f;
            if (ok && c_ok) v[u] = BF16 ? bf16_to_f32(((const uint16_t*)feat)[pu * Cf + c]) : ((const float*)feat)[pu * Cf + c];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + u < w_end) {
                if (id[u] != cur) { flush(); cur = id[u]; run = 0.0f; run_n = 0; }
                amax = max(amax, __float_as_uint(v[u]) & 0x7fffffffu);
                run += v[u];
                run_n += 1;
            }
        }
    }
    flush();
    if (amax >= 0x47000000u) atomicOr(err, 1);
    __syncthreads();
    seg_flush_table(g, tab, cnt, slice, id_off, acc_lo, acc_hi, gcnt, [](int ch) { return ch; });
}

// k = fp32( (hi * 2^32 + lo) * 2^-32 ) / (count + 1e-6)   (pretrain_trainer.py:462); count as the fp32 row sum the reference forms
__global__ __launch_bounds__(THREADS) void segmean_fx_finalize_kernel(const u64_t* __restrict__ acc_lo, const u64_t* __restrict__ acc_hi,
                                                                      const int* __restrict__ gcnt, const int* __restrict__ err,
                                                                      float* __restrict__ k, float* __restrict__ count, int S, int Cf) {
    const int64_t n = (int64_t)S * Cf;
    const bool bad = *err != 0;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const int64_t s = i / Cf;
        const double tot = (double)(long long)acc_hi[i] * 4294967296.0 + (double)acc_lo[i];
        const float sum = (float)(tot * (1.0 / 4294967296.0));
        const float cn = (float)gcnt[s];
        k[i] = bad ? __uint_as_float(0x7fc00000u) : sum / __fadd_rn(cn, 1e-6f);
        if (i - s * Cf == 0) count[s] = cn;
    }
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

size_t oess_masked_stats_doubles(int n_slices) { return n_slices > 0 ? (size_t)n_slices * 4 * (1 + K2_MAX_ROWS) : 0; }

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
    const int64_t L = (int64_t)Cs * HW, in_stride = (int64_t)Ctot * HW, in_off = (int64_t)c0 * HW;
    const int vec = ((L & 3) == 0) && ((in_stride & 3) == 0) && ((in_off & 3) == 0) && (((uintptr_t)in & 15) == 0);
    if (B > K2_MAX_ROWS) return OESS_EINVAL;
    const dim3 grid = norm_grid(L, B, vec);
    hipLaunchKernelGGL(norm_stats_kernel, grid, dim3(THREADS), 0, st, in, L, (int64_t)B, in_stride, in_off, vec, stats);
    hipLaunchKernelGGL(norm_stats_finalize_kernel, dim3(1), dim3(THREADS), 0, st, stats, (int)(grid.x * grid.y));
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_masked_stats_slices_f32(const float* in, int B, int Ctot, int Cs, int n_slices, int64_t HW, double* stats,
                                 oess_stream_t stream) {
    if (!in || !stats || B <= 0 || Ctot <= 0 || Cs <= 0 || n_slices <= 0 || (int64_t)n_slices * Cs > Ctot || HW <= 0 || B > K2_MAX_ROWS ||
        n_slices > 65535)
        return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t L = (int64_t)Cs * HW, in_stride = (int64_t)Ctot * HW;
    const int vec = ((L & 3) == 0) && ((in_stride & 3) == 0) && (((uintptr_t)in & 15) == 0);
    dim3 grid = norm_grid(L, B, vec);
    grid.z = (unsigned)n_slices;
    hipLaunchKernelGGL(norm_stats_kernel, grid, dim3(THREADS), 0, st, in, L, (int64_t)B, in_stride, (int64_t)0, vec, stats);
    hipLaunchKernelGGL(norm_stats_finalize_kernel, dim3((unsigned)n_slices), dim3(THREADS), 0, st, stats, (int)(grid.x * grid.y));
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

size_t oess_segment_mean_fwd_workspace_bytes(int S, int Cf) {
    if (S <= 0 || Cf <= 0) return 0;
    return align_up((size_t)S * Cf * 16 + (size_t)S * 4 + 4, 16);
}

int oess_segment_mean_fwd(const void* feat, int is_bf16, const int64_t* ids, int64_t P, int64_t pixels_per_sample,
                          int superpixel_size, int Cf, int S, float* k, float* count, void* workspace, size_t workspace_bytes,
                          oess_stream_t stream) {
    if (!feat || !ids || !k || !count || !workspace || P <= 0 || pixels_per_sample <= 0 || Cf <= 0 || S <= 0) return OESS_EINVAL;
    if (P % pixels_per_sample != 0 || ((uintptr_t)workspace & 15) != 0) return OESS_EINVAL;
    const size_t need = oess_segment_mean_fwd_workspace_bytes(S, Cf);
    if (workspace_bytes < need) return OESS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    OESS_HIP(hipMemsetAsync(workspace, 0, need, st));
    u64_t* acc_lo = (u64_t*)workspace;
    u64_t* acc_hi = acc_lo + (size_t)S * Cf;
    int* gcnt = (int*)(acc_hi + (size_t)S * Cf);
    int* err = gcnt + S;
    const int64_t B = P / pixels_per_sample;
    SegGeom g;
    g.P = P; g.pps = pixels_per_sample; g.sps = superpixel_size; g.Cf = Cf; g.S = S;
    g.nslice = (Cf + SEGF_CH - 1) / SEGF_CH;
    {   // pixels per workgroup: aim at three full rounds of one workgroup per CU over all (chunk, slice) pairs
        long long chunks = (3LL * num_cus() / g.nslice + B / 2) / B;
        if (chunks < 1) chunks = 1;
        long long q = (pixels_per_sample + chunks - 1) / chunks;
        q = (q + 63) / 64 * 64;
        if (q < 512) q = 512;
        if (q > SEGF_MAX_PPW) q = SEGF_MAX_PPW;
        g.pix_per_wg = (int)q;
    }
    const int64_t vchunks = (pixels_per_sample + g.pix_per_wg - 1) / g.pix_per_wg;
    const int64_t nwg = B * vchunks * g.nslice;
    if (nwg > 0x7fffffffLL) return OESS_EINVAL;
    const dim3 grid((unsigned)nwg);
    const int cpl = is_bf16 ? 8 : 4;
    const bool vec = (Cf % cpl == 0) && ((uintptr_t)feat & 15) == 0;
#define OESS_SEGF(KERNEL)                                                                                                   \
    {                                                                                                                       \
        (void)hipFuncSetAttribute((const void*)&KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEGF_LDS);       \
        hipLaunchKernelGGL(KERNEL, grid, dim3(SEGF_THREADS), SEGF_LDS, st, feat, ids, g, acc_lo, acc_hi, gcnt, err);       \
    }
    if (vec) { if (is_bf16) OESS_SEGF(segmean_fwd_fx_kernel<true>) else OESS_SEGF(segmean_fwd_fx_kernel<false>) }
    else { if (is_bf16) OESS_SEGF(segmean_fwd_fx_scalar_kernel<true>) else OESS_SEGF(segmean_fwd_fx_scalar_kernel<false>) }
#undef OESS_SEGF
    hipLaunchKernelGGL(segmean_fx_finalize_kernel, dim3(stream_grid((int64_t)S * Cf, THREADS)), dim3(THREADS), 0, st, acc_lo, acc_hi,
                       gcnt, err, k, count, S, Cf);
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
