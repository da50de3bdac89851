// Kernels of the MaskCLIP ViT-B/16 image tower that are not convolutions / GEMMs (models/maskclip_model.py:448-541,
// 545-851 in the reference; the GEMMs run on conv_fwd.hip as 1x1 convolutions over the token axis):
//   * LayerNorm over the channel axis of a [rows x C] bf16 token matrix (eps 1e-6 in the ViT)
//   * multi-head self attention softmax(Q K^T / sqrt(d)) V for head dimension 64 from the fused in_proj output
//     [B, L, 3C] (nn.MultiheadAttention's packed q | k | v layout), fp32 online softmax.
// The attention is an MFMA flash-attention kernel (a first VALU version, 4 lanes per query, took 16-18 ms per tower forward
// against 1.25 ms and has been removed).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "oess.h"
#include "oess_common.h"

namespace {
using namespace oess;
constexpr int THREADS = 256;

// one wave per row; lane handles channels lane, lane + 64, ...  (C <= 64 * 32)
__global__ __launch_bounds__(THREADS) void layernorm_kernel(const uint16_t* __restrict__ x, int64_t xs, int64_t rows, int C,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, uint16_t* __restrict__ y, int64_t ys) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int MAXE = 32;
    for (int64_t r = (int64_t)blockIdx.x * (THREADS / 64) + wave; r < rows; r += (int64_t)gridDim.x * (THREADS / 64)) {
        float v[MAXE];                                       // fully unrolled + uniform guards: stays in registers
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
            v[i] = 0.f;
            if (i * 64 < C) { const int c = i * 64 + lane; if (c < C) { v[i] = bf16_to_f32(x[r * xs + c]); s += v[i]; } }
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXE; ++i)
            if (i * 64 < C) { const int c = i * 64 + lane; if (c < C) { const float d = v[i] - mean; q += d * d; } }
        const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);         // biased variance, like nn.LayerNorm
#pragma unroll
        for (int i = 0; i < MAXE; ++i)
            if (i * 64 < C) {
                const int c = i * 64 + lane;
                if (c < C) y[r * ys + c] = f32_to_bf16((v[i] - mean) * rstd * gamma[c] + beta[c]);
            }
    }
}

// Same, 16-byte accesses: lane owns 8 consecutive channels of chunk lane, lane + 64, ...  (C % 8 == 0, 16-byte aligned rows).
// The 2-byte form above moved 1.0 TB/s on the ViT-B/16 token matrix (27 us per call, 26 calls per tower forward).
__global__ __launch_bounds__(THREADS) void layernorm_vec_kernel(const uint16_t* __restrict__ x, int64_t xs, int64_t rows, int C,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float eps, uint16_t* __restrict__ y, int64_t ys) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int MAXC = 4;                                  // chunks of 8 channels per lane: C <= 2048
    const int nchunk = C >> 3;
    union P8 { uint4 q; uint16_t h[8]; };
    for (int64_t r = (int64_t)blockIdx.x * (THREADS / 64) + wave; r < rows; r += (int64_t)gridDim.x * (THREADS / 64)) {
        P8 v[MAXC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = i * 64 + lane;
            v[i].q = make_uint4(0u, 0u, 0u, 0u);
            if (ch < nchunk) {
                v[i].q = *reinterpret_cast<const uint4*>(x + r * xs + ch * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) s += bf16_to_f32(v[i].h[k]);
            }
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
            if (i * 64 + lane < nchunk) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = bf16_to_f32(v[i].h[k]) - mean; q += d * d; }
            }
        const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);         // biased variance, like nn.LayerNorm
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = i * 64 + lane;
            if (ch < nchunk) {
                const float4 g0 = *reinterpret_cast<const float4*>(gamma + ch * 8), g1 = *reinterpret_cast<const float4*>(gamma + ch * 8 + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(beta + ch * 8), b1 = *reinterpret_cast<const float4*>(beta + ch * 8 + 4);
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                float o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (bf16_to_f32(v[i].h[k]) - mean) * rstd * gg[k] + bb[k];
                *reinterpret_cast<uint4*>(y + r * ys + ch * 8) = pack_bf16x8(o);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA flash attention for head dimension 64 (v_mfma_f32_32x32x16_bf16).
// A wave owns 32 queries; per tile of 32 keys it computes the TRANSPOSED score tile S^T = K Q^T (4 MFMAs), so a lane
// (query column q = lane & 31, half hi = lane >> 5) holds 16 scores of ITS query in registers:
// keys K(e) = (e & 3) + 8 (e >> 2) + 4 hi.  The online-softmax max / sum are then register reductions plus one
// cross-half shuffle.  P^T needs no data movement to become the B operand of O^T = V^T P^T (4 MFMAs): MFMA k-slot
// (kk, hi, j) is simply DEFINED as key (j & 3) + 8 (2 kk + (j >> 2)) + 4 hi - the lane's own registers 8 kk .. 8 kk + 7 -
// and the V tile is stored in LDS transposed with its key axis permuted the same way.
// LDS pitches (72 / 40 elements) make both fragment reads conflict-free ds_read_b128.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) short abf16x8_t;
typedef __attribute__((ext_vector_type(16))) float af32x16_t;
constexpr int AQ = 128;                  // queries per workgroup (4 waves x 32)
constexpr int KP = 72, VP = 40;          // LDS pitches in elements

__device__ __forceinline__ int vt_pos(int key) {          // LDS column of a key inside the permuted V^T tile
    const int hi = (key >> 2) & 1, low = key & 3, g = key >> 3;
    return (g >> 1) * 16 + hi * 8 + low + 4 * (g & 1);
}

__global__ __launch_bounds__(THREADS) void attention_d64_mfma_kernel(const uint16_t* __restrict__ qkv, int64_t qs, int B, int L,
                                                                     int heads, float scale, uint16_t* __restrict__ out, int64_t os) {
    __shared__ __attribute__((aligned(16))) uint16_t lk[32 * KP];
    __shared__ __attribute__((aligned(16))) uint16_t lvt[64 * VP];
    const int C = heads * 64;
    const int qblocks = (L + AQ - 1) / AQ;
    int bid = blockIdx.x;
    const int qb = bid % qblocks; bid /= qblocks;
    const int h = bid % heads;
    const int b = bid / heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qcol = lane & 31, hi = lane >> 5;
    const int qi = qb * AQ + wave * 32 + qcol;
    const uint16_t* base = qkv + (int64_t)b * L * qs + h * 64;
    // Q fragments (B operand of S^T): Q[q][16 kk + 8 hi .. + 8]
    abf16x8_t qf[4];
    {
        const uint16_t* qp = base + (int64_t)(qi < L ? qi : L - 1) * qs;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const abf16x8_t*>(qp + 16 * kk + 8 * hi);
    }
    af32x16_t o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    float m = -INFINITY, l = 0.f;
    // staging lanes: thread -> (key row, 8-channel chunk)
    const int srow = threadIdx.x >> 3, sch = threadIdx.x & 7;
    uint4 kreg = make_uint4(0u, 0u, 0u, 0u), vreg = kreg;
    auto gload = [&](int k0) {
        const int kj = k0 + srow;
        kreg = make_uint4(0u, 0u, 0u, 0u); vreg = kreg;
        if (kj < L) {
            const uint16_t* rp = base + (int64_t)kj * qs + sch * 8;
            kreg = *reinterpret_cast<const uint4*>(rp + C);
            vreg = *reinterpret_cast<const uint4*>(rp + 2 * C);
        }
    };
    gload(0);
    for (int k0 = 0; k0 < L; k0 += 32) {
        __syncthreads();                                    // previous tile fully consumed
        *reinterpret_cast<uint4*>(&lk[srow * KP + sch * 8]) = kreg;
        {
            union { uint4 q; uint16_t hh[8]; } u;
            u.q = vreg;
            const int pc = vt_pos(srow);
#pragma unroll
            for (int i = 0; i < 8; ++i) lvt[(sch * 8 + i) * VP + pc] = u.hh[i];
        }
        __syncthreads();
        if (k0 + 32 < L) gload(k0 + 32);                    // next tile's global loads fly under this tile's MFMAs
        // S^T = K Q^T
        af32x16_t st;
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const abf16x8_t kf = *reinterpret_cast<const abf16x8_t*>(&lk[qcol * KP + 16 * kk + 8 * hi]);   // row = key (lane & 31)
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st, 0, 0, 0);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            if (key >= L) st[e] = -INFINITY;
            tmax = fmaxf(tmax, st[e]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax);
        const float resc = __expf((m - m_new) * scale);
        float p[16];
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { p[e] = __expf((st[e] - m_new) * scale); ps += p[e]; }
        l = l * resc + ps;                                  // per-half partial sum; halves are combined at the end
        m = m_new;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[e] *= resc; o1[e] *= resc; }
        // O^T += V^T P^T
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            union { abf16x8_t v; uint32_t w[4]; } pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) pf.w[j] = pack_bf16x2(p[8 * kk + 2 * j], p[8 * kk + 2 * j + 1]);
            const abf16x8_t v0 = *reinterpret_cast<const abf16x8_t*>(&lvt[qcol * VP + 16 * kk + 8 * hi]);          // d = lane & 31
            const abf16x8_t v1 = *reinterpret_cast<const abf16x8_t*>(&lvt[(32 + qcol) * VP + 16 * kk + 8 * hi]);   // d = 32 + (lane & 31)
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf.v, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf.v, o1, 0, 0, 0);
        }
    }
    l += __shfl_xor(l, 32, 64);
    if (qi < L) {
        const float inv = 1.0f / l;
        uint16_t* op = out + ((int64_t)b * L + qi) * os + h * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 a, c;
            a.x = pack_bf16x2(o0[4 * g] * inv, o0[4 * g + 1] * inv);
            a.y = pack_bf16x2(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            c.x = pack_bf16x2(o1[4 * g] * inv, o1[4 * g + 1] * inv);
            c.y = pack_bf16x2(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(op + 8 * g + 4 * hi) = a;            // d = 8 g + 4 hi + (0..3)
            *reinterpret_cast<uint2*>(op + 32 + 8 * g + 4 * hi) = c;       // d = 32 + ...
        }
    }
}
}  // namespace

extern "C" {

int oess_layernorm_bf16(const void* x, long long x_row_stride, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                        void* y, long long y_row_stride, oess_stream_t stream) {
    if (!x || !y || !gamma || !beta || rows <= 0 || C <= 0 || C > 2048 || x_row_stride < C || y_row_stride < C || eps <= 0.f)
        return OESS_EINVAL;
    int64_t g = (rows + 3) / 4;
    if (g > 65536) g = 65536;
    const bool vec = (C & 7) == 0 && (x_row_stride & 7) == 0 && (y_row_stride & 7) == 0 &&
                     ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(layernorm_vec_kernel, dim3((unsigned)g), dim3(THREADS), 0, (hipStream_t)stream, (const uint16_t*)x,
                           (int64_t)x_row_stride, rows, C, gamma, beta, eps, (uint16_t*)y, (int64_t)y_row_stride);
    else
        hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)g), dim3(THREADS), 0, (hipStream_t)stream, (const uint16_t*)x,
                           (int64_t)x_row_stride, rows, C, gamma, beta, eps, (uint16_t*)y, (int64_t)y_row_stride);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_attention_d64_bf16(const void* qkv, long long qkv_row_stride, int B, int L, int heads, float scale, void* out,
                            long long out_row_stride, oess_stream_t stream) {
    if (!qkv || !out || B <= 0 || L <= 0 || heads <= 0 || qkv_row_stride < 3ll * heads * 64 || out_row_stride < heads * 64 ||
        (qkv_row_stride & 7) || (out_row_stride & 7) || ((uintptr_t)qkv & 15) || ((uintptr_t)out & 15))
        return OESS_EINVAL;
    {
        const long long blocks = (long long)B * heads * ((L + AQ - 1) / AQ);
        if (blocks > 0x7fffffffll) return OESS_EINVAL;
        hipLaunchKernelGGL(attention_d64_mfma_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, (hipStream_t)stream,
                           (const uint16_t*)qkv, (int64_t)qkv_row_stride, B, L, heads, scale, (uint16_t*)out, (int64_t)out_row_stride);
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
}

}  // extern "C"
