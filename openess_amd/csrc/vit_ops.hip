// Kernels of the MaskCLIP ViT-B/16 image tower that are not convolutions / GEMMs (models/maskclip_model.py:448-541,
// 545-851 in the reference; the GEMMs run on conv_fwd.hip as 1x1 convolutions over the token axis):
//   * LayerNorm over the channel axis of a [rows x C] bf16 token matrix (eps 1e-6 in the ViT)
//   * multi-head self attention softmax(Q K^T / sqrt(d)) V for head dimension 64 from the fused in_proj output
//     [B, L, 3C] (nn.MultiheadAttention's packed q | k | v layout), fp32 online softmax.
// The attention is an MFMA flash-attention kernel (a first VALU version, 4 lanes per query, took 16-18 ms per tower forward
// and has been removed; the round-2 MFMA kernel with 32-key tiles, two barriers per tile and transposed V stores took 1.26 ms).
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
// A wave owns 32 queries; per tile of 64 keys it computes the TRANSPOSED score tiles S^T = K Q^T (2 x 4 MFMAs), so a lane
// (query column q = lane & 31, half hi = lane >> 5) holds 2 x 16 scores of ITS query in registers; the online-softmax max /
// sum are register reductions plus one cross-half shuffle.  P^T then has to be the B operand of O^T = V^T P^T (2 x 4 MFMAs),
// whose k-slot j of lane (q, hi) is key 16 ks + 8 hi + j.  Instead of shuffling the scores into that order, the K tile is
// stored in LDS with its rows PERMUTED (key bits 2 and 3 swapped inside each 32-key block): score register e = 8 kk + j of
// lane (q, hi) is then key 32 sb + 16 kk + 8 hi + j, exactly the lane's own k-slots -- P^T needs no data movement.
// The V tile stays row-major [key][d]; its transposed A fragments come from ds_read_b64_tr_b16 (semantics in conv_wgrad.hip),
// so nothing is transposed on the store side (round 2 stored V^T with eight 2-byte LDS writes per thread and key tile).
// Two LDS stages: the next tile's K / V rows travel HBM -> registers under the MFMAs and are parked in the other stage, ONE
// barrier per 64 keys (round 2: two per 32 keys).  LDS pitches 144 B (K, ds_read_b128) and 192 B (V, transposing reads).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) short abf16x8_t;
typedef __attribute__((ext_vector_type(4))) short abf16x4_t;
typedef __attribute__((ext_vector_type(16))) float af32x16_t;
constexpr int AQ = 128;                  // queries per workgroup (4 waves x 32)
constexpr int AK = 64;                   // keys per tile
constexpr int KPB = 144, VPB = 192;      // LDS row pitches in bytes
constexpr int KST = AK * KPB, VST = AK * VPB;

__device__ __forceinline__ abf16x4_t att_tr_read(uint32_t lds_addr) {
    abf16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    return v;
}

__global__ __launch_bounds__(THREADS, 3) void attention_d64_mfma_kernel(const uint16_t* __restrict__ qkv, int64_t qs, int B, int L,
                                                                     int heads, float scale, uint16_t* __restrict__ out, int64_t os) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * KST + 2 * VST];          // [K stage 0][K stage 1][V stage 0][V stage 1]
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
    // staging: thread -> rows (tid >> 3) and (tid >> 3) + 32 of the tile, 16-byte chunk tid & 7
    const int srow = threadIdx.x >> 3, sch = threadIdx.x & 7;
    // LDS row of key r of a 32-key block: bits 2 and 3 of r swapped (see header)
    const int prow = (srow & ~12) | ((srow & 4) << 1) | ((srow & 8) >> 1);
    uint4 kreg[2], vreg[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kj = k0 + srow + 32 * i;
            kreg[i] = make_uint4(0u, 0u, 0u, 0u); vreg[i] = kreg[i];
            if (kj < L) {
                const uint16_t* rp = base + (int64_t)kj * qs + sch * 8;
                kreg[i] = *reinterpret_cast<const uint4*>(rp + C);
                vreg[i] = *reinterpret_cast<const uint4*>(rp + 2 * C);
            }
        }
    };
    auto park = [&](int stage) {
        unsigned char* lk = lds + stage * KST;
        unsigned char* lv = lds + 2 * KST + stage * VST;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<uint4*>(lk + (prow + 32 * i) * KPB + sch * 16) = kreg[i];
            *reinterpret_cast<uint4*>(lv + (srow + 32 * i) * VPB + sch * 16) = vreg[i];
        }
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    // transposing-read lane geometry (conv_wgrad.hip): 16-lane group g4: columns 16 (g4 & 1) + 4 (li & 3), rows 8 (g4 >> 1) + (li >> 2)
    const int g4 = lane >> 4, li = lane & 15;
    const uint32_t tr_off = (uint32_t)(((g4 >> 1) * 8 + (li >> 2)) * VPB + ((g4 & 1) * 16 + (li & 3) * 4) * 2);
    const int ntiles = (L + AK - 1) / AK;
    gload(0);
    park(0);
    __syncthreads();
    if (ntiles > 1) gload(AK);
    for (int it = 0; it < ntiles; ++it) {
        const int k0 = it * AK, stage = it & 1;
        const unsigned char* lk = lds + stage * KST;
        const uint32_t lv = lds0 + 2 * KST + stage * VST;
        // S^T = K Q^T for the two 32-key blocks of the tile
        af32x16_t st[2];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) st[sb][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const abf16x8_t kf = *reinterpret_cast<const abf16x8_t*>(lk + (32 * sb + qcol) * KPB + (16 * kk + 8 * hi) * 2);
                st[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st[sb], 0, 0, 0);
            }
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = k0 + 32 * sb + 16 * (e >> 3) + 8 * hi + (e & 7);       // permuted K rows: register e is this key
                if (key >= L) st[sb][e] = -INFINITY;
                tmax = fmaxf(tmax, st[sb][e]);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax);
        const float resc = __expf((m - m_new) * scale);
        float ps = 0.f;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int e = 0; e < 16; ++e) { st[sb][e] = __expf((st[sb][e] - m_new) * scale); ps += st[sb][e]; }
        l = l * resc + ps;                                  // per-half partial sum; halves are combined at the end
        m = m_new;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[e] *= resc; o1[e] *= resc; }
        // O^T += V^T P^T: k-step ks covers keys 16 ks .. 16 ks + 15 of the tile = registers 8 (ks & 1) .. + 7 of block ks >> 1
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            union { abf16x8_t v; uint32_t w[4]; } pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) pf.w[j] = pack_bf16x2(st[ks >> 1][8 * (ks & 1) + 2 * j], st[ks >> 1][8 * (ks & 1) + 2 * j + 1]);
            const uint32_t vb = lv + (uint32_t)(16 * ks) * VPB + tr_off;
            abf16x4_t a0l = att_tr_read(vb), a0h = att_tr_read(vb + 4 * VPB);                 // d = lane & 31
            abf16x4_t a1l = att_tr_read(vb + 64), a1h = att_tr_read(vb + 64 + 4 * VPB);       // d = 32 + (lane & 31)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0l), "+v"(a0h), "+v"(a1l), "+v"(a1h) :: "memory");
            const abf16x8_t v0 = __builtin_shufflevector(a0l, a0h, 0, 1, 2, 3, 4, 5, 6, 7);
            const abf16x8_t v1 = __builtin_shufflevector(a1l, a1h, 0, 1, 2, 3, 4, 5, 6, 7);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf.v, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf.v, o1, 0, 0, 0);
        }
        if (it + 1 < ntiles) park(stage ^ 1);              // the other stage was last read in iteration it - 1: every wave is past it
        __syncthreads();
        if (it + 2 < ntiles) gload(k0 + 2 * AK);           // flies under the next tile's MFMAs
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
