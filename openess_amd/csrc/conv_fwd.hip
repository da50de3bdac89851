// Implicit-GEMM 2-D convolution for gfx950 (bf16 in, fp32 MFMA accumulate).
//
//   out[m, n] = act( sum_k A[m, k] * Wp[n, k] + bias[n] ),   m = (b, oy, ox),  k = (r, s, ci)
//   A[m, k]  = in[b, oy*stride - pad + r*dil, ox*stride - pad + s*dil, ci]   (0 outside the image)
//
// Layout: activations are NHWC bf16 with an explicit pixel stride (so a conv can read or write a
// channel slice of a wider concat buffer: skip connections and ConvLSTM cat(x, h) need no copy).
// Weights are pre-packed once to Wp[Npad][Kpad] bf16, K ordered (r, s, ci), zero padded to the
// tile sizes, so the B operand is a plain row-major panel.
//
// Kernel: 128 x BN output tile per 256-thread workgroup (4 waves), BK = 64.
//   - A and B K-slabs are gathered with 16-byte loads (8 channels of one filter tap) into registers
//     and written to an XOR-swizzled LDS image (16-byte chunk c of row r lives at c ^ ((r>>1)&7):
//     conflict-free ds_read_b128 for the 32x32x16 fragment pattern);
//   - next slab's global loads are issued before the MFMA block of the current slab (register
//     staging: the zero fill of padded taps needs per-lane predication, which LDS-DMA cannot do);
//   - v_mfma_f32_32x32x16_bf16, each wave owns a 64 x 64 (BN=128), 32 x 64 (BN=64) or 32 x 32
//     (BN=32) accumulator block;
//   - epilogue: bias + optional ReLU in fp32, convert to bf16, transpose through LDS and store
//     whole NHWC rows with 16-byte stores (or fp32 direct stores for the small logits heads).
//   - workgroup ids are remapped so that the n-tiles of one m-tile land on the same XCD (shared L2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "oess.h"
#include "oess_common.h"

namespace {
using namespace oess;

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;     // 8 bf16 = 4 VGPRs (MFMA A/B operand)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;    // 32x32 accumulator fragment
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t; // 16-byte staging register (native vector: stays in VGPRs)

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int CONV_THREADS = 256;

struct ConvArgs {
    const uint16_t* in;      // NHWC bf16
    const uint16_t* w;       // packed [Npad][Kpad]
    const float* bias;       // [Cout] or null
    uint16_t* out;           // NHWC bf16 (or null when out_f32 is set)
    float* out_f32;          // NHWC fp32 alternative output
    const uint16_t* residual;  // optional NHWC bf16 tensor added before the activation (same pixel stride as out)
    long long in_pix_stride, out_pix_stride, res_pix_stride;
    int B, H, W, Cin;        // input geometry; Cin % 8 == 0
    int Ho, Wo, Cout;
    int R, S, stride, pad, dil;
    int Kpad;                // multiple of BK
    int M;                   // B*Ho*Wo
    int relu;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int BN>
__global__ __launch_bounds__(CONV_THREADS) void conv_fwd_kernel(ConvArgs a) {
    // wave layout: BN=128 -> 2x2 waves of 64x64; BN=64 -> 4x1 waves of 32x64; BN=32 -> 4x1 waves of 32x32
    constexpr int WAVES_N = (BN == 128) ? 2 : 1;
    constexpr int WAVES_M = 4 / WAVES_N;
    constexpr int WM = BM / WAVES_M;          // 64 or 32
    constexpr int WN = BN / WAVES_N;          // 64, 64 or 32
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int B_ROWS_PER_THREAD = BN / 32;   // 16-byte chunks of the B slab per thread

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: 2 x { A slab [BM][8] chunks, B slab [BN][8] chunks } (double buffered), then the tap table
    constexpr int STAGE_CHUNKS = (BM + BN) * 8;
    u32x4_t* lbase = reinterpret_cast<u32x4_t*>(smem);
    int2* ltab = reinterpret_cast<int2*>(smem + 2 * STAGE_CHUNKS * 16);      // [Kpad/8] {element offset, dy | dx<<16}

    // ---- XCD-aware tile mapping (bijective): consecutive logical tiles share an XCD's L2
    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int KT = a.Kpad / BK;

    // ---- tap table, built once per workgroup: chunk kc -> (r, s, channel chunk).  Keeps the two integer
    //      divisions out of the K loop (they were ~40 % of its VALU work).
    {
        const int cpt = a.Cin >> 3, ntaps = a.R * a.S;
        for (int kc = tid; kc < KT * 8; kc += CONV_THREADS) {
            const int tap = kc / cpt, cc = kc - tap * cpt;
            const int r = tap / a.S, s = tap - r * a.S;
            int2 e;
            if (tap < ntaps) {
                const int dy = r * a.dil, dx = s * a.dil;
                e.x = (dy * a.W + dx) * (int)a.in_pix_stride + cc * 8;
                e.y = (dy & 0xffff) | (dx << 16);
            } else {
                e.x = 0;
                e.y = 0x7fff | (0x7fff << 16);                  // far outside: fails every bounds test
            }
            ltab[kc] = e;
        }
    }

    // ---- per-thread gather state: chunk column c (fixed), rows (tid>>3) + 32*i
    const int c = tid & 7;
    const int row0 = tid >> 3;
    int iy0[4], ix0[4];
    const uint16_t* rowptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + row0 + 32 * i;
        const bool valid = m < a.M;
        const int mm = valid ? m : 0;
        const int hw = a.Ho * a.Wo;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        iy0[i] = valid ? oy * a.stride - a.pad : -0x4000;       // invalid rows fail the bounds test
        ix0[i] = ox * a.stride - a.pad;
        rowptr[i] = a.in + (((long long)b * a.H + (oy * a.stride - a.pad)) * a.W + ix0[i]) * a.in_pix_stride;
    }
    const uint16_t* wrow = a.w + (size_t)(n0 + row0) * a.Kpad + c * 8;

    u32x4_t ra[4], rb[B_ROWS_PER_THREAD];
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};
    // global -> registers for K-slab KT_IDX.  Loads are unconditional (clamped to the tensor base) and
    // zeroed by select afterwards, so the four gathers issue back to back without exec-mask branches.
#define OESS_GLOAD(KT_IDX)                                                                                          \
    {                                                                                                               \
        const int2 e_ = ltab[(KT_IDX) * 8 + c];                                                                     \
        const int dy_ = (int)(short)(e_.y & 0xffff), dx_ = e_.y >> 16;                                              \
        bool ok_[4];                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                             \
            ok_[i] = (unsigned)(iy0[i] + dy_) < (unsigned)a.H && (unsigned)(ix0[i] + dx_) < (unsigned)a.W;          \
            const uint16_t* src_ = ok_[i] ? rowptr[i] + e_.x : a.in;                                                \
            ra[i] = *reinterpret_cast<const u32x4_t*>(src_);                                                        \
        }                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < B_ROWS_PER_THREAD; ++i)                                               \
            rb[i] = *reinterpret_cast<const u32x4_t*>(wrow + (size_t)(32 * i) * a.Kpad + (size_t)(KT_IDX) * BK);    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) ra[i] = ok_[i] ? ra[i] : zero4;                               \
    }
#define OESS_LSTORE(BUF)                                                                               \
    {                                                                                                  \
        u32x4_t* lA_ = lbase + (BUF) * STAGE_CHUNKS;                                                   \
        u32x4_t* lB_ = lA_ + BM * 8;                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
            const int r_ = row0 + 32 * i;                                                              \
            lA_[r_ * 8 + swz(r_, c)] = ra[i];                                                          \
        }                                                                                              \
        _Pragma("unroll") for (int i = 0; i < B_ROWS_PER_THREAD; ++i) {                                \
            const int r_ = row0 + 32 * i;                                                              \
            lB_[r_ * 8 + swz(r_, c)] = rb[i];                                                          \
        }                                                                                              \
    }

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    __syncthreads();                                   // tap table visible
    OESS_GLOAD(0)
    OESS_LSTORE(0)
    __syncthreads();
    if (KT > 1) OESS_GLOAD(1)
    // double-buffered LDS: ONE barrier per K-slab.  compute(buf) -> store next slab into the other buffer ->
    // barrier -> issue the global loads two slabs ahead (they fly under the next compute).
    for (int kt = 0; kt < KT; ++kt) {
        const u32x4_t* lA = lbase + (kt & 1) * STAGE_CHUNKS;
        const u32x4_t* lB = lA + BM * 8;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8_t fa[MT], fb[NT];
            const int chunk = ks * 2 + (lane >> 5);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int r = wm * WM + i * 32 + (lane & 31);
                fa[i] = *reinterpret_cast<const bf16x8_t*>(&lA[r * 8 + swz(r, chunk)]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int r = wn * WN + j * 32 + (lane & 31);
                fb[j] = *reinterpret_cast<const bf16x8_t*>(&lB[r * 8 + swz(r, chunk)]);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) {
            OESS_LSTORE((kt + 1) & 1)
            __syncthreads();
            if (kt + 2 < KT) OESS_GLOAD(kt + 2)
        }
    }
    __syncthreads();                                   // all LDS reads done before the epilogue reuses smem

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane & 31, row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5)
    const int ncol_l = lane & 31;
    if (a.out_f32) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * WN + j * 32 + ncol_l;
                const float bv = (a.bias && n < a.Cout) ? a.bias[n] : 0.0f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (m < a.M && n < a.Cout) {
                        float v = acc[i][j][e] + bv;
                        if (a.relu) v = fmaxf(v, 0.0f);
                        a.out_f32[(long long)m * a.out_pix_stride + n] = v;
                    }
                }
            }
        return;
    }
    // bf16 path: stage the tile as [BM][BN] bf16 in LDS (row pitch BN*2 + 16 bytes against bank conflicts)
    uint16_t* lC = reinterpret_cast<uint16_t*>(smem);
    constexpr int PITCH = BN + 8;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int nl = wn * WN + j * 32 + ncol_l;
            const int n = n0 + nl;
            const float bv = (a.bias && n < a.Cout) ? a.bias[n] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ml = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                lC[ml * PITCH + nl] = f32_to_bf16(acc[i][j][e] + bv);      // activation applied after the residual
            }
        }
    __syncthreads();
    constexpr int CHUNKS_N = BN / 8;                       // 16-byte chunks per tile row
    for (int idx = tid; idx < BM * CHUNKS_N; idx += CONV_THREADS) {
        const int ml = idx / CHUNKS_N, cn = idx - ml * CHUNKS_N;
        const int m = m0 + ml, n = n0 + cn * 8;
        if (m >= a.M || n >= a.Cout) continue;
        uint4 v = *reinterpret_cast<const uint4*>(&lC[ml * PITCH + cn * 8]);
        uint16_t* dst = a.out + (long long)m * a.out_pix_stride + n;
        union { uint4 q4; uint16_t h[8]; } u, rs;
        u.q4 = v;
        rs.q4 = make_uint4(0u, 0u, 0u, 0u);
        const bool full = n + 8 <= a.Cout;
        if (a.residual) {
            if (full) rs.q4 = *reinterpret_cast<const uint4*>(a.residual + (long long)m * a.res_pix_stride + n);
            else {
#pragma unroll
                for (int q = 0; q < 8; ++q) if (n + q < a.Cout) rs.h[q] = a.residual[(long long)m * a.res_pix_stride + n + q];
            }
        }
        if (a.residual || a.relu) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float f = bf16_to_f32(u.h[q]);
                if (a.residual) f += bf16_to_f32(rs.h[q]);
                if (a.relu) f = fmaxf(f, 0.0f);
                u.h[q] = f32_to_bf16(f);
            }
        }
        if (full) {
            *reinterpret_cast<uint4*>(dst) = u.q4;
        } else {                                   // ragged channel tail (Cout % 8 != 0)
#pragma unroll
            for (int q = 0; q < 8; ++q) if (n + q < a.Cout) dst[q] = u.h[q];
        }
    }
}

// ---- weight packing: OIHW fp32 (PyTorch Conv2d.weight) -> Wp[Npad][Kpad] bf16, k = (r, s, ci)
// flip != 0 produces the data-gradient operator: Wp[ci][(R-1-r, S-1-s), co] (rotated, in/out swapped).
__global__ void pack_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp, int Cout, int Cin, int R,
                                   int S, int Cin_pad, int Kpad, int Npad, int flip) {
    const long long total = (long long)Npad * Kpad;
    const int Nlog = flip ? Cin : Cout;      // logical output channels of the packed operator
    const int Klog_c = flip ? Cout : Cin;    // logical input channels
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / Kpad), k = (int)(i - (long long)n * Kpad);
        const int tap = k / Cin_pad, ci = k - tap * Cin_pad;
        float v = 0.0f;
        if (n < Nlog && tap < R * S && ci < Klog_c) {
            const int r = tap / S, s = tap - r * S;
            if (!flip) v = w[(((long long)n * Cin + ci) * R + r) * S + s];
            else v = w[(((long long)ci * Cin + n) * R + (R - 1 - r)) * S + (S - 1 - s)];
        }
        wp[i] = f32_to_bf16(v);
    }
}

}  // namespace

extern "C" {

int oess_conv2d_pack_weight(const float* w_oihw, int Cout, int Cin, int R, int S, int flip_for_dgrad, void* packed,
                            size_t packed_bytes, oess_stream_t stream) {
    if (!w_oihw || !packed || Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0) return OESS_EINVAL;
    const int n_log = flip_for_dgrad ? Cin : Cout, c_log = flip_for_dgrad ? Cout : Cin;
    const int cin_pad = (c_log + 7) / 8 * 8;
    const int kpad = (R * S * cin_pad + BK - 1) / BK * BK;
    const int npad = (n_log + 127) / 128 * 128;
    if (packed_bytes < (size_t)npad * kpad * 2) return OESS_ENOMEM;
    const long long total = (long long)npad * kpad;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, (uint16_t*)packed, Cout,
                       Cin, R, S, cin_pad, kpad, npad, flip_for_dgrad);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

size_t oess_conv2d_packed_bytes(int Cout, int Cin, int R, int S, int flip_for_dgrad) {
    if (Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0) return 0;
    const int n_log = flip_for_dgrad ? Cin : Cout, c_log = flip_for_dgrad ? Cout : Cin;
    const int cin_pad = (c_log + 7) / 8 * 8;
    const int kpad = (R * S * cin_pad + BK - 1) / BK * BK;
    const int npad = (n_log + 127) / 128 * 128;
    return (size_t)npad * kpad * 2;
}

int oess_conv2d_fwd_bf16(const void* in, long long in_pix_stride, int B, int H, int W, int Cin, const void* w_packed,
                         const float* bias, int Cout, int R, int S, int stride, int pad, int dil, int relu,
                         const void* residual, long long res_pix_stride, void* out_bf16, float* out_f32,
                         long long out_pix_stride, oess_stream_t stream) {
    if (!in || !w_packed || (!out_bf16 && !out_f32) || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 7) || Cout <= 0 ||
        R <= 0 || S <= 0 || stride <= 0 || pad < 0 || dil <= 0)
        return OESS_EINVAL;
    if ((in_pix_stride & 7) || in_pix_stride < Cin || out_pix_stride < Cout) return OESS_EINVAL;
    if (out_bf16 && !out_f32 && (out_pix_stride & 7) && (Cout & 7) == 0) return OESS_EINVAL;
    if (residual && ((res_pix_stride & 7) || out_f32)) return OESS_EINVAL;
    ConvArgs a;
    a.in = (const uint16_t*)in; a.w = (const uint16_t*)w_packed; a.bias = bias;
    a.out = out_f32 ? nullptr : (uint16_t*)out_bf16; a.out_f32 = out_f32;
    a.residual = (const uint16_t*)residual;
    a.in_pix_stride = in_pix_stride; a.out_pix_stride = out_pix_stride; a.res_pix_stride = res_pix_stride;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.dil = dil;
    a.Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
    a.Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
    if (a.Ho <= 0 || a.Wo <= 0) return OESS_EINVAL;
    a.Kpad = (R * S * Cin + BK - 1) / BK * BK;
    const long long M = (long long)B * a.Ho * a.Wo;
    if (M > 0x7fffffffll) return OESS_EINVAL;
    a.M = (int)M;
    a.relu = relu;
    a.tiles_m = (a.M + BM - 1) / BM;
    hipStream_t st = (hipStream_t)stream;
    {   // > 64 KiB of dynamic LDS needs an explicit opt-in (once per process)
        static bool attr_done = false;
        if (!attr_done) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fwd_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_done = true;
        }
    }
    // the packed weight has Npad = multiple of 128 rows, so any BN <= 128 tiles it safely
    if (Cout > 64) {
        a.tiles_n = (Cout + 127) / 128;
        const size_t lds = (size_t)2 * (BM + 128) * 8 * 16 + (size_t)(a.Kpad / 8) * 8;   // 64 KiB staging + tap table
        hipLaunchKernelGGL(conv_fwd_kernel<128>, dim3(a.tiles_m * a.tiles_n), dim3(CONV_THREADS), lds, st, a);
    } else if (Cout > 32) {
        a.tiles_n = 1;
        const size_t lds = (size_t)2 * (BM + 64) * 8 * 16 + (size_t)(a.Kpad / 8) * 8;
        hipLaunchKernelGGL(conv_fwd_kernel<64>, dim3(a.tiles_m * a.tiles_n), dim3(CONV_THREADS), lds, st, a);
    } else {
        a.tiles_n = 1;
        const size_t lds = (size_t)2 * (BM + 32) * 8 * 16 + (size_t)(a.Kpad / 8) * 8;
        hipLaunchKernelGGL(conv_fwd_kernel<32>, dim3(a.tiles_m * a.tiles_n), dim3(CONV_THREADS), lds, st, a);
    }
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // extern "C"
