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
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <map>
#include <mutex>
#include <vector>
#include "oess.h"
#include "oess_common.h"

namespace {
#include <type_traits>
#include <utility>
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
    float* stats;              // optional [tiles_m][2][Cout] per-tile column sums / sums of squares of the fp32 result
    long long in_pix_stride, out_pix_stride, res_pix_stride;
    int B, H, W, Cin;        // input geometry; Cin % 8 == 0
    int Ho, Wo, Cout;
    int R, S, stride, pad, dil;
    int Kpad;                // multiple of BK
    int M;                   // B*Ho*Wo
    int relu;
    int tiles_m, tiles_n;
    // fused ConvLSTM epilogue (EPI == 1): Cout = 4*lstm_C gate-interleaved rows (n' = 4*hc + gate)
    const float* lstm_prev;    // [M][C] fp32 previous cell state or null (= zero state)
    float* lstm_cell;          // [M][C] fp32 new cell state (may alias lstm_prev: a tile only touches its own block)
    uint16_t* lstm_h;          // hidden output, bf16, pixel stride lstm_h_stride (must NOT alias the conv input)
    long long lstm_h_stride;
    int lstm_C;
    unsigned inv_cpt, inv_s;   // exact small-range reciprocals: kc / cpt == (kc * inv_cpt) >> 20, tap / S == (tap * inv_s) >> 16
    // split-K (small-M, long-K layers: DeepLab's ASPP at output stride 16): grid.y = ksplit, workgroup (tile, z) reduces
    // K-slabs [z * kt_per, (z + 1) * kt_per) and writes its fp32 accumulators to partial[z][M][Cout]; splitk_reduce_kernel
    // adds the slices in a fixed order and does what the epilogue would have done (bias / residual / activation / tile stats)
    float* partial;
    int ksplit, kt_per;
    // row-halo kernel: exact reciprocals of W and W + dil for the small per-lane quotients of its prologue (x < 256:
    // x / d == umulhi(x, floor(2^32 / d) + 1) whenever x * d < 2^32); 0 = divide
    unsigned mg_w, mg_wd;
};

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// 16-byte store of a finished output row piece.  OESS_OUT_STORE: 0 = plain (write-back in the XCD's L2), 1 = non-temporal,
// 2 = agent scope (write-through).  The end of a kernel writes the XCD L2s' dirty lines back before the next kernel of the
// stream may start (8 non-coherent L2s): the fewer dirty lines a kernel leaves, the shorter the gap behind it.
#define OESS_OUT_STORE 0
__device__ __forceinline__ void out_store16(void* p, uint4 v) {
#if OESS_OUT_STORE == 1
    __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_t*>(p));
#elif OESS_OUT_STORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(u32x4_t{v.x, v.y, v.z, v.w}) : "memory");
#else
    *reinterpret_cast<uint4*>(p) = v;
#endif
}

// threads per workgroup of the LDS-DMA kernel by tile height: 64- and 128-row tiles 4 waves, 256-row tiles 8 waves
constexpr int conv_tile_threads(int bmx) { return bmx == 256 ? 512 : 256; }

// epilogue activation: 0 none, 1 ReLU, 2 GELU (exact erf form = nn.GELU(), the ViT FFN of models/maskclip_model.py)
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. 2^-15 of the bf16 result's ulp): one exp, one rcp, five FMAs instead
// of the ~50-instruction library erff.  The GELU epilogue of the ViT's fc1 (256 x 256 tiles, one workgroup per CU, nothing to
// hide an epilogue behind) spent a third of its workgroup lifetime in erff: 128 calls per thread.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float conv_act(float v, int mode) {
    if (mode == 1) return fmaxf(v, 0.0f);
    if (mode == 2) return 0.5f * v * (1.0f + fast_erf(v * 0.70710678118654752f));
    return v;
}

// PITCH: row pitch (elements) of the bf16 LDS image (BN + 8: padded, conflict-free 16-byte row reads).
template <int BMX, int BN, int PITCH = BN + 8, int NTHREADS = conv_tile_threads(BMX), int WAVES_N = (BN == 128) ? 2 : 1>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a,
                                              f32x16_t (&acc)[BMX / ((NTHREADS / 64) / WAVES_N) / 32][(BN / WAVES_N) / 32],
                                              unsigned char* smem, int m0, int n0, int wm, int wn, int lane, int tid,
                                              float* red_override = nullptr) {
    constexpr int WAVES_M = (NTHREADS / 64) / WAVES_N;
    constexpr int WM = BMX / WAVES_M;
    constexpr int WN = BN / WAVES_N;
    constexpr int MT = WM / 32, NT = WN / 32;
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
                        v = conv_act(v, a.relu);
                        a.out_f32[(long long)m * a.out_pix_stride + n] = v;
                    }
                }
            }
        return;
    }
    // bf16 path: stage the tile as [BM][BN] bf16 in LDS (row pitch BN*2 + 16 bytes against bank conflicts)
    uint16_t* lC = reinterpret_cast<uint16_t*>(smem);
    float* red = red_override ? red_override : reinterpret_cast<float*>(smem + BMX * PITCH * 2);   // [WAVES_M][BN][2] (BatchNorm partials)
    // bf16 image + (optionally) per-column sum / sum of squares over this tile's rows of the values AS STORED (rounded to bf16): the
    // statistics then describe exactly the tensor that BatchNorm normalises afterwards (sum of xhat == 0 over the stored values),
    // which the backward needs -- with statistics of the un-rounded accumulators the residual mean of the rounding errors times
    // d(beta) leaks into d(gamma), a second noise term as large as the rounding noise itself on common-mode gradients (measured:
    // BatchNorm weight-gradient cosine 0.74 -> 0.51 on the DeepLab test).  Rows >= M are exact zeros (their A rows were zero
    // filled; stats are only requested for bias-free convs).  One rounding per value serves both the image and the sums.
    auto stage = [&](auto with_stats) __attribute__((always_inline)) {
        constexpr bool WS = decltype(with_stats)::value;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int nl = wn * WN + j * 32 + ncol_l;
            const int n = n0 + nl;
            const float bv = (a.bias && n < a.Cout) ? a.bias[n] : 0.0f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ml = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    const uint32_t pk = pack_bf16x2(acc[i][j][e] + bv, 0.0f);      // activation applied after the residual
                    lC[ml * PITCH + nl] = (uint16_t)pk;
                    if constexpr (WS) {
                        const float v = __uint_as_float(pk << 16);
                        s1 += v; s2 += v * v;
                    }
                }
            if constexpr (WS) {
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lane < 32) {
                    const int col = wn * WN + j * 32 + lane;
                    red[(wm * BN + col) * 2 + 0] = s1;
                    red[(wm * BN + col) * 2 + 1] = s2;
                }
            }
        }
    };
    // Residual rows of this thread's output pieces are requested all at once between the staging writes and the barrier (the
    // accumulators are dead there, so the ITERS 16-byte pieces -- 16 on the 256 x 256 tile, 8 / 4 on the 128- / 64-row tiles --
    // reuse their registers): one L2 / HBM latency under the barrier instead of one per store-loop iteration (the 512 -> 2048
    // layer at M = 140 800 ran 657 us with a residual against 417 us without one: a workgroup that owns its CU has no other
    // wave to hide the loads).  Requested before the staging they would cost the 128-row kernels their third wave per SIMD.
    constexpr int CHUNKS_N = BN / 8;                       // 16-byte chunks per tile row
    constexpr int ITERS = (BMX * CHUNKS_N + NTHREADS - 1) / NTHREADS;
    if (a.stats) stage(std::true_type{});
    else stage(std::false_type{});
    uint4 rsv[ITERS];
    if (a.residual) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = tid + it * NTHREADS;
            const int ml = idx / CHUNKS_N, cn = idx - ml * CHUNKS_N;
            const int m = m0 + ml, n = n0 + cn * 8;
            rsv[it] = make_uint4(0u, 0u, 0u, 0u);
            if (idx < BMX * CHUNKS_N && m < a.M && n + 8 <= a.Cout)
                rsv[it] = *reinterpret_cast<const uint4*>(a.residual + (long long)m * a.res_pix_stride + n);
        }
    }
    __syncthreads();
    if (a.stats && tid < BN) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) { s1 += red[(w * BN + tid) * 2]; s2 += red[(w * BN + tid) * 2 + 1]; }
        const int n = n0 + tid;
        if (n < a.Cout) {
            // tile_stats rows are per 128 output rows; a 256-row tile fills row 2t and zeroes row 2t+1
            const int trow = (m0 / BMX) * (BMX / 128);
            a.stats[((size_t)trow * 2 + 0) * a.Cout + n] = s1;
            a.stats[((size_t)trow * 2 + 1) * a.Cout + n] = s2;
            if (BMX == 256 && m0 + 128 < a.M) {
                a.stats[((size_t)(trow + 1) * 2 + 0) * a.Cout + n] = 0.f;
                a.stats[((size_t)(trow + 1) * 2 + 1) * a.Cout + n] = 0.f;
            }
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * NTHREADS;
        if (idx >= BMX * CHUNKS_N) break;
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
            if (full) rs.q4 = rsv[it];
            else {
#pragma unroll
                for (int q = 0; q < 8; ++q) if (n + q < a.Cout) rs.h[q] = a.residual[(long long)m * a.res_pix_stride + n + q];
            }
        }
        if (a.residual || a.relu) {
            float f[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f[q] = bf16_to_f32(u.h[q]);
                if (a.residual) f[q] += bf16_to_f32(rs.h[q]);
                f[q] = conv_act(f[q], a.relu);
            }
            u.q4 = pack_bf16x8(f);
        }
        if (full) {
            out_store16(dst, u.q4);
        } else {                                   // ragged channel tail (Cout % 8 != 0)
#pragma unroll
            for (int q = 0; q < 8; ++q) if (n + q < a.Cout) dst[q] = u.h[q];
        }
    }
}


// ---- fused ConvLSTM cell update (e2vid/model/submodules.py:205-212) straight from the accumulators.
// The gate convolution is run TRANSPOSED (D^T = W * A^T: the packed weight is the MFMA A operand), so a lane
// holds, for ONE pixel (col = lane & 31), rows (e&3) + 8*(e>>2) + 4*(lane>>5) of the 32-row n block; with the
// weight rows packed gate-interleaved (n' = 4*hc + gate) the four registers e = 4*q .. 4*q+3 are exactly the
// (in, remember, out, cell) pre-activations of hidden channel 2*q + (lane>>5): the LSTM algebra is lane local,
// the 4C-channel gate tensor never exists in memory, and c / h leave through padded LDS images as full rows.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// Previous cell state of a 128-row x 32-hidden-channel tile, fetched COALESCED (whole 128-byte rows, float4 per lane)
// at kernel start so the loads retire under the K loop; the epilogue redistributes it through LDS.  (Read in place by
// the lanes that own the gates it would be 16 loads per lane touching 32 different lines each, issued after the K loop.)
struct LstmPrefetch { float4 v[4]; };
__device__ __forceinline__ void lstm_prefetch(const ConvArgs& a, LstmPrefetch& p, int m0, int n0, int tid) {
    const int hc0 = n0 >> 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = tid + 256 * k, row = idx >> 3, c4 = idx & 7;
        const int m = m0 + row;
        p.v[k] = (a.lstm_prev && m < a.M) ? *reinterpret_cast<const float4*>(a.lstm_prev + (long long)m * a.lstm_C + hc0 + c4 * 4)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// accumulator start values = the gate biases (EPI == 1 kernels that pass ADD_BIAS = false to the epilogue): the 64 bias
// loads per lane leave the epilogue and hide under the first operand fetch
template <int MT, int NT>
__device__ __forceinline__ void lstm_bias_init(const ConvArgs& a, f32x16_t (&acc)[MT][NT], int n0, int wn, int lane) {
    const int hi = lane >> 5, hc0 = n0 >> 2, C = a.lstm_C;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int hc = hc0 + wn * 8 * NT + j * 8 + 2 * q + hi;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float b = a.bias ? a.bias[g * C + hc] : 0.0f;
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j][q * 4 + g] = b;
            }
        }
}

template <int MT = 2, int NT = 2, bool PREF = false, bool ADD_BIAS = true>
__device__ __forceinline__ void lstm_epilogue(const ConvArgs& a, f32x16_t (&acc)[MT][NT], unsigned char* smem, int m0, int n0,
                                              int wm, int wn, int lane, int tid, const LstmPrefetch* pref = nullptr) {
    // 2 x 2 waves; workgroup tile = 64*MT rows x 64*NT gate columns = 16*NT hidden channels
    constexpr int ROWS = 64 * MT, HC = 16 * NT;
    constexpr int CP = HC + 1, HP = HC + 2;                // LDS pitches: fp32 cell image, bf16 hidden image
    static_assert(!PREF || (MT == 2 && NT == 2), "prefetch layout is the 128 x 128 tile's");
    float* lc = reinterpret_cast<float*>(smem);            // [ROWS][CP]
    uint16_t* lh = reinterpret_cast<uint16_t*>(smem + ROWS * CP * 4);    // [ROWS][HP]
    const int C = a.lstm_C;
    const int hc0 = n0 >> 2;                               // first hidden channel of this tile
    const int p = lane & 31, hi = lane >> 5;
    if constexpr (PREF) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + 256 * k, row = idx >> 3, c4 = idx & 7;
            float* d = lc + row * CP + c4 * 4;
            d[0] = pref->v[k].x; d[1] = pref->v[k].y; d[2] = pref->v[k].z; d[3] = pref->v[k].w;
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int ml = wm * 32 * MT + i * 32 + p;
        const int m = m0 + ml;
        const bool valid = m < a.M;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int hcl = wn * 8 * NT + j * 8 + 2 * q + hi;
                const int hc = hc0 + hcl;
                float gi = acc[i][j][q * 4 + 0], gr = acc[i][j][q * 4 + 1], go = acc[i][j][q * 4 + 2], gc = acc[i][j][q * 4 + 3];
                if (ADD_BIAS && a.bias) { gi += a.bias[hc]; gr += a.bias[C + hc]; go += a.bias[2 * C + hc]; gc += a.bias[3 * C + hc]; }
                float pc;
                if constexpr (PREF) pc = lc[ml * CP + hcl];              // this lane is the slot's only reader and writer
                else pc = (a.lstm_prev && valid) ? a.lstm_prev[(long long)m * C + hc] : 0.0f;
                const float nc = fast_sigmoid(gr) * pc + fast_sigmoid(gi) * fast_tanh(gc);     // submodules.py:211
                const float hv = fast_sigmoid(go) * fast_tanh(nc);                              // submodules.py:212
                lc[ml * CP + hcl] = nc;
                lh[ml * HP + hcl] = (uint16_t)pack_bf16x2(hv, 0.0f);
            }
    }
    __syncthreads();
    // 16-byte stores when rows are 16-byte aligned (always for the E2VID state buffers); LDS pitches are odd -> dword reads
    const bool vec_ok = (C & 3) == 0 && (a.lstm_h_stride & 7) == 0 && (((uintptr_t)a.lstm_h | (uintptr_t)a.lstm_cell) & 15) == 0;
    if (vec_ok) {
#pragma unroll
        for (int idx = tid; idx < ROWS * (HC / 4); idx += 256) {       // cell: float4 per lane, HC/4 lanes per row
            const int row = idx / (HC / 4), c4 = idx - row * (HC / 4);
            const int m = m0 + row;
            const float* sp = lc + row * CP + c4 * 4;
            if (m < a.M) out_store16(a.lstm_cell + (long long)m * C + hc0 + c4 * 4, make_uint4(__float_as_uint(sp[0]), __float_as_uint(sp[1]), __float_as_uint(sp[2]), __float_as_uint(sp[3])));
        }
        const uint32_t* lhv = reinterpret_cast<const uint32_t*>(lh);
#pragma unroll
        for (int idx = tid; idx < ROWS * (HC / 8); idx += 256) {       // hidden: 8 bf16 per lane
            const int row = idx / (HC / 8), c8 = idx - row * (HC / 8);
            const int m = m0 + row;
            const uint32_t* sp = lhv + row * (HP / 2) + c8 * 4;
            if (m < a.M) out_store16(a.lstm_h + (long long)m * a.lstm_h_stride + hc0 + c8 * 8, make_uint4(sp[0], sp[1], sp[2], sp[3]));
        }
        return;
    }
#pragma unroll 4
    for (int idx = tid; idx < ROWS * HC; idx += 256) {     // cell: ROWS x HC fp32, whole rows per lane group
        const int row = idx / HC, col = idx - row * HC;
        const int m = m0 + row;
        if (m < a.M) a.lstm_cell[(long long)m * C + hc0 + col] = lc[row * CP + col];
    }
    const uint32_t* lh32 = reinterpret_cast<const uint32_t*>(lh);
#pragma unroll 4
    for (int idx = tid; idx < ROWS * (HC / 2); idx += 256) {   // hidden: ROWS x HC/2 dwords (2 bf16 each)
        const int row = idx / (HC / 2), col = idx - row * (HC / 2);
        const int m = m0 + row;
        if (m < a.M) *reinterpret_cast<uint32_t*>(a.lstm_h + (long long)m * a.lstm_h_stride + hc0 + col * 2) = lh32[row * (HP / 2) + col];
    }
}

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

    conv_epilogue<BM, BN>(a, acc, smem, m0, n0, wm, wn, lane, tid);
}


// =================================================================================================
// v2: LDS-DMA pipeline.  Both operand slabs are moved HBM -> LDS by `buffer_load_dwordx4 ... lds`
// (no staging VGPRs, no ds_write pass, fully asynchronous).  Probed on MI355X
// (tools/probes/lds_dma_probe.hip): the destination is M0-base + lane*16 (lane-linear per wave) and an
// out-of-range voffset WRITES ZEROS, which is exactly the zero padding an implicit GEMM needs: padded
// taps and rows beyond M simply get voffset = 0x80000000.  The XOR swizzle is applied on the SOURCE
// side (lane p of a wave instruction fetches the chunk that belongs at LDS slot p), reads are unchanged.
// NSTAGE-deep LDS ring, ONE raw s_barrier per K-slab, counted vmcnt so that NSTAGE-2 slabs stay in
// flight across the barrier (a __syncthreads would drain them: cdna_hip_programming.md section 5).
// =================================================================================================
// FASTK: Cin % 64 == 0, i.e. every 64-wide K-slab lies inside ONE filter tap -> the tap decode is wave-uniform
// (scalar) and the per-lane part of a gather address is a constant.
// NTHREADS = 256 with a 256 x 256 tile gives the vendor-GEMM shape: 4 waves, each a 128 x 128 wave tile (16 MFMAs per
// 8 fragment reads instead of 4 per 4 -> half the LDS read bytes per FLOP), 2 x 64 KB ring, 1 workgroup per CU.
template <int BMX, int BN, int NSTAGE, bool FASTK, int EPI = 0, int NTHREADS = conv_tile_threads(BMX)>
__global__ __launch_bounds__(NTHREADS) void conv_fwd_dma_kernel(ConvArgs a) {
    constexpr int NWAVES = NTHREADS / 64;            // 64 / 128-row tile: 4 waves, 256 x 128 tile: 8 waves
    constexpr int WAVES_N = (BN >= 128) ? 2 : 1;
    constexpr int WAVES_M = NWAVES / WAVES_N;
    constexpr int WM = BMX / WAVES_M;
    constexpr int WN = BN / WAVES_N;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int A_INSTR = BMX * 8 / 64 / NWAVES;   // = 4: BMX rows x 8 chunks / 64 lanes / waves
    constexpr int B_INSTR = BN * 8 / 64 / NWAVES;    // BN rows x 8 chunks / 64 lanes / waves
    constexpr int IPS = A_INSTR + B_INSTR;           // DMA instructions per thread per stage
    constexpr int STAGE_BYTES = (BMX + BN) * 8 * 16;
    constexpr int NFRAG = MT + NT;                   // ds_read_b128 per k-step

    // ONE LDS array (a second __shared__ object makes hipcc drain vmcnt before every ds_read)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
    const int m0 = tile_m * BMX, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // K-slab range of this workgroup: everything, or slice blockIdx.y of a split-K launch
    const int kbeg = (EPI == 0 && a.partial) ? (int)blockIdx.y * a.kt_per : 0;
    const int KT = (EPI == 0 && a.partial) ? ((kbeg + a.kt_per < a.Kpad / BK) ? kbeg + a.kt_per : a.Kpad / BK) : a.Kpad / BK;
    const int cpt = a.Cin >> 3, ntaps = a.R * a.S;

    // buffer descriptors (wave-uniform kernel arguments only)
    const long long in_bytes = (((long long)a.B * a.H * a.W - 1) * a.in_pix_stride + a.Cin) * 2;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7ffffff0, 0x00020000);

    // lane geometry of one wave-level DMA instruction: 8 rows x 8 chunk slots
    const int lrow = lane >> 3, slot = lane & 7;
    int iy0[A_INSTR], ix0[A_INSTR], rowoff[A_INSTR], csrc[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int r = (wave * A_INSTR + i) * 8 + lrow;           // row of the tile this lane fetches
        const int m = m0 + r;
        const bool valid = m < a.M;
        const int mm = valid ? m : 0;
        const int hw = a.Ho * a.Wo;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        iy0[i] = valid ? oy * a.stride - a.pad : -0x4000;
        ix0[i] = ox * a.stride - a.pad;
        rowoff[i] = (int)((((long long)b * a.H + (oy * a.stride - a.pad)) * a.W + ix0[i]) * a.in_pix_stride * 2);
        csrc[i] = slot ^ ((r >> 1) & 7);                           // source chunk that lives at this LDS slot
    }
    int boff[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int r = (wave * B_INSTR + i) * 8 + lrow;
        boff[i] = ((n0 + r) * a.Kpad + (slot ^ ((r >> 1) & 7)) * 8) * 2;
    }

    // DMA issue for K-slab kt.  Tap arithmetic uses exact reciprocals (host-verified): no LDS table reads here,
    // because hipcc drains vmcnt(0) in front of any compiler-visible LDS read while an LDS-DMA is in flight.
    auto issue = [&](int kt) {
        unsigned char* st = smem + (kt % NSTAGE) * STAGE_BYTES;
        if constexpr (FASTK) {
            // scalar tap decode for the whole slab
            const unsigned kc0 = (unsigned)(kt * 8);
            const unsigned tap = (kc0 * a.inv_cpt) >> 20;
            const int cc0 = (int)(kc0 - tap * cpt);
            const unsigned r = (tap * a.inv_s) >> 16;
            const int sx = (int)(tap - r * a.S);
            const int dy = (int)r * a.dil, dx = sx * a.dil;
            const int tapoff = ((dy * a.W + dx) * (int)a.in_pix_stride + cc0 * 8) * 2;
            const bool tap_ok = (int)tap < ntaps;
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i) {
                const bool ok = tap_ok && (unsigned)(iy0[i] + dy) < (unsigned)a.H && (unsigned)(ix0[i] + dx) < (unsigned)a.W;
                const unsigned voff = ok ? (unsigned)(rowoff[i] + csrc[i] * 16 + tapoff) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave * A_INSTR + i) * 1024),
                                                         16, voff, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i) {
                const unsigned kc = (unsigned)(kt * 8 + csrc[i]);
                const unsigned tap = (kc * a.inv_cpt) >> 20;
                const int cc = (int)(kc - tap * cpt);
                const unsigned r = (tap * a.inv_s) >> 16;
                const int sx = (int)(tap - r * a.S);
                const int dy = (int)r * a.dil, dx = sx * a.dil;
                const bool ok = (int)tap < ntaps && (unsigned)(iy0[i] + dy) < (unsigned)a.H && (unsigned)(ix0[i] + dx) < (unsigned)a.W;
                const unsigned voff = ok ? (unsigned)(rowoff[i] + ((dy * a.W + dx) * (int)a.in_pix_stride + cc * 8) * 2) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave * A_INSTR + i) * 1024),
                                                         16, voff, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(st + BMX * 128 + (wave * B_INSTR + i) * 1024),
                                                     16, (unsigned)(boff[i] + kt * BK * 2), 0, 0, 0);
    };

    f32x16_t acc[MT][NT];
    if constexpr (EPI == 1) {
        lstm_bias_init<MT, NT>(a, acc, n0, wn, lane);
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    }

    // fragment addresses (bytes from the stage base), fixed over the K loop: row r, chunk slot swz(r, ks*2 + half)
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    uint32_t fa_off[MT], fb_off[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) fa_off[i] = (uint32_t)(wm * WM + i * 32 + (lane & 31)) * 128;
#pragma unroll
    for (int j = 0; j < NT; ++j) fb_off[j] = (uint32_t)(BMX * 128 + (wn * WN + j * 32 + (lane & 31)) * 128);
    const int half = lane >> 5;
    // slot of chunk (ks*2+half) in row r: (ks*2+half) ^ ((r>>1)&7); (r>>1)&7 == ((lane&31)>>1)&7 for every tile row here
    const int rsw = ((lane & 31) >> 1) & 7;

#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (kbeg + s < KT) issue(kbeg + s);
    constexpr bool LSTM_PREF = (EPI == 1 && MT == 2 && NT == 2);
    LstmPrefetch pref;
    if constexpr (LSTM_PREF) lstm_prefetch(a, pref, m0, n0, tid);

#define OESS_FRAG_READ(DST_A, DST_B, KS)                                                                         \
    {                                                                                                            \
        const uint32_t sl_ = (uint32_t)((((KS) * 2 + half) ^ rsw) * 16);                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                           \
            asm volatile("ds_read_b128 %0, %1" : "=v"(DST_A[i]) : "v"(stage_ + fa_off[i] + sl_) : "memory");   \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                           \
            asm volatile("ds_read_b128 %0, %1" : "=v"(DST_B[j]) : "v"(stage_ + fb_off[j] + sl_) : "memory");   \
    }
#define OESS_FRAG_MMA(SRC_A, SRC_B)                                                                              \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                           \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                       \
                acc[i][j] = (EPI == 1) ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(SRC_B[j], SRC_A[i], acc[i][j], 0, 0, 0)     \
                                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(SRC_A[i], SRC_B[j], acc[i][j], 0, 0, 0);   \
    }
    // wait until only N_ LDS reads remain outstanding; the "+v" operands tie later uses of the fragments to the wait
#define OESS_WAIT_FRAGS(N_, FA_, FB_)                                                                            \
    {                                                                                                            \
        if constexpr (MT == 4 && NT == 4)                                                                        \
            asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(FA_[0]), "+v"(FA_[1]), "+v"(FA_[2]), "+v"(FA_[3]),       \
                         "+v"(FB_[0]), "+v"(FB_[1]), "+v"(FB_[2]), "+v"(FB_[3]) : "n"(N_) : "memory");           \
        else if constexpr (MT == 2 && NT == 4)                                                                   \
            asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(FA_[0]), "+v"(FA_[1]), "+v"(FB_[0]), "+v"(FB_[1]), "+v"(FB_[2]), "+v"(FB_[3]) : "n"(N_) : "memory"); \
        else if constexpr (MT == 2 && NT == 2)                                                                   \
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(FA_[0]), "+v"(FA_[1]), "+v"(FB_[0]), "+v"(FB_[1]) : "n"(N_) : "memory"); \
        else if constexpr (MT == 1 && NT == 2)                                                                   \
            asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(FA_[0]), "+v"(FB_[0]), "+v"(FB_[1]) : "n"(N_) : "memory"); \
        else                                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(FA_[0]), "+v"(FB_[0]) : "n"(N_) : "memory");             \
    }

    for (int kt = kbeg; kt < KT; ++kt) {
        // retire slab kt: at most NSTAGE-2 younger slabs may stay in flight (fewer at the tail)
        if (kt + NSTAGE - 2 < KT) {
            if constexpr (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr ((NSTAGE - 2) * IPS == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if constexpr ((NSTAGE - 2) * IPS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr ((NSTAGE - 2) * IPS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr ((NSTAGE - 2) * IPS == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if constexpr ((NSTAGE - 2) * IPS == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr ((NSTAGE - 2) * IPS == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                   // slab kt complete for every wave; buffer of slab kt-1 is free
        if (kt + NSTAGE - 1 < KT) issue(kt + NSTAGE - 1);
        const uint32_t stage_ = lds0 + (uint32_t)((kt % NSTAGE) * STAGE_BYTES);
        bf16x8_t fa0[MT], fb0[NT], fa1[MT], fb1[NT];
        // register double-buffered fragments: reads of k-step ks+1 are in flight under the MFMAs of k-step ks
        __builtin_amdgcn_s_setprio(3);
        OESS_FRAG_READ(fa0, fb0, 0)
        OESS_FRAG_READ(fa1, fb1, 1)
        OESS_WAIT_FRAGS(NFRAG, fa0, fb0)
        OESS_FRAG_MMA(fa0, fb0)
        OESS_FRAG_READ(fa0, fb0, 2)
        OESS_WAIT_FRAGS(NFRAG, fa1, fb1)
        OESS_FRAG_MMA(fa1, fb1)
        OESS_FRAG_READ(fa1, fb1, 3)
        OESS_WAIT_FRAGS(NFRAG, fa0, fb0)
        OESS_FRAG_MMA(fa0, fb0)
        OESS_WAIT_FRAGS(0, fa1, fb1)
        OESS_FRAG_MMA(fa1, fb1)
        __builtin_amdgcn_s_setprio(0);
    }
#undef OESS_FRAG_READ
#undef OESS_FRAG_MMA
#undef OESS_WAIT_FRAGS
    __syncthreads();

    if constexpr (LSTM_PREF) lstm_epilogue<MT, NT, true, false>(a, acc, smem, m0, n0, wm, wn, lane, tid, &pref);
    else if constexpr (EPI == 1) lstm_epilogue<MT, NT, false, false>(a, acc, smem, m0, n0, wm, wn, lane, tid);
    else {
        if (a.partial) {        // split-K slice: raw fp32 accumulators, 128-byte row segments per half wave
            float* dst = a.partial + (size_t)blockIdx.y * (size_t)a.M * a.Cout;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int n = n0 + wn * WN + j * 32 + (lane & 31);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        if (m < a.M && n < a.Cout) dst[(size_t)m * a.Cout + n] = acc[i][j][e];
                    }
                }
            return;
        }
        conv_epilogue<BMX, BN, BN + 8, NTHREADS, WAVES_N>(a, acc, smem, m0, n0, wm, wn, lane, tid);
    }
}

// Split-K tail: out[m][n] = act(sum_z partial[z][m][n] + bias[n] [+ residual]) as bf16 NHWC, slices added in z order
// (deterministic), plus the per-128-row-tile column sums / sums of squares of the fp32 result that the conv epilogue
// provides for BatchNorm (same [tiles_m][2][Cout] layout).  grid = (tiles_m, ceil(Cout / 64)); thread = (row lane, 4 columns).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvArgs a) {
    __shared__ float red[16][64][2];
    const int tile = blockIdx.x, n = blockIdx.y * 64 + (threadIdx.x & 15) * 4, rl = threadIdx.x >> 4;
    const size_t MN = (size_t)a.M * a.Cout;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const bool n_ok = n < a.Cout;               // Cout % 4 == 0 (host-checked): a float4 is inside the row or not at all
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias && n_ok) { bv[0] = a.bias[n]; bv[1] = a.bias[n + 1]; bv[2] = a.bias[n + 2]; bv[3] = a.bias[n + 3]; }
    for (int r = rl; r < 128; r += 16) {
        const int m = tile * 128 + r;
        if (m >= a.M || !n_ok) continue;
        const float* p = a.partial + (size_t)m * a.Cout + n;
        float4 v = *reinterpret_cast<const float4*>(p);
        for (int z = 1; z < a.ksplit; ++z) {
            const float4 w = *reinterpret_cast<const float4*>(p + (size_t)z * MN);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        float f[4] = {v.x + bv[0], v.y + bv[1], v.z + bv[2], v.w + bv[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float q = bf16_to_f32(f32_to_bf16(f[k])); s1[k] += q; s2[k] += q * q; }   // statistics of the stored values
        if (a.out_f32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) f[k] = conv_act(f[k], a.relu);
            *reinterpret_cast<float4*>(a.out_f32 + (long long)m * a.out_pix_stride + n) = make_float4(f[0], f[1], f[2], f[3]);
            continue;
        }
        if (a.residual || a.relu) {
            // the one-pass epilogue rounds the conv result to bf16 BEFORE the residual add: keep that rounding point
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = bf16_to_f32(f32_to_bf16(f[k]));
                if (a.residual) t += bf16_to_f32(a.residual[(long long)m * a.res_pix_stride + n + k]);
                f[k] = conv_act(t, a.relu);
            }
        }
        uint2 o;
        o.x = pack_bf16x2(f[0], f[1]);
        o.y = pack_bf16x2(f[2], f[3]);
        *reinterpret_cast<uint2*>(a.out + (long long)m * a.out_pix_stride + n) = o;
    }
    if (!a.stats) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[rl][(threadIdx.x & 15) * 4 + k][0] = s1[k]; red[rl][(threadIdx.x & 15) * 4 + k][1] = s2[k]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int nn = blockIdx.y * 64 + threadIdx.x;
        if (nn < a.Cout) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { t1 += red[r][threadIdx.x][0]; t2 += red[r][threadIdx.x][1]; }
            a.stats[((size_t)tile * 2 + 0) * a.Cout + nn] = t1;
            a.stats[((size_t)tile * 2 + 1) * a.Cout + nn] = t2;
        }
    }
}

// =================================================================================================
// 3x3 / stride-1 / "same" convolutions with ROW-HALO REUSE of the pixel operand (conv3x3_halo_kernel).
//
// In the implicit GEMM above, every K-slab = (filter tap, 64 input channels) fetches its own 128 x 64 im2col block, yet the
// blocks of the three taps of one filter ROW (dx = 0, 1, 2) are the same 128 pixels shifted by `dil` pixels.  Here the
// K loop runs in (dy, channel chunk, dx) order -- the standard packed weight already holds each (tap, chunk) as 64
// contiguous k, so no new packing -- and the pixel operand of the three dx taps is ONE halo buffer in LDS:
//   halo rows = [dil lead pixels][segment 0][dil gap][segment 1][dil gap] ... [last segment][dil trail pixels]
// where a segment is a run of tile pixels inside one image row; the gap rows are written as zeros by out-of-range DMA
// offsets, so a side tap that crosses an image-row (or image) boundary lands on zeros exactly like the im2col padding.
// Tile pixel i sits at halo row hrow(i); tap dx reads row hrow(i) + (dx - 1) * dil.  Per K-slab the workgroup now moves
// 16 KB of weights + a third of a <= 20 KB halo instead of 32 KB: ~30 % fewer L2->LDS bytes and LDS-DMA writes on the
// layers that make up most of the step (ConvLSTM gates, decoder and teacher 3x3 convs).
// LDS: 2 halo buffers (2 x 20 KB) + 2 weight stages (2 x 16 KB) = 72 KB -> still two workgroups per CU.
// =================================================================================================
constexpr int HALO_ROWS = 160;

// one 128 x 128 tile (`bid` = tile index after the XCD remap); smem = [halo 0][halo 1][weights 0][weights 1]
// BMX = 256 (EPI = 1 only; round 5): an 8-wave workgroup owns 256 pixels x 128 gate columns -- the weight slab is
// fetched once per 256 pixels (150 instead of 92 FLOP per L2 -> LDS byte, 3.7 instead of 5.7 DMA instructions per wave and slab),
// one workgroup per CU (2 x 40 KB halo + 2 x 16 KB weights), the wave tile stays 64 x 64.
constexpr int HALO_ROWS_256 = 320;
constexpr int LSTM_EPI_HALF = 26624;                     // LDS of one 128-row half of the ConvLSTM epilogue (25 600 B used)
template <int EPI, int BMX = 128>
__device__ __forceinline__ void conv3x3_halo_tile(const ConvArgs& a, const int bid, unsigned char* smem) {
    constexpr int BN = 128, NWAVES = BMX / 32, WAVES_N = 2, WM = 64, WN = 64, MT = 2, NT = 2;
    constexpr int HROWS = (BMX == 128) ? HALO_ROWS : HALO_ROWS_256;
    constexpr int H_INSTR = HROWS / 8 / NWAVES;          // 5 DMA instructions per thread per halo
    constexpr int B_INSTR = BN * 8 / 64 / NWAVES;        // 4 (2 with eight waves)
    constexpr int HALO_BYTES = HROWS * 128, BST_BYTES = BN * 128;
    static_assert(BMX == 128 || EPI == 1, "the 256-row tile exists for the fused ConvLSTM only");
    constexpr int NFRAG = MT + NT;

    const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
    const int m0 = tile_m * BMX, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // Loop-invariant scalars of the K loop, pinned in SGPRs.  In the grouped kernel `a` is g.a[p] with a run-time p, i.e. kernel-
    // argument MEMORY: hipcc treats such loads as free to rematerialise and re-issued s_load_dword a.Cin / a.H / a.in_pix_stride
    // + s_waitcnt lgkmcnt(0) in front of every slab's barrier (round-5 disassembly: three scalar-cache round trips per macro step
    // on the kernel that owns 41 % of the step).  The empty asm makes the values opaque, so they stay in registers.
    int Cin_s = a.Cin, H_s = a.H, ips_s = (int)a.in_pix_stride;
    asm volatile("" : "+s"(Cin_s), "+s"(H_s), "+s"(ips_s));
    const int nch = Cin_s >> 6;                          // 64-channel chunks
    const int NJ = 3 * nch;                              // macro steps (dy, chunk); 3 K-slabs each
    const int W = a.W, dil = a.dil, wd = W + dil;

    const long long in_bytes = (((long long)a.B * a.H * a.W - 1) * a.in_pix_stride + a.Cin) * 2;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7ffffff0, 0x00020000);

    // tile origin (wave-uniform)
    const int hw = a.H * W;
    const int b0 = m0 / hw, rem0 = m0 - b0 * hw;
    const int oy0 = rem0 / W, ox0 = rem0 - oy0 * W;
    const int L0 = (W - ox0 < BMX) ? W - ox0 : BMX;      // tile pixels in the first image row

    // ---- halo DMA geometry: lane (lrow, slot) of instruction q writes halo row h = q*8 + lrow, 16-byte slot `slot`
    const int lrow = lane >> 3, slot = lane & 7;
    int hy[H_INSTR], hoff[H_INSTR];
#pragma unroll
    for (int i = 0; i < H_INSTR; ++i) {
        const int h = (wave * H_INSTR + i) * 8 + lrow;
        const int hp = h - dil;
        int m_seg, px, drow;                             // first tile pixel of the row's segment, x coordinate of this halo row, image rows below the tile's first
        if (hp < L0 + dil) { m_seg = m0; px = ox0 + hp; drow = 0; }
        else {
            const int h2 = hp - (L0 + dil);
            const int q = a.mg_wd ? (int)__umulhi((unsigned)h2, a.mg_wd) : h2 / wd, r = h2 - q * wd;
            m_seg = m0 + L0 + q * W; px = r; drow = q + 1;
        }
        const bool valid = m_seg < a.M && (m_seg == m0 || m_seg - m0 < BMX) && (unsigned)px < (unsigned)W;
        // segment q starts an image row: its row is the tile's first row + drow, carried into the next image(s) -- no division
        int oy = oy0 + drow;
        const long long grow = (long long)b0 * a.H + oy;  // row index over the whole batch
        while (oy >= a.H) oy -= a.H;
        hy[i] = valid ? oy : -0x4000;
        hoff[i] = valid ? (int)((grow * W + px) * a.in_pix_stride * 2) + (slot ^ ((h >> 1) & 7)) * 16 : 0;
    }
    int boff[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int r = (wave * B_INSTR + i) * 8 + lrow;
        boff[i] = ((n0 + r) * a.Kpad + (slot ^ ((r >> 1) & 7)) * 8) * 2;
    }
    // halo part `part` (instructions [i0, i1)) of macro step j -> halo buffer j & 1
    auto issue_halo = [&](int j, int dy, int cc, int i0, int i1) {        // (dy, cc) = (j / nch, j % nch), kept by the caller
        const int ddy = (dy - 1) * dil;
        const int tapoff = (ddy * W * ips_s + cc * 64) * 2;
        unsigned char* st = smem + (j & 1) * HALO_BYTES;
#pragma unroll
        for (int i = 0; i < H_INSTR; ++i) {
            if (i < i0 || i >= i1) continue;
            const bool ok = (unsigned)(hy[i] + ddy) < (unsigned)H_s;
            const unsigned voff = ok ? (unsigned)(hoff[i] + tapoff) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave * H_INSTR + i) * 1024),
                                                     16, voff, 0, 0, 0);
        }
    };
    // weight slab of (macro step j, dx) -> weight stage kt & 1, kt = 3*j + dx
    auto issue_w = [&](int j, int dy, int cc, int dx) {
        const int koff = ((dy * 3 + dx) * Cin_s + cc * 64) * 2;
        unsigned char* st = smem + 2 * HALO_BYTES + ((3 * j + dx) & 1) * BST_BYTES;
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(st + (wave * B_INSTR + i) * 1024),
                                                     16, (unsigned)(boff[i] + koff), 0, 0, 0);
    };

    f32x16_t acc[MT][NT];
    if constexpr (EPI == 1) {
        lstm_bias_init<MT, NT>(a, acc, n0, wn, lane);
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    }

    // ---- fragment addresses.  Pixel fragments: halo row of tile pixel (wm*64 + i*32 + lane&31), shifted per dx tap
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    uint32_t fa_row[MT][3], fa_sw[MT][3], fb_off[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = wm * WM + i * 32 + (lane & 31);
        int hr;
        if (r < L0) hr = r;                              // (+ dil lead rows, - dil for the dx = 0 tap)
        else {
            const int t = r - L0, q = a.mg_w ? (int)__umulhi((unsigned)t, a.mg_w) : t / W, rr = t - q * W;
            hr = L0 + dil + q * wd + rr;
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int h = hr + dx * dil;
            fa_row[i][dx] = (uint32_t)h * 128;
            fa_sw[i][dx] = (uint32_t)((h >> 1) & 7);
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) fb_off[j] = (uint32_t)(2 * HALO_BYTES + (wn * WN + j * 32 + (lane & 31)) * 128);
    const uint32_t half = (uint32_t)(lane >> 5);
    const uint32_t rswb = (uint32_t)(((lane & 31) >> 1) & 7);

    // Fragment addresses, complete: one VGPR per (tile row block, dx tap, k-step) for the pixel operand and per (column block,
    // k-step) for the weights; the halo buffer (j & 1) and the weight stage ((j + dx) & 1) enter as the ds_read's IMMEDIATE offset
    // (the macro-step loop is unrolled by the parity of j), so a fragment read costs no VALU instruction.  Round-5 PMC: the
    // kernel issues ~80 non-MFMA instructions per 16 MFMAs per wave, the most an in-order wave hides (MI355X_MICROARCH.md, "one
    // wave per SIMD: <= 5 single-issue instructions hidden per MFMA gap"); 30 of them were the v_add_u32 of these addresses.
    uint32_t fa_addr[MT][3][4], fb_addr[NT][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint32_t c_ = (uint32_t)(ks * 2) + half;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                fa_addr[i][dx][ks] = lds0 + fa_row[i][dx] + ((c_ ^ fa_sw[i][dx]) << 4);
                asm volatile("" : "+v"(fa_addr[i][dx][ks]));       // keep it in its register (not recomputed in the loop)
            }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            fb_addr[j][ks] = lds0 + fb_off[j] + ((c_ ^ rswb) << 4);
            asm volatile("" : "+v"(fb_addr[j][ks]));
        }
    }

    issue_halo(0, 0, 0, 0, H_INSTR);
    issue_w(0, 0, 0, 0);
    constexpr bool LSTM_PREF = (EPI == 1);
    LstmPrefetch pref;
    const int ehalf = (BMX == 256) ? (wm >> 1) : 0;      // 256-row tile: the epilogue runs as two independent 128-row halves
    if constexpr (LSTM_PREF) lstm_prefetch(a, pref, m0 + ehalf * 128, n0, tid & 255);

#define OESS_HFRAG_READ(DST_A, DST_B, KS, DX)                                                                    \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                           \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST_A[i]) : "v"(fa_addr[i][DX][KS]), "n"(HOFF) : "memory"); \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                           \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST_B[j]) : "v"(fb_addr[j][KS]), "n"(WOFF) : "memory"); \
    }
#define OESS_HFRAG_MMA(SRC_A, SRC_B)                                                                             \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                           \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                       \
                acc[i][j] = (EPI == 1) ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(SRC_B[j], SRC_A[i], acc[i][j], 0, 0, 0)     \
                                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(SRC_A[i], SRC_B[j], acc[i][j], 0, 0, 0);   \
    }
#define OESS_HWAIT(N_, FA_, FB_)                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(FA_[0]), "+v"(FA_[1]), "+v"(FB_[0]), "+v"(FB_[1]) : "n"(N_) : "memory");

    int dy_c = 0, cc_c = 0;                              // (dy, chunk) of macro step j, carried instead of divided out
    int dy_n = 0, cc_n = 0;
    // one K-slab (macro step j of parity PAR, tap dx): barrier, next operands on their way, 16 MFMAs
    auto slab = [&](auto par_c, auto dx_c, int j) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value, dx = decltype(dx_c)::value;
        constexpr int HOFF = PAR * HALO_BYTES, WOFF = ((PAR + dx) & 1) * BST_BYTES;       // (3 j + dx) & 1 == (j + dx) & 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // slab (j, dx) complete for every wave; the other buffers are free
        // next weight slab, and a third of the next macro step's halo, travel under this slab's MFMAs
        if (dx < 2) issue_w(j, dy_c, cc_c, dx + 1);
        else if (j + 1 < NJ) issue_w(j + 1, dy_n, cc_n, 0);
        if (j + 1 < NJ) {
            if (dx == 0) issue_halo(j + 1, dy_n, cc_n, 0, 2);
            else if (dx == 1) issue_halo(j + 1, dy_n, cc_n, 2, 4);
            else issue_halo(j + 1, dy_n, cc_n, 4, H_INSTR);
        }
        bf16x8_t fa0[MT], fb0[NT], fa1[MT], fb1[NT];
        __builtin_amdgcn_s_setprio(3);
        OESS_HFRAG_READ(fa0, fb0, 0, dx)
        OESS_HFRAG_READ(fa1, fb1, 1, dx)
        OESS_HWAIT(NFRAG, fa0, fb0)
        OESS_HFRAG_MMA(fa0, fb0)
        OESS_HFRAG_READ(fa0, fb0, 2, dx)
        OESS_HWAIT(NFRAG, fa1, fb1)
        OESS_HFRAG_MMA(fa1, fb1)
        OESS_HFRAG_READ(fa1, fb1, 3, dx)
        OESS_HWAIT(NFRAG, fa0, fb0)
        OESS_HFRAG_MMA(fa0, fb0)
        OESS_HWAIT(0, fa1, fb1)
        OESS_HFRAG_MMA(fa1, fb1)
        __builtin_amdgcn_s_setprio(0);
    };
    auto macro_step = [&](auto par_c, int j) __attribute__((always_inline)) {
        dy_n = dy_c; cc_n = cc_c + 1;                    // macro step j + 1
        if (cc_n == nch) { cc_n = 0; ++dy_n; }
        slab(par_c, std::integral_constant<int, 0>{}, j);
        slab(par_c, std::integral_constant<int, 1>{}, j);
        slab(par_c, std::integral_constant<int, 2>{}, j);
        dy_c = dy_n; cc_c = cc_n;
    };
    for (int j = 0; j < NJ; j += 2) {
        macro_step(std::integral_constant<int, 0>{}, j);
        if (j + 1 < NJ) macro_step(std::integral_constant<int, 1>{}, j + 1);
    }
#undef OESS_HFRAG_READ
#undef OESS_HFRAG_MMA
#undef OESS_HWAIT
    __syncthreads();

    if constexpr (EPI == 1) lstm_epilogue<MT, NT, true, false>(a, acc, smem + ehalf * LSTM_EPI_HALF, m0 + ehalf * 128, n0, wm & 1, wn, lane,
                                                                 tid & 255, &pref);
    else conv_epilogue<BMX, BN, BN + 8, 256, WAVES_N>(a, acc, smem, m0, n0, wm, wn, lane, tid);
}

template <int EPI>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    conv3x3_halo_tile<EPI>(a, bid, smem);
}

// Up to three INDEPENDENT problems of the kernel above in one launch (the three ConvLSTM levels of E2VID's recurrent encoder
// on the skewed schedule: level l works on sub-window s - l, e2vid/model/unet.py mirror).  Alone, the levels are 17.2 /
// 8.6 / 4.3 rounds of tiles over the 512 workgroup slots and each pays its own partial last round and launch gap; together
// they are 30.1 rounds.  The host orders the problems by K, longest tiles first, so that the launch ends on the short ones.
// Workgroup blockIdx = 8 * idx + xcd: problem p owns idx in [start8[p], start8[p + 1]) on every XCD, and inside it the tiles
// are dealt to the XCDs in contiguous chunks exactly as the single-problem kernel does (neighbouring tiles share halo rows
// and weight slabs in that XCD's L2).
struct ConvGroup {
    ConvArgs a[3];
    int start8[4];           // problem p owns idx in [start8[p], start8[p + 1]); absent problems: empty ranges at the end
};
template <int EPI>
__global__ __launch_bounds__(256) void conv3x3_halo_group_kernel(ConvGroup g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    // (alternating the short-K tiles with the long-K ones, so that a CU's two workgroups differ in kind, was measured: -2.8 %
    //  against this problem-after-problem order -- the problems' weight slabs and halos evict each other from the XCD's L2)
    const int p = (idx >= g.start8[1] ? 1 : 0) + (idx >= g.start8[2] ? 1 : 0);
    const int li = idx - g.start8[p];
    const ConvArgs& a = g.a[p];
    const int nwg = a.tiles_m * a.tiles_n;
    const int q = nwg >> 3, r = nwg & 7;
    if (li >= q + (xcd < r ? 1 : 0)) return;             // padding of a problem whose tile count is not a multiple of 8
    conv3x3_halo_tile<EPI>(a, (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + li, smem);
}

// the same grouped launch on 256 x 128 tiles (8 waves, one workgroup per CU): tiles_m of every problem counts 256-row tiles
__global__ __launch_bounds__(512) void conv3x3_halo256_group_kernel(ConvGroup g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int p = (idx >= g.start8[1] ? 1 : 0) + (idx >= g.start8[2] ? 1 : 0);
    const int li = idx - g.start8[p];
    const ConvArgs& a = g.a[p];
    const int nwg = a.tiles_m * a.tiles_n;
    const int q = nwg >> 3, r = nwg & 7;
    if (li >= q + (xcd < r ? 1 : 0)) return;
    conv3x3_halo_tile<1, 256>(a, (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + li, smem);
}

#include "conv_lstm_w128.h"
#include "conv_w128_gemm.h"
#include "conv3x3_w128.h"

// =================================================================================================
// 5x5 / stride-2 / pad-2 convolutions with a 2-D INPUT HALO in LDS (conv5x5s2_halo_kernel): E2VID's three encoder
// ConvLayers (e2vid/model/unet.py: 32->64 @440x640, 64->128 @220x320, 128->256 @110x160, 20 launches each per step).
//
// In the implicit GEMM every (tap, channel chunk) K-slab fetches its own im2col block, so an input pixel travels L2 -> LDS
// 25 / 4 = 6.25 times per output-channel tile.  Here a workgroup owns an 8 x 16 patch of output pixels x 64 output channels
// and keeps the 19 x 35 input pixels under it (32 channels = 64 bytes each) in LDS for all 25 taps: 5.2 input pixels per
// output pixel instead of 25, and tiles of 128 x 64 instead of 128 x 128 (4 400 / 2 200 / 1 120 workgroups on the three
// layers instead of 550-1 100 tiles that quantise badly over the 512 slots).
// Stride 2 would make the 16 lanes of a fragment-read phase touch only every second 64-byte block (half the banks), so the
// halo is stored as TWO COLUMN-PARITY PLANES: plane p holds input columns 2 xi + p, and tap column s reads plane s & 1 at
// xi = px + (s >> 1) -- consecutive output pixels are consecutive blocks again.  Block (p, hy, xi) keeps its four 16-byte
// chunks at slot = chunk ^ (((xi >> 2) & 1) | (((hy >> 1) & 1) << 1)); a fragment row r of an M-tile is the pixel
// (py = 2 i + ((r >> 3) & 1), px = (r & 7) + 8 (r >> 4)), so a phase (16 lanes) = 8 pixels x 2 patch rows = 4 lanes in each
// of the 4 bank quarters with 4 different slots: conflict-free.  Weights: one (tap, 32-channel) slab of 64 rows x 64 bytes
// per tap in a 6-deep LDS-DMA ring (5 slabs = ~1.5 us of taps in flight), slot = chunk ^ ((row >> 2) & 3).
// Four waves, each a 64-pixel x 32-channel wave tile (fragments of the next k-step in flight under the current MFMAs); LDS 44 KB
// halo + 24 KB ring = 68 KB -> two workgroups per CU = two waves per SIMD from DIFFERENT workgroups, so one workgroup's halo
// fetch and epilogue run under the other's K loop (with two-wave workgroups, one wave per SIMD, the three phases of a tile
// simply added up: 130 us on the 32->64 layer, 92 us with the epilogue compiled out, against 118 us for the im2col kernel).
// Channel chunks (Cin = 64 / 128) reload the halo; the weight ring runs on across the chunk boundary.
// =================================================================================================
constexpr int S2_PH = 8, S2_PW = 16;                                 // output patch
constexpr int S2_HH = 2 * S2_PH + 3, S2_XW = 18;                     // 19 halo rows; 18 blocks per plane row (35 columns)
constexpr int S2_BLOCKS = 2 * S2_HH * S2_XW;                         // 684 blocks of 64 bytes
constexpr int S2_HINSTR = (S2_BLOCKS + 63) / 64;                     // 11 wave-level DMA instructions (16 blocks each) per wave, 4 waves
constexpr int S2_HALO_BYTES = S2_HINSTR * 4 * 1024;                  // 45 056
constexpr int S2_SLAB = 64 * 64;
constexpr int S2_RING_PLAIN = 6, S2_RING_FUSED = 5;                  // weight slabs in the ring
constexpr int S2_LDS = S2_HALO_BYTES + S2_RING_PLAIN * S2_SLAB;      // 69 632
// fused E2VID head (5x5 stride 1, 8 -> 32 channels) in front of the 32 -> 64 encoder: the voxel patch under the halo
constexpr int S2_VH = S2_HH + 4, S2_VW = 2 * S2_PW + 3 + 4;          // 23 x 39 voxel pixels of 16 bytes (8 channels)
constexpr int S2_VINSTR = (S2_VH * S2_VW + 255) / 256;               // 4 wave-level DMA instructions per wave (64 pixels each)
constexpr int S2_VTOTAL = (S2_VH * S2_VW + 63) / 64;                 // 15 instructions in all
constexpr int S2_VOX_BYTES = S2_VTOTAL * 1024;                       // 15 360
constexpr int S2_LDS_FUSED = S2_HALO_BYTES + S2_RING_FUSED * S2_SLAB + S2_VOX_BYTES;    // 80 896: two workgroups per CU
struct S2Head {              // FUSED: the encoder's input is relu(conv5x5(x8, hw) + hb), computed per patch into the LDS halo
    const uint16_t* x8;      // NHWC bf16, 8 channels (event bins zero-padded), pixel stride x8_stride elements
    long long x8_stride;
    const uint16_t* hw;      // packed head weight [128][256] (rows 0..31 used, k = tap * 8 + channel)
    const float* hb;         // [32] or null
    int relu;
    // ev != null: the 8-channel input is formed on the fly from the fp32 NCHW event tensor (EventPreprocessor apply + NHWC8
    // bf16 re-layout of oess_event_slice_to_nhwc8_bf16, same arithmetic): x8 is not read
    const float* ev;         // [B][Ctot][H][W] fp32
    int Ctot, c0, Cs, normalize;
    const double* stats;     // {sum, sumsq, nnz} of the slice
};
constexpr int S2_IMG_PITCH = 72;                                     // bf16 output image [128 pixels][64 + 8]

template <typename F, int... Is>
__device__ __forceinline__ void s2_for_taps(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }

// one 8 x 16-pixel x 64-channel tile (`bid` = tile index after the XCD remap)
template <bool FUSED>
__device__ __forceinline__ void conv5x5s2_halo_tile(const ConvArgs& a, const S2Head& hd, const int bid, unsigned char* smem) {
    constexpr int S2_RING = FUSED ? S2_RING_FUSED : S2_RING_PLAIN;
    const int tile_n = bid % a.tiles_n;
    int patch = bid / a.tiles_n;
    const int tiles_x = (a.Wo + S2_PW - 1) / S2_PW, tiles_y = (a.Ho + S2_PH - 1) / S2_PH;
    const int ox0 = (patch % tiles_x) * S2_PW; patch /= tiles_x;
    const int oy0 = (patch % tiles_y) * S2_PH;
    const int b = patch / tiles_y;
    const int n0 = tile_n * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                         // wave tile: patch rows 4 wm .. 4 wm + 3 (64 pixels) x channels 32 wn .. + 31
    const int nchunks = a.Cin >> 5;
    const int NS = nchunks * 25;                                     // weight slabs of this tile

    const long long in_bytes = FUSED ? (((long long)a.B * a.H * a.W - 1) * hd.x8_stride + 8) * 2
                                     : (((long long)a.B * a.H * a.W - 1) * a.in_pix_stride + a.Cin) * 2;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(FUSED ? (void*)hd.x8 : (void*)a.in, 0, (FUSED && !hd.x8) ? 0 : (int)in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7ffffff0, 0x00020000);

    // ---- halo DMA geometry: lane (block lb = lane >> 2, slot = lane & 3) of instruction i writes block u = (wave*11 + i)*16 + lb
    unsigned hoff[FUSED ? 1 : S2_HINSTR];
#pragma unroll
    for (int i = 0; i < (FUSED ? 0 : S2_HINSTR); ++i) {
        const int u = (wave * S2_HINSTR + i) * 16 + (lane >> 2), slot = lane & 3;
        const int p = u / (S2_HH * S2_XW), rem = u - p * (S2_HH * S2_XW);
        const int hy = rem / S2_XW, xi = rem - hy * S2_XW;
        const int hx = 2 * xi + p;
        const int f = ((xi >> 2) & 1) | (((hy >> 1) & 1) << 1);
        const int iy = 2 * oy0 - 2 + hy, ix = 2 * ox0 - 2 + hx;
        const bool ok = u < S2_BLOCKS && hx < 2 * S2_PW + 3 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        hoff[i] = ok ? (unsigned)((((long long)b * a.H + iy) * a.W + ix) * a.in_pix_stride * 2 + ((slot ^ f) << 4)) : 0x80000000u;
    }
    // ---- weight DMA geometry: this wave writes rows wave*16 + (lane >> 2) of every slab
    unsigned boff;
    {
        const int row = wave * 16 + (lane >> 2), slot = lane & 3;
        boff = (unsigned)(((n0 + row) * a.Kpad + ((slot ^ ((row >> 2) & 3)) << 3)) * 2);
    }
    auto issue_halo = [&](int cc) {
#pragma unroll
        for (int i = 0; i < (FUSED ? 0 : S2_HINSTR); ++i) {
            const unsigned voff = hoff[i] == 0x80000000u ? 0x80000000u : hoff[i] + (unsigned)(cc * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(smem + (wave * S2_HINSTR + i) * 1024),
                                                     16, voff, 0, 0, 0);
        }
    };
    // slab s = (chunk, tap) -> ring buffer `buf`; slabs beyond the tile are dummy (out-of-range source = zero fill) so that
    // the number of DMA instructions in flight is the same at every tap
    auto issue_w = [&](int s, int buf) {
        const int cc = s / 25, tap = s - cc * 25;
        const unsigned voff = s < NS ? boff + (unsigned)((tap * a.Cin + cc * 32) * 2) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(smem + S2_HALO_BYTES + buf * S2_SLAB + wave * 1024),
                                                 16, voff, 0, 0, 0);
    };

    // accumulators: acc[i] = channels (rows) x pixels (columns) of M-tile i -- operands swapped, so that a lane holds four
    // CONSECUTIVE channels of one pixel and the epilogue writes 8-byte pieces
    f32x16_t acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;

    // ---- fragment addresses.  slot << 4 = (ks << 5) ^ ((half ^ f) << 4): the lane part ((half ^ f) << 4) is folded into one base
    //      register per (M-tile, s >> 1, parity of r >> 1) -- 12 registers -- the tap's block offset is an instruction immediate
    //      and k-step 1 flips bit 5 (bases are multiples of 64 bytes + the slot bits, so the XOR never carries)
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int r31 = lane & 31, half = lane >> 5;
    const int a_px = (r31 & 7) + 8 * (r31 >> 4);
    uint32_t va[2][3][2], vb;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int py = 4 * wm + 2 * i + ((r31 >> 3) & 1);
#pragma unroll
        for (int sv = 0; sv < 3; ++sv)
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                const int f = (((a_px + sv) >> 2) & 1) | (((py + rp) & 1) << 1);
                va[i][sv][rp] = lds0 + (uint32_t)((2 * py * S2_XW + a_px) * 64 + ((half ^ f) << 4));
            }
    }
    {
        const int n = wn * 32 + r31;
        vb = lds0 + (uint32_t)(S2_HALO_BYTES + n * 64 + ((half ^ ((n >> 2) & 3)) << 4));
    }

#define S2_READ(FA, FB, R_, S_, KS_, BUF_)                                                                              \
    {                                                                                                                   \
        constexpr int blk_ = ((((S_) & 1) * S2_HH + (R_)) * S2_XW + ((S_) >> 1)) * 64;                                   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                   \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(FA[i]) : "v"(va[i][(S_) >> 1][((R_) >> 1) & 1] ^ (uint32_t)((KS_) << 5)), "n"(blk_) : "memory"); \
        asm volatile("ds_read_b128 %0, %1" : "=v"(FB) : "v"((vb ^ (uint32_t)((KS_) << 5)) + (uint32_t)((BUF_) * S2_SLAB)) : "memory");  \
    }
#define S2_MMA(FA, FB)                                                                                                  \
    {                                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                   \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB, FA[i], acc[i], 0, 0, 0);                               \
    }
#define S2_WAIT(N_, FA, FB) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(FA[0]), "+v"(FA[1]), "+v"(FB) : "n"(N_) : "memory");

    if constexpr (FUSED) {
        // ---- E2VID head in front of the encoder: relu(conv5x5(x8) + hb) for the 19 x 35 halo pixels, written straight into the
        //      halo planes (zeros outside the image = the encoder's own padding).  Head weights: 13 k-steps of two taps x 8
        //      channels, kept in registers (MFMA A operand: rows = 32 head channels); pixels are the B operand, read from the
        //      23 x 39 voxel patch in LDS; a lane ends up with four consecutive channels of one halo pixel -> 8-byte LDS writes.
        unsigned char* vox = smem + S2_HALO_BYTES + S2_RING_FUSED * S2_SLAB;
        bf16x8_t wf[13];
        {
            const int p32 = lane & 31, hi = lane >> 5;
#pragma unroll
            for (int ks = 0; ks < 13; ++ks) wf[ks] = *reinterpret_cast<const bf16x8_t*>(hd.hw + (size_t)p32 * 256 + (ks * 2 + hi) * 8);
        }
        if (hd.ev) {
            // voxel patch straight from the fp32 event tensor: the slice's EventPreprocessor normalisation (inference_utils.py:80-85,
            // the float32 operation order of norm_to_nhwc8_kernel) and the 8-channel bf16 packing happen here, per patch
#pragma unroll
            for (int s = 0; s < S2_RING - 1; ++s) issue_w(s, s);
            const double nnz = hd.stats ? hd.stats[2] : 0.0;
            const bool active = hd.normalize && nnz > 0.0;
            float mean = 0.f, stdv = 1.f;
            if (active) {
                const float nf = (float)nnz;
                mean = (float)hd.stats[0] / nf;
                stdv = sqrtf(__fsub_rn((float)hd.stats[1] / nf, __fmul_rn(mean, mean)));
            }
            const long long hw_ = (long long)a.H * a.W;
            float raw[4][5];
            bool okv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = tid + 256 * i;
                const int vy = v / S2_VW, vx = v - vy * S2_VW;
                const int iy = 2 * oy0 - 4 + vy, ix = 2 * ox0 - 4 + vx;
                okv[i] = v < S2_VH * S2_VW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const float* src = hd.ev + ((long long)b * hd.Ctot + hd.c0) * hw_ + (long long)iy * a.W + ix;
#pragma unroll
                for (int c = 0; c < 5; ++c) raw[i][c] = (okv[i] && c < hd.Cs) ? src[c * hw_] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = tid + 256 * i;
                float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    float q = raw[i][c];
                    if (active) q = __fmul_rn((q != 0.0f) ? 1.0f : 0.0f, __fsub_rn(q, mean)) / stdv;
                    f[c] = (okv[i] && c < hd.Cs) ? q : 0.0f;
                }
                const u32x4_t o = {pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
                if (v < S2_VTOTAL * 64)
                    asm volatile("ds_write_b128 %0, %1" :: "v"((uint32_t)(uintptr_t)vox + (uint32_t)(v * 16)), "v"(o) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int i = 0; i < S2_VINSTR; ++i) {
                const int v = (wave * S2_VINSTR + i) * 64 + lane;
                const int vy = v / S2_VW, vx = v - vy * S2_VW;
                const int iy = 2 * oy0 - 4 + vy, ix = 2 * ox0 - 4 + vx;
                const bool ok = v < S2_VH * S2_VW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned voff = ok ? (unsigned)((((long long)b * a.H + iy) * a.W + ix) * hd.x8_stride * 2) : 0x80000000u;
                if (wave * S2_VINSTR + i < S2_VTOTAL)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vox + (wave * S2_VINSTR + i) * 1024), 16, voff, 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < S2_RING - 1; ++s) issue_w(s, s);
        }
        float bq[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) bq[q][k] = hd.hb ? hd.hb[8 * q + 4 * (lane >> 5) + k] : 0.0f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // voxel patch (and the first weight slabs) are in LDS
        const uint32_t vox0 = (uint32_t)(uintptr_t)vox, pl0 = (uint32_t)(uintptr_t)smem;
        const int hi = lane >> 5;
        // two M-tiles (64 halo pixels) per pass: two independent accumulators, two reads in flight under two MFMAs
        static_assert(((S2_BLOCKS + 31) / 32) % 2 == 0, "M-tiles are taken in pairs");
        for (int pr = wave; pr < (S2_BLOCKS + 31) / 64; pr += 4) {
            int uu_[2], hy_[2], xi_[2];
            bool inside_[2];
            uint32_t vbase[2];
#pragma unroll
            for (int z = 0; z < 2; ++z) {
                const int u = (2 * pr + z) * 32 + (lane & 31);
                uu_[z] = u;
                const int uu = u < S2_BLOCKS ? u : 0;
                const int p = uu / (S2_HH * S2_XW), rem = uu - p * (S2_HH * S2_XW);
                const int hy = rem / S2_XW, xi = rem - hy * S2_XW;
                int hx = 2 * xi + p;
                inside_[z] = u < S2_BLOCKS && hx < 2 * S2_PW + 3 && (unsigned)(2 * oy0 - 2 + hy) < (unsigned)a.H &&
                             (unsigned)(2 * ox0 - 2 + hx) < (unsigned)a.W;
                hx = hx < 2 * S2_PW + 3 ? hx : 0;
                hy_[z] = hy; xi_[z] = xi;
                vbase[z] = vox0 + (uint32_t)((hy * S2_VW + hx) * 16);
            }
            f32x16_t hacc[2];
#pragma unroll
            for (int z = 0; z < 2; ++z)
#pragma unroll
                for (int e = 0; e < 16; ++e) hacc[z][e] = 0.0f;
            bf16x8_t pa[2], pb[2];
            // k-step ks: this half wave's tap = 2 ks + hi (tap 25 carries zero weights: any address)
#define S2_HREAD(DST, KS_)                                                                                             \
            {                                                                                                         \
                constexpr int t0_ = 2 * (KS_), t1_ = 2 * (KS_) + 1 < 25 ? 2 * (KS_) + 1 : 0;                            \
                const uint32_t off_ = hi ? (uint32_t)(((t1_ / 5) * S2_VW + t1_ % 5) * 16) : (uint32_t)(((t0_ / 5) * S2_VW + t0_ % 5) * 16); \
                asm volatile("ds_read_b128 %0, %1" : "=v"(DST[0]) : "v"(vbase[0] + off_) : "memory");                  \
                asm volatile("ds_read_b128 %0, %1" : "=v"(DST[1]) : "v"(vbase[1] + off_) : "memory");                  \
            }
            S2_HREAD(pa, 0)
            s2_for_taps([&](auto kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (ks + 1 < 13) {
                    if constexpr (ks & 1) { S2_HREAD(pa, ks + 1) } else { S2_HREAD(pb, ks + 1) }
                    if constexpr (ks & 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(pb[0]), "+v"(pb[1]) :: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(pa[0]), "+v"(pa[1]) :: "memory");
                } else {
                    if constexpr (ks & 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pb[0]), "+v"(pb[1]) :: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pa[0]), "+v"(pa[1]) :: "memory");
                }
#pragma unroll
                for (int z = 0; z < 2; ++z)
                    hacc[z] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], (ks & 1) ? pb[z] : pa[z], hacc[z], 0, 0, 0);
            }, std::make_integer_sequence<int, 13>{});
#undef S2_HREAD
#pragma unroll
            for (int z = 0; z < 2; ++z) {
                if (uu_[z] >= S2_BLOCKS) continue;
                const uint32_t f = (uint32_t)(((xi_[z] >> 2) & 1) | (((hy_[z] >> 1) & 1) << 1));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = hacc[z][q * 4 + k] + bq[q][k];
                        if (hd.relu) v[k] = fmaxf(v[k], 0.0f);
                        v[k] = inside_[z] ? v[k] : 0.0f;
                    }
                    const uint2 o = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    asm volatile("ds_write_b64 %0, %1" :: "v"(pl0 + (uint32_t)(uu_[z] * 64) + (((uint32_t)q ^ f) << 4) + (uint32_t)(8 * hi)), "v"(o) : "memory");
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
#pragma unroll
        for (int s = 0; s < S2_RING - 1; ++s) issue_w(s, s);
    }
    int sg = 0, buf = 0;                                             // global slab index and its ring buffer
    for (int cc = 0; cc < nchunks; ++cc) {
        if (cc > 0) __builtin_amdgcn_s_barrier();                    // every wave is done with the previous chunk's halo
        issue_halo(cc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // halo and every slab issued so far are in LDS
        bf16x8_t fa0[2], fb0, fa1[2], fb1;
        S2_READ(fa0, fb0, 0, 0, 0, buf)
        // the 25 taps, unrolled with compile-time (r, s): tap offsets are instruction immediates
        s2_for_taps([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value, r = t / 5, s = t - r * 5;
            if (t > 0) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(S2_RING - 3) : "memory");     // slab sg + 1 has landed (RING - 3 younger slabs may be in flight)
                __builtin_amdgcn_s_barrier();                        // ... for every wave; the buffer of slab sg - 1 is free
            }
            const int nb = buf + 1 == S2_RING ? 0 : buf + 1;         // buffer of slab sg + 1
            issue_w(sg + S2_RING - 1, buf == 0 ? S2_RING - 1 : buf - 1);
            __builtin_amdgcn_s_setprio(3);
            S2_READ(fa1, fb1, r, s, 1, buf)
            S2_WAIT(3, fa0, fb0)
            S2_MMA(fa0, fb0)
            if constexpr (t < 24) {
                constexpr int r2 = (t + 1) / 5, s2 = (t + 1) - r2 * 5;
                S2_READ(fa0, fb0, r2, s2, 0, nb)
                S2_WAIT(3, fa1, fb1)
            } else {
                S2_WAIT(0, fa1, fb1)
            }
            S2_MMA(fa1, fb1)
            __builtin_amdgcn_s_setprio(0);
            ++sg; buf = nb;
        }, std::make_integer_sequence<int, 25>{});
    }
#undef S2_READ
#undef S2_MMA
#undef S2_WAIT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // dummy tail slabs have been written (ring region only)
    __syncthreads();

    // ---- epilogue: bias + activation -> bf16 image [128 pixels][64 channels] in LDS (8-byte pieces: a lane holds channels
    //      32 wn + 8 q + 4 half + {0..3} of pixel r31 of each M-tile) -> 16-byte row stores
    uint16_t* img = reinterpret_cast<uint16_t*>(smem);
    {
        float bq[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) bq[q][k] = a.bias ? a.bias[n0 + wn * 32 + 8 * q + 4 * half + k] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int py = 4 * wm + 2 * i + ((r31 >> 3) & 1);
            uint16_t* dst = img + (py * S2_PW + a_px) * S2_IMG_PITCH + wn * 32 + 4 * half;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = acc[i][q * 4 + k] + bq[q][k];
                    if (a.relu) v[k] = fmaxf(v[k], 0.0f);
                }
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(dst + 8 * q) = o;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = tid + k * 256, pl = idx >> 3, c = idx & 7;
        const int oy = oy0 + (pl >> 4), ox = ox0 + (pl & 15);
        if (oy < a.Ho && ox < a.Wo)
            out_store16(a.out + (((long long)b * a.Ho + oy) * a.Wo + ox) * a.out_pix_stride + n0 + c * 8,
                        *reinterpret_cast<const uint4*>(img + pl * S2_IMG_PITCH + c * 8));
    }
}

template <bool FUSED>
__global__ __launch_bounds__(256, 2) void conv5x5s2_halo_kernel(ConvArgs a, S2Head hd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    conv5x5s2_halo_tile<FUSED>(a, hd, bid, smem);
}

// Two independent problems of the plain kernel in one launch (encoder convs of levels 1 and 2 on the skewed schedule of the
// recurrent encoder: 2 240 + 1 120 workgroups = 4.4 + 2.2 rounds over the 512 slots alone, 6.6 together); mapping as in
// conv3x3_halo_group_kernel, longest K first.
struct S2Group {
    ConvArgs a[2];
    int start8[3];
};
__global__ __launch_bounds__(256, 2) void conv5x5s2_halo_group_kernel(S2Group g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int p = idx >= g.start8[1] ? 1 : 0;
    const int li = idx - g.start8[p];
    const ConvArgs& a = g.a[p];
    const int nwg = a.tiles_m * a.tiles_n;
    const int q = nwg >> 3, r = nwg & 7;
    if (li >= q + (xcd < r ? 1 : 0)) return;
    conv5x5s2_halo_tile<false>(a, S2Head{}, (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + li, smem);
}

// =================================================================================================
// v4: BK = 32 slabs in a 4-deep LDS ring (same 64 KB per workgroup, still 2 workgroups per CU).
// The 2-stage BK = 64 kernel has ONE slab in flight per workgroup and must hide the whole L2 -> LDS round trip
// (~1.0-1.3 us under load) behind one slab of MFMAs (0.54 us when two workgroups share the CU): it is latency bound
// (tools/conv_ablate.py).  Here three 32-wide slabs are in flight (48 KB instead of 32 KB per workgroup) and a slab
// is issued three compute phases before it is needed.  64-byte LDS rows: chunk c of row r lives at
// c ^ ((r >> 2) & 3) (conflict free for the ds_read_b128 lane groups), one wave DMA instruction covers 16 rows.
// =================================================================================================
template <int BN, bool FASTK, int EPI = 0, int NST = 4>
__global__ __launch_bounds__(256) void conv_fwd_dma32_kernel(ConvArgs a) {
    constexpr int BMX = 128, BKS = 32, NWAVES = 4;
    constexpr int WAVES_N = (BN == 128) ? 2 : 1;
    constexpr int WAVES_M = NWAVES / WAVES_N;
    constexpr int WM = BMX / WAVES_M;
    constexpr int WN = BN / WAVES_N;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int A_INSTR = BMX / 16 / NWAVES;        // 2: 16 rows x 64 B per wave instruction
    constexpr int B_INSTR = BN / 16 / NWAVES;         // 2 (BN = 128) or 1 (BN = 64)
    constexpr int IPS = A_INSTR + B_INSTR;
    constexpr int A_BYTES = BMX * 64;
    constexpr int STAGE_BYTES = (BMX + BN) * 64;
    constexpr int NFRAG = MT + NT;
    static_assert(BN == 128 || BN == 64, "BN");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
    const int m0 = tile_m * BMX, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int KT = a.Kpad / BKS;
    const int cpt = a.Cin >> 3, ntaps = a.R * a.S;

    const long long in_bytes = (((long long)a.B * a.H * a.W - 1) * a.in_pix_stride + a.Cin) * 2;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7ffffff0, 0x00020000);

    const int lrow = lane >> 2, slot = lane & 3;       // 16 rows x 4 chunk slots per wave instruction
    int iy0[A_INSTR], ix0[A_INSTR], rowoff[A_INSTR], csrc[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int r = (wave * A_INSTR + i) * 16 + lrow;
        const int m = m0 + r;
        const bool valid = m < a.M;
        const int mm = valid ? m : 0;
        const int hw = a.Ho * a.Wo;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        iy0[i] = valid ? oy * a.stride - a.pad : -0x4000;
        ix0[i] = ox * a.stride - a.pad;
        rowoff[i] = (int)((((long long)b * a.H + (oy * a.stride - a.pad)) * a.W + ix0[i]) * a.in_pix_stride * 2);
        csrc[i] = slot ^ ((r >> 2) & 3);
    }
    int boff[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int r = (wave * B_INSTR + i) * 16 + lrow;
        boff[i] = ((n0 + r) * a.Kpad + (slot ^ ((r >> 2) & 3)) * 8) * 2;
    }

    auto issue = [&](int kt) {
        unsigned char* st = smem + (kt % NST) * STAGE_BYTES;
        if constexpr (FASTK) {
            const unsigned kc0 = (unsigned)(kt * 4);
            const unsigned tap = (kc0 * a.inv_cpt) >> 20;
            const int cc0 = (int)(kc0 - tap * cpt);
            const unsigned r = (tap * a.inv_s) >> 16;
            const int sx = (int)(tap - r * a.S);
            const int dy = (int)r * a.dil, dx = sx * a.dil;
            const int tapoff = ((dy * a.W + dx) * (int)a.in_pix_stride + cc0 * 8) * 2;
            const bool tap_ok = (int)tap < ntaps;
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i) {
                const bool ok = tap_ok && (unsigned)(iy0[i] + dy) < (unsigned)a.H && (unsigned)(ix0[i] + dx) < (unsigned)a.W;
                const unsigned voff = ok ? (unsigned)(rowoff[i] + csrc[i] * 16 + tapoff) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave * A_INSTR + i) * 1024),
                                                         16, voff, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i) {
                const unsigned kc = (unsigned)(kt * 4 + csrc[i]);
                const unsigned tap = (kc * a.inv_cpt) >> 20;
                const int cc = (int)(kc - tap * cpt);
                const unsigned r = (tap * a.inv_s) >> 16;
                const int sx = (int)(tap - r * a.S);
                const int dy = (int)r * a.dil, dx = sx * a.dil;
                const bool ok = (int)tap < ntaps && (unsigned)(iy0[i] + dy) < (unsigned)a.H && (unsigned)(ix0[i] + dx) < (unsigned)a.W;
                const unsigned voff = ok ? (unsigned)(rowoff[i] + ((dy * a.W + dx) * (int)a.in_pix_stride + cc * 8) * 2) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave * A_INSTR + i) * 1024),
                                                         16, voff, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(st + A_BYTES + (wave * B_INSTR + i) * 1024),
                                                     16, (unsigned)(boff[i] + kt * BKS * 2), 0, 0, 0);
    };

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    uint32_t fa_off[MT], fb_off[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) fa_off[i] = (uint32_t)(wm * WM + i * 32 + (lane & 31)) * 64;
#pragma unroll
    for (int j = 0; j < NT; ++j) fb_off[j] = (uint32_t)(A_BYTES + (wn * WN + j * 32 + (lane & 31)) * 64);
    const int half = lane >> 5;
    const int rsw = ((lane & 31) >> 2) & 3;
    const uint32_t sl0 = (uint32_t)(((0 + half) ^ rsw) * 16), sl1 = (uint32_t)(((2 + half) ^ rsw) * 16);

#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < KT) issue(s);

#define OESS_FR32(DST_A, DST_B, SL)                                                                              \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                           \
            asm volatile("ds_read_b128 %0, %1" : "=v"(DST_A[i]) : "v"(stage_ + fa_off[i] + SL) : "memory");     \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                           \
            asm volatile("ds_read_b128 %0, %1" : "=v"(DST_B[j]) : "v"(stage_ + fb_off[j] + SL) : "memory");     \
    }
#define OESS_MMA32(SRC_A, SRC_B)                                                                                 \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                           \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                       \
                acc[i][j] = (EPI == 1) ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(SRC_B[j], SRC_A[i], acc[i][j], 0, 0, 0)     \
                                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(SRC_A[i], SRC_B[j], acc[i][j], 0, 0, 0);   \
    }
#define OESS_WAIT32(N_, FA_, FB_)                                                                                \
    {                                                                                                            \
        if constexpr (MT == 2 && NT == 2)                                                                        \
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(FA_[0]), "+v"(FA_[1]), "+v"(FB_[0]), "+v"(FB_[1]) : "n"(N_) : "memory"); \
        else                                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(FA_[0]), "+v"(FB_[0]), "+v"(FB_[1]) : "n"(N_) : "memory"); \
    }

    for (int kt = 0; kt < KT; ++kt) {
        // retire slab kt; up to two younger slabs stay in flight across the barrier
        if (NST == 4 && kt + 2 < KT) {
            if constexpr (IPS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else if (kt + 1 < KT) {
            if constexpr (IPS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                   // slab kt visible to every wave; the stage of slab kt-1 is free
        if (kt + NST - 1 < KT) issue(kt + NST - 1);
        const uint32_t stage_ = lds0 + (uint32_t)((kt % NST) * STAGE_BYTES);
        bf16x8_t fa0[MT], fb0[NT], fa1[MT], fb1[NT];
        OESS_FR32(fa0, fb0, sl0)
        OESS_FR32(fa1, fb1, sl1)
        OESS_WAIT32(NFRAG, fa0, fb0)
        OESS_MMA32(fa0, fb0)
        OESS_WAIT32(0, fa1, fb1)
        OESS_MMA32(fa1, fb1)
    }
#undef OESS_FR32
#undef OESS_MMA32
#undef OESS_WAIT32
    __syncthreads();

    if constexpr (EPI == 1) lstm_epilogue(a, acc, smem, m0, n0, wm, wn, lane, tid);
    else conv_epilogue<BMX, BN>(a, acc, smem, m0, n0, wm, wn, lane, tid);
}


// =================================================================================================
// Small-Cin convolution (Cin == 8: one 16-byte chunk per pixel, e.g. the E2VID head on the 5-bin voxel grid padded
// to 8 channels; stride 1, dilation 1, R*S <= 32 taps, Cout <= 32).
// As an implicit GEMM this layer is an im2col blow-up: every output pixel pulls R*S x 16 B through L2 -> LDS (400 B for
// 16 B of unique input at 5x5), and the generic kernel runs at the L2 gather rate.  Here a workgroup stages the input
// HALO tile ((8+R-1) x (64+S-1) pixels x 16 B, ~13 KB) in LDS once and the MFMA B-operand (pixels) is read straight
// out of it: lane p of k-step ks reads the chunk of pixel (ty + r, tx + s), (r, s) = tap ks*2 + (lane>>5).  Consecutive
// lanes read consecutive 16-byte chunks: conflict free.  The whole packed weight (32 x 256 bf16) lives in 64 VGPRs.
// The product is computed transposed (weights = MFMA A operand), so a lane owns 4 consecutive output channels of one
// pixel per register quad and stores them as 8-byte pieces.
// =================================================================================================
template <int R, int S>
__global__ __launch_bounds__(256, 2) void conv_smallcin_kernel(ConvArgs a) {
    // PERSISTENT over tiles: the weights are fetched once per workgroup, and the halo of the NEXT tile travels
    // HBM -> registers while the current tile is multiplied and stored, so a tile costs its LDS / MFMA / store work and
    // not a load round trip on top (one tile per workgroup measured 111 us for 180 MB of traffic: 3x the HBM time).
    constexpr int TH = 8, TW = 64, HW_ = TW + S - 1, HH_ = TH + R - 1, NTAP = R * S;
    constexpr int KS = (NTAP + 1) / 2;                       // k-steps of 16 = 2 taps; taps >= NTAP carry zero weights
    constexpr int OP = 36;                                   // output image pitch in elements (32 ch + 4: conflict-free 8-byte writes)
    constexpr int HALO_N = HH_ * HW_, HALO_BYTES = HALO_N * 16, IMG_BYTES = TH * TW * OP * 2;
    constexpr int HPT = (HALO_N + 255) / 256;                // halo chunks per thread
    __shared__ __attribute__((aligned(16))) unsigned char sm[HALO_BYTES + IMG_BYTES];
    u32x4_t* halo = reinterpret_cast<u32x4_t*>(sm);
    uint16_t* img = reinterpret_cast<uint16_t*>(sm + HALO_BYTES);
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    const int ntiles = a.B * tiles_y * tiles_x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, hi = lane >> 5;

    // weights: lane (n = p, k-half = hi) keeps its fragments for the whole kernel
    bf16x8_t wf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        wf[ks] = *reinterpret_cast<const bf16x8_t*>(a.w + (size_t)p * a.Kpad + (ks * 2 + hi) * 8);

    const u32x4_t zero4 = {0u, 0u, 0u, 0u};
    // halo of tile t -> registers (zero outside the image / beyond the tile list)
    auto fetch = [&](int t, u32x4_t (&h)[HPT]) {
        int bid = t;
        const int tx0 = (bid % tiles_x) * TW; bid /= tiles_x;
        const int ty0 = (bid % tiles_y) * TH;
        const int b = bid / tiles_y;
#pragma unroll
        for (int k = 0; k < HPT; ++k) {
            const int i = tid + k * 256;
            const int hy = i / HW_, hx = i - hy * HW_;
            const int iy = ty0 - a.pad + hy, ix = tx0 - a.pad + hx;
            h[k] = zero4;
            if (t < ntiles && i < HALO_N && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                h[k] = *reinterpret_cast<const u32x4_t*>(a.in + (((long long)b * a.H + iy) * a.W + ix) * a.in_pix_stride);
        }
    };
    const int cchunks = (a.Cout + 7) >> 3;
    const bool al16 = (((uintptr_t)a.out) & 15) == 0 && (a.out_pix_stride & 7) == 0;
    float bq[4][4];                                          // this lane's 16 output-channel biases
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = 8 * q + 4 * hi + k;
            bq[q][k] = (a.bias && ch < a.Cout) ? a.bias[ch] : 0.0f;
        }

    u32x4_t hreg[HPT];
    fetch((int)blockIdx.x, hreg);
    auto park = [&]() {
#pragma unroll
        for (int k = 0; k < HPT; ++k) {
            const int i = tid + k * 256;
            if (i < HALO_N) halo[i] = hreg[k];
        }
    };
    park();
    for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x) {
        int bid = t;
        const int tx0 = (bid % tiles_x) * TW; bid /= tiles_x;
        const int ty0 = (bid % tiles_y) * TH;
        const int b = bid / tiles_y;
        // LDS-only barriers: __syncthreads() would also drain vmcnt, i.e. wait for the previous tile's stores to be acknowledged
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // halo complete; the previous tile's image reads are done too
        fetch(t + (int)gridDim.x, hreg);                     // next tile's halo, in flight under the MFMA phase

        // wave w: output rows 2w, 2w+1 of the tile, two 32-pixel halves each
        f32x16_t acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[u][e] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            int tap = ks * 2 + hi;
            tap = tap < NTAP ? tap : 0;                      // zero weights there: any in-range address will do
            const int r = tap / S, sx = tap - r * S;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ty = wave * 2 + (u >> 1), tx = (u & 1) * 32 + p;
                const u32x4_t q = halo[(ty + r) * HW_ + tx + sx];
                bf16x8_t af;
                __builtin_memcpy(&af, &q, 16);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], af, acc[u], 0, 0, 0);
            }
        }
        // epilogue: lane (pixel p of the m-tile, hi) holds channels (e&3) + 8*(e>>2) + 4*hi -> 8-byte pieces into the image
        // the activation mode is wave-uniform: branch ONCE around the whole block (per-value mode tests inlined erff 64 times)
        auto to_image = [&](auto act) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pl = (wave * 2 + (u >> 1)) * TW + (u & 1) * 32 + p;      // pixel index inside the tile
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = 8 * q + 4 * hi;
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = act(acc[u][q * 4 + k] + bq[q][k]);
                    uint2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(img + pl * OP + ch) = o;
                }
            }
        };
        if (a.relu == 1) to_image([](float x) { return fmaxf(x, 0.0f); });
        else if (a.relu == 2) to_image([](float x) { return conv_act(x, 2); });
        else to_image([](float x) { return x; });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // image complete (and every halo read of this tile retired)
        // Park the next halo BEFORE this tile's stores: vmcnt is one in-order counter for loads and stores, so waiting for
        // the halo registers after the stores would wait for their acknowledgement too (a store round trip per tile).
        park();
        // coalesced stores: 4 lanes x 16 bytes per pixel row (Cout <= 32), consecutive lanes = consecutive pixels
#pragma unroll 2
        for (int i = tid; i < TH * TW * 4; i += 256) {
            const int pl = i >> 2, cc = i & 3;
            if (cc >= cchunks) continue;
            const int oy = ty0 + pl / TW, ox = tx0 + (pl % TW);
            if (oy >= a.Ho || ox >= a.Wo) continue;
            uint16_t* dst = a.out + (((long long)b * a.Ho + oy) * a.Wo + ox) * a.out_pix_stride + cc * 8;
            const uint2 lo = *reinterpret_cast<const uint2*>(img + pl * OP + cc * 8);
            const uint2 hi2 = *reinterpret_cast<const uint2*>(img + pl * OP + cc * 8 + 4);
            if (cc * 8 + 8 <= a.Cout && al16) {
                *reinterpret_cast<uint4*>(dst) = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
            } else {                                         // ragged channel tail (Cout % 8 == 4) or 8-byte aligned rows
                *reinterpret_cast<uint2*>(dst) = lo;
                if (cc * 8 + 4 < a.Cout) *reinterpret_cast<uint2*>(dst + 4) = hi2;
            }
        }
        // no barrier here: the image is only rewritten after the next iteration's first barrier
    }
}

// ---- weight packing: OIHW fp32 (PyTorch Conv2d.weight) -> Wp[Npad][Kpad] bf16, k = (r, s, ci)
// flip != 0 produces the data-gradient operator: Wp[ci][(R-1-r, S-1-s), co] (rotated, in/out swapped).
__global__ void pack_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp, int Cout, int Cin, int R,
                                   int S, int Cin_pad, int Kpad, int Npad, int flip) {
    const long long total = (long long)Npad * Kpad;
    const int Nlog = flip == 1 ? Cin : Cout;      // logical output channels of the packed operator
    const int Klog_c = flip == 1 ? Cout : Cin;    // logical input channels
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / Kpad), k = (int)(i - (long long)n * Kpad);
        const int tap = k / Cin_pad, ci = k - tap * Cin_pad;
        float v = 0.0f;
        if (n < Nlog && tap < R * S && ci < Klog_c) {
            const int r = tap / S, s = tap - r * S;
            if (flip == 2) {                    // ConvLSTM gate interleave: packed row n = 4*hc + gate <- Conv2d row gate*C + hc
                const int src = (n & 3) * (Cout >> 2) + (n >> 2);
                v = w[(((long long)src * Cin + ci) * R + r) * S + s];
            } else if (!flip) v = w[(((long long)n * Cin + ci) * R + r) * S + s];
            else v = w[(((long long)ci * Cin + n) * R + (R - 1 - r)) * S + (S - 1 - s)];
        }
        wp[i] = f32_to_bf16(v);
    }
}

// Many weights in ONE launch (every trainable conv of a model after an optimiser step: ~125 separate 6 us launches per
// frame2recon step otherwise).  Two table-driven kernels; table[p] = eight 64-bit words in device memory, the last one the
// problem's first workgroup; a workgroup finds its problem by a binary search over that column (problems differ by 500x in
// size: a fixed number of workgroups per problem left the chip idle behind the two largest operands).
//   pack_fwd_multi_kernel : {w OIHW fp32, packed, Cout, Cin, R, S, -, first block}: a thread forms 8 consecutive k of one packed
//                           row (same output channel, same tap, 8 input channels) -> ONE 16-byte store and one index decode per
//                           8 elements (the single-weight kernel above decodes every element with a 64-bit division).
//   pack_flip_multi_kernel: {packed forward operand, packed data-gradient operand, Cout, Cin, R, S, -, first block}: the
//                           data-gradient operator Wf[ci][(R-1-r, S-1-s), co] is the forward operand W[co][(r, s), ci] with
//                           (co, ci) transposed per tap -- a 64 x 64 bf16 tile transpose through LDS, both sides in 128-byte
//                           rows (from the fp32 OIHW tensor every element would be a separate 32-byte sector).
// Both need Cout % 8 == 0 and Cin % 8 == 0 (no channel padding inside a row) and only write the valid region: the zero
// padding of the buffers was written by the first, single-weight packing and is never touched again.
constexpr int PACK_CHUNK = 256 * 8 * 4;               // packed elements per workgroup of the forward kernel
__device__ __forceinline__ const long long* pack_find(const long long* table, int n) {
    int lo = 0, hi = n;                                   // largest p with first_block[p] <= blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (table[(size_t)mid * 8 + 7] <= (long long)blockIdx.x) lo = mid; else hi = mid;
    }
    return table + (size_t)lo * 8;
}
__global__ __launch_bounds__(256) void pack_fwd_multi_kernel(const long long* __restrict__ table, int n) {
    const long long* d = pack_find(table, n);
    const float* w = reinterpret_cast<const float*>(d[0]);
    uint16_t* wp = reinterpret_cast<uint16_t*>(d[1]);
    const int Cout = (int)d[2], Cin = (int)d[3], R = (int)d[4], S = (int)d[5];
    const int RS = R * S;
    const int Kpad = (RS * Cin + BK - 1) / BK * BK;                    // Cin % 8 == 0: Cin_pad == Cin
    const int k8_per_row = RS * Cin / 8;                                // valid 8-element groups of a packed row
    const long long groups = (long long)Cout * k8_per_row;
    const long long g0 = ((long long)blockIdx.x - d[7]) * (PACK_CHUNK / 8);
#pragma unroll
    for (int u = 0; u < PACK_CHUNK / 8 / 256; ++u) {
        const long long gi = g0 + u * 256 + threadIdx.x;
        if (gi >= groups) break;
        const int nn = (int)(gi / k8_per_row), k8 = (int)(gi - (long long)nn * k8_per_row);
        const int k = k8 * 8, tap = k / Cin, ci = k - tap * Cin;
        const float* src = w + ((long long)nn * Cin + ci) * RS + tap;   // element (nn, ci + q, tap) at src[q * RS]
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = src[(long long)q * RS];
        *reinterpret_cast<uint4*>(wp + (long long)nn * Kpad + k) = pack_bf16x8(v);
    }
}
__global__ __launch_bounds__(256) void pack_flip_multi_kernel(const long long* __restrict__ table, int n) {
    __shared__ uint16_t tile[64][64 + 8];
    const long long* d = pack_find(table, n);
    const uint16_t* wf = reinterpret_cast<const uint16_t*>(d[0]);       // forward operand [co][tap * Cin + ci]
    uint16_t* wb = reinterpret_cast<uint16_t*>(d[1]);                   // data-gradient operand [ci][tap' * Cout + co]
    const int Cout = (int)d[2], Cin = (int)d[3], R = (int)d[4], S = (int)d[5];
    const int RS = R * S;
    const int KpadF = (RS * Cin + BK - 1) / BK * BK, KpadB = (RS * Cout + BK - 1) / BK * BK;
    const int tco = (Cout + 63) / 64, tci = (Cin + 63) / 64;
    int b = (int)((long long)blockIdx.x - d[7]);                        // (tap, co tile, ci tile)
    const int ci_t = b % tci; b /= tci;
    const int co_t = b % tco; const int tap = b / tco;
    const int co0 = co_t * 64, ci0 = ci_t * 64;
    // load: rows co0 .. co0+63, 64 ci each (8 chunks of 16 bytes): 512 chunks, two per thread
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = u * 256 + threadIdx.x, row = c >> 3, ch = c & 7;
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (co0 + row < Cout && ci0 + ch * 8 < Cin)
            q = *reinterpret_cast<const uint4*>(wf + (long long)(co0 + row) * KpadF + (long long)tap * Cin + ci0 + ch * 8);
        *reinterpret_cast<uint4*>(&tile[row][ch * 8]) = q;
    }
    __syncthreads();
    const int tapb = RS - 1 - tap;                                      // (R-1-r) * S + (S-1-s)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = u * 256 + threadIdx.x, row = c >> 3, ch = c & 7;  // row = ci within the tile, ch = group of 8 co
        if (ci0 + row < Cin && co0 + ch * 8 < Cout) {
            union { uint4 q; uint16_t h[8]; } o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.h[e] = tile[ch * 8 + e][row];
            *reinterpret_cast<uint4*>(wb + (long long)(ci0 + row) * KpadB + (long long)tapb * Cout + co0 + ch * 8) = o.q;
        }
    }
}

}  // namespace

extern "C" {

long long oess_conv2d_pack_multi_blocks(int Cout, int Cin, int R, int S, int flip_from_packed) {
    if (Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0 || (Cout & 7) || (Cin & 7)) return 0;
    if (flip_from_packed) return (long long)R * S * ((Cout + 63) / 64) * ((Cin + 63) / 64);
    const long long groups = (long long)Cout * (R * S * Cin / 8);
    return (groups + PACK_CHUNK / 8 - 1) / (PACK_CHUNK / 8);
}

int oess_conv2d_pack_weight_multi(const long long* table_dev, int n, long long total_blocks, int flip_from_packed,
                                  oess_stream_t stream) {
    if (!table_dev || n <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffll) return OESS_EINVAL;
    if (flip_from_packed)
        hipLaunchKernelGGL(pack_flip_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n);
    else
        hipLaunchKernelGGL(pack_fwd_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_conv2d_pack_weight(const float* w_oihw, int Cout, int Cin, int R, int S, int flip_for_dgrad, void* packed,
                            size_t packed_bytes, oess_stream_t stream) {
    if (!w_oihw || !packed || Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0) return OESS_EINVAL;
    if (flip_for_dgrad == 2 && (Cout & 3)) return OESS_EINVAL;
    const int n_log = flip_for_dgrad == 1 ? Cin : Cout, c_log = flip_for_dgrad == 1 ? Cout : Cin;
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
    const int n_log = flip_for_dgrad == 1 ? Cin : Cout, c_log = flip_for_dgrad == 1 ? Cout : Cin;
    const int cin_pad = (c_log + 7) / 8 * 8;
    const int kpad = (R * S * cin_pad + BK - 1) / BK * BK;
    const int npad = (n_log + 127) / 128 * 128;
    return (size_t)npad * kpad * 2;
}

}  // extern "C"

namespace {
struct LstmOut { const float* prev; float* cell; void* h; long long h_stride; int C; };

void conv_set_attrs() {
    static bool set_on[64] = {false};     // the attribute is per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    const bool attrs_set = dev >= 0 && dev < 64 && set_on[dev];
    if (!attrs_set) {       // > 64 KiB of dynamic LDS needs an explicit opt-in
        const void* fns[] = {(const void*)&conv_fwd_kernel<128>, (const void*)&conv_fwd_kernel<64>, (const void*)&conv_fwd_kernel<32>,
                             (const void*)&conv_fwd_dma_kernel<128, 128, 2, false>, (const void*)&conv_fwd_dma_kernel<128, 64, 2, false>, (const void*)&conv_fwd_dma_kernel<128, 32, 2, false>,
                             (const void*)&conv_fwd_dma_kernel<128, 128, 2, true>, (const void*)&conv_fwd_dma_kernel<128, 64, 2, true>, (const void*)&conv_fwd_dma_kernel<128, 32, 2, true>,
                             (const void*)&conv_fwd_dma_kernel<64, 128, 2, false>, (const void*)&conv_fwd_dma_kernel<64, 128, 2, true>,
                             (const void*)&conv_fwd_dma_kernel<128, 128, 2, false, 1>, (const void*)&conv_fwd_dma_kernel<128, 128, 2, true, 1>,
                             (const void*)&conv3x3_halo_kernel<0>, (const void*)&conv3x3_halo_kernel<1>, (const void*)&conv3x3_halo_group_kernel<1>,
                             (const void*)&conv3x3_halo256_group_kernel, (const void*)&conv3x3_lstm_w128_kernel, (const void*)&conv1x1_w128_kernel, (const void*)&conv3x3_w128_kernel,
                             (const void*)&conv_fwd_dma_kernel<256, 256, 2, true>,
                             (const void*)&conv_fwd_dma_kernel<128, 128, 4, true, 0, 512>,
                             (const void*)&conv_fwd_dma32_kernel<128, true, 0, 3>, (const void*)&conv5x5s2_halo_kernel<false>,
                             (const void*)&conv5x5s2_halo_kernel<true>, (const void*)&conv5x5s2_halo_group_kernel};
        for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 64) set_on[dev] = true;
    }
}

int conv_fwd_impl(const void* in, long long in_pix_stride, int B, int H, int W, int Cin, const void* w_packed,
                  const float* bias, int Cout, int R, int S, int stride, int pad, int dil, int relu,
                  const void* residual, long long res_pix_stride, void* out_bf16, float* out_f32,
                  long long out_pix_stride, float* tile_stats, const LstmOut* lstm, oess_stream_t stream,
                  void* workspace = nullptr, size_t workspace_bytes = 0, size_t* want_workspace = nullptr,
                  ConvArgs* capture = nullptr) {
    // capture != null (ConvLSTM only): fill *capture with the launch arguments of the row-halo kernel instead of launching it;
    // OESS_EINVAL when the geometry takes another kernel (the caller then launches the problems one by one)
    if (want_workspace) *want_workspace = 0;
    if (lstm) {
        if (!lstm->cell || !lstm->h || lstm->C <= 0 || (lstm->C & 31) || Cout != 4 * lstm->C || (lstm->h_stride & 1) ||
            lstm->h_stride < lstm->C || residual || relu || tile_stats || out_f32 || stride != 1)
            return OESS_EINVAL;
        out_bf16 = lstm->h;             // satisfies the generic output checks below; the conv epilogue is not used
        out_pix_stride = Cout;
    }
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
    a.stats = tile_stats;
    if (tile_stats && (bias || out_f32 || residual || relu)) return OESS_EINVAL;
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
    a.lstm_prev = lstm ? lstm->prev : nullptr; a.lstm_cell = lstm ? lstm->cell : nullptr;
    a.lstm_h = lstm ? (uint16_t*)lstm->h : nullptr; a.lstm_h_stride = lstm ? lstm->h_stride : 0; a.lstm_C = lstm ? lstm->C : 0;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.partial = nullptr; a.ksplit = 1; a.kt_per = 0;
    a.mg_w = (W >= 2 && 256ll * W < 0x100000000ll) ? (unsigned)(0x100000000ull / (unsigned)W) + 1u : 0u;
    a.mg_wd = (256ll * (W + dil) < 0x100000000ll) ? (unsigned)(0x100000000ull / (unsigned)(W + dil)) + 1u : 0u;
    hipStream_t st = (hipStream_t)stream;
    conv_set_attrs();
    // The LDS-DMA kernels address the input with 32-bit buffer offsets and decode filter taps with exact small-range
    // reciprocals (verified here over the whole range); anything outside takes the register-staged generic kernel.
    const long long in_extent = (((long long)B * H * W - 1) * in_pix_stride + Cin) * 2;
    bool dma_ok = in_extent < 0x7ffffff0ll;
    {
        const unsigned cpt = (unsigned)(Cin >> 3), nkc = (unsigned)(a.Kpad / 8);
        a.inv_cpt = ((1u << 20) + cpt - 1) / cpt;
        a.inv_s = ((1u << 16) + (unsigned)S - 1) / (unsigned)S;
        bool exact = nkc < 4096;
        for (unsigned kc = 0; exact && kc < nkc; ++kc) exact = ((kc * a.inv_cpt) >> 20) == kc / cpt;
        const unsigned maxtap = nkc / cpt + 1;
        for (unsigned t = 0; exact && t <= maxtap; ++t) exact = ((t * a.inv_s) >> 16) == t / (unsigned)S;
        if (!exact) dma_ok = false;
    }
    // the packed weight has Npad = multiple of 128 rows, so any BN <= 128 tiles it safely
    const int bn = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
    a.tiles_n = (Cout + bn - 1) / bn;
    const dim3 grid(a.tiles_m * a.tiles_n), block(CONV_THREADS);
    const bool fastk = (Cin % 64) == 0, fastk32 = (Cin % 32) == 0;
    const size_t epi = (size_t)BM * (bn + 8) * 2 + 4096;     // output image + BatchNorm partials
    // tile-quantisation model shared by the rules below: 128-row tiles run two workgroups per CU (512 slots), 64-row tiles
    // three (768 slots) at 0.88 of the per-tile efficiency (measured)
    const long long t128 = (long long)a.tiles_m * a.tiles_n;
    const long long t64 = (long long)((a.M + 63) / 64) * a.tiles_n;
    const double e128 = (double)t128 / (double)(((t128 + 511) / 512) * 512);
    const double e64 = 0.88 * (double)t64 / (double)(((t64 + 767) / 768) * 768);
    const bool want64 = bn == 128 && !tile_stats && !lstm && e64 > e128 * 1.04;

    // (1) Cin == 8 stencil layers (E2VID head): LDS halo tile instead of the im2col gather
    if (!capture && !lstm && Cin == 8 && stride == 1 && dil == 1 && R == 5 && S == 5 && Cout <= 32 && (Cout & 3) == 0 && !residual && !out_f32 &&
        !tile_stats && (out_pix_stride & 3) == 0 && a.Kpad == 256 && dma_ok) {
        if (want_workspace) return OESS_OK;
        const int tiles = B * ((a.Ho + 7) / 8) * ((a.Wo + 63) / 64);
        hipLaunchKernelGGL((conv_smallcin_kernel<5, 5>), dim3(tiles < 512 ? tiles : 512), dim3(256), 0, st, a);   // persistent: 2 workgroups per CU
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
    if (!dma_ok) {
        if (want_workspace) return OESS_OK;
        if (lstm || capture) return OESS_EINVAL;       // the fused ConvLSTM epilogue exists only in the LDS-DMA kernels
        const size_t tab = (size_t)(a.Kpad / 8) * 8;
        size_t lds = (size_t)2 * (BM + bn) * 8 * 16 + tab;
        if (lds < epi) lds = epi;
        if (bn == 128) hipLaunchKernelGGL(conv_fwd_kernel<128>, grid, block, lds, st, a);
        else if (bn == 64) hipLaunchKernelGGL(conv_fwd_kernel<64>, grid, block, lds, st, a);
        else hipLaunchKernelGGL(conv_fwd_kernel<32>, grid, block, lds, st, a);
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
    // (1b) 5x5 stride-2 pad-2 layers (E2VID's encoder ConvLayers): 2-D input halo in LDS, 8 x 16-pixel x 64-channel tiles
    if (!lstm && R == 5 && S == 5 && stride == 2 && pad == 2 && dil == 1 && (Cin & 31) == 0 && (Cout & 63) == 0 && !tile_stats &&
        !residual && !out_f32 && relu != 2 && (out_pix_stride & 7) == 0 && (((uintptr_t)out_bf16) & 15) == 0 && a.Kpad >= 25 * Cin) {
        if (want_workspace) return OESS_OK;
        a.tiles_n = Cout / 64;
        a.tiles_m = B * ((a.Ho + S2_PH - 1) / S2_PH) * ((a.Wo + S2_PW - 1) / S2_PW);
        if (capture) { *capture = a; return OESS_OK; }
        hipLaunchKernelGGL((conv5x5s2_halo_kernel<false>), dim3(a.tiles_m * a.tiles_n), dim3(256), S2_LDS, st, a, S2Head{});
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
    if (capture && !lstm) return OESS_EINVAL;          // (non-ConvLSTM captures are for the stride-2 group launch only)
    // (1c) SMALL MAPS (at most one 128 x 128 tile per CU: DeepLabv3's OS16 half, M = 8 960), K >= 16 slabs, not a row-halo 3x3:
    //      with a single workgroup on a CU nothing is lost by giving it the whole LDS, so the ring is 4 deep (three slabs in
    //      flight).  Measured on the DeepLabv3 forward (round 5, same box, ring 2 / 3 / 4): 1024->256 36.1 / 31.6 / 31.4 us,
    //      2048->256 42.1 / 36.2 / 35.4, 1280->256 30.6-43.8 / 26.3 / 26.7, 256->1024 25.1 / 21.1 / 23.8; the 3x3 layers of the
    //      same maps do not move (45 / 64 / 73 us either way: they stay on the row-halo kernel) -- the small-map K loop is not
    //      waiting for the operand round trip alone (profiles/r05_deep_ring_ab.txt).  EIGHT waves (32 x 64 wave tiles, two waves
    //      per SIMD: one wave's DMA issue runs under the other's MFMAs, which a lone 4-wave workgroup cannot do): 1024->256
    //      32.3 -> 29.4 us, 2048->256 36.0 -> 30.1, 1280->256 28.7 -> 23.1, 256->1024 22.5 -> 20.4 (same box, alternating).
    {
        const bool is3x3halo = R == 3 && S == 3 && stride == 1 && pad == dil;
        if (!capture && !lstm && bn == 128 && fastk && t128 <= 256 && a.Kpad / BK >= 16 && a.Kpad / BK < 200 && !is3x3halo) {
            if (want_workspace) return OESS_OK;
            size_t lds = (size_t)4 * (BM + 128) * 8 * 16;
            if (lds < epi) lds = epi;
            hipLaunchKernelGGL((conv_fwd_dma_kernel<128, 128, 4, true, 0, 512>), grid, dim3(512), lds, st, a);
            OESS_HIP(hipGetLastError());
            return OESS_OK;
        }
    }
    // (2a) the same layers in front of a BatchNorm (raw bf16 result + tile statistics: conv2 of the frozen teacher's dilated
    //      bottlenecks), Cout % 256 == 0, >= 2 tiles of 256 x 256 per CU: persistent workgroups on 128 x 128 wave tiles
    //      (conv3x3_w128.h).  OESS_W128_CONV3=0 keeps rule (2) (A/B).
    if (!capture && !lstm && R == 3 && S == 3 && stride == 1 && pad == dil && fastk && (Cout % 256) == 0 && a.Kpad == 9 * Cin && a.Ho == H && a.Wo == W &&
        !bias && !relu && !residual && !out_f32 && (out_pix_stride & 7) == 0 && (((uintptr_t)out_bf16) & 15) == 0 && a.mg_w && a.mg_wd) {
        const int use3 = [] { const char* e = getenv("OESS_W128_CONV3"); return e ? atoi(e) : 1; }();
        const long long t256 = (long long)((a.M + 255) / 256) * (Cout / 256);
        const long long out_extent = ((long long)a.M - 1) * out_pix_stride * 2 + (long long)Cout * 2;
        const int breaks = (256 + W - 2) / W;
        if (use3 && t256 >= 2ll * num_cus() && dil + 255 + dil * breaks + dil + 1 <= W128_HROWS && breaks + 1 + 2 * dil <= H &&
            out_extent < 0x7ffffff0ll && (long long)Cout * a.Kpad * 2 < 0x7ffffff0ll && (long long)H * W * W < 0x100000000ll) {
            if (want_workspace) return OESS_OK;
            a.tiles_m = (a.M + 255) / 256; a.tiles_n = Cout / 256;
            hipLaunchKernelGGL(conv3x3_w128_kernel, dim3(num_cus() / 8 * 8), dim3(256), (size_t)W128_OPER, st, a);
            OESS_HIP(hipGetLastError());
            return OESS_OK;
        }
    }
    // (2) 3x3 stride-1 'same' convolutions with Cin % 64 == 0: row-halo reuse of the pixel operand, unless the 64-row tiling
    //     is what the layer wants (tile quantisation of small maps)
    if (R == 3 && S == 3 && stride == 1 && pad == dil && fastk && bn == 128 && a.Kpad == 9 * Cin && a.Ho == H && a.Wo == W &&
        (dil + 127 + dil * ((BM + W - 2) / W) + dil + 1) <= HALO_ROWS && !want64) {
        if (want_workspace) return OESS_OK;
        if (capture) { *capture = a; return OESS_OK; }
        const size_t lds = (size_t)2 * HALO_ROWS * 128 + (size_t)2 * 128 * 128;
        if (lstm) hipLaunchKernelGGL((conv3x3_halo_kernel<1>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<0>), grid, block, lds, st, a);
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
    // (3) fused ConvLSTM cell update on geometries the halo kernel does not take
    if (capture) return OESS_EINVAL;
    if (lstm) {
        const size_t lds = (size_t)2 * (BM + 128) * 8 * 16;
        if (fastk) hipLaunchKernelGGL((conv_fwd_dma_kernel<128, 128, 2, true, 1>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((conv_fwd_dma_kernel<128, 128, 2, false, 1>), grid, block, lds, st, a);
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
    // (3a) split-K for small-M / long-K layers (DeepLabv3's ASPP at output stride 16: M = 8 960 = 70 row tiles x 2 column
    //      tiles = 140 workgroups for 512 slots, K = 18 432 = 288 slabs each: 272 us at 311 TFLOP/s).  Model (us): a workgroup
    //      spends 1.1 per slab + 6 fixed; the fp32 slices cost a write and a read of ks * M * Cout * 4 bytes at ~3 TB/s plus
    //      one launch.  Taken when it beats both the one-pass 128-row tiling and the 64-row tiling by 15 %.
    if (!lstm && bn == 128 && (Cout & 3) == 0 && a.Kpad / BK >= 32 && t128 <= 384 && (!out_f32 || (out_pix_stride & 3) == 0)) {
        const int KTall = a.Kpad / BK;
        auto rounds = [](long long wg, long long slots) { return (double)((wg + slots - 1) / slots); };
        const double t_one = rounds(t128, 512) * (KTall * 1.1 + 6.0);
        const double t_64 = rounds(t64, 768) * (KTall * 0.62 + 6.0);
        int best = 1;
        double tbest = (tile_stats ? t_one : (t_one < t_64 ? t_one : t_64)) / 1.15;
        for (int ks = 2; ks <= 8; ++ks) {
            const int per = (KTall + ks - 1) / ks;
            if (per < 8 || (long long)per * (ks - 1) >= KTall) continue;       // every slice non-empty
            const double tk = rounds(t128 * ks, 512) * (per * 1.1 + 6.0) + 2.0 * ks * (double)a.M * Cout * 4.0 / 3.0e6 + 5.0;
            if (tk < tbest) { tbest = tk; best = ks; }
        }
        if (best > 1) {
            const size_t need = (size_t)best * (size_t)a.M * Cout * sizeof(float);
            if (want_workspace) { *want_workspace = need; return OESS_OK; }
            if (workspace && workspace_bytes >= need) {
                a.partial = (float*)workspace; a.ksplit = best; a.kt_per = (KTall + best - 1) / best;
                const size_t lds = (size_t)2 * (BM + 128) * 8 * 16;
                const dim3 gridk(a.tiles_m * a.tiles_n, best);
                if (fastk) hipLaunchKernelGGL((conv_fwd_dma_kernel<128, 128, 2, true>), gridk, block, lds, st, a);
                else hipLaunchKernelGGL((conv_fwd_dma_kernel<128, 128, 2, false>), gridk, block, lds, st, a);
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3(a.tiles_m, (Cout + 63) / 64), dim3(256), 0, st, a);
                OESS_HIP(hipGetLastError());
                return OESS_OK;
            }
        }
    }
    if (want_workspace) return OESS_OK;
    // (3b) large plain-GEMM layers (1x1, Cin % 64 == 0, Cout % 256 == 0, K >= 256): 256 x 256 tiles on ONE 8-wave workgroup per
    //      CU, wave tile 64 x 128 (24 fragment reads per 32 MFMAs instead of 16 per 16, half the L2 -> LDS bytes per FLOP).
    //      Same template as rule (6).  Measured against the 128 x 128 tiling on the teacher's layers at M = 140 800:
    //      512->2048 556 -> 451 us, 1024->2048 831 -> 680, 2048->512 396 -> 362, 256->1024 179 -> 169; ViT fc1 (M = 8 968,
    //      432 tiles) 65.8 -> 62.5 without its GELU epilogue (with it, round 5: 88 us on either tiling).  Below ~1.5 rounds of 256 tiles the coarser quantisation loses (324 tiles: 51.7 -> 59.5 us).
    //      (Round 5: under the concurrent step schedule these 128 KB / 512-thread workgroups wait for a CU free of ConvLSTM
    //      workgroups and run 2.2 x longer than alone; the 128 x 128 tiling, which co-resides, was measured there as well:
    //      193.0 vs 194.8 event-frames/s, 318 vs 324 on frame2recon_full -- the big tile stays.)
    if (!lstm && bn == 128 && fastk && (Cout % 256) == 0 && a.Kpad >= 256 && R == 1 && S == 1 && stride == 1) {
        const long long t256 = (long long)((a.M + 255) / 256) * (Cout / 256);
        // (3a) the same layers in front of a BatchNorm (raw bf16 result + tile statistics; the frozen teacher's conv1 / conv3 /
        //      downsample layers) or with a bias / ReLU (its 2048 -> 256 decoder layer), >= 2 tiles per CU: persistent workgroups on
        //      128 x 128 wave tiles (conv_w128_gemm.h).
        //      OESS_W128_GEMM=0 keeps rule (3b) (A/B).
        const int use_w128 = [] { const char* e = getenv("OESS_W128_GEMM"); return e ? atoi(e) : 1; }();
        const long long out_extent = ((long long)a.M - 1) * out_pix_stride * 2 + (long long)Cout * 2;
        static const long long min_t = [] { const char* e = getenv("OESS_W128_MIN_TILES"); return e ? atoll(e) : 0ll; }();      // A/B knob
        if (use_w128 && t256 >= (min_t > 0 ? min_t : 2ll * num_cus()) && (relu == 0 || relu == 1) && !residual && !out_f32 && a.Kpad == Cin && (out_pix_stride & 7) == 0 &&
            (((uintptr_t)out_bf16) & 15) == 0 && out_extent < 0x7ffffff0ll && (long long)Cout * a.Kpad * 2 < 0x7ffffff0ll) {
            a.tiles_m = (a.M + 255) / 256; a.tiles_n = Cout / 256;
            // non-temporal result stores where the result is >= 4 x the input (256 -> 1024 168 -> 150 us, 512 -> 2048 402 -> 376;
            // 2 x and reducing layers lose 2-10 % with them: EXPERIMENTS R6-8).  OESS_W128_NT = 0 / 1 forces (A/B).
            static const int nt_env = [] { const char* e = getenv("OESS_W128_NT"); return e ? atoi(e) : -1; }();
            a.ksplit = nt_env >= 0 ? nt_env : (Cout >= 4 * Cin);
            hipLaunchKernelGGL(conv1x1_w128_kernel, dim3(num_cus() / 8 * 8), dim3(256), (size_t)G128_LDS, st, a);
            OESS_HIP(hipGetLastError());
            return OESS_OK;
        }
        if (getenv("OESS_W128_WHY"))
            fprintf(stderr, "[oess] 1x1 %d -> %d M %d not on conv1x1_w128_kernel: t256 %lld bias %d relu %d residual %d out_f32 %d Kpad %d ops %lld align %d\n", Cin, Cout, a.M,
                    t256, bias != nullptr, relu, residual != nullptr, out_f32 != nullptr, a.Kpad, (long long)out_pix_stride, (int)(((uintptr_t)out_bf16) & 15));
        if (t256 >= 400) {
            a.tiles_m = (a.M + 255) / 256; a.tiles_n = Cout / 256;
            size_t lds = (size_t)2 * 512 * 128;                                          // 2 stages x (256 + 256) rows x 128 B
            const size_t epi256 = (size_t)256 * (256 + 8) * 2 + (size_t)4 * 256 * 2 * 4 + 256;   // output image + BatchNorm partials
            if (lds < epi256) lds = epi256;
            hipLaunchKernelGGL((conv_fwd_dma_kernel<256, 256, 2, true>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a);
            OESS_HIP(hipGetLastError());
            return OESS_OK;
        }
    }
    // (4) short reductions (K <= 256, the 1x1 bottleneck convs): a workgroup lives ~6 us of which the K loop is a fraction, so
    //     residency matters more than the main loop: BK = 32 slabs in a 3-deep ring = 48 KB -> 3 workgroups per CU
    //     (K = 64: 0.121 -> 0.096 ms, K = 256: 0.0414 -> 0.0369 ms; from K = 512 up the BK = 64 kernel wins again)
    if (bn == 128 && fastk32 && a.Kpad <= 256) {
        size_t lds3 = (size_t)3 * (BM + 128) * 64;
        if (lds3 < epi) lds3 = epi;
        hipLaunchKernelGGL((conv_fwd_dma32_kernel<128, true, 0, 3>), grid, block, lds3, st, a);
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
    // (5) 64 x 128 tiles (48 KB of LDS: 3 workgroups per CU) when the 128-row tiling leaves most of its last round of
    //     workgroups empty: e.g. 550 tiles over 512 slots run as two rounds at 54 % - 1100 half tiles over 768 slots do not
    if (want64) {
        a.tiles_m = (a.M + 63) / 64;
        const dim3 grid64(a.tiles_m * a.tiles_n);
        size_t lds = (size_t)2 * (64 + 128) * 8 * 16;
        const size_t epi64 = (size_t)64 * (128 + 8) * 2 + 4096;
        if (lds < epi64) lds = epi64;
        if (fastk) hipLaunchKernelGGL((conv_fwd_dma_kernel<64, 128, 2, true>), grid64, block, lds, st, a);
        else hipLaunchKernelGGL((conv_fwd_dma_kernel<64, 128, 2, false>), grid64, block, lds, st, a);
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
    // (6) the general LDS-DMA kernel: 128 x {128, 64, 32} tiles, 2-stage ring, 2 workgroups per CU
    {
        size_t lds = (size_t)2 * (BM + bn) * 8 * 16;
        if (lds < epi) lds = epi;
#define OESS_LAUNCH_DMA(BN_)                                                                                  \
        if (fastk) hipLaunchKernelGGL((conv_fwd_dma_kernel<128, BN_, 2, true>), grid, block, lds, st, a);        \
        else hipLaunchKernelGGL((conv_fwd_dma_kernel<128, BN_, 2, false>), grid, block, lds, st, a);
        if (bn == 128) { OESS_LAUNCH_DMA(128) }
        else if (bn == 64) { OESS_LAUNCH_DMA(64) }
        else { OESS_LAUNCH_DMA(32) }
#undef OESS_LAUNCH_DMA
    }
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}
}  // namespace

extern "C" {

int oess_conv2d_fwd_bf16(const void* in, long long in_pix_stride, int B, int H, int W, int Cin, const void* w_packed,
                         const float* bias, int Cout, int R, int S, int stride, int pad, int dil, int relu,
                         const void* residual, long long res_pix_stride, void* out_bf16, float* out_f32,
                         long long out_pix_stride, float* tile_stats, void* workspace, size_t workspace_bytes,
                         oess_stream_t stream) {
    return conv_fwd_impl(in, in_pix_stride, B, H, W, Cin, w_packed, bias, Cout, R, S, stride, pad, dil, relu, residual,
                         res_pix_stride, out_bf16, out_f32, out_pix_stride, tile_stats, nullptr, stream, workspace, workspace_bytes);
}

size_t oess_conv2d_fwd_workspace_bytes(int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil,
                                       int with_tile_stats, int out_is_f32) {
    // runs the dispatch rules of oess_conv2d_fwd_bf16 on dummy (never dereferenced) pointers up to the split-K decision
    size_t need = 0;
    static const char dummy[16] = {0};
    const int rc = conv_fwd_impl(dummy, Cin, B, H, W, Cin, dummy, nullptr, Cout, R, S, stride, pad, dil, 0, nullptr, 0,
                                 out_is_f32 ? nullptr : (void*)dummy, out_is_f32 ? (float*)dummy : nullptr, (Cout + 7) / 8 * 8,
                                 with_tile_stats ? (float*)dummy : nullptr, nullptr, nullptr, nullptr, 0, &need);
    return rc == OESS_OK ? need : 0;
}

int oess_convlstm_fused_bf16(const void* in, long long in_pix_stride, int B, int H, int W, int Cin, const void* w_packed_gates,
                             const float* bias, int C_hidden, int R, int S, int pad, const float* prev_cell, float* cell,
                             void* hidden, long long hidden_pix_stride, oess_stream_t stream) {
    if (!in || !hidden) return OESS_EINVAL;
    {   // the hidden output must not alias the convolution input (neighbouring tiles still read the old state)
        const char* i0 = (const char*)in; const char* i1 = i0 + ((long long)B * H * W - 1) * in_pix_stride * 2 + (long long)Cin * 2;
        const char* h0 = (const char*)hidden; const char* h1 = h0 + ((long long)B * H * W - 1) * hidden_pix_stride * 2 + (long long)C_hidden * 2;
        if (h0 < i1 && i0 < h1) return OESS_EINVAL;
    }
    LstmOut l{prev_cell, cell, hidden, hidden_pix_stride, C_hidden};
    return conv_fwd_impl(in, in_pix_stride, B, H, W, Cin, w_packed_gates, bias, 4 * C_hidden, R, S, 1, pad, 1, 0, nullptr, 0,
                         nullptr, nullptr, 0, nullptr, &l, stream);
}

int oess_convlstm_fused_group_bf16(const oess_convlstm_desc_t* d, int n, oess_stream_t stream) {
    if (!d || n <= 0 || n > 3) return OESS_EINVAL;
    auto span = [](const void* p, long long pixels, long long stride, int c, const char** lo, const char** hi) {
        *lo = (const char*)p; *hi = *lo + (pixels - 1) * stride * 2 + (long long)c * 2;
    };
    for (int i = 0; i < n; ++i) {
        if (!d[i].in || !d[i].hidden || !d[i].cell || d[i].B <= 0 || d[i].H <= 0 || d[i].W <= 0) return OESS_EINVAL;
        const long long px = (long long)d[i].B * d[i].H * d[i].W;
        const char *h0, *h1, *c0 = (const char*)d[i].cell, *c1 = c0 + px * d[i].C_hidden * 4;
        span(d[i].hidden, px, d[i].hidden_pix_stride, d[i].C_hidden, &h0, &h1);
        for (int j = 0; j < n; ++j) {       // the problems run concurrently: no output of one may overlap anything of another
            const long long pj = (long long)d[j].B * d[j].H * d[j].W;
            const char *i0, *i1, *g0, *g1, *e0 = (const char*)d[j].cell, *e1 = e0 + pj * d[j].C_hidden * 4;
            span(d[j].in, pj, d[j].in_pix_stride, d[j].Cin, &i0, &i1);
            span(d[j].hidden, pj, d[j].hidden_pix_stride, d[j].C_hidden, &g0, &g1);
            if (h0 < i1 && i0 < h1) return OESS_EINVAL;                   // (j == i: the single-problem rule)
            if (j != i && ((h0 < g1 && g0 < h1) || (c0 < e1 && e0 < c1) || (c0 < i1 && i0 < c1))) return OESS_EINVAL;
        }
    }
    ConvArgs args[3];
    bool grouped = n >= 2;
    for (int i = 0; i < n && grouped; ++i) {
        LstmOut l{d[i].prev_cell, d[i].cell, d[i].hidden, d[i].hidden_pix_stride, d[i].C_hidden};
        grouped = conv_fwd_impl(d[i].in, d[i].in_pix_stride, d[i].B, d[i].H, d[i].W, d[i].Cin, d[i].w_packed_gates, d[i].bias,
                                4 * d[i].C_hidden, d[i].R, d[i].S, 1, d[i].pad, 1, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, &l,
                                stream, nullptr, 0, nullptr, &args[i]) == OESS_OK;
    }
    if (!grouped) {
        for (int i = 0; i < n; ++i) {
            const int rc = oess_convlstm_fused_bf16(d[i].in, d[i].in_pix_stride, d[i].B, d[i].H, d[i].W, d[i].Cin, d[i].w_packed_gates,
                                                    d[i].bias, d[i].C_hidden, d[i].R, d[i].S, d[i].pad, d[i].prev_cell, d[i].cell,
                                                    d[i].hidden, d[i].hidden_pix_stride, stream);
            if (rc != OESS_OK) return rc;
        }
        return OESS_OK;
    }
    int order[3] = {0, 1, 2};                  // longest K first: the launch ends on the short tiles
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && args[order[j]].Kpad > args[order[j - 1]].Kpad; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    ConvGroup g;
    memset(&g, 0, sizeof(g));
    int at = 0;
    for (int i = 0; i < 3; ++i) {
        g.start8[i] = at;
        if (i < n) {
            g.a[i] = args[order[i]];
            at += (g.a[i].tiles_m * g.a[i].tiles_n + 7) / 8;
        }
    }
    g.start8[3] = at;
    // 256 x 128 tiles on ONE 8-wave workgroup per CU (default since round 5).  Alone the grouped launch is 2-3 % slower on them
    // than on two 128 x 128 workgroups per CU (one workgroup per CU exposes the cell-update epilogue; fewer operand bytes per FLOP
    // buy nothing), but the product schedule runs it next to the teacher's and the decoder's kernels, and one 112 KB workgroup
    // leaves them 48 KB of LDS and 24 wave slots per CU where two 72 KB workgroups leave 16 KB: +1.4-1.6 % on the step on three
    // boxes (EXPERIMENTS R5-3b).  OESS_LSTM256 = 0 restores the 128 x 128 tiles (A/B), 2 = only the problems with >= 30 K-slabs.
    static const int use256 = [] { const char* e = getenv("OESS_LSTM256"); return e ? atoi(e) : 1; }();
    if (use256) {
        ConvGroup big, small;
        memset(&big, 0, sizeof(big)); memset(&small, 0, sizeof(small));
        int nb = 0, ns = 0;
        for (int i = 0; i < n; ++i) {
            const ConvArgs& a = g.a[i];
            const bool fits = a.R == 3 && a.dil == 1 && (a.dil + 255 + a.dil * ((256 + a.W - 2) / a.W) + a.dil + 1) <= HALO_ROWS_256 &&
                              a.tiles_n * 128 == a.Cout;
            if (fits && (use256 == 1 || a.Kpad / BK >= 30)) big.a[nb++] = a; else small.a[ns++] = a;
        }
        auto layout = [](ConvGroup& q, int cnt, int rows) {
            int at_ = 0;
            for (int i = 0; i < 3; ++i) {
                q.start8[i] = at_;
                if (i < cnt) {
                    q.a[i].tiles_m = (q.a[i].M + rows - 1) / rows;
                    at_ += (q.a[i].tiles_m * q.a[i].tiles_n + 7) / 8;
                }
            }
            q.start8[3] = at_;
            return at_;
        };
        if (nb) {
            const int atb = layout(big, nb, 256);
            const size_t lds256 = (size_t)2 * HALO_ROWS_256 * 128 + (size_t)2 * 128 * 128;
            hipLaunchKernelGGL(conv3x3_halo256_group_kernel, dim3(8 * atb), dim3(512), lds256, (hipStream_t)stream, big);
        }
        if (ns) {
            const int ats = layout(small, ns, 128);
            const size_t lds128 = (size_t)2 * HALO_ROWS * 128 + (size_t)2 * 128 * 128;
            hipLaunchKernelGGL((conv3x3_halo_group_kernel<1>), dim3(8 * ats), dim3(CONV_THREADS), lds128, (hipStream_t)stream, small);
        }
        OESS_HIP(hipGetLastError());
        return OESS_OK;
    }
    const size_t lds = (size_t)2 * HALO_ROWS * 128 + (size_t)2 * 128 * 128;
    hipLaunchKernelGGL((conv3x3_halo_group_kernel<1>), dim3(8 * at), dim3(CONV_THREADS), lds, (hipStream_t)stream, g);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // extern "C"

// ---- ConvLSTM on 128 x 128 wave tiles with the cell state in the kernel's own layout (conv_lstm_w128.h)
namespace {
struct W128Sched { int* dev; int stride; int grid; };
// Static tile lists: workgroup b (one per CU; XCD b % 8 by the dispatch order, used for locality only) takes tiles of "its" XCD's
// contiguous chunk of every problem -- neighbouring tiles share halo rows and weight slabs in that XCD's L2, as in the one-tile
// kernels -- dealt longest-K first onto the least loaded of the XCD's workgroups (cost = slabs x measured cycles per slab + a
// per-tile constant), i.e. what a dynamic queue would do, without an atomic and a reset per launch.
// pure host part: flat[grid][stride] tile lists ((problem << 24) | tile, -1 = end), every tile of every problem exactly once
static int w128_tile_lists(const ConvArgs* a, int n, int grid, std::vector<int>* flat, int* stride_out) {
    if (grid < 8 || grid % 8) return OESS_EINVAL;
    const int per = grid / 8;
    std::vector<std::vector<int>> lists(grid);
    std::vector<long long> load(grid, 0);
    for (int x = 0; x < 8; ++x)
        for (int p = 0; p < n; ++p) {
            const int nwg = a[p].tiles_m * a[p].tiles_n, q = nwg >> 3, r = nwg & 7;
            const int cnt = q + (x < r ? 1 : 0), base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
            const long long cost = (long long)(a[p].Cin / 64) * 9 * 2330 + 20000;     // measured: cycles per K-slab, per-tile set-up + cell update (profiles/r06_w128_v2_stamps.txt)
            for (int li = 0; li < cnt; ++li) {
                int best = 0;
                for (int c = 1; c < per; ++c) if (load[c * 8 + x] < load[best * 8 + x]) best = c;
                lists[best * 8 + x].push_back((p << 24) | (base + li));
                load[best * 8 + x] += cost;
            }
        }
    size_t longest = 0;
    for (auto& l : lists) longest = l.size() > longest ? l.size() : longest;
    if (longest > (size_t)W128_MAX_LIST) return OESS_EINVAL;
    const int stride = (int)longest + 1;
    flat->assign((size_t)grid * stride, -1);
    for (int b = 0; b < grid; ++b) for (size_t k = 0; k < lists[b].size(); ++k) (*flat)[(size_t)b * stride + k] = lists[b][k];
    *stride_out = stride;
    return OESS_OK;
}
static int w128_schedule(const ConvArgs* a, int n, W128Sched* out) {
    static std::mutex mu;
    static std::map<std::vector<int>, W128Sched> cache;
    std::vector<int> key;
    // OESS_W128_GRID (A/B): fewer persistent workgroups than CUs leaves whole CUs to the kernels of the other streams of the product
    // schedule (a 152 KB / 512-register workgroup shares its CU with nothing)
    static const int grid_env = [] { const char* e = getenv("OESS_W128_GRID"); return e ? atoi(e) : 0; }();
    const int grid = (grid_env >= 8 && grid_env <= num_cus()) ? grid_env / 8 * 8 : num_cus();
    int dev = 0;
    (void)hipGetDevice(&dev);
    key.push_back(dev);                        // the list lives in this device's memory
    key.push_back(grid);
    for (int i = 0; i < n; ++i) { key.push_back(a[i].tiles_m); key.push_back(a[i].tiles_n); key.push_back(a[i].Cin); }
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) { *out = it->second; return OESS_OK; }
    std::vector<int> flat;
    int stride = 0;
    if (w128_tile_lists(a, n, grid, &flat, &stride) != OESS_OK) return OESS_EINVAL;
    W128Sched sc{nullptr, stride, grid};
    if (hipMalloc((void**)&sc.dev, flat.size() * sizeof(int) + (size_t)grid * 4 * 16 * sizeof(float)) != hipSuccess) return OESS_ELAUNCH;
    if (hipMemcpy(sc.dev, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return OESS_ELAUNCH;
    cache[key] = sc;
    *out = sc;
    return OESS_OK;
}
}  // namespace

extern "C" {
int oess_convlstm_w128_tile_lists(const int* tiles_m, const int* tiles_n, const int* cin, int n, int grid, int* lists, int capacity, int* stride) {
    if (!tiles_m || !tiles_n || !cin || !stride || n <= 0 || n > 3) return OESS_EINVAL;
    ConvArgs a[3];
    memset(a, 0, sizeof(a));
    for (int i = 0; i < n; ++i) { a[i].tiles_m = tiles_m[i]; a[i].tiles_n = tiles_n[i]; a[i].Cin = cin[i]; }
    std::vector<int> flat;
    const int rc = w128_tile_lists(a, n, grid, &flat, stride);
    if (rc != OESS_OK) return rc;
    if (lists) {
        if ((size_t)capacity < flat.size()) return OESS_ENOMEM;
        memcpy(lists, flat.data(), flat.size() * sizeof(int));
    }
    return OESS_OK;
}

size_t oess_convlstm_w128_cell_bytes(long long pixels, int C_hidden) {
    if (pixels <= 0 || C_hidden <= 0 || (C_hidden & 63)) return 0;
    return (size_t)((pixels + 255) / 256 * 256) * (size_t)C_hidden * 4;
}

int oess_convlstm_w128_cell_relayout(const float* src, float* dst, long long pixels, int C_hidden, int to_tiled, oess_stream_t stream) {
    if (!src || !dst || src == dst || pixels <= 0 || C_hidden <= 0 || (C_hidden & 63)) return OESS_EINVAL;
    const long long total = (pixels + 255) / 256 * 256 * C_hidden;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(w128_cell_relayout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, pixels, C_hidden, to_tiled);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_convlstm_w128_group_bf16(const oess_convlstm_desc_t* d, int n, oess_stream_t stream) {
    if (!d || n <= 0 || n > 3) return OESS_EINVAL;
    auto span = [](const void* p, long long pixels, long long stride, int c, const char** lo, const char** hi) {
        *lo = (const char*)p; *hi = *lo + (pixels - 1) * stride * 2 + (long long)c * 2;
    };
    int ctot = 0;
    for (int i = 0; i < n; ++i) {
        if (!d[i].in || !d[i].hidden || !d[i].cell || d[i].B <= 0 || d[i].H < 8 || d[i].W <= 0 || d[i].R != 3 || d[i].S != 3 || d[i].pad != 1 ||
            d[i].C_hidden <= 0 || (d[i].C_hidden & 63) || d[i].Cin <= 0 || (d[i].Cin & 63))
            return OESS_EINVAL;
        ctot += 4 * d[i].C_hidden;
        const long long px = (long long)d[i].B * d[i].H * d[i].W;
        const size_t cb = oess_convlstm_w128_cell_bytes(px, d[i].C_hidden);
        if (cb >= 0x7fffffffull || px * d[i].W >= 0x100000000ll) return OESS_EINVAL;
        if (d[i].prev_cell && d[i].prev_cell != d[i].cell) {
            const char *p0 = (const char*)d[i].prev_cell, *c0 = (const char*)d[i].cell;
            if (p0 < c0 + cb && c0 < p0 + cb) return OESS_EINVAL;         // a tile updates its own block in place or elsewhere, not a shifted copy
        }
        const char *h0, *h1, *c0 = (const char*)d[i].cell, *c1 = c0 + cb;
        span(d[i].hidden, px, d[i].hidden_pix_stride, d[i].C_hidden, &h0, &h1);
        if ((h1 - h0) >= 0x7fffffffll) return OESS_EINVAL;
        for (int j = 0; j < n; ++j) {
            const long long pj = (long long)d[j].B * d[j].H * d[j].W;
            const char *i0, *i1, *g0, *g1, *e0 = (const char*)d[j].cell, *e1 = e0 + oess_convlstm_w128_cell_bytes(pj, d[j].C_hidden);
            span(d[j].in, pj, d[j].in_pix_stride, d[j].Cin, &i0, &i1);
            span(d[j].hidden, pj, d[j].hidden_pix_stride, d[j].C_hidden, &g0, &g1);
            if (h0 < i1 && i0 < h1) return OESS_EINVAL;
            if (c0 < i1 && i0 < c1) return OESS_EINVAL;
            if (j != i && ((h0 < g1 && g0 < h1) || (c0 < e1 && e0 < c1))) return OESS_EINVAL;
        }
    }
    if (ctot > W128_BIAS_FLOATS) return OESS_EINVAL;
    ConvArgs args[3];
    for (int i = 0; i < n; ++i) {
        LstmOut l{d[i].prev_cell, d[i].cell, d[i].hidden, d[i].hidden_pix_stride, d[i].C_hidden};
        if (conv_fwd_impl(d[i].in, d[i].in_pix_stride, d[i].B, d[i].H, d[i].W, d[i].Cin, d[i].w_packed_gates, d[i].bias, 4 * d[i].C_hidden,
                          3, 3, 1, 1, 1, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, &l, stream, nullptr, 0, nullptr, &args[i]) != OESS_OK)
            return OESS_EINVAL;
        ConvArgs& a = args[i];
        // halo rows of a 256-pixel tile; a tile may wrap into the next image at most once (rows per tile <= H)
        if (a.Kpad != 9 * a.Cin || (1 + 255 + ((256 + a.W - 2) / a.W) + 1 + 1) > W128_HROWS || ((256 + a.W - 2) / a.W + 1) > a.H || !a.mg_w || !a.mg_wd)
            return OESS_EINVAL;
        a.tiles_m = (a.M + 255) / 256;
        a.tiles_n = a.Cout / 256;
    }
    int order[3] = {0, 1, 2};                  // longest K first
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && args[order[j]].Kpad > args[order[j - 1]].Kpad; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    W128Group g;
    memset(&g, 0, sizeof(g));
    for (int i = 0; i < n; ++i) g.a[i] = args[order[i]];
    g.n = n;
    W128Sched sc;
    const int rc = w128_schedule(g.a, n, &sc);
    if (rc != OESS_OK) return rc;
    g.sched = sc.dev; g.sched_stride = sc.stride;
    hipLaunchKernelGGL(conv3x3_lstm_w128_kernel, dim3(sc.grid), dim3(256), (size_t)W128_LDS, (hipStream_t)stream, g);
    OESS_HIP(hipGetLastError());
#if (W128_ABL & 8192)
    {
        std::vector<float> st((size_t)sc.grid * 64);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(st.data(), sc.dev + (size_t)sc.grid * sc.stride, st.size() * 4, hipMemcpyDeviceToHost);
        double sum[16] = {0};
        for (int b = 0; b < sc.grid; ++b) for (int k = 0; k < 16; ++k) sum[k] += st[((size_t)b * 4) * 16 + k];
        fprintf(stderr, "w128 stamps (wave 0, cycles per workgroup, mean of %d): dx0 G0..G3 | dx1 | dx2 | setup+fill-issue  first-operand-wait  cell-update  tiles\n  ", sc.grid);
        double loop = 0;
        for (int k = 0; k < 12; ++k) { fprintf(stderr, "%9.0f%s", sum[k] / sc.grid, (k & 3) == 3 ? " |" : ""); loop += sum[k] / sc.grid; }
        fprintf(stderr, " %9.0f %9.0f %9.0f %5.1f   loop %9.0f\n", sum[12] / sc.grid, sum[13] / sc.grid, sum[14] / sc.grid, sum[15] / sc.grid, loop);
        double mn = 1e30, mx = 0, mean = 0;
        for (int b = 0; b < sc.grid; ++b) {
            double t = 0;
            for (int k = 0; k < 15; ++k) t += st[((size_t)b * 4) * 16 + k];
            mn = t < mn ? t : mn; mx = t > mx ? t : mx; mean += t / sc.grid;
        }
        fprintf(stderr, "  stamped cycles per workgroup: min %.0f mean %.0f max %.0f\n", mn, mean, mx);
    }
#endif
    return OESS_OK;
}
}  // extern "C"

extern "C" {
int oess_conv5x5s2_group_bf16(const oess_conv_s2_desc_t* d, int n, oess_stream_t stream) {
    if (!d || n <= 0 || n > 2) return OESS_EINVAL;
    for (int i = 0; i < n; ++i)
        if (!d[i].in || !d[i].w_packed || !d[i].out || d[i].B <= 0 || d[i].H <= 0 || d[i].W <= 0) return OESS_EINVAL;
    if (n == 2) {       // the problems run concurrently: neither output may overlap the other's input or output
        const char *lo[2][2], *hi[2][2];
        long long st[2][2], cb[2][2];                      // pixel stride and channel bytes of {in, out}
        for (int i = 0; i < 2; ++i) {
            const long long pin = (long long)d[i].B * d[i].H * d[i].W;
            const long long pout = (long long)d[i].B * ((d[i].H - 1) / 2 + 1) * ((d[i].W - 1) / 2 + 1);
            st[i][0] = d[i].in_pix_stride * 2; cb[i][0] = (long long)d[i].Cin * 2;
            st[i][1] = d[i].out_pix_stride * 2; cb[i][1] = (long long)d[i].Cout * 2;
            lo[i][0] = (const char*)d[i].in; hi[i][0] = lo[i][0] + (pin - 1) * st[i][0] + cb[i][0];
            lo[i][1] = (const char*)d[i].out; hi[i][1] = lo[i][1] + (pout - 1) * st[i][1] + cb[i][1];
        }
        for (int i = 0; i < 2; ++i)
            for (int k = 0; k < 2; ++k) {
                if (!(lo[i][1] < hi[1 - i][k] && lo[1 - i][k] < hi[i][1])) continue;
                // overlapping spans are fine when the two views are disjoint CHANNEL SLICES of one pixel grid (the x half and
                // the h half of a ConvLSTM cat(x, h) buffer): same pixel stride S, offsets differing by delta (mod S) with
                // delta >= bytes of the first and delta + bytes of the second <= S
                const long long S = st[i][1];
                if (S != st[1 - i][k] || S <= 0) return OESS_EINVAL;
                const long long diff = (long long)(lo[1 - i][k] - lo[i][1]);
                const long long delta = ((diff % S) + S) % S;
                if (!(delta >= cb[i][1] && delta + cb[1 - i][k] <= S)) return OESS_EINVAL;
            }
    }
    ConvArgs args[2];
    bool grouped = n == 2;
    for (int i = 0; i < n && grouped; ++i)
        grouped = conv_fwd_impl(d[i].in, d[i].in_pix_stride, d[i].B, d[i].H, d[i].W, d[i].Cin, d[i].w_packed, d[i].bias, d[i].Cout, 5, 5, 2,
                                2, 1, d[i].relu, nullptr, 0, d[i].out, nullptr, d[i].out_pix_stride, nullptr, nullptr, stream, nullptr, 0,
                                nullptr, &args[i]) == OESS_OK;
    if (!grouped) {
        for (int i = 0; i < n; ++i) {
            const int rc = conv_fwd_impl(d[i].in, d[i].in_pix_stride, d[i].B, d[i].H, d[i].W, d[i].Cin, d[i].w_packed, d[i].bias,
                                         d[i].Cout, 5, 5, 2, 2, 1, d[i].relu, nullptr, 0, d[i].out, nullptr, d[i].out_pix_stride, nullptr,
                                         nullptr, stream);
            if (rc != OESS_OK) return rc;
        }
        return OESS_OK;
    }
    S2Group g;
    memset(&g, 0, sizeof(g));
    const int first = args[1].Cin > args[0].Cin ? 1 : 0;           // longest K first
    g.a[0] = args[first]; g.a[1] = args[1 - first];
    g.start8[0] = 0;
    g.start8[1] = (g.a[0].tiles_m * g.a[0].tiles_n + 7) / 8;
    g.start8[2] = g.start8[1] + (g.a[1].tiles_m * g.a[1].tiles_n + 7) / 8;
    hipLaunchKernelGGL(conv5x5s2_halo_group_kernel, dim3(8 * g.start8[2]), dim3(256), S2_LDS, (hipStream_t)stream, g);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

static int e2vid_head_enc0_launch(const S2Head& hd, int B, int H, int W, const void* enc_w_packed, const float* enc_bias, int enc_relu,
                                  void* out, long long out_pix_stride, oess_stream_t stream) {
    conv_set_attrs();
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.w = (const uint16_t*)enc_w_packed; a.bias = enc_bias; a.out = (uint16_t*)out; a.out_pix_stride = out_pix_stride;
    a.B = B; a.H = H; a.W = W; a.Cin = 32; a.Cout = 64; a.R = 5; a.S = 5; a.stride = 2; a.pad = 2; a.dil = 1;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.Kpad = (25 * 32 + BK - 1) / BK * BK;
    a.M = B * a.Ho * a.Wo; a.relu = enc_relu;
    a.tiles_n = 1;
    a.tiles_m = B * ((a.Ho + S2_PH - 1) / S2_PH) * ((a.Wo + S2_PW - 1) / S2_PW);
    hipLaunchKernelGGL((conv5x5s2_halo_kernel<true>), dim3(a.tiles_m), dim3(256), S2_LDS_FUSED, (hipStream_t)stream, a, hd);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

int oess_e2vid_events_head_enc0_bf16(const float* events, int B, int Ctot, int c0, int Cs, int H, int W, const double* stats,
                                     int normalize, const void* head_w_packed, const float* head_bias, int head_relu,
                                     const void* enc_w_packed, const float* enc_bias, int enc_relu, void* out,
                                     long long out_pix_stride, oess_stream_t stream) {
    if (!events || !head_w_packed || !enc_w_packed || !out || B <= 0 || H <= 0 || W <= 0 || Ctot <= 0 || c0 < 0 || Cs <= 0 ||
        Cs > 5 || c0 + Cs > Ctot || (normalize && !stats) || (out_pix_stride & 7) || out_pix_stride < 64 || (((uintptr_t)out) & 15) ||
        (unsigned)head_relu > 1u || (unsigned)enc_relu > 1u)
        return OESS_EINVAL;
    S2Head hd{nullptr, 8, (const uint16_t*)head_w_packed, head_bias, head_relu, events, Ctot, c0, Cs, normalize, stats};
    return e2vid_head_enc0_launch(hd, B, H, W, enc_w_packed, enc_bias, enc_relu, out, out_pix_stride, stream);
}

int oess_e2vid_head_enc0_bf16(const void* x8, long long x8_pix_stride, int B, int H, int W, const void* head_w_packed,
                              const float* head_bias, int head_relu, const void* enc_w_packed, const float* enc_bias, int enc_relu,
                              void* out, long long out_pix_stride, oess_stream_t stream) {
    if (!x8 || !head_w_packed || !enc_w_packed || !out || B <= 0 || H <= 0 || W <= 0 || (x8_pix_stride & 7) || x8_pix_stride < 8 ||
        (out_pix_stride & 7) || out_pix_stride < 64 || (((uintptr_t)out) & 15) || (unsigned)head_relu > 1u || (unsigned)enc_relu > 1u)
        return OESS_EINVAL;
    if ((((long long)B * H * W - 1) * x8_pix_stride + 8) * 2 >= 0x7ffffff0ll) return OESS_EINVAL;     // 32-bit buffer offsets
    S2Head hd{(const uint16_t*)x8, x8_pix_stride, (const uint16_t*)head_w_packed, head_bias, head_relu, nullptr, 0, 0, 0, 0, nullptr};
    return e2vid_head_enc0_launch(hd, B, H, W, enc_w_packed, enc_bias, enc_relu, out, out_pix_stride, stream);
}

}  // extern "C"
