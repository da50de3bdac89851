// Weight gradient of a 2-D convolution on gfx950 (bf16 operands, fp32 MFMA accumulate, fp32 result).
//
//   dW[co][ci][r][s] = sum over (b, oy, ox) of  dY[b,oy,ox,co] * X[b, oy*stride-pad+r*dil, ox*stride-pad+s*dil, ci]
//
// GEMM view: D[co][kk] = sum_p A[co][p] * Bm[p][kk], kk = (r, s, ci), reduction over output pixels p.
// Both operands are stored pixel-major in HBM (NHWC), i.e. with the REDUCTION index strided, so the MFMA
// fragments (8 consecutive k per lane) need a transpose.  It is done by the LDS transpose-read
// ds_read_b64_tr_b16 (semantics probed on MI355X, tools/probes/tr_probe.hip: within a 16-lane group, lane
// i supplies the address of 4 contiguous bf16 and receives element (i&3) of lanes (i>>2)+{0,4,8,12}); with
// lane address = &T[k0 + (i>>2)][n0 + 4*(i&3)] lane i receives T[k0..k0+3][n0+i]: a 4x16 transposed block.
// LDS tiles keep the natural [pixel][channel] order (row pitch 320 B = 64 mod 256: conflict-free tr reads).
//
// Workgroup: 128 (co) x 128 (kk) tile, 4 waves of 64x64, K-step = 64 output pixels of ONE output row
// (no per-pixel div/mod), split-K over output rows, fp32 atomics into the OIHW gradient (pre-zeroed).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "oess.h"
#include "oess_common.h"

namespace {
using namespace oess;
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

constexpr int TM = 128, TN = 128, KP = 64;
constexpr int PITCH = 320;                 // bytes per pixel row of an LDS tile (128 ch * 2 B + 64 B skew)
constexpr int WG = 256;

struct WgradArgs {
    const uint16_t* x;  long long xps;      // input activations NHWC bf16
    const uint16_t* dy; long long dps;      // output gradient NHWC bf16
    float* part;                            // split-K partials [splits][tiles_m*128][tiles_n*128] fp32 (plain stores)
    int B, H, W, Cin, Cin_x;                // Cin = weight's input channels; Cin_x = channels present in x (>= Cin, %8 == 0)
    int Ho, Wo, Cout;
    int R, S, stride, pad, dil;
    int Kdim;                               // R*S*Cin_x
    int rows_total, rows_per_split;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ bf16x4_t tr_read(uint32_t lds_addr) {
    bf16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    return v;
}

// TMV = output channels per tile: 128 (2 x 2 waves of 64 x 64), 64 (2 x 2 waves of 32 x 64) or 32 (1 x 4 waves of
// 32 x 32).  The narrow tiles are for the decoder's 32 / 64-channel layers at 440 x 640 and 220 x 320, where a 128-row
// tile spends 75 % / 50 % of its MFMAs on zero rows and the kernel was MFMA bound on padding (370 us for 41 GFLOP).
template <int TMV>
__global__ __launch_bounds__(WG) void conv_wgrad_kernel(WgradArgs a) {
    constexpr int WAVES_M = (TMV == 32) ? 1 : 2, WAVES_N = 4 / WAVES_M;
    constexpr int WMV = TMV / WAVES_M, WNV = TN / WAVES_N;       // wave tile
    constexpr int MT = WMV / 32, NT = WNV / 32;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KP * PITCH];
    unsigned char* lA = smem;                      // dY tile  [KP pixels][TMV co]
    unsigned char* lB = smem + KP * PITCH;         // X  tile  [KP pixels][128 kk]
    const int tile = blockIdx.x;
    const int tile_n = tile % a.tiles_n, tile_m = tile / a.tiles_n;
    const int co0 = tile_m * TMV, kk0 = tile_n * TN;
    const int split = blockIdx.y;
    const int row_beg = split * a.rows_per_split;
    int row_end = row_beg + a.rows_per_split;
    if (row_end > a.rows_total) row_end = a.rows_total;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // ---- global->LDS staging roles: 16 chunk columns x 16 pixel rows per pass, 4 passes
    const int ccol = tid & 15, prow = tid >> 4;
    // A (dY): channel chunk
    const int a_co = co0 + ccol * 8;
    const bool a_ok = a_co < a.Cout && ccol * 8 < TMV;            // Cout % 8 == 0 is required by the host wrapper
    // B (X): fixed (tap, channel chunk) of this thread
    const int kk = kk0 + ccol * 8;
    const int cpt = a.Cin_x;                        // kk = tap * Cin_x + ci
    const int tap = kk / cpt, ci0 = kk - tap * cpt;
    const bool b_ok = kk < a.Kdim;
    const int r = tap / a.S, s = tap - r * a.S;
    const int dyo = r * a.dil - a.pad, dxo = s * a.dil - a.pad;

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // tr-read lane geometry (see header): 16-lane group g4, i0 = 16*(g4&1), k-offset = 8*(g4>>1)
    const int g4 = lane >> 4, li = lane & 15;
    const uint32_t ldsA = (uint32_t)(uintptr_t)lA, ldsB = (uint32_t)(uintptr_t)lB;
    const uint32_t tr_row = (uint32_t)((g4 >> 1) * 8 + (li >> 2));
    const uint32_t tr_col = (uint32_t)((g4 & 1) * 16 + (li & 3) * 4);

    const u32x4_t zero4 = {0u, 0u, 0u, 0u};
    const int spr = (a.Wo + KP - 1) / KP;                    // K-steps per output row
    const int nsteps = (row_end - row_beg) * spr;
    u32x4_t ra[4], rb[4];
#define OESS_WG_LOAD(STEP)                                                                                          \
    {                                                                                                               \
        const int row_ = row_beg + (STEP) / spr, ox0_ = ((STEP) % spr) * KP;                                        \
        const int b_ = row_ / a.Ho, oy_ = row_ - b_ * a.Ho;                                                         \
        const int iy_ = oy_ * a.stride + dyo;                                                                       \
        const bool iy_ok_ = iy_ >= 0 && iy_ < a.H;                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                             \
            const int ox = ox0_ + prow + 16 * i;                                                                    \
            const bool pv = ox < a.Wo;                                                                              \
            const long long opix = ((long long)b_ * a.Ho + oy_) * a.Wo + (pv ? ox : 0);                             \
            const u32x4_t* pa = reinterpret_cast<const u32x4_t*>(a.dy + opix * a.dps + (a_ok ? a_co : 0));         \
            ra[i] = (pv && a_ok) ? *pa : zero4;                                                                     \
            const int ix = ox * a.stride + dxo;                                                                     \
            const bool ok = pv && b_ok && iy_ok_ && ix >= 0 && ix < a.W;                                            \
            const long long ipix = ((long long)b_ * a.H + (ok ? iy_ : 0)) * a.W + (ok ? ix : 0);                    \
            const u32x4_t* pb = reinterpret_cast<const u32x4_t*>(a.x + ipix * a.xps + (b_ok ? ci0 : 0));           \
            rb[i] = ok ? *pb : zero4;                                                                               \
        }                                                                                                           \
    }
    if (nsteps > 0) OESS_WG_LOAD(0)
    for (int step = 0; step < nsteps; ++step) {
        __syncthreads();                            // previous K-step's LDS reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pr = prow + 16 * i;
            if (ccol * 8 < TMV) *reinterpret_cast<u32x4_t*>(lA + pr * PITCH + ccol * 16) = ra[i];
            *reinterpret_cast<u32x4_t*>(lB + pr * PITCH + ccol * 16) = rb[i];
        }
        __syncthreads();
        if (step + 1 < nsteps) OESS_WG_LOAD(step + 1)   // next slab's HBM loads fly under the MFMAs below
        {
#pragma unroll
            for (int ks = 0; ks < KP / 16; ++ks) {
                bf16x4_t al[MT], ah[MT], bl[NT], bh[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const uint32_t base = ldsA + (uint32_t)(ks * 16 + tr_row) * PITCH + (uint32_t)(wm * WMV + i * 32 + tr_col) * 2;
                    al[i] = tr_read(base);
                    ah[i] = tr_read(base + 4 * PITCH);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const uint32_t base = ldsB + (uint32_t)(ks * 16 + tr_row) * PITCH + (uint32_t)(wn * WNV + j * 32 + tr_col) * 2;
                    bl[j] = tr_read(base);
                    bh[j] = tr_read(base + 4 * PITCH);
                }
                // the "+v" operands tie every later use of the results to this wait (hipcc does not track
                // inline-asm LDS reads: cdna_hip_programming.md 5.4 rule 18)
                if constexpr (MT == 2 && NT == 2)
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1])
                                 :: "memory");
                else if constexpr (MT == 1 && NT == 2)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al[0]), "+v"(ah[0]), "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]) :: "memory");
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al[0]), "+v"(ah[0]), "+v"(bl[0]), "+v"(bh[0]) :: "memory");
                bf16x8_t fa[MT], fb[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[i] = __builtin_shufflevector(al[i], ah[i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[j] = __builtin_shufflevector(bl[j], bh[j], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
    }
#undef OESS_WG_LOAD
    // ---- epilogue: plain coalesced stores of the partial tile (no atomics); reduced by wgrad_reduce_kernel
    const int ldn = a.tiles_n * TN;
    float* P = a.part + (size_t)split * a.tiles_m * TMV * ldn;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int kq = kk0 + wn * WNV + j * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wm * WMV + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                P[(size_t)co * ldn + kq] = acc[i][j][e];
            }
        }
}

// =================================================================================================
// 128-row tiles, LDS-DMA form.  Same tile, same K-step (64 output pixels of one output row), same partial layout as
// conv_wgrad_kernel<128>; what changes is how the two operand tiles reach LDS and how the fragments are read:
//  * both tiles travel HBM/L2 -> LDS by `buffer_load_dwordx4 ... lds` (out-of-range offsets write the zero padding), into a
//    2-stage ring, ONE raw barrier per K-step, the next step's fetch issued right behind it (the register-staged form paid
//    8 ds_write_b128 per thread and two __syncthreads per step: its LDS write port time alone, 32 KB at ~79 B/clk, was 0.8 of
//    the step's MFMA time);
//  * rows are 256 bytes with no skew (a wave-level DMA instruction fills four whole rows); chunk c of pixel row r lives at slot
//    c ^ ((r & 3) << 2), which makes the 32 lanes of a ds_read_b64_tr_b16 group (4 rows x 4 chunks) hit all 64 banks once;
//  * fragments of k-step ks + 1 are read under the MFMAs of k-step ks.
// LDS 2 x 32 KB -> two workgroups per CU.
// =================================================================================================
// TMV = 128 / 64 / 32 output channels per tile as in conv_wgrad_kernel (2 x 2 waves of 64 x 64 / 32 x 64, 1 x 4 waves of 32 x 32):
// the dY tile keeps its 256-byte rows, lanes whose chunk lies beyond TMV channels fetch nothing (zero fill).
template <int TMV>
__global__ __launch_bounds__(WG, 2) void conv_wgrad_dma_kernel(WgradArgs a) {
    constexpr int WAVES_M = (TMV == 32) ? 1 : 2, WAVES_N = 4 / WAVES_M;
    constexpr int WMV = TMV / WAVES_M, WNV = TN / WAVES_N;
    constexpr int MT = WMV / 32, NT = WNV / 32;
    constexpr int ROWB = 256, TILE_B = KP * ROWB, STAGE_B = 2 * TILE_B;         // A tile + B tile per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tile = blockIdx.x;
    const int tile_n = tile % a.tiles_n, tile_m = tile / a.tiles_n;
    const int co0 = tile_m * TMV, kk0 = tile_n * TN;
    const int split = blockIdx.y;
    const int row_beg = split * a.rows_per_split;
    int row_end = row_beg + a.rows_per_split;
    if (row_end > a.rows_total) row_end = a.rows_total;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const long long x_bytes = (((long long)a.B * a.H * a.W - 1) * a.xps + a.Cin_x) * 2;
    const long long d_bytes = (((long long)a.B * a.Ho * a.Wo - 1) * a.dps + a.Cout) * 2;
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)d_bytes, 0x00020000);

    // ---- DMA roles: instruction i of this wave fills pixel rows (wave*4 + i)*4 + (lane >> 4), slot = lane & 15
    const int slot = lane & 15, lr = lane >> 4;
    const int csrc = slot ^ (lr << 2);                       // (row & 3) == lr for every row this lane fills
    const int a_co = co0 + csrc * 8;
    const bool a_ok = a_co < a.Cout && csrc * 8 < TMV;
    const int kk = kk0 + csrc * 8;
    const int tap = kk / a.Cin_x, ci0 = kk - tap * a.Cin_x;
    const bool b_ok = kk < a.Kdim;
    const int r = tap / a.S, s = tap - r * a.S;
    const int dyo = r * a.dil - a.pad, dxo = s * a.dil - a.pad;
    // K-steps walk the split's output pixels in FLAT order, 64 at a time across row (and image) boundaries: a step is always
    // full except the split's last one (one-row steps waste 37 % of the loads and MFMAs on 80- and 40-pixel-wide maps).
    // Each lane carries the (image, row, column) of its four pixel rows and advances them by 64 pixels per step.
    const long long pix_beg = (long long)row_beg * a.Wo, pix_end = (long long)row_end * a.Wo;
    const int nsteps = (int)((pix_end - pix_beg + KP - 1) / KP);
    int pb[4], py[4], px[4];
    long long left[4];                                       // pixels from this lane's pixel row to the end of the split
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pr = (wave * 4 + i) * 4 + lr;
        const long long p = pix_beg + pr;
        const long long row = p / a.Wo;
        px[i] = (int)(p - row * a.Wo);
        pb[i] = (int)(row / a.Ho);
        py[i] = (int)(row - (long long)pb[i] * a.Ho);
        left[i] = pix_end - p;
    }

    auto issue = [&](int step) {
        unsigned char* st = smem + (step & 1) * STAGE_B;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool pv = left[i] > 0;
            const int ox = px[i], oy_ = py[i], b_ = pb[i];
            const unsigned va = (pv && a_ok) ? (unsigned)(((((long long)b_ * a.Ho + oy_) * a.Wo + ox) * a.dps + a_co) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (__attribute__((address_space(3))) void*)(st + (wave * 4 + i) * 1024), 16, va, 0, 0, 0);
            const int iy = oy_ * a.stride + dyo, ix = ox * a.stride + dxo;
            const bool ok = pv && b_ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned vb = ok ? (unsigned)(((((long long)b_ * a.H + iy) * a.W + ix) * a.xps + ci0) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(st + TILE_B + (wave * 4 + i) * 1024), 16, vb, 0, 0, 0);
            // advance this pixel row by one K-step
            left[i] -= KP;
            px[i] += KP;
            while (px[i] >= a.Wo) {
                px[i] -= a.Wo;
                if (++py[i] == a.Ho) { py[i] = 0; ++pb[i]; }
            }
        }
    };

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // ---- tr-read geometry (file header): 16-lane group g4, row = 8*(g4>>1) + (li>>2), column = 16*(g4&1) + 4*(li&3)
    const int g4 = lane >> 4, li = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t tr_row = (uint32_t)((g4 >> 1) * 8 + (li >> 2));
    const int tr_col = (g4 & 1) * 16 + (li & 3) * 4;         // element column inside a 32-wide sub-tile
    const uint32_t sw = (uint32_t)(((li >> 2) & 3) << 2);     // (row & 3) << 2: rows are ks*16 + tr_row (+4)
    uint32_t offA[MT], offB[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int col = wm * WMV + i * 32 + tr_col;          // element column of the A tile (co)
        offA[i] = tr_row * ROWB + ((((uint32_t)col >> 3) ^ sw) << 4) + (uint32_t)((col & 7) * 2);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = wn * WNV + j * 32 + tr_col;          // element column of the B tile (kk)
        offB[j] = (uint32_t)TILE_B + tr_row * ROWB + ((((uint32_t)col >> 3) ^ sw) << 4) + (uint32_t)((col & 7) * 2);
    }
#define OESS_WTR(DST, ADDR) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(DST) : "v"(ADDR) : "memory")
#define OESS_WREAD(AL, AH, BL, BH, KS)                                                                               \
    {                                                                                                               \
        const uint32_t kb_ = stage_ + (uint32_t)((KS) * 16 * ROWB);                                                 \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) { OESS_WTR(AL[i], kb_ + offA[i]); OESS_WTR(AH[i], kb_ + offA[i] + 4 * ROWB); } \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) { OESS_WTR(BL[j], kb_ + offB[j]); OESS_WTR(BH[j], kb_ + offB[j] + 4 * ROWB); } \
    }
    // wait for every outstanding LDS read; the "+v" operands tie later uses of the fragments to the wait
#define OESS_WWAIT(AL, AH, BL, BH)                                                                                  \
    {                                                                                                               \
        if constexpr (MT == 2 && NT == 2)                                                                           \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(AL[0]), "+v"(AH[0]), "+v"(AL[1]), "+v"(AH[1]), "+v"(BL[0]), "+v"(BH[0]), "+v"(BL[1]), "+v"(BH[1]) :: "memory"); \
        else if constexpr (MT == 1 && NT == 2)                                                                      \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(AL[0]), "+v"(AH[0]), "+v"(BL[0]), "+v"(BH[0]), "+v"(BL[1]), "+v"(BH[1]) :: "memory"); \
        else                                                                                                        \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(AL[0]), "+v"(AH[0]), "+v"(BL[0]), "+v"(BH[0]) :: "memory");  \
    }
#define OESS_WMMA(AL, AH, BL, BH)                                                                                   \
    {                                                                                                               \
        bf16x8_t fa_[MT], fb_[NT];                                                                                  \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) fa_[i] = __builtin_shufflevector(AL[i], AH[i], 0, 1, 2, 3, 4, 5, 6, 7); \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) fb_[j] = __builtin_shufflevector(BL[j], BH[j], 0, 1, 2, 3, 4, 5, 6, 7); \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                              \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                          \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i], fb_[j], acc[i][j], 0, 0, 0);             \
    }

    if (nsteps > 0) issue(0);
    for (int step = 0; step < nsteps; ++step) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // step's tiles are in LDS for every wave; the other stage is free
        if (step + 1 < nsteps) issue(step + 1);
        const uint32_t stage_ = lds0 + (uint32_t)((step & 1) * STAGE_B);
        bf16x4_t al0[MT], ah0[MT], bl0[NT], bh0[NT], al1[MT], ah1[MT], bl1[NT], bh1[NT];
        __builtin_amdgcn_s_setprio(3);
        // at most 8 LDS reads outstanding (lgkmcnt is a 4-bit counter): wait for k-step ks, issue ks + 1, multiply ks
        OESS_WREAD(al0, ah0, bl0, bh0, 0)
        OESS_WWAIT(al0, ah0, bl0, bh0)
        OESS_WREAD(al1, ah1, bl1, bh1, 1)
        OESS_WMMA(al0, ah0, bl0, bh0)
        OESS_WWAIT(al1, ah1, bl1, bh1)
        OESS_WREAD(al0, ah0, bl0, bh0, 2)
        OESS_WMMA(al1, ah1, bl1, bh1)
        OESS_WWAIT(al0, ah0, bl0, bh0)
        OESS_WREAD(al1, ah1, bl1, bh1, 3)
        OESS_WMMA(al0, ah0, bl0, bh0)
        OESS_WWAIT(al1, ah1, bl1, bh1)
        OESS_WMMA(al1, ah1, bl1, bh1)
        __builtin_amdgcn_s_setprio(0);
    }
#undef OESS_WTR
#undef OESS_WREAD
#undef OESS_WWAIT
#undef OESS_WMMA
    // ---- epilogue: plain stores of the partial tile; reduced by wgrad_reduce_kernel
    const int ldn = a.tiles_n * TN;
    float* P = a.part + (size_t)split * a.tiles_m * TMV * ldn;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int kq = kk0 + wn * WNV + j * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wm * WMV + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                P[(size_t)co * ldn + kq] = acc[i][j][e];
            }
        }
}

// =================================================================================================
// Narrow layers at full resolution (the decoder's 3x3 convs with Cout <= 64: 440x640 64->32, 220x320 128->64): the WHOLE
// gradient of a 64-channel input chunk -- TMV output channels x (9 taps x 64 ci) -- lives in one workgroup's accumulators.
// The 128-wide (tap, channel) tiling read dY once per tile and X once per tap pair (26 FLOP per byte moved to LDS for TMV = 32).
// Here a workgroup walks DOWN a 64-pixel-wide strip of one image; a K-step = one output row of the strip fetches one row of dY
// (64 px x TMV) and ONE new row of X (66 px x 64 ch; rows oy-1, oy are already in LDS: ring of 6 rows, dY ring of 3, two steps
// in flight), and the nine taps' B operands are shifted windows of the three resident X rows: X is read once.
// LDS rows are 128 bytes (one pixel x 64 channels); chunk c of pixel row p sits at slot c ^ (((p >> 1) & 1) << 2): any four
// consecutive pixel rows x four chunks (one ds_read_b64_tr_b16 lane group, whatever the tap shift) cover all 64 banks once
// (SQ_LDS_BANK_CONFLICT = 0).  Four waves share the 18 (tap, 32-channel) sub-tiles round-robin; waves 2 and 3 carry a fifth,
// unused one so that the hot loop has no wave-dependent branch (its first form spent 118 scalar branches per step on them).
// Work unit = (64-channel chunk, strip, row range); one partial tile per unit.
// =================================================================================================
constexpr int ST_XROW = 9 * 1024;                            // 72 pixel-row slots of 128 B (66 used): 9 wave-level DMA instructions
constexpr int ST_XRING = 6, ST_DRING = 3, ST_DROW = KP * 128;
constexpr int ST_LDS = ST_XRING * ST_XROW + ST_DRING * ST_DROW;   // 79 872 -> two workgroups per CU

template <int TMV>
__global__ __launch_bounds__(WG, 2) void conv_wgrad_strip_kernel(WgradArgs a, int row_splits, int rows_per) {
    constexpr int MT = TMV / 32;
    constexpr int NSUB = 18, SPW = (NSUB + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int chunk = blockIdx.x;
    const int ci_base = chunk * 64;
    const int unit = blockIdx.y;                             // (strip, row range)
    const int strip = unit / row_splits, rs = unit - strip * row_splits;
    const int spr = (a.Wo + KP - 1) / KP;
    const int b_ = strip / spr, ox0 = (strip - b_ * spr) * KP;
    const int r0 = rs * rows_per;
    int r1 = r0 + rows_per;
    if (r1 > a.Ho) r1 = a.Ho;
    const int nsteps = r1 - r0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const long long x_bytes = (((long long)a.B * a.H * a.W - 1) * a.xps + a.Cin_x) * 2;
    const long long d_bytes = (((long long)a.B * a.Ho * a.Wo - 1) * a.dps + a.Cout) * 2;
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)d_bytes, 0x00020000);

    const int slot = lane & 7, lr = lane >> 3;
    // dY row: instructions wave*2 + {0,1}: pixels (wave*2 + i)*8 + lr
    int a_px[2], a_coff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int px = (wave * 2 + i) * 8 + lr;
        const int c = slot ^ (((px >> 1) & 1) << 2);
        a_px[i] = px;
        a_coff[i] = (c * 8 < TMV && c * 8 < a.Cout) ? c * 8 : -1;
    }
    // X row: instructions n = wave + 4*i (n < 9): halo pixels q = n*8 + lr (q = 0 is column ox0 - 1)
    int h_q[3], h_coff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = (wave + 4 * i) * 8 + lr;
        const int c = slot ^ (((q >> 1) & 1) << 2);
        h_q[i] = q;
        h_coff[i] = (q < KP + 2 && ci_base + c * 8 < a.Cin_x) ? ci_base + c * 8 : -1;
    }
    auto issue_x = [&](int iy) {                             // image row iy (may be -1 or H: zero fill) -> ring slot (iy + 1) % 6
        unsigned char* dst = smem + ((iy + 1) % ST_XRING) * ST_XROW;
        const bool row_ok = (unsigned)iy < (unsigned)a.H;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (wave + 4 * i >= 9) continue;
            const int ix = ox0 - 1 + h_q[i];
            const bool ok = row_ok && h_coff[i] >= 0 && (unsigned)ix < (unsigned)a.W;
            const unsigned vb = ok ? (unsigned)(((((long long)b_ * a.H + iy) * a.W + ix) * a.xps + h_coff[i]) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(dst + (wave + 4 * i) * 1024), 16, vb, 0, 0, 0);
        }
    };
    auto issue_d = [&](int oy) {                             // output row oy -> ring slot oy % 3 (rows >= r1: zero fill, never used)
        unsigned char* dst = smem + ST_XRING * ST_XROW + (oy % ST_DRING) * ST_DROW;
        const bool row_ok = oy < r1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ox = ox0 + a_px[i];
            const bool ok = row_ok && a_coff[i] >= 0 && ox < a.Wo;
            const unsigned va = ok ? (unsigned)(((((long long)b_ * a.Ho + oy) * a.Wo + ox) * a.dps + a_coff[i]) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (__attribute__((address_space(3))) void*)(dst + (wave * 2 + i) * 1024), 16, va, 0, 0, 0);
        }
    };

    f32x16_t acc[SPW][MT];
#pragma unroll
    for (int t = 0; t < SPW; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][i][e] = 0.0f;

    const int g4 = lane >> 4, li = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int tr_row = (g4 >> 1) * 8 + (li >> 2);
    const int tr_col = (g4 & 1) * 16 + (li & 3) * 4;
    uint32_t offA[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int col = i * 32 + tr_col;
        offA[i] = (uint32_t)(tr_row * 128 + ((((col >> 3) ^ (((tr_row >> 1) & 1) << 2))) << 4) + (col & 7) * 2);
    }
    uint32_t offB[3][2];
#pragma unroll
    for (int sx = 0; sx < 3; ++sx)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int pr = tr_row + sx, col = hf * 32 + tr_col;
            offB[sx][hf] = (uint32_t)(pr * 128 + ((((col >> 3) ^ (((pr >> 1) & 1) << 2))) << 4) + (col & 7) * 2);
        }
#define OESS_FTR2(DST, ADDR) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(DST) : "v"(ADDR) : "memory")
    // this wave's sub-tiles t = wave + 4k: tap = t >> 1 (X row r = tap / 3, shift sx = tap % 3), channel half = t & 1; t >= 18 (the
    // fifth sub-tile of waves 2, 3) is computed on sub-tile 0's operands and dropped at the end
    int sub_r[SPW];
    uint32_t sub_off[SPW];
#pragma unroll
    for (int k = 0; k < SPW; ++k) {
        const int t = wave + 4 * k < NSUB ? wave + 4 * k : 0;
        const int tp = t >> 1, r = tp / 3, sx = tp - r * 3, hf = t & 1;
        sub_r[k] = r;
        sub_off[k] = hf ? (sx == 0 ? offB[0][1] : (sx == 1 ? offB[1][1] : offB[2][1])) : (sx == 0 ? offB[0][0] : (sx == 1 ? offB[1][0] : offB[2][0]));
    }

    // prologue: rows r0-1, r0 | group 0 = (X row r0+1, dY r0) | group 1 = (X row r0+2, dY r0+1)
    issue_x(r0 - 1); issue_x(r0);
    issue_x(r0 + 1); issue_d(r0);
    issue_x(r0 + 2); issue_d(r0 + 1);
    for (int step = 0; step < nsteps; ++step) {
        const int oy = r0 + step;
        // group `step` has landed; group step+1 (this wave: 3 or 2 X instructions + 2 dY instructions) may stay in flight
        if (wave == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // ... for every wave; the slots of rows oy-3 / dY oy-1 are free
        issue_x(oy + 3); issue_d(oy + 2);
        const uint32_t abase = lds0 + (uint32_t)(ST_XRING * ST_XROW + (oy % ST_DRING) * ST_DROW);
        uint32_t xb[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) xb[r] = lds0 + (uint32_t)(((oy + r) % ST_XRING) * ST_XROW);      // image row oy - 1 + r
        __builtin_amdgcn_s_setprio(3);
        bf16x4_t al[2][MT], ah[2][MT], bl[2][SPW], bh[2][SPW];
        uint32_t bb[SPW];                                    // this step's B base of every sub-tile (ring slot of its X row)
#pragma unroll
        for (int k = 0; k < SPW; ++k) bb[k] = (sub_r[k] == 0 ? xb[0] : (sub_r[k] == 1 ? xb[1] : xb[2])) + sub_off[k];
        auto read_ks = [&](int set, int ks) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                OESS_FTR2(al[set][i], abase + offA[i] + (uint32_t)(ks * 16 * 128));
                OESS_FTR2(ah[set][i], abase + offA[i] + (uint32_t)(ks * 16 * 128 + 4 * 128));
            }
#pragma unroll
            for (int k = 0; k < SPW; ++k) {
                OESS_FTR2(bl[set][k], bb[k] + (uint32_t)(ks * 16 * 128));
                OESS_FTR2(bh[set][k], bb[k] + (uint32_t)(ks * 16 * 128 + 4 * 128));
            }
        };
        auto wait_all = [&](int set) __attribute__((always_inline)) {
            if constexpr (MT == 2)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al[set][0]), "+v"(ah[set][0]), "+v"(al[set][1]), "+v"(ah[set][1]), "+v"(bl[set][0]), "+v"(bh[set][0]),
                             "+v"(bl[set][1]), "+v"(bh[set][1]), "+v"(bl[set][2]), "+v"(bh[set][2]), "+v"(bl[set][3]), "+v"(bh[set][3]), "+v"(bl[set][4]), "+v"(bh[set][4]) :: "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al[set][0]), "+v"(ah[set][0]), "+v"(bl[set][0]), "+v"(bh[set][0]), "+v"(bl[set][1]), "+v"(bh[set][1]),
                             "+v"(bl[set][2]), "+v"(bh[set][2]), "+v"(bl[set][3]), "+v"(bh[set][3]), "+v"(bl[set][4]), "+v"(bh[set][4]) :: "memory");
        };
        auto mma = [&](int set) __attribute__((always_inline)) {
            bf16x8_t fa[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = __builtin_shufflevector(al[set][i], ah[set][i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int k = 0; k < SPW; ++k) {
                const bf16x8_t fb = __builtin_shufflevector(bl[set][k], bh[set][k], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[k][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb, acc[k][i], 0, 0, 0);
            }
        };
        read_ks(0, 0);
        wait_all(0); read_ks(1, 1); mma(0);
        wait_all(1); read_ks(0, 2); mma(1);
        wait_all(0); read_ks(1, 3); mma(0);
        wait_all(1); mma(1);
        __builtin_amdgcn_s_setprio(0);
    }
#undef OESS_FTR2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the look-ahead groups beyond the last row (zero fills) have landed
    const int ldn = a.tiles_n * TN;
    float* P = a.part + (size_t)unit * a.tiles_m * TMV * ldn;
#pragma unroll
    for (int k = 0; k < SPW; ++k) {
        const int t = wave + 4 * k;
        if (t >= NSUB) continue;
        const int tp = t >> 1, hf = t & 1;
        const int ci = ci_base + hf * 32 + (lane & 31);
        if (ci >= a.Cin_x) continue;
        const int kq = tp * a.Cin_x + ci;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                P[(size_t)co * ldn + kq] = acc[k][i][e];
            }
    }
}

// dW[co][ci][r][s] = sum over splits of part[split][co][(r,s,ci)]
// Threads walk the PARTIAL layout (kq = (r,s,ci) contiguous): the `splits` reads per element are coalesced and independent
// (unrolled by 4); the one write per element is the scattered side.  (Walking the OIHW layout made every read of a 3x3 layer a
// Cin-strided gather: 22-25 us per launch for a few MB, 60 launches per frame2recon step.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int splits, int Mpad, int ldn,
                                                           int Cout, int Cin, int Cin_x, int RS, float* __restrict__ dw) {
    const int kdim = RS * Cin_x;
    const long long total = (long long)Cout * kdim;
    const size_t sstride = (size_t)Mpad * ldn;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int co = (int)(i / kdim), kq = (int)(i - (long long)co * kdim);
        const int tp = kq / Cin_x, ci = kq - tp * Cin_x;
        if (ci >= Cin) continue;
        const float* p = part + (size_t)co * ldn + kq;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = 0;
        for (; k + 3 < splits; k += 4) {
            s0 += p[(size_t)k * sstride]; s1 += p[(size_t)(k + 1) * sstride];
            s2 += p[(size_t)(k + 2) * sstride]; s3 += p[(size_t)(k + 3) * sstride];
        }
        for (; k < splits; ++k) s0 += p[(size_t)k * sstride];
        dw[((size_t)co * Cin + ci) * RS + tp] = (s0 + s1) + (s2 + s3);
    }
}

// Many slices of a small gradient (the strip / whole-gradient kernels: ~500 slices of 18 432 elements):
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ part, int splits, int Mpad, int ldn,
                                                                int Cout, int Cin, int Cin_x, int RS, float* __restrict__ dw) {
    // 32 consecutive (r,s,ci) elements x 8 split lanes per workgroup: lane l adds splits l, l + 8, ... in order (four independent
    // chains), the eight lane sums are added in lane order through LDS -- a fixed association whatever the launch geometry.
    // (One thread per element walked all `splits` slices itself: with the strip / whole-gradient kernels' ~500 slices of a
    // 18 432-element gradient that was 72 workgroups x 480 dependent reads = ~120 us, more than the gradient kernel.)
    __shared__ float red[8][32];
    const int kdim = RS * Cin_x;
    const long long total = (long long)Cout * kdim;
    const size_t sstride = (size_t)Mpad * ldn;
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    for (long long i0 = (long long)blockIdx.x * 32; i0 < total; i0 += (long long)gridDim.x * 32) {
        const long long i = i0 + el;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int co = 0, tp = 0, ci = Cin;
        if (i < total) {
            co = (int)(i / kdim);
            const int kq = (int)(i - (long long)co * kdim);
            tp = kq / Cin_x; ci = kq - tp * Cin_x;
            if (ci < Cin) {
                const float* p = part + (size_t)co * ldn + kq;
                int k = sl;
                for (; k + 24 < splits; k += 32) {
                    s0 += p[(size_t)k * sstride]; s1 += p[(size_t)(k + 8) * sstride];
                    s2 += p[(size_t)(k + 16) * sstride]; s3 += p[(size_t)(k + 24) * sstride];
                }
                for (; k < splits; k += 8) s0 += p[(size_t)k * sstride];
            }
        }
        red[sl][el] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (sl == 0 && i < total && ci < Cin) {
            float t = red[0][el];
#pragma unroll
            for (int l = 1; l < 8; ++l) t += red[l][el];
            dw[((size_t)co * Cin + ci) * RS + tp] = t;
        }
        __syncthreads();
    }
}

static inline size_t per_split_bytes(const WgradArgs& a, int tmv) { return (size_t)a.tiles_m * tmv * a.tiles_n * TN * sizeof(float); }

}  // namespace

extern "C" {

int oess_conv2d_wgrad_bf16(const void* x, long long x_pix_stride, int B, int H, int W, int Cin_x, const void* dy,
                           long long dy_pix_stride, int Cout, int Cin, int R, int S, int stride, int pad, int dil,
                           float* dw_oihw, void* workspace, size_t workspace_bytes, oess_stream_t stream) {
    if (!x || !dy || !dw_oihw || !workspace || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin_x < Cin || (Cin_x & 7) || Cout <= 0 ||
        (Cout & 7) || R <= 0 || S <= 0 || stride <= 0 || pad < 0 || dil <= 0 || (x_pix_stride & 7) || (dy_pix_stride & 7))
        return OESS_EINVAL;
    WgradArgs a;
    a.x = (const uint16_t*)x; a.xps = x_pix_stride; a.dy = (const uint16_t*)dy; a.dps = dy_pix_stride; a.part = (float*)workspace;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cin_x = Cin_x; a.Cout = Cout;
    a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.dil = dil;
    a.Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
    a.Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
    if (a.Ho <= 0 || a.Wo <= 0) return OESS_EINVAL;
    a.Kdim = R * S * Cin_x;
    const int tmv = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : TM);      // narrow layers: 32- / 64-row tiles (-10 % on the decoder's 32/64-channel layers)
    a.tiles_m = (Cout + tmv - 1) / tmv;
    a.tiles_n = (a.Kdim + TN - 1) / TN;
    a.rows_total = B * a.Ho;
    const int tiles = a.tiles_m * a.tiles_n;
    const int target = 512;
    const long long x_bytes = (((long long)B * H * W - 1) * x_pix_stride + Cin_x) * 2;
    const long long d_bytes = (((long long)B * a.Ho * a.Wo - 1) * dy_pix_stride + Cout) * 2;
    const bool dma = x_bytes < 0x7ffffff0ll && d_bytes < 0x7ffffff0ll;      // 32-bit buffer offsets reach both tensors
    // split-K over output rows.  Register-staged form: ~512 workgroups (1024: +3 % step time on frame2recon from the larger
    // partial-sum traffic).  LDS-DMA form (64 KB of LDS: exactly two workgroups per CU = 512 slots): whichever of floor / ceil
    // (512 / tiles) needs fewer rounds per split -- 36 tiles x 15 splits = 540 workgroups was a 1.05-round launch
    int splits = (target + tiles - 1) / tiles;
    if (dma && tiles <= target) {
        const int lo = target / tiles, hi = splits;
        auto cost = [&](int sp) { return (double)(((long long)tiles * sp + target - 1) / target) / (double)sp; };
        splits = cost(lo) < cost(hi) ? lo : hi;
    }
    const size_t per_split = (size_t)a.tiles_m * tmv * a.tiles_n * TN * sizeof(float);
    if (per_split > workspace_bytes) return OESS_ENOMEM;
    if ((size_t)splits * per_split > workspace_bytes) splits = (int)(workspace_bytes / per_split);
    if (splits > a.rows_total) splits = a.rows_total;
    if (splits < 1) splits = 1;
    a.rows_per_split = (a.rows_total + splits - 1) / splits;
    splits = (a.rows_total + a.rows_per_split - 1) / a.rows_per_split;
    hipStream_t st = (hipStream_t)stream;
    // narrow full-resolution 3x3 layers: the whole (Cout x 9 x 64-channel) gradient per workgroup, X fetched once as a halo
    // (measured against the 128-wide tiling: 440x640 64->32 390 -> 342 us, 220x320 128->64 267 -> 182; 220x320 64->64 118 -> 131:
    //  one 64-channel chunk with 64 output channels stays on the generic form)
    if (dma && R == 3 && S == 3 && stride == 1 && pad == 1 && dil == 1 && Cout <= 64 && (Cin_x % 64) == 0 && a.Wo >= KP &&
        (long long)a.rows_total * a.Wo >= 100000) {
        const int chunks = Cin_x / 64;
        const int spr = (a.Wo + KP - 1) / KP, strips = B * spr;
        int rsplit = (target / chunks) / strips;                     // row ranges per strip: at most 512 workgroups in all (one round)
        if (rsplit < 1) rsplit = 1;
        if (rsplit > a.Ho) rsplit = a.Ho;
        int rows_per = (a.Ho + rsplit - 1) / rsplit;
        rsplit = (a.Ho + rows_per - 1) / rows_per;
        const long long units = (long long)strips * rsplit;
        if (units * per_split_bytes(a, tmv) <= workspace_bytes && units <= 65535) {
            static bool attr2 = false;
            if (!attr2) {
                (void)hipFuncSetAttribute((const void*)&conv_wgrad_strip_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)&conv_wgrad_strip_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr2 = true;
            }
            if (tmv == 32) hipLaunchKernelGGL(conv_wgrad_strip_kernel<32>, dim3(chunks, (unsigned)units), dim3(WG), ST_LDS, st, a, rsplit, rows_per);
            else hipLaunchKernelGGL(conv_wgrad_strip_kernel<64>, dim3(chunks, (unsigned)units), dim3(WG), ST_LDS, st, a, rsplit, rows_per);
            splits = (int)units;
            goto reduce;
        }
    }
    {
    const dim3 grid(tiles, splits);
    if (dma) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)&conv_wgrad_dma_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)&conv_wgrad_dma_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)&conv_wgrad_dma_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr = true;
        }
        const size_t lds = (size_t)2 * 2 * KP * 256;
        if (tmv == 32) hipLaunchKernelGGL(conv_wgrad_dma_kernel<32>, grid, dim3(WG), lds, st, a);
        else if (tmv == 64) hipLaunchKernelGGL(conv_wgrad_dma_kernel<64>, grid, dim3(WG), lds, st, a);
        else hipLaunchKernelGGL(conv_wgrad_dma_kernel<128>, grid, dim3(WG), lds, st, a);
    } else {
        if (tmv == 32) hipLaunchKernelGGL(conv_wgrad_kernel<32>, grid, dim3(WG), 0, st, a);
        else if (tmv == 64) hipLaunchKernelGGL(conv_wgrad_kernel<64>, grid, dim3(WG), 0, st, a);
        else hipLaunchKernelGGL(conv_wgrad_kernel<128>, grid, dim3(WG), 0, st, a);
    }
    }
reduce:
    const long long total = (long long)Cout * a.Kdim;
    if (splits >= 16) {         // many slices: 8 split lanes per element (fixed-order combine)
        long long rg = (total + 31) / 32;
        if (rg > 8192) rg = 8192;
        hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)rg), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                           splits, a.tiles_m * tmv, a.tiles_n * TN, Cout, Cin, Cin_x, R * S, dw_oihw);
    } else {
        long long rg = (total + 255) / 256;
        if (rg > 4096) rg = 4096;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rg), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, splits,
                           a.tiles_m * tmv, a.tiles_n * TN, Cout, Cin, Cin_x, R * S, dw_oihw);
    }
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // extern "C"
