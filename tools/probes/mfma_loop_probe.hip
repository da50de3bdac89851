// Probe: what does the conv main loop's COMPUTE side cost on gfx950, piece by piece?
//   mode 0: 16 x v_mfma_f32_32x32x16_bf16 per iteration, nothing else (MFMA pipe ceiling, 2 waves/SIMD)
//   mode 1: + the 16 ds_read_b128 fragment reads with the pipelined lgkmcnt waits of conv_fwd_dma_kernel
//   mode 2: mode 0 + one s_barrier per iteration
//   mode 3: mode 1 + one s_barrier per iteration  (= the conv K loop without its DMA)
//   mode 4: mode 3 with MT=4,NT=2 wave tile (128 x 64 per wave, 32 MFMA + 24 reads per iteration)
// Prints TFLOP/s per mode; WGS_PER_CU is set through the dynamic LDS size (64 KB -> 2 per CU, 128 KB -> 1).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int MODE, int MT, int NT>
__global__ __launch_bounds__(256) void k(float* out, int iters, int rnd, const uint32_t* gsrc, uint32_t gbytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 256) {
        uint32_t h = (uint32_t)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // random sign + mantissa, exponent near 1.0 (values in +-[0.5, 2)) - like normalised activations
        uint32_t v = 0x3f803f80u;
        if (rnd == 1) v = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u);
        if (rnd == 2) v = h;                                                     // every bit random (NaN/Inf/denormals included)
        if (rnd == 3) {                                                          // randn-like: geometric exponent spread below 2.0
            uint32_t h2 = h * 747796405u + 2891336453u; h2 ^= h2 >> 16;
            const uint32_t e0 = 127u - (uint32_t)min(__builtin_ctz(h2 | 0x100u), 8), e1 = 127u - (uint32_t)min(__builtin_ctz((h2 >> 16) | 0x100u), 8);
            v = (h & 0x807f807fu) | (e0 << 7) | (e1 << 23);
        }
        ((uint32_t*)smem)[i] = v;
    }
    __syncthreads();
    f32x16_t acc[MT][NT];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int rsw = ((lane & 31) >> 1) & 7, half = lane >> 5;
    uint32_t fa_off[MT], fb_off[NT];
    for (int i = 0; i < MT; ++i) fa_off[i] = (uint32_t)(((wave >> 1) * 32 * MT + i * 32 + (lane & 31)) & 127) * 128;
    for (int j = 0; j < NT; ++j) fb_off[j] = (uint32_t)(16384 + (((wave & 1) * 32 * NT + j * 32 + (lane & 31)) & 127) * 128);
    bf16x8_t fa0[MT], fb0[NT], fa1[MT], fb1[NT];
    for (int i = 0; i < MT; ++i) { fa0[i] = *(bf16x8_t*)(smem + fa_off[i] + half * 16); fa1[i] = *(bf16x8_t*)(smem + fa_off[i] + 32 + half * 16); }
    for (int j = 0; j < NT; ++j) { fb0[j] = *(bf16x8_t*)(smem + fb_off[j] + half * 16); fb1[j] = *(bf16x8_t*)(smem + fb_off[j] + 32 + half * 16); }
    constexpr bool RD = (MODE == 1 || MODE >= 3), BAR = (MODE >= 2), DMA = (MODE >= 5);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)gsrc, 0, (int)gbytes, 0x00020000);
    // mode 5: 8 LDS-DMA instructions per wave per iteration (one 32 KB stage per workgroup), 2-stage ring like dma2;
    // mode 6: same with a 3-slab-deep ring (needs 96 KB -> 1 workgroup per CU).  Source walks a gbytes window (L2 resident).
    constexpr int NST = (MODE == 6) ? 3 : 2;
    auto issue = [&](int kt) {
        if constexpr (DMA) {
            const uint32_t base = (uint32_t)(((uint64_t)(blockIdx.x * 7 + kt) * 32768u) % gbytes);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (kt % NST) * 32768 + (wave * 8 + i) * 1024),
                                                         16, base + (wave * 8 + i) * 1024 + lane * 16, 0, 0, 0);
        }
    };
    if constexpr (DMA) for (int s_ = 0; s_ < NST - 1; ++s_) issue(s_);
#define FRAG_READ(DA, DB, KS)                                                                                  \
    if constexpr (RD) {                                                                                        \
        const uint32_t sl_ = (uint32_t)((((KS) * 2 + half) ^ rsw) * 16);                                       \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                         \
            asm volatile("ds_read_b128 %0, %1" : "=v"(DA[i]) : "v"(stage_ + fa_off[i] + sl_) : "memory");      \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                         \
            asm volatile("ds_read_b128 %0, %1" : "=v"(DB[j]) : "v"(stage_ + fb_off[j] + sl_) : "memory");      \
    }
#define FRAG_MMA(SA, SB)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j)             \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SA[i], SB[j], acc[i][j], 0, 0, 0);
#define WAITF(N_, FA_, FB_)                                                                                    \
    if constexpr (RD) {                                                                                        \
        if constexpr (MT == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(FA_[0]), "+v"(FA_[1]), "+v"(FB_[0]), "+v"(FB_[1]) : "n"(N_) : "memory"); \
        else asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(FA_[0]), "+v"(FA_[1]), "+v"(FA_[2]), "+v"(FA_[3]), "+v"(FB_[0]), "+v"(FB_[1]) : "n"(N_) : "memory"); \
    }
    constexpr int NF = MT + NT;
    for (int kt = 0; kt < iters; ++kt) {
        if constexpr (DMA) { if constexpr (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        if constexpr (BAR) __builtin_amdgcn_s_barrier();
        if constexpr (DMA) issue(kt + NST - 1);
        const uint32_t stage_ = lds0 + (uint32_t)((kt % NST) * 32768);
        FRAG_READ(fa0, fb0, 0)
        FRAG_READ(fa1, fb1, 1)
        WAITF(NF, fa0, fb0)
        FRAG_MMA(fa0, fb0)
        FRAG_READ(fa0, fb0, 2)
        WAITF(NF, fa1, fb1)
        FRAG_MMA(fa1, fb1)
        FRAG_READ(fa1, fb1, 3)
        WAITF(NF, fa0, fb0)
        FRAG_MMA(fa0, fb0)
        WAITF(0, fa1, fb1)
        FRAG_MMA(fa1, fb1)
    }
    float s = 0.f;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE, int MT, int NT>
static void run(const char* name, int lds, float* d, int rnd, int iters, const uint32_t* g = nullptr, uint32_t gb = 0, int grid_override = 0, int reps = 3) {
    int grid = 256 * (lds > 80 * 1024 ? 1 : 2) * 4;
    if (grid_override) grid = grid_override;
    hipFuncSetAttribute((const void*)&k<MODE, MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, MT, NT>), dim3(grid), dim3(256), lds, 0, d, iters, rnd, g, gb);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<MODE, MT, NT>), dim3(grid), dim3(256), lds, 0, d, iters, rnd, g, gb);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double fl = (double)grid * 4 * iters * 4 * MT * NT * 32768.0;
    printf("%-44s rnd %d iters %5d lds %3d KB  %.3f ms  %.0f TFLOP/s\n", name, rnd, iters, lds / 1024, ms, fl / ms / 1e9);
}
int main(int argc, char** argv) {
    float* d; hipMalloc(&d, 4096);
    uint32_t* g; const uint32_t gb = 64u << 20; hipMalloc(&g, gb); hipMemset(g, 0x3f, gb);
    if (argc > 1) {   // conv-like launch shapes: many short workgroups
        for (int iters : {9, 36, 144, 576}) {
            run<3, 2, 2>("K loop, grid 4400 (short workgroups)", 64 * 1024, d, 1, iters, g, gb, 4400, 20);
            run<3, 2, 2>("K loop, grid 512 x 8", 64 * 1024, d, 1, iters, g, gb, 4096, 20);
            run<3, 2, 2>("K loop, grid 512", 64 * 1024, d, 1, iters * 8, g, gb, 512, 20);
        }
        return 0;
    }
    for (int rnd : {0, 1, 2, 3}) for (int iters : {2000}) {
        const int lds = 64 * 1024;
        run<0, 2, 2>("mfma only (64x64 wave tile)", lds, d, rnd, iters);
        run<3, 2, 2>("mfma + reads + barrier (= conv K loop)", lds, d, rnd, iters);
        run<4, 4, 2>("128x64 wave tile: reads + barrier", lds, d, rnd, iters);
        if (iters == 2000) {
            run<5, 2, 2>("K loop + LDS-DMA 2-stage, 2 WG/CU", lds, d, rnd, iters, g, gb);
            run<5, 2, 2>("K loop + LDS-DMA 2-stage, 16 MB window", lds, d, rnd, iters, g, 16u << 20);
            run<6, 2, 2>("K loop + LDS-DMA 3-stage, 1 WG/CU", 96 * 1024, d, rnd, iters, g, gb);
        }
    }
    return 0;
}
