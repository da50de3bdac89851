// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS[i] = i (uint16).  Lane l passes byte address addr(l).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(uint16_t* out, int mode) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    int l = threadIdx.x;
    // mode 0: addr = l*8 bytes (each lane points at 4 consecutive elements)
    // mode 1: row-major [16 rows][64 cols] matrix (128 B rows): lane -> row (l&15)... addr = (l&15)*128 + (l>>4)*8
    // mode 2: addr = (l>>4)*128*... [k rows of 16 cols]: 4 rows x 16 cols blocks: addr = ((l>>4)*4)*32 + (l&15)*2?? must be 8B aligned -> skip
    uint32_t addr;
    if (mode == 0) addr = l * 8;
    else if (mode == 1) addr = (l & 15) * 128 + (l >> 4) * 8;
    else addr = (l & 3) * 8 + ((l >> 2) & 3) * 128 + (l >> 4) * 512;   // 4 lanes cover 16 cols of a row; 4 rows per 16-lane group (row pitch 128 B)
    uint32_t base = (uint32_t)(uintptr_t)lds;   // LDS offset (low 32 bits of the local pointer)
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]); printf("%s", (l % 4 == 3) ? "\n" : "   |"); }
    }
    return 0;
}
