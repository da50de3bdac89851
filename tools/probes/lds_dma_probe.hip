// Probe: buffer_load_dwordx4 ... lds (LDS-DMA) on gfx950: destination = M0 base + lane*16; do out-of-range lanes
// write zeros (buffer OOB semantics) or leave LDS untouched?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const uint32_t* in, uint32_t nbytes, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* l = (uint32_t*)smem;
    for (int i = threadIdx.x; i < 1024; i += 256) l[i] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nbytes, 0x00020000);
    uint32_t voff = (255 - threadIdx.x) * 16;            // reversed source order: lane-linear destination check
    if ((threadIdx.x % 5) == 3) voff = 0x7ffffff0u;      // out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + (threadIdx.x >> 6) * 1024), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = l[i];
}
int main() {
    uint32_t *din, *dout, h[1024], hin[1024];
    for (int i = 0; i < 1024; ++i) hin[i] = 1000 + i;
    hipMalloc(&din, 4096); hipMalloc(&dout, 4096);
    hipMemcpy(din, hin, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, din, 4096u, dout);
    hipMemcpy(h, dout, 4096, hipMemcpyDeviceToHost);
    for (int t = 0; t < 16; ++t) printf("lane %2d (src chunk %3d%s): %08x %u %u %u\n", t, 255 - t, (t % 5 == 3) ? " OOB" : "", h[t * 4], h[t * 4], h[t * 4 + 1], h[t * 4 + 3]);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        uint32_t exp0 = (t % 5 == 3) ? 0u : 1000 + (255 - t) * 4;
        if (h[t * 4] != exp0) bad++;
    }
    printf("mismatches vs (zero-fill OOB, lane-linear dest): %d\n", bad);
    return 0;
}
