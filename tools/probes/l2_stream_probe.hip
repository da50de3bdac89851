// How fast can a CU pull operands out of L2?  Two ways of streaming 16 bytes per lane from an L2-resident, L1-missing
// region (2 MB shared by all workgroups, each starting at its own offset), 256 threads per workgroup, 2 workgroups per CU:
//   mode 0  buffer_load_dwordx4 ... lds     (LDS-DMA, what the conv kernels use)
//   mode 1  global_load_dwordx4 -> VGPR     (result xor-reduced, nothing stored)
// Prints chip-wide TB/s and bytes per clock per CU (at the nominal 2.4 GHz).
// Measured (MI355X): both paths deliver the same stream: 17-22 TB/s with one workgroup per CU, 29-30 TB/s with two,
// 31-32 TB/s with four (~50 B/clk/CU); 16 KB in flight per workgroup is enough.  The LDS-DMA path is not the slower one,
// and the conv kernels' 14-15 TB/s operand stream next to their MFMA phase is half of what L2 can deliver.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ src, uint32_t region_bytes, int iters, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // INFLIGHT x 4 KB
    const int tid = threadIdx.x, wave = tid >> 6;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)region_bytes, 0x00020000);
    uint32_t off = (uint32_t)(((uint64_t)blockIdx.x * 36864u) % region_bytes);
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 v[INFLIGHT];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            const uint32_t o = (off + (uint32_t)u * 4096u + (uint32_t)tid * 16u) % region_bytes;
            if (MODE == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + u * 4096 + wave * 1024), 16, o, 0, 0, 0);
            else
                v[u] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(src) + o);
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            if (MODE == 1) acc ^= v[u];
        }
        off = (off + INFLIGHT * 4096u) % region_bytes;
    }
    if (MODE == 0) { __syncthreads(); acc = *reinterpret_cast<u32x4*>(smem + tid * 16); }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int MODE, int INFLIGHT>
float run(const uint32_t* src, uint32_t region, uint32_t* out, int grid) {
    const int iters = 2000 / INFLIGHT;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, INFLIGHT>), dim3(grid), dim3(256), INFLIGHT * 4096, 0, src, region, iters, out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, INFLIGHT>), dim3(grid), dim3(256), INFLIGHT * 4096, 0, src, region, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 5.0 * grid * (double)iters * INFLIGHT * 4096.0;
    return (float)(bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const uint32_t region = 2u << 20;
    uint32_t *src, *out;
    CK(hipMalloc(&src, region)); CK(hipMalloc(&out, 64)); CK(hipMemset(src, 1, region));
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
        const int grid = 256 * wgs_per_cu;
        printf("%d workgroup(s) per CU:\n", wgs_per_cu);
        float r;
#define ROW(M, N) r = run<M, N>(src, region, out, grid); printf("  mode %d, %2d x 4 KB in flight per workgroup: %6.2f TB/s = %5.1f B/clk/CU\n", M, N, r, r * 1e12 / 256 / 2.4e9);
        ROW(0, 4) ROW(0, 8) ROW(0, 16)
        ROW(1, 4) ROW(1, 8) ROW(1, 16)
    }
    return 0;
}
