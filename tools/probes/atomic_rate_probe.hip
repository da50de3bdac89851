// How fast are no-return integer atomics into a voxel-grid-sized region on gfx950?  (Round-5 question: an INPUT-stationary
// voxelizer -- 8 global int32 atomics per event into a per-sub-window fixed-point grid -- needs 128 M of them per batch of 8.)
//   mode 0  agent-scope global_atomic_add_u32 (memory-side across the 8 non-coherent XCD L2s), region = `regions` grids of
//           5 x 480 x 640 ints; every workgroup walks events of one grid (grid = blockIdx % regions)
//   mode 1  workgroup-scope atomics (performed in the issuing XCD's L2); grid g is only ever touched from XCD g % 8
//           (blockIdx % 8 placement heuristic; a timing probe, results are checked against mode 0's sums)
//   mode 2  mode 0 with plain (racy) load-add-store instead of atomics: the traffic floor of the same access pattern
// Each "event" adds to 8 corners (x, x+1) x (y, y+1) x (t, t+1) like the tri-linear splat.
// usage: ./atomic_rate_probe [events_per_grid=100000] [regions=16] [launch_grids=160]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int C = 5, H = 480, W = 640;

__device__ __forceinline__ unsigned hash(unsigned a) { a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16; return a; }

template <int MODE>
__global__ __launch_bounds__(256) void k(int* __restrict__ grids, int regions, int events_per_grid, int wgs_per_grid) {
    const int job = blockIdx.x / wgs_per_grid, part = blockIdx.x % wgs_per_grid;
    // mode 1: jobs are dealt so that job j runs on XCD (blockIdx % 8): grid index = job mapped to a region owned by that XCD
    int g;
    if (MODE == 1) { const int xcd = blockIdx.x & 7; g = (xcd + 8 * ((blockIdx.x >> 3) / wgs_per_grid)) % regions; }
    else g = job % regions;
    int* grid = grids + (size_t)g * C * H * W;
    const int per = (events_per_grid + wgs_per_grid - 1) / wgs_per_grid;
    const int e0 = (MODE == 1 ? ((blockIdx.x >> 3) % wgs_per_grid) : part) * per;
    for (int e = e0 + threadIdx.x; e < e0 + per && e < events_per_grid; e += 256) {
        const unsigned h = hash((unsigned)e * 2654435761u + (unsigned)job * 40503u);
        const int x = h % (W - 1), y = (h >> 10) % (H - 1), t = (h >> 20) % (C - 1);
        const int base = (t * H + y) * W + x;
        const int w = (int)(h >> 24) + 1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            int* p = grid + base + (c & 1) + ((c >> 1) & 1) * W + (c >> 2) * H * W;
            if (MODE == 0) (void)__hip_atomic_fetch_add(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 1) (void)__hip_atomic_fetch_add(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else *p = *p + w;
        }
    }
}

int main(int argc, char** argv) {
    const int epg = argc > 1 ? atoi(argv[1]) : 100000, regions = argc > 2 ? atoi(argv[2]) : 16, jobs = argc > 3 ? atoi(argv[3]) : 160;
    int* grids;
    const size_t gbytes = (size_t)regions * C * H * W * 4;
    CK(hipMalloc(&grids, gbytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<long long> sums(3, 0);
    for (int wpg : {8, 16, 48}) {
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipMemset(grids, 0, gbytes));
            CK(hipDeviceSynchronize());
            const int nb = ((jobs * wpg + 7) / 8) * 8;
            float best = 1e9f;
            for (int it = 0; it < 4; ++it) {
                CK(hipEventRecord(a));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, grids, regions, epg, wpg);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, grids, regions, epg, wpg);
                else hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, grids, regions, epg, wpg);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            std::vector<int> host((size_t)C * H * W);
            CK(hipMemcpy(host.data(), grids, host.size() * 4, hipMemcpyDeviceToHost));
            long long s = 0; for (int v : host) s += v;
            const double n_at = (double)jobs * epg * 8;
            printf("mode %d  wgs/grid %2d  regions %d (%.0f MB)  %8.3f ms  %7.1f G atomics/s  checksum(grid 0) %lld\n", mode, wpg, regions,
                   gbytes / 1e6, best, n_at / best / 1e6, s);
        }
    }
    return 0;
}
