// What does a grid of many small workgroups cost on gfx950 before it does any work?  The voxelizer's splat kernel launches
// one 256-thread workgroup with a 20 KB LDS tile per (segment, tile) item - 44 800 of them at the BASELINE size - and its
// run time barely moves when the splat itself or the output write is removed.  This probe times the skeleton:
//   mode 0  empty kernel
//   mode 1  + zero-fill of the LDS tile and a barrier
//   mode 2  + one dependent global load per lane (a run table row) and a wave scan
//   mode 3  + a gather of 16-byte records addressed by the loaded value
//   mode 4  + the 20 KB tile written to global memory (float4 per lane), contiguous per workgroup
//   mode 5  the write alone (no LDS, no barrier)
//   mode 6  mode 4 with non-temporal stores
//   mode 7  mode 4 in the voxel grid's layout: 64-pixel x 16-row x 5-plane tiles of a 640-wide image (256 B segments)
//   mode 8  mode 7 with non-temporal stores
// for several workgroup sizes / LDS sizes.   usage: ./wg_floor_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void skel(const int* __restrict__ table, const float4* __restrict__ recs, float* __restrict__ out, int lds_words,
                     int n_recs) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const int item = blockIdx.x;
    if (MODE == 0) return;
    if (MODE == 5) {
        float* o = out + (size_t)item * lds_words;
        for (int i = threadIdx.x * 4; i < lds_words; i += blockDim.x * 4) *reinterpret_cast<float4*>(&o[i]) = make_float4(1.f, 2.f, 3.f, 4.f);
        return;
    }
    int v = 0;
    if (MODE >= 2) {
        v = table[(size_t)item * 64 + (threadIdx.x & 63)];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(v, off, 64); if ((int)(threadIdx.x & 63) >= off) v += y; }
    }
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE >= 3) r = recs[(unsigned)(v * 977 + item * 131 + threadIdx.x) % (unsigned)n_recs];
    for (int i = threadIdx.x * 2; i < lds_words; i += blockDim.x * 2) *reinterpret_cast<int2*>(&lds[i]) = make_int2(0, 0);
    __syncthreads();
    if (MODE >= 3) atomicAdd(&lds[(threadIdx.x * 37 + (int)r.x) % lds_words], (int)r.y + 1);
    __syncthreads();
    if (MODE >= 7) {
        // items = segments x (10 x 28) tiles; rows of 16 float4, lds_words / 1024 planes of 440 x 640
        const int rows = 16, planes = lds_words / (rows * 64);
        const int s = item / 280, tile = item - s * 280, ty = tile / 10, tx = tile - ty * 10;
        for (int i = threadIdx.x; i < lds_words / 4; i += blockDim.x) {
            const int q = i & 15, rc = i >> 4, rr = rc & (rows - 1), c = rc / rows;
            const int yy = ty * rows + rr;
            if (yy < 440) {
                const int4 a = *reinterpret_cast<const int4*>(&lds[i * 4]);
                float4* o = reinterpret_cast<float4*>(&out[((size_t)(s * planes + c) * 440 + yy) * 640 + tx * 64 + q * 4]);
                const float4 v = make_float4((float)a.x, (float)a.y, (float)a.z, (float)a.w);
                if (MODE == 8) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(o)); else *o = v;
            }
        }
    } else if (MODE >= 4) {
        float* o = out + (size_t)item * lds_words;
        for (int i = threadIdx.x * 4; i < lds_words; i += blockDim.x * 4) {
            const int4 a = *reinterpret_cast<const int4*>(&lds[i]);
            const float4 v = make_float4((float)a.x, (float)a.y, (float)a.z, (float)a.w);
            if (MODE == 6) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(&o[i])); else *reinterpret_cast<float4*>(&o[i]) = v;
        }
    } else if (lds[threadIdx.x] == 0x12345) out[threadIdx.x] = 1.f;
}

template <int MODE>
float run(int items, int threads, int lds_bytes, const int* table, const float4* recs, float* out, int n_recs) {
    hipFuncSetAttribute((const void*)&skel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(skel<MODE>, dim3(items), dim3(threads), lds_bytes, 0, table, recs, out, lds_bytes / 4, n_recs);
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(skel<MODE>, dim3(items), dim3(threads), lds_bytes, 0, table, recs, out, lds_bytes / 4, n_recs);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / it * 1000.f;
}

int main() {
    const int items = 44800, n_recs = 18 << 20;
    int* table; float4* recs; float* out;
    CK(hipMalloc(&table, (size_t)items * 64 * 4 * 4)); CK(hipMalloc(&recs, (size_t)n_recs * 16)); CK(hipMalloc(&out, (size_t)items * 4 * 20480));
    CK(hipMemset(table, 0, (size_t)items * 64 * 4 * 4)); CK(hipMemset(recs, 0, (size_t)n_recs * 16));
    struct Cfg { int items, threads, lds; } cfgs[] = {{items, 256, 20480}, {items, 256, 4096}, {items * 2, 128, 10240}, {items / 2, 512, 40960},
                                                      {items / 2, 256, 40960}, {items / 4, 1024, 81920}};
    for (auto c : cfgs) {
        printf("items %6d threads %4d lds %6d B:", c.items, c.threads, c.lds);
        printf("  empty %6.1f", run<0>(c.items, c.threads, c.lds, table, recs, out, n_recs));
        printf("  +zero %6.1f", run<1>(c.items, c.threads, c.lds, table, recs, out, n_recs));
        printf("  +table %6.1f", run<2>(c.items, c.threads, c.lds, table, recs, out, n_recs));
        printf("  +gather %6.1f", run<3>(c.items, c.threads, c.lds, table, recs, out, n_recs));
        printf("  +write %6.1f", run<4>(c.items, c.threads, c.lds, table, recs, out, n_recs));
        printf("  write-only %6.1f", run<5>(c.items, c.threads, c.lds, table, recs, out, n_recs));
        printf("  nt %6.1f", run<6>(c.items, c.threads, c.lds, table, recs, out, n_recs));
        if (c.items == items && c.lds == 20480) {
            printf("  tiled %6.1f", run<7>(c.items, c.threads, c.lds, table, recs, out, n_recs));
            printf("  tiled-nt %6.1f", run<8>(c.items, c.threads, c.lds, table, recs, out, n_recs));
        }
        printf(" us\n");
    }
    return 0;
}
