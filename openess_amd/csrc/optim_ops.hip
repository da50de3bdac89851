// a18 optimiser step: torch.optim.AdamW (training/pretrain_trainer.py:231-243, decoupled weight decay, no amsgrad) for a
// whole parameter list in ONE launch.  Same per-element operation order as ATen's _multi_tensor_adamw:
//   p *= 1 - lr * wd;  m += (g - m) * (1 - b1);  v = v * b2 + (1 - b2) * g * g;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// (compiled with -ffp-contract=off and correctly rounded division / sqrt, so it tracks the library to the last ulp).
// table: [n_tensors][5] int64 on the device = {param, grad, exp_avg, exp_avg_sq pointers, numel};
// chunk_map: [n_chunks][2] int32 = {tensor index, chunk index}; one workgroup per chunk of `chunk_elems` elements.
// HBM-bound: 4 reads + 3 writes of 4 bytes per parameter.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "oess.h"
#include "oess_common.h"

namespace {
constexpr int THREADS = 256;

__global__ __launch_bounds__(THREADS) void adamw_multi_kernel(const int64_t* __restrict__ table, const int32_t* __restrict__ chunk_map,
                                                              int chunk_elems, float decay, float b1c, float b2, float b2c,
                                                              float step_size, float bc2_sqrt, float eps) {
    const int t = chunk_map[blockIdx.x * 2], c = chunk_map[blockIdx.x * 2 + 1];
    float* __restrict__ p = reinterpret_cast<float*>(table[t * 5 + 0]);
    const float* __restrict__ g = reinterpret_cast<const float*>(table[t * 5 + 1]);
    float* __restrict__ m = reinterpret_cast<float*>(table[t * 5 + 2]);
    float* __restrict__ v = reinterpret_cast<float*>(table[t * 5 + 3]);
    const int64_t n = table[t * 5 + 4];
    const int64_t beg = (int64_t)c * chunk_elems;
    int64_t end = beg + chunk_elems;
    if (end > n) end = n;
    for (int64_t i = beg + threadIdx.x; i < end; i += THREADS) {
        const float gi = g[i];
        const float pi = p[i] * decay;
        const float mi = m[i] + (gi - m[i]) * b1c;
        const float vi = v[i] * b2 + b2c * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}
}  // namespace

extern "C" {

int oess_adamw_multi_f32(const int64_t* table, int n_tensors, const int32_t* chunk_map, int n_chunks, int chunk_elems, double lr,
                         double beta1, double beta2, double eps, double weight_decay, double bias_correction1,
                         double bias_correction2_sqrt, oess_stream_t stream) {
    if (!table || !chunk_map || n_tensors <= 0 || n_chunks <= 0 || chunk_elems <= 0 || bias_correction1 <= 0.0 || bias_correction2_sqrt <= 0.0)
        return OESS_EINVAL;
    // scalar coefficients in double (as the Python optimiser computes them), rounded to fp32 once
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)n_chunks), dim3(THREADS), 0, (hipStream_t)stream, table, chunk_map, chunk_elems,
                       (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)(lr / bias_correction1), (float)bias_correction2_sqrt, (float)eps);
    OESS_HIP(hipGetLastError());
    return OESS_OK;
}

}  // extern "C"
