// Optimiser step over the flat parameter buffer (include/igcn.h: igcn_adam_step).
#include "common.h"

__global__ void k_adam_tick(int32_t* step) { *step += 1; }

// torch.optim.Adam (amsgrad=False, weight_decay=0, maximize=False), single-tensor formulation:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void k_adam(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, const int32_t* __restrict__ step, float lr, float b1, float b2,
                       float eps, float gscale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float t = (float)(*step);
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  const float gi = g[i] * gscale;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[i] -= (lr / bc1) * (mi / denom);
}

extern "C" int igcn_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                              int32_t* step, float lr, float beta1, float beta2, float eps, float grad_scale,
                              void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, st, step);
  if (n > 0)
    hipLaunchKernelGGL(k_adam, dim3((unsigned)igcn_cdiv(n, 256)), dim3(256), 0, st, n, param, grad, exp_avg,
                       exp_avg_sq, step, lr, beta1, beta2, eps, grad_scale);
  IGCN_CHECK_LAUNCH("adam_step");
  return IGCN_OK;
}

// ---- multi-tensor variants: one launch walks a device table of tensors --------------------------------
// table[t] = {param, grad, exp_avg, exp_avg_sq} (device pointers as int64), numel[t].  grad == 0: skipped
// (torch.optim.Adam skips parameters whose .grad is None).  The table lets autograd hand over freshly
// written gradient tensors (no AccumulateGrad add into a pre-zeroed flat buffer, no zero_grad memset).
__global__ void __launch_bounds__(256)
k_adam_multi(const int64_t* __restrict__ table, const int64_t* __restrict__ numel,
             const int32_t* __restrict__ step, float lr, float b1, float b2, float eps, float gscale) {
  const int t = blockIdx.y;
  const int64_t n = numel[t];
  if ((int64_t)blockIdx.x * 256 >= n) return;       // most tensors are tiny: their surplus workgroups leave at once
  const float* g = reinterpret_cast<const float*>(table[4 * t + 1]);
  if (g == nullptr) return;
  float* p = reinterpret_cast<float*>(table[4 * t]);
  float* m = reinterpret_cast<float*>(table[4 * t + 2]);
  float* v = reinterpret_cast<float*>(table[4 * t + 3]);
  const float ts = (float)(*step);
  const float bc1 = 1.f - powf(b1, ts), bc2s = sqrtf(1.f - powf(b2, ts));
#pragma unroll 4
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= (lr / bc1) * (mi / (sqrtf(vi) / bc2s + eps));
  }
}

extern "C" int igcn_adam_step_multi(int n_tensors, const int64_t* table, const int64_t* numel, int32_t* step,
                                    float lr, float beta1, float beta2, float eps, float grad_scale,
                                    void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, st, step);
  if (n_tensors > 0)
    hipLaunchKernelGGL(k_adam_multi, dim3(96, n_tensors), dim3(256), 0, st, table, numel, step, lr, beta1, beta2,
                       eps, grad_scale);
  IGCN_CHECK_LAUNCH("adam_step_multi");
  return IGCN_OK;
}

// dst_flat[off[t] .. off[t]+numel[t]) = grad tensor t (zeros when it has no gradient): packs the gradients
// into the flat bucket that the data-parallel all-reduce exchanges.
__global__ void __launch_bounds__(256)
k_pack_grads(const int64_t* __restrict__ table, const int64_t* __restrict__ numel,
             const int64_t* __restrict__ offset, float* __restrict__ flat) {
  const int t = blockIdx.y;
  const float* g = reinterpret_cast<const float*>(table[4 * t + 1]);
  const int64_t n = numel[t];
  float* dst = flat + offset[t];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dst[i] = g ? g[i] : 0.f;
}

extern "C" int igcn_pack_grads(int n_tensors, const int64_t* table, const int64_t* numel, const int64_t* offset,
                               float* flat, void* stream) {
  if (n_tensors > 0)
    hipLaunchKernelGGL(k_pack_grads, dim3(32, n_tensors), dim3(256), 0, (hipStream_t)stream, table, numel, offset,
                       flat);
  IGCN_CHECK_LAUNCH("pack_grads");
  return IGCN_OK;
}
