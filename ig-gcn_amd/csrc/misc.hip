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
