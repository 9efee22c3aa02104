// loss_probability (kernel/sgcn_img_snp.py:153-181) riding in the mask kernels (igcn_edge_mask_{fwd,bwd}_reg, and the
// front kernel of csrc/sgcn_fused.hip): the same term / derivative as csrc/loss.hip's stand-alone regulariser.  For p in (0,1):
//   r(p) = l1 p - ent (p log(p + eps) + (1 - p) log(1 - p + eps))
#pragma once
#include "common.h"
__device__ __forceinline__ float em_reg_term(float p, float l1, float ent, float eps) {
  return l1 * fabsf(p) - ent * (p * logf(p + eps) + (1.f - p) * logf((1.f - p) + eps));
}
__device__ __forceinline__ float em_reg_grad(float p, float l1, float ent, float eps) {
  return l1 - ent * (logf(p + eps) + p / (p + eps) - logf((1.f - p) + eps) - (1.f - p) / ((1.f - p) + eps));
}
