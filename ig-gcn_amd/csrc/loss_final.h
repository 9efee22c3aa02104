// The last step of the train step's loss (kernel/train_eval_sgcn_img_snps.py:525-543) when its terms arrive as PARTIAL SUMS
// from several launches — the fused output-heads + loss kernel (k_head_loss_fwd: rows of [ce, mi, mse, rec] sums), the Gram
// loss kernel (rows of [consist_1, orth_1, consist_2, orth_2]) and the mask regulariser (loss_probability's partials):
// column sums, the seven lam-weighted terms and their weighted sum.  Nothing of the backward depends on it (every
// gradient is known where its term is computed), so it runs wherever a workgroup is free: as an entry of the backward's
// deferred flush (k_multi_reduce, plan.hip) or as a launch of its own (k_loss_final, loss.hip).
#pragma once
#include "common.h"

// wts [10] (device): lam[0..5], hp_ce, hp_mi, B, NR.  out [8]: loss, terms[7] = {ce, mi, reg, prob, recon, cluster, orth}.
// 256 threads; lds >= 9 * 4 floats.
__device__ __forceinline__ void loss_final_body(const float* __restrict__ parts, int nparts, const float* __restrict__ gram,
                                                int gram_rows, const float* __restrict__ prob, int prob_rows,
                                                const float* __restrict__ wts, float* __restrict__ out, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = (int)(blockDim.x >> 6);
  float s[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) s[j] = 0.f;
  for (int r = tid; r < nparts; r += (int)blockDim.x) {
    const float4 v = *reinterpret_cast<const float4*>(parts + (int64_t)r * 4);
    s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
  }
  for (int r = tid; r < gram_rows; r += (int)blockDim.x) {
#pragma unroll
    for (int j = 0; j < 4; ++j) s[4 + j] += gram[(int64_t)r * 4 + j];
  }
  for (int r = tid; r < prob_rows; r += (int)blockDim.x) s[8] += prob[r];
#pragma unroll
  for (int j = 0; j < 9; ++j) s[j] = wave_sum(s[j]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 9; ++j) lds[j * 4 + wv] = s[j];
  }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      float t = 0.f;
      for (int i = 0; i < nw && i < 4; ++i) t += lds[j * 4 + i];
      s[j] = t;
    }
    const float lam0 = wts[0], B = wts[8], NR = wts[9];
    float t[7];
    t[0] = lam0 != 0.f ? lam0 * (s[0] / B) : 0.f;
    t[1] = lam0 != 0.f ? lam0 * (s[1] / B) : 0.f;
    t[2] = wts[1] * (s[2] / (2.f * B * NR));
    t[3] = wts[2] * s[8];
    t[4] = wts[3] * (s[3] * 0.5f);
    t[5] = wts[4] * ((s[4] + s[6]) * 0.5f);
    t[6] = wts[5] * s[5];
    for (int k = 0; k < 7; ++k) out[1 + k] = t[k];
    out[0] = wts[6] * t[0] + wts[7] * t[1] + t[2] + t[3] + t[4] + t[5] + t[6];
  }
}
