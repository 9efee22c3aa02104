// Loss terms of the train step as first-class kernels:
//  * mask regulariser  — loss_probability, kernel/sgcn_img_snp.py:153-181
//  * Gram-form batch losses — consist_loss :183-196 and OrthogonalConstraint :198-205 from ONE B x B Gram
//    matrix G = s s^T of the fused features (the reference forms an (R*D) x (R*D) product for the latter).
#include "common.h"

// -------------------------------------------------------------------------------------------------
// mask regulariser.  For p in (0,1):  r(p) = l1*p + ent*( -(p log(p+eps) + (1-p) log(1-p+eps)) ), averaged.
//   loss = mean r_x(sigmoid(prob)) + mean r_e(e) + mean r_x(sigmoid(snps_prob))
// element ranges: [0,n_prob) prob logits, [n_prob, n_prob+n_edge) edge mask values, then snps logits.
// -------------------------------------------------------------------------------------------------
struct MaskRegArgs {
  const float *prob, *e, *snps;
  int64_t n_prob, n_edge, n_snps;
  float l1_x, ent_x, l1_e, ent_e, eps;
};

__device__ __forceinline__ float reg_term(float p, float l1, float ent, float eps) {
  return l1 * fabsf(p) - ent * (p * logf(p + eps) + (1.f - p) * logf((1.f - p) + eps));
}

__device__ __forceinline__ float reg_grad(float p, float l1, float ent, float eps) {
  // d/dp of reg_term (p > 0)
  return l1 - ent * (logf(p + eps) + p / (p + eps) - logf((1.f - p) + eps) - (1.f - p) / ((1.f - p) + eps));
}

__global__ void __launch_bounds__(256) k_mask_reg_fwd(MaskRegArgs a, float* __restrict__ partial) {
  __shared__ float red[16];
  const int64_t total = a.n_prob + a.n_edge + a.n_snps;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    if (i < a.n_prob) {
      const float p = 1.f / (1.f + expf(-a.prob[i]));
      acc += reg_term(p, a.l1_x, a.ent_x, a.eps) / (float)a.n_prob;
    } else if (i < a.n_prob + a.n_edge) {
      acc += reg_term(a.e[i - a.n_prob], a.l1_e, a.ent_e, a.eps) / (float)a.n_edge;
    } else {
      const float p = 1.f / (1.f + expf(-a.snps[i - a.n_prob - a.n_edge]));
      acc += reg_term(p, a.l1_x, a.ent_x, a.eps) / (float)a.n_snps;
    }
  }
  acc = block_sum_all(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ void k_mask_reg_bwd(MaskRegArgs a, const float* __restrict__ gout, float* __restrict__ dprob,
                               float* __restrict__ de, float* __restrict__ dsnps) {
  const int64_t total = a.n_prob + a.n_edge + a.n_snps;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float g = gout[0];
  if (i < a.n_prob) {
    const float p = 1.f / (1.f + expf(-a.prob[i]));
    dprob[i] = g * reg_grad(p, a.l1_x, a.ent_x, a.eps) * p * (1.f - p) / (float)a.n_prob;
  } else if (i < a.n_prob + a.n_edge) {
    const int64_t k = i - a.n_prob;
    de[k] = g * reg_grad(a.e[k], a.l1_e, a.ent_e, a.eps) / (float)a.n_edge;
  } else {
    const int64_t k = i - a.n_prob - a.n_edge;
    const float p = 1.f / (1.f + expf(-a.snps[k]));
    dsnps[k] = g * reg_grad(p, a.l1_x, a.ent_x, a.eps) * p * (1.f - p) / (float)a.n_snps;
  }
}

#define MR_BLOCKS 1024        // upper bound; scratch holds this many partials

// workgroups of igcn_mask_reg_fwd = block partials it leaves in scratch[0 .. blocks): ~2k elements per workgroup, at
// most MR_BLOCKS of them (the dense stress shape has 8.4 M edge-mask values: 128 workgroups left half the chip idle)
extern "C" int igcn_mask_reg_blocks(int64_t n_total) {
  int64_t blocks = igcn_cdiv(n_total, 2048);
  return (int)(blocks < 1 ? 1 : (blocks > MR_BLOCKS ? MR_BLOCKS : blocks));
}

// loss == NULL: the block partials stay in scratch and their consumer sums them (igcn_loss_head_fwd, prob_rows):
// one launch less on the step's critical path.
extern "C" int igcn_mask_reg_fwd(int64_t n_prob, int64_t n_edge, int64_t n_snps, const float* prob, const float* e,
                                 const float* snps, float l1_x, float ent_x, float l1_e, float ent_e, float eps,
                                 float* loss /*[1] or NULL*/, float* scratch /*[1024]*/, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  MaskRegArgs a{prob, e, snps, n_prob, n_edge, n_snps, l1_x, ent_x, l1_e, ent_e, eps};
  const int blocks = igcn_mask_reg_blocks(n_prob + n_edge + n_snps);
  hipLaunchKernelGGL(k_mask_reg_fwd, dim3((unsigned)blocks), dim3(256), 0, st, a, scratch);
  IGCN_CHECK_LAUNCH("mask_reg_fwd");
  if (loss == nullptr) return IGCN_OK;
  return igcn_launch_reduce_rows(scratch, blocks, 1, 1, loss, 0, st);
}

extern "C" int igcn_mask_reg_bwd(int64_t n_prob, int64_t n_edge, int64_t n_snps, const float* prob, const float* e,
                                 const float* snps, float l1_x, float ent_x, float l1_e, float ent_e, float eps,
                                 const float* gout /*[1] device*/, float* dprob, float* de, float* dsnps,
                                 void* stream) {
  hipStream_t st = (hipStream_t)stream;
  MaskRegArgs a{prob, e, snps, n_prob, n_edge, n_snps, l1_x, ent_x, l1_e, ent_e, eps};
  const int64_t total = n_prob + n_edge + n_snps;
  if (total == 0) return IGCN_OK;
  hipLaunchKernelGGL(k_mask_reg_bwd, dim3((unsigned)igcn_cdiv(total, 256)), dim3(256), 0, st, a, gout, dprob, de,
                     dsnps);
  IGCN_CHECK_LAUNCH("mask_reg_bwd");
  return IGCN_OK;
}

// -------------------------------------------------------------------------------------------------
// Gram-form batch losses on G = s s^T  (B x B, symmetric), Lap = diag(W 1) - W  (B x B):
//   out[0] = consist = sum_ij Lap_ij G_ij / B^2                       ( = tr(s^T Lap s)/B^2 )
//   out[1] = orth    = ( sum_ij G_ij^2/(G_ii G_jj) - 2 sum_i 1 + RD ) / B^2
//            ( = ||Wn^T Wn - I||_F^2 / B^2 with Wn the row-normalised s, via ||Wn^T Wn||_F = ||Wn Wn^T||_F )
// backward writes S = dG + dG^T so that ds = S s :
//   dG_ij = gc*Lap_ij/B^2 + go*( 2 G_ij/(G_ii G_jj)  [i != j]  ;  -2 sum_{k != i} G_ik^2/(G_ii^2 G_kk)  [i == j] )/B^2
// -------------------------------------------------------------------------------------------------
// `groups` Gram matrices (the passes of a batched sweep) per launch: blockIdx.y = group, one Laplacian for all
// RBF: the Laplacian is not read but MADE here — W_ij = exp(-gamma ||t_i - t_j||^2) (t NULL: W = 1) evaluated inside
// the row walk, Lap = diag(W 1) - W written by group 0's workgroups for the backward: the stand-alone k_rbf_laplacian
// launch in front disappears from the train step.
// UNIT (RBF only): the launch also writes what k_gram_loss_bwd would for the upstream gradient `unit` (gc, go per group,
// known on the host: a train step's d loss / d (consist, orth) are the loss weights) — S = dG + dG^T per group, from the
// same expressions in the same order, so a step whose upstream IS that has no Gram-loss backward launch.
struct GramUnit { float gc[4], go[4]; };
template <bool RBF>
__global__ void __launch_bounds__(256)
k_gram_loss_fwd(int B, int RD, const float* __restrict__ Gall, const float* __restrict__ Lap,
                float* __restrict__ partial /*[B, 2 groups]*/, const float* __restrict__ t, int T, float gamma,
                float* __restrict__ lap_out, GramUnit unit, float* __restrict__ Sall) {
  __shared__ float red[16];
  const int i = blockIdx.x, grp = blockIdx.y, groups = gridDim.y;
  const float* G = Gall + (int64_t)grp * B * B;
  const float gii = G[(int64_t)i * B + i];
  const bool UNIT = RBF && Sall != nullptr;
  float* S = UNIT ? Sall + (int64_t)grp * B * B : nullptr;
  const float gc = UNIT ? unit.gc[grp] : 0.f, go = UNIT ? unit.go[grp] : 0.f, b2u = (float)B * (float)B;
  float c = 0.f, o = 0.f, rs = 0.f, dsum = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) {
    const float g = G[(int64_t)i * B + j];
    if (RBF) {
      float w = 1.f;
      if (t) {
        float d2 = 0.f;
#pragma unroll 10
        for (int k = 0; k < T; ++k) {
          const float d = t[(int64_t)i * T + k] - t[(int64_t)j * T + k];
          d2 += d * d;
        }
        w = expf(-gamma * d2);
      }
      rs += w;
      if (j != i) {
        c -= w * g;
        if (grp == 0) lap_out[(int64_t)i * B + j] = -w;
        if (UNIT) {                                               // k_gram_loss_bwd's row walk (Lap_ij = Lap_ji = -w)
          const float gjj = G[(int64_t)j * B + j];
          dsum += g * g / (gii * gii * gjj);
          S[(int64_t)i * B + j] = (gc * ((-w) + (-w)) + go * 4.f * g / (gii * gjj)) / b2u;
        }
      }
    } else {
      c += Lap[(int64_t)i * B + j] * g;
    }
    o += g * g / (gii * G[(int64_t)j * B + j]);
  }
  c = block_sum_all(c, red);
  o = block_sum_all(o, red);
  if (RBF) rs = block_sum_all(rs, red);
  if (UNIT) dsum = block_sum_all(dsum, red);
  if (threadIdx.x == 0) {
    if (RBF) {
      const float dii = rs - 1.f;                               // diag(W 1) - W_ii, W_ii = exp(0)
      c += dii * gii;
      if (grp == 0) lap_out[(int64_t)i * B + i] = dii;
      if (UNIT) S[(int64_t)i * B + i] = 2.f * (gc * dii - go * 2.f * dsum) / b2u;
    }
    const float b2 = (float)B * (float)B;
    partial[(int64_t)i * 2 * groups + 2 * grp] = c / b2;
    partial[(int64_t)i * 2 * groups + 2 * grp + 1] = (o - 2.f + (float)RD / (float)B) / b2;   // -2B + RD, spread over rows
  }
}

__global__ void __launch_bounds__(256)
k_gram_loss_bwd(int B, const float* __restrict__ Gall, const float* __restrict__ Lap, const float* __restrict__ gout,
                float* __restrict__ Sall) {
  __shared__ float red[16];
  const int i = blockIdx.x, grp = blockIdx.y;
  const float* G = Gall + (int64_t)grp * B * B;
  float* S = Sall + (int64_t)grp * B * B;
  const float gc = gout[2 * grp], go = gout[2 * grp + 1];
  const float b2 = (float)B * (float)B;
  const float gii = G[(int64_t)i * B + i];
  float dsum = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) {
    if (j == i) continue;
    const float g = G[(int64_t)i * B + j], gjj = G[(int64_t)j * B + j];
    dsum += g * g / (gii * gii * gjj);
    // off-diagonal: dG_ij + dG_ji (both symmetric expressions)
    S[(int64_t)i * B + j] = (gc * (Lap[(int64_t)i * B + j] + Lap[(int64_t)j * B + i]) + go * 4.f * g / (gii * gjj)) / b2;
  }
  dsum = block_sum_all(dsum, red);
  if (threadIdx.x == 0) S[(int64_t)i * B + i] = 2.f * (gc * Lap[(int64_t)i * B + i] - go * 2.f * dsum) / b2;
}

// out == NULL: the row partials [B][groups*2] stay in scratch for the consumer to sum (igcn_loss_head_fwd, gram_rows).
extern "C" int igcn_gram_loss_fwd(int B, int RD, int groups, const float* G /*[groups,B,B]*/, const float* Lap,
                                  float* out /*[groups,2] or NULL*/, float* scratch /*[2 B groups]*/, void* stream) {
  IGCN_REQUIRE(B > 0 && groups >= 1 && groups <= 64, "gram_loss_fwd: bad B / groups");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_gram_loss_fwd<false>, dim3(B, groups), dim3(256), 0, st, B, RD, G, Lap, scratch, nullptr, 0, 0.f,
                     nullptr, GramUnit{}, nullptr);
  IGCN_CHECK_LAUNCH("gram_loss_fwd");
  if (out == nullptr) return IGCN_OK;
  return igcn_launch_reduce_rows(scratch, B, 2 * groups, 2 * groups, out, 0, st);
}

// The same with the RBF Laplacian of consist_loss (util/image_cluster.py:15-31; tsne [B, T] or NULL: W = 1) built inside:
// lap_out [B, B] is an OUTPUT (what igcn_gram_loss_bwd reads).
extern "C" int igcn_gram_loss_fwd_rbf(int B, int RD, int groups, const float* G /*[groups,B,B]*/, const float* tsne, int T,
                                      float gamma, float* lap_out, float* out /*[groups,2] or NULL*/,
                                      float* scratch /*[2 B groups]*/, void* stream) {
  IGCN_REQUIRE(B > 0 && groups >= 1 && groups <= 64 && lap_out && (tsne == nullptr || T > 0), "gram_loss_fwd_rbf: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_gram_loss_fwd<true>, dim3(B, groups), dim3(256), 0, st, B, RD, G, nullptr, scratch, tsne, T, gamma,
                     lap_out, GramUnit{}, nullptr);
  IGCN_CHECK_LAUNCH("gram_loss_fwd_rbf");
  if (out == nullptr) return IGCN_OK;
  return igcn_launch_reduce_rows(scratch, B, 2 * groups, 2 * groups, out, 0, st);
}

// igcn_gram_loss_fwd_rbf that also writes S [groups, B, B] = what igcn_gram_loss_bwd returns for the upstream gradient
// gout [groups, 2] given HERE, on the host (bit for bit: the same expressions in the same order).
extern "C" int igcn_gram_loss_fwd_rbf_unit(int B, int RD, int groups, const float* G, const float* tsne, int T, float gamma,
                                           float* lap_out, float* out, float* scratch, const float* gout /*HOST*/,
                                           float* S, void* stream) {
  IGCN_REQUIRE(B > 0 && groups >= 1 && groups <= 4 && lap_out && gout && S && (tsne == nullptr || T > 0),
               "gram_loss_fwd_rbf_unit: bad arguments (groups <= 4)");
  hipStream_t st = (hipStream_t)stream;
  GramUnit u{};
  for (int g = 0; g < groups; ++g) {
    u.gc[g] = gout[2 * g];
    u.go[g] = gout[2 * g + 1];
  }
  hipLaunchKernelGGL(k_gram_loss_fwd<true>, dim3(B, groups), dim3(256), 0, st, B, RD, G, nullptr, scratch, tsne, T, gamma,
                     lap_out, u, S);
  IGCN_CHECK_LAUNCH("gram_loss_fwd_rbf_unit");
  if (out == nullptr) return IGCN_OK;
  return igcn_launch_reduce_rows(scratch, B, 2 * groups, 2 * groups, out, 0, st);
}

extern "C" int igcn_gram_loss_bwd(int B, int groups, const float* G, const float* Lap,
                                  const float* gout /*[groups,2] device*/, float* S /*[groups,B,B]*/, void* stream) {
  IGCN_REQUIRE(B > 0 && groups >= 1 && groups <= 64, "gram_loss_bwd: bad B / groups");
  hipLaunchKernelGGL(k_gram_loss_bwd, dim3(B, groups), dim3(256), 0, (hipStream_t)stream, B, G, Lap, gout, S);
  IGCN_CHECK_LAUNCH("gram_loss_bwd");
  return IGCN_OK;
}

// Lap = diag(W 1) - W,  W_ij = exp(-gamma * ||t_i - t_j||^2)   (rbf_kernel_torch, util/image_cluster.py:15-31);
// t == NULL gives W = 1 (the non-soft branch of consist_loss).  One block per row i.
__global__ void __launch_bounds__(256)
k_rbf_laplacian(int B, int T, float gamma, const float* __restrict__ t, float* __restrict__ Lap) {
  __shared__ float red[16];
  const int i = blockIdx.x;
  float rs = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) {
    float w = 1.f;
    if (t) {
      float d2 = 0.f;
#pragma unroll 10
      for (int k = 0; k < T; ++k) {
        const float d = t[(int64_t)i * T + k] - t[(int64_t)j * T + k];
        d2 += d * d;
      }
      w = expf(-gamma * d2);
    }
    rs += w;
    if (j != i) Lap[(int64_t)i * B + j] = -w;
  }
  rs = block_sum_all(rs, red);
  if (threadIdx.x == 0) {
    float wii = 1.f;     // exp(0)
    Lap[(int64_t)i * B + i] = rs - wii;
  }
}

extern "C" int igcn_rbf_laplacian(int B, int T, float gamma, const float* t, float* Lap, void* stream) {
  IGCN_REQUIRE(B > 0, "rbf_laplacian: bad B");
  hipLaunchKernelGGL(k_rbf_laplacian, dim3(B), dim3(256), 0, (hipStream_t)stream, B, T, gamma, t, Lap);
  IGCN_CHECK_LAUNCH("rbf_laplacian");
  return IGCN_OK;
}

// -------------------------------------------------------------------------------------------------
// Loss head of train() (kernel/train_eval_sgcn_img_snps.py:525-543) on the STACKED outputs of the batched sweep
// (rows [0,B): plain pass, rows [B,2B): isExplain pass), one launch per direction instead of ~40 scalar-sized ones:
//   ce = nll(logp[:B], y)   mi = nll(logp[B:], y)                      (:525-526; mean over the batch)
//   reg = (mse(reg[:B], clin) + mse(reg[B:], clin)) / 2                (:527)  = mean over all 2B*NR elements
//   recon = (sum (x_hat[:B]-snps)^2 + sum (x_hat[B:]-snps)^2) / 2      (:530)
//   cluster = (consist[0] + consist[1]) / 2   orth = orth[0]           (:533-538; gram = igcn_gram_loss_fwd outputs)
//   terms = {lam0*ce, lam0*mi, lam1*reg, lam2*prob, lam3*recon, lam4*cluster, lam5*orth}
//   loss = hp_ce*terms[0] + hp_mi*terms[1] + terms[2..6]               (:543)
// -------------------------------------------------------------------------------------------------
struct LossHeadW { float lam[6]; float hp_ce, hp_mi; };
// d loss / d input for an upstream gradient of ONE, written by the forward itself (igcn_loss_head_fwd_grads): every
// element's gradient is known where its forward term is computed, so a train step (whose d loss / d loss IS one) needs
// no backward launch for the loss head.  All NULL: plain forward.
struct LossHeadGrads { float *dlogp, *dreg, *dxhat, *dgram, *dprob; };

//   from_logits: `logp` holds the raw class scores; log_softmax (sgcn_img_snp.py:305) is taken here, row by row, and
//   written to logp_out [2B, C] (the model output, and what the backward reads).  gram [gram_rows][4] and prob
//   [prob_rows] may be the un-reduced partials of igcn_gram_loss_fwd / igcn_mask_reg_fwd: their rows are summed here.
__global__ void __launch_bounds__(1024)
k_loss_head_fwd(int B, int C, int NR, int S, const float* __restrict__ logp, int from_logits,
                float* __restrict__ logp_out, const int64_t* __restrict__ y,
                const float* __restrict__ reg, const float* __restrict__ clin, const float* __restrict__ x_hat,
                const float* __restrict__ snps, const float* __restrict__ gram, int gram_rows,
                const float* __restrict__ prob, int prob_rows,
                LossHeadW w, float* __restrict__ loss, float* __restrict__ terms, LossHeadGrads gr) {
  __shared__ float red[9][16];
  const int tid = threadIdx.x;
  float ce = 0.f, mi = 0.f, mse = 0.f, rec = 0.f;
  float gs[4] = {0.f, 0.f, 0.f, 0.f}, ps = 0.f;
  for (int r = tid; r < gram_rows; r += 1024) {
#pragma unroll
    for (int j = 0; j < 4; ++j) gs[j] += gram[(int64_t)r * 4 + j];
  }
  for (int r = tid; r < prob_rows; r += 1024) ps += prob[r];
  if (from_logits) {
    for (int row = tid; row < 2 * B; row += 1024) {
      const float* xr = logp + (int64_t)row * C;
      float m = -INFINITY;
      for (int c = 0; c < C; ++c) m = fmaxf(m, xr[c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(xr[c] - m);
      const float lse = logf(se);
      for (int c = 0; c < C; ++c) logp_out[(int64_t)row * C + c] = (xr[c] - m) - lse;
      if (gr.dlogp) {                                    // nll o log_softmax: wt / B (softmax - onehot)
        const float wt = (row < B ? w.hp_ce : w.hp_mi) * w.lam[0];
        const int64_t yc = y[row < B ? row : row - B];
        for (int c = 0; c < C; ++c)
          gr.dlogp[(int64_t)row * C + c] = wt != 0.f ? wt / (float)B * (expf((xr[c] - m) - lse) - (yc == c ? 1.f : 0.f)) : 0.f;
      }
      if (w.lam[0] != 0.f) {
        const int b = row < B ? row : row - B;
        const int64_t c = y[b];
        if (c < 0 || c >= C) {
          ce = __builtin_nanf("");
        } else {
          const float v = (xr[c] - m) - lse;
          if (row < B) ce -= v; else mi -= v;
        }
      }
    }
  }
  // lam[0] == 0 (main.py's default): the reference sets loss_ce = loss_mi = 0.0 outright (:540-542) — the class
  // scores are not read at all, so a non-finite log-probability cannot poison the loss through 0 * inf
  if (w.lam[0] != 0.f && !from_logits) {
    for (int b = tid; b < B; b += 1024) {
      const int64_t c = y[b];
      if (c < 0 || c >= C) {             // F.nll_loss raises on the host; a kernel cannot: poison the loss instead of
        ce = __builtin_nanf("");         // reading out of bounds (the NaN is what the caller sees)
        continue;
      }
      ce -= logp[(int64_t)b * C + c];
      mi -= logp[(int64_t)(B + b) * C + c];
    }
  }
  if (gr.dlogp && !from_logits) {                        // log-probabilities given: d nll = -wt / B at the label
    for (int i = tid; i < 2 * B * C; i += 1024) {
      const int row = i / C, c = i % C;
      const float wt = (row < B ? w.hp_ce : w.hp_mi) * w.lam[0];
      gr.dlogp[i] = (y[row < B ? row : row - B] == c) ? -wt / (float)B : 0.f;
    }
  }
  const int nreg = B * NR, nrec = B * S;
  for (int i0 = tid; i0 < 2 * nreg; i0 += 4 * 1024) {     // (batches of four slots, as the reconstruction walk below)
    float rv[4], cv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 1024, ic = i < 2 * nreg ? i : tid;
      rv[u] = reg[ic];
      cv[u] = clin[ic < nreg ? ic : ic - nreg];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 1024;
      if (i < 2 * nreg) {
        const float d = rv[u] - cv[u];
        mse += d * d;
        if (gr.dreg) gr.dreg[i] = w.lam[1] * 2.f * d / (float)(2 * nreg);
      }
    }
  }
  // one workgroup walks everything: 16 bytes per lane and the whole walk unrolled, so that it is one batch of loads
  // instead of a chain of dependent trips (the kernel is pure latency)
  if ((nrec & 3) == 0 && (((uintptr_t)x_hat | (uintptr_t)snps | (uintptr_t)gr.dxhat) & 15) == 0) {
    const int nq = nrec / 4;
    // batches of eight slots, every load of a batch issued before the first use: out-of-range slots re-read slot `tid`
    // and count as zero.  [`#pragma unroll 8` on the plain loop: 6.75 trips per thread at B = 256 = no full batch, and
    // the remainder loop the compiler adds runs one load per trip, each waiting out its own round trip.]
    for (int i0 = tid; i0 < 2 * nq; i0 += 8 * 1024) {
      float4 a[8], s4[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 1024, ic = i < 2 * nq ? i : tid;
        a[u] = reinterpret_cast<const float4*>(x_hat)[ic];
        s4[u] = reinterpret_cast<const float4*>(snps)[ic < nq ? ic : ic - nq];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 1024;
        if (i < 2 * nq) {
          const float d0 = a[u].x - s4[u].x, d1 = a[u].y - s4[u].y, d2 = a[u].z - s4[u].z, d3 = a[u].w - s4[u].w;
          rec += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
          if (gr.dxhat)
            reinterpret_cast<float4*>(gr.dxhat)[i] = make_float4(w.lam[3] * d0, w.lam[3] * d1, w.lam[3] * d2, w.lam[3] * d3);
        }
      }
    }
  } else {
#pragma unroll 8
    for (int i = tid; i < 2 * nrec; i += 1024) {
      const float d = x_hat[i] - snps[i < nrec ? i : i - nrec];
      rec += d * d;
      if (gr.dxhat) gr.dxhat[i] = w.lam[3] * d;
    }
  }
  // the four sums through LDS together: two barriers instead of eight
  {
    const int lane = tid & 63, wv = tid >> 6;
    ce = wave_sum(ce); mi = wave_sum(mi); mse = wave_sum(mse); rec = wave_sum(rec);
    ps = wave_sum(ps);
#pragma unroll
    for (int j = 0; j < 4; ++j) gs[j] = wave_sum(gs[j]);
    if (lane == 0) {
      red[0][wv] = ce; red[1][wv] = mi; red[2][wv] = mse; red[3][wv] = rec; red[4][wv] = ps;
#pragma unroll
      for (int j = 0; j < 4; ++j) red[5 + j][wv] = gs[j];
    }
    __syncthreads();
    if (tid == 0) {
      ce = mi = mse = rec = ps = 0.f;
      gs[0] = gs[1] = gs[2] = gs[3] = 0.f;
      for (int i = 0; i < 16; ++i) {
        ce += red[0][i]; mi += red[1][i]; mse += red[2][i]; rec += red[3][i]; ps += red[4][i];
#pragma unroll
        for (int j = 0; j < 4; ++j) gs[j] += red[5 + j][i];
      }
    }
  }
  if (tid == 0) {
    float t[7];
    t[0] = w.lam[0] != 0.f ? w.lam[0] * (ce / (float)B) : 0.f;
    t[1] = w.lam[0] != 0.f ? w.lam[0] * (mi / (float)B) : 0.f;
    t[2] = w.lam[1] * (mse / (float)(2 * nreg));
    t[3] = w.lam[2] * ps;
    t[4] = w.lam[3] * (rec * 0.5f);
    t[5] = w.lam[4] * ((gs[0] + gs[2]) * 0.5f);
    t[6] = w.lam[5] * gs[1];
    for (int k = 0; k < 7; ++k) terms[k] = t[k];
    loss[0] = w.hp_ce * t[0] + w.hp_mi * t[1] + t[2] + t[3] + t[4] + t[5] + t[6];
    if (gr.dgram) {
      gr.dgram[0] = w.lam[4] * 0.5f; gr.dgram[1] = w.lam[5]; gr.dgram[2] = w.lam[4] * 0.5f; gr.dgram[3] = 0.f;
      gr.dprob[0] = w.lam[2];
    }
  }
}

__global__ void __launch_bounds__(256)
k_loss_head_bwd(int B, int C, int NR, int S, const int64_t* __restrict__ y, const float* __restrict__ reg,
                const float* __restrict__ clin, const float* __restrict__ x_hat, const float* __restrict__ snps,
                const float* __restrict__ logp /* forward took raw scores: dlogp is the gradient of THOSE; or NULL */,
                LossHeadW w, const float* __restrict__ gout, float* __restrict__ dlogp, float* __restrict__ dreg,
                float* __restrict__ dxhat, float* __restrict__ dgram, float* __restrict__ dprob) {
  const float g = gout[0];
  const int n0 = 2 * B * C, n1 = 2 * B * NR, n2 = 2 * B * S;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n0) {
    const int row = i / C, c = i % C, b = row < B ? row : row - B;
    const float wt = (row < B ? w.hp_ce : w.hp_mi) * w.lam[0];
    const float hit = (y[b] == c) ? 1.f : 0.f;
    if (logp == nullptr) dlogp[i] = -g * wt / (float)B * hit;
    else dlogp[i] = wt != 0.f ? g * wt / (float)B * (expf(logp[i]) - hit) : 0.f;      // nll o log_softmax
    return;
  }
  i -= n0;
  if (i < n1) {
    const int h = n1 / 2;
    dreg[i] = g * w.lam[1] * 2.f * (reg[i] - clin[i < h ? i : i - h]) / (float)n1;
    return;
  }
  i -= n1;
  if (i < n2) {
    const int h = n2 / 2;
    dxhat[i] = g * w.lam[3] * (x_hat[i] - snps[i < h ? i : i - h]);
    return;
  }
  i -= n2;
  if (i == 0) {
    dgram[0] = g * w.lam[4] * 0.5f;
    dgram[1] = g * w.lam[5];
    dgram[2] = g * w.lam[4] * 0.5f;
    dgram[3] = 0.f;
    dprob[0] = g * w.lam[2];
  }
}

static LossHeadW loss_head_w(const float* lam6, float hp_ce, float hp_mi) {
  LossHeadW w;
  for (int k = 0; k < 6; ++k) w.lam[k] = lam6[k];
  w.hp_ce = hp_ce;
  w.hp_mi = hp_mi;
  return w;
}

extern "C" int igcn_loss_head_fwd(int B, int C, int NR, int S, const float* logp, int from_logits, float* logp_out,
                                  const int64_t* y, const float* reg, const float* clin, const float* x_hat,
                                  const float* snps, const float* gram, int gram_rows, const float* prob,
                                  int prob_rows, const float* lam6 /*HOST [6]*/, float hp_ce, float hp_mi,
                                  float* loss /*[1]*/, float* terms /*[7]*/, void* stream) {
  IGCN_REQUIRE(B > 0 && C > 0 && NR > 0 && S > 0 && gram_rows >= 1 && prob_rows >= 1, "loss_head_fwd: bad sizes");
  IGCN_REQUIRE(!from_logits || logp_out != nullptr, "loss_head_fwd: from_logits needs logp_out");
  hipLaunchKernelGGL(k_loss_head_fwd, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, C, NR, S, logp, from_logits,
                     logp_out, y, reg, clin, x_hat, snps, gram, gram_rows, prob, prob_rows,
                     loss_head_w(lam6, hp_ce, hp_mi), loss, terms, LossHeadGrads{});
  IGCN_CHECK_LAUNCH("loss_head_fwd");
  return IGCN_OK;
}

// The forward that also writes what igcn_loss_head_bwd would for gout = 1: dlogp [2B,C] (the gradient of the raw scores
// when from_logits), dreg [2B,NR], dxhat [2B,S], dgram [4], dprob [1].  In a train step d loss / d loss IS one: no
// backward launch for the loss head.
extern "C" int igcn_loss_head_fwd_grads(int B, int C, int NR, int S, const float* logp, int from_logits, float* logp_out,
                                        const int64_t* y, const float* reg, const float* clin, const float* x_hat,
                                        const float* snps, const float* gram, int gram_rows, const float* prob,
                                        int prob_rows, const float* lam6 /*HOST [6]*/, float hp_ce, float hp_mi,
                                        float* loss, float* terms, float* dlogp, float* dreg, float* dxhat, float* dgram,
                                        float* dprob, void* stream) {
  IGCN_REQUIRE(B > 0 && C > 0 && NR > 0 && S > 0 && gram_rows >= 1 && prob_rows >= 1, "loss_head_fwd_grads: bad sizes");
  IGCN_REQUIRE(!from_logits || logp_out != nullptr, "loss_head_fwd_grads: from_logits needs logp_out");
  IGCN_REQUIRE(dlogp && dreg && dxhat && dgram && dprob, "loss_head_fwd_grads: null gradient output");
  hipLaunchKernelGGL(k_loss_head_fwd, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, C, NR, S, logp, from_logits,
                     logp_out, y, reg, clin, x_hat, snps, gram, gram_rows, prob, prob_rows,
                     loss_head_w(lam6, hp_ce, hp_mi), loss, terms, LossHeadGrads{dlogp, dreg, dxhat, dgram, dprob});
  IGCN_CHECK_LAUNCH("loss_head_fwd_grads");
  return IGCN_OK;
}

extern "C" int igcn_loss_head_bwd(int B, int C, int NR, int S, const int64_t* y, const float* reg, const float* clin,
                                  const float* x_hat, const float* snps, const float* logp /*[2B,C] or NULL*/,
                                  const float* lam6 /*HOST [6]*/, float hp_ce,
                                  float hp_mi, const float* gout /*[1] device*/, float* dlogp, float* dreg,
                                  float* dxhat, float* dgram /*[4]*/, float* dprob /*[1]*/, void* stream) {
  IGCN_REQUIRE(B > 0 && C > 0 && NR > 0 && S > 0, "loss_head_bwd: bad sizes");
  const int64_t total = (int64_t)2 * B * (C + NR + S) + 1;
  hipLaunchKernelGGL(k_loss_head_bwd, dim3((unsigned)igcn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, B, C,
                     NR, S, y, reg, clin, x_hat, snps, logp, loss_head_w(lam6, hp_ce, hp_mi), gout, dlogp, dreg, dxhat,
                     dgram, dprob);
  IGCN_CHECK_LAUNCH("loss_head_bwd");
  return IGCN_OK;
}
