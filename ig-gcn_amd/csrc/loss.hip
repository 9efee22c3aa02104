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
// (row i of group grp; the body of k_gram_loss_fwd, also one role of k_head_loss_gram_fwd)
template <bool RBF>
__device__ __forceinline__ void gram_loss_fwd_body(const int i, const int grp, const int groups, int B, int RD,
                                                   const float* __restrict__ Gall, const float* __restrict__ Lap,
                                                   float* __restrict__ partial, const float* __restrict__ t, int T,
                                                   float gamma, float* __restrict__ lap_out, const GramUnit& unit,
                                                   float* __restrict__ Sall, float* red) {
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

template <bool RBF>
__global__ void __launch_bounds__(256)
k_gram_loss_fwd(int B, int RD, const float* __restrict__ Gall, const float* __restrict__ Lap,
                float* __restrict__ partial /*[B, 2 groups]*/, const float* __restrict__ t, int T, float gamma,
                float* __restrict__ lap_out, GramUnit unit, float* __restrict__ Sall) {
  __shared__ float red[16];
  gram_loss_fwd_body<RBF>((int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, B, RD, Gall, Lap, partial, t, T, gamma, lap_out,
                          unit, Sall, red);
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

// -------------------------------------------------------------------------------------------------
// The OUTPUT HEADS and the loss head of a train step as one multi-workgroup launch.  lin2 / lin2_regr (64 -> 3 each,
// kernel/sgcn_img_snp.py:289-290,300-301), log_softmax (:305), the cross-entropy / regression / reconstruction terms of
// train() (:525-530) AND the backward of all of it for an upstream gradient of one were four launches on the step's
// critical path — k_small_linear_fwd (4.9 us), k_loss_head_fwd (ONE workgroup, 10.7 us), k_small_linear_bwd (5.1 us) — for
// a few hundred KB: a workgroup here owns 256 / (K / 4) rows of the stacked sweep, computes their scores, the softmax,
// the three row-wise loss sums, d loss / d (scores, regression outputs), and from those — they are in registers — the
// gradients of the two layers' inputs and its rows' share of their weight / bias gradients (partial rows for the deferred
// reduction).  The loss VALUE is only read by the host: its last step (loss_final.h) rides in the backward's flush.
// -------------------------------------------------------------------------------------------------
#include "loss_final.h"
#define HL_MAXC 4
struct HeadLossArgs {
  int B, K, C, NR, S;
  const float *x1, *keep1, *W1, *b1;      // classifier head: features [2B, K] (dropout factors or NULL), lin2 [C, K], [C]
  const float *x2, *keep2, *W2, *b2;      // regression head: features [2B, K], lin2_regr [NR, K], [NR]
  const int64_t* y;                        // [B]
  const float *clin, *x_hat, *snps;        // [B, NR], [2B, S], [B, S]
  LossHeadW w;
  float *logp_out, *reg_out;               // [2B, C] log_softmax, [2B, NR]
  float *dx1, *dx2, *dxhat;                // d loss / d (features, x_hat) for an upstream gradient of one
  float *parts;                            // [blocks][4]: sums of -logp[y] (plain | masked pass), (reg - clin)^2, (x_hat - snps)^2
  float *wpart;                            // [blocks][C K + C + NR K + NR]: the layers' weight | bias gradient partials
  float *dgram, *dprob;                    // [4], [1]: d loss / d (Gram terms, regulariser)
};

#define HL_RED_FLOATS (256 * 4 * HL_MAXC + 256 * HL_MAXC)
__device__ __forceinline__ void head_loss_body(const unsigned blk, const HeadLossArgs& a, float* red) {
  const int K = a.K, kq = K / 4, q = threadIdx.x % kq, rl = threadIdx.x / kq, rpb = 256 / kq;
  const int rows = 2 * a.B;
  const int r = (int)blk * rpb + rl;
  const bool live = rl < rpb && r < rows;
  const int b = r < a.B ? r : r - a.B;
  float ce = 0.f, mi = 0.f, mse = 0.f, rec = 0.f;
  float4 gw1[HL_MAXC], gw2[HL_MAXC];
  float gb1[HL_MAXC], gb2[HL_MAXC];
#pragma unroll
  for (int c = 0; c < HL_MAXC; ++c) {
    gw1[c] = gw2[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    gb1[c] = gb2[c] = 0.f;
  }
  if (live) {
    // every load of the row up front: features, dropout factors, both layers' weight quads, label, targets
    float4 x1 = *reinterpret_cast<const float4*>(a.x1 + (int64_t)r * K + 4 * q);
    float4 x2 = *reinterpret_cast<const float4*>(a.x2 + (int64_t)r * K + 4 * q);
    float4 k1 = make_float4(1.f, 1.f, 1.f, 1.f), k2 = k1;
    if (a.keep1) k1 = *reinterpret_cast<const float4*>(a.keep1 + (int64_t)r * K + 4 * q);
    if (a.keep2) k2 = *reinterpret_cast<const float4*>(a.keep2 + (int64_t)r * K + 4 * q);
    float4 w1[HL_MAXC], w2[HL_MAXC];
    float tgt[HL_MAXC];
#pragma unroll
    for (int c = 0; c < HL_MAXC; ++c) {
      w1[c] = c < a.C ? *reinterpret_cast<const float4*>(a.W1 + c * K + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      w2[c] = c < a.NR ? *reinterpret_cast<const float4*>(a.W2 + c * K + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      tgt[c] = c < a.NR ? a.clin[(int64_t)b * a.NR + c] : 0.f;
    }
    const int64_t yc = a.y[b];
    x1.x *= k1.x; x1.y *= k1.y; x1.z *= k1.z; x1.w *= k1.w;           // dropout of the input, fused: x * keep
    x2.x *= k2.x; x2.y *= k2.y; x2.z *= k2.z; x2.w *= k2.w;
    float s1[HL_MAXC], s2[HL_MAXC];
#pragma unroll
    for (int c = 0; c < HL_MAXC; ++c) {
      s1[c] = (x1.x * w1[c].x + x1.y * w1[c].y) + (x1.z * w1[c].z + x1.w * w1[c].w);      // (k_small_linear_fwd's order)
      s2[c] = (x2.x * w2[c].x + x2.y * w2[c].y) + (x2.z * w2[c].z + x2.w * w2[c].w);
      for (int o = 1; o < kq; o <<= 1) {
        s1[c] += __shfl_xor(s1[c], o, 64);
        s2[c] += __shfl_xor(s2[c], o, 64);
      }
      s1[c] += (c < a.C && a.b1) ? a.b1[c] : 0.f;
      s2[c] += (c < a.NR && a.b2) ? a.b2[c] : 0.f;
    }
    // log_softmax of the scores, the cross-entropy term and its gradient (k_loss_head_fwd, from_logits)
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < HL_MAXC; ++c)
      if (c < a.C) m = fmaxf(m, s1[c]);
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < HL_MAXC; ++c)
      if (c < a.C) se += expf(s1[c] - m);
    const float lse = logf(se);
    const float wt = (r < a.B ? a.w.hp_ce : a.w.hp_mi) * a.w.lam[0];
    float d1[HL_MAXC], d2[HL_MAXC];
#pragma unroll
    for (int c = 0; c < HL_MAXC; ++c) {
      const float lp = (s1[c] - m) - lse;
      d1[c] = (c < a.C && wt != 0.f) ? wt / (float)a.B * (expf(lp) - (yc == c ? 1.f : 0.f)) : 0.f;
      const float dr = s2[c] - tgt[c];
      d2[c] = c < a.NR ? a.w.lam[1] * 2.f * dr / (float)(2 * a.B * a.NR) : 0.f;
      if (q == 0) {
        if (c < a.C) a.logp_out[(int64_t)r * a.C + c] = lp;
        if (c < a.NR) {
          a.reg_out[(int64_t)r * a.NR + c] = s2[c];
          mse += dr * dr;
        }
        if (c < a.C && a.w.lam[0] != 0.f && yc == c) {
          if (r < a.B) ce -= lp; else mi -= lp;
        }
      }
    }
    if (q == 0 && a.w.lam[0] != 0.f && (yc < 0 || yc >= a.C)) ce = __builtin_nanf("");      // (a label outside the classes)
    // backward of the two layers for exactly those upstream gradients: dx = (d W) * keep, dW += d x^T, db += d
    float4 e1 = make_float4(0.f, 0.f, 0.f, 0.f), e2 = e1;
#pragma unroll
    for (int c = 0; c < HL_MAXC; ++c) {
      e1.x += d1[c] * w1[c].x; e1.y += d1[c] * w1[c].y; e1.z += d1[c] * w1[c].z; e1.w += d1[c] * w1[c].w;
      e2.x += d2[c] * w2[c].x; e2.y += d2[c] * w2[c].y; e2.z += d2[c] * w2[c].z; e2.w += d2[c] * w2[c].w;
      gw1[c] = make_float4(d1[c] * x1.x, d1[c] * x1.y, d1[c] * x1.z, d1[c] * x1.w);
      gw2[c] = make_float4(d2[c] * x2.x, d2[c] * x2.y, d2[c] * x2.z, d2[c] * x2.w);
      gb1[c] = d1[c];
      gb2[c] = d2[c];
    }
    *reinterpret_cast<float4*>(a.dx1 + (int64_t)r * K + 4 * q) = make_float4(e1.x * k1.x, e1.y * k1.y, e1.z * k1.z, e1.w * k1.w);
    *reinterpret_cast<float4*>(a.dx2 + (int64_t)r * K + 4 * q) = make_float4(e2.x * k2.x, e2.y * k2.y, e2.z * k2.z, e2.w * k2.w);
  }
  // reconstruction term of the block's rows (:530) and its gradient
  {
    const int64_t e0 = (int64_t)blk * rpb * a.S, e1 = min((int64_t)rows, (int64_t)(blk + 1) * rpb) * a.S;
    const int64_t half = (int64_t)a.B * a.S;
    for (int64_t i = e0 + threadIdx.x; i < e1; i += 256) {
      const float d = a.x_hat[i] - a.snps[i < half ? i : i - half];
      rec += d * d;
      a.dxhat[i] = a.w.lam[3] * d;
    }
  }
  // the block's four loss sums
  {
    ce = wave_sum(ce); mi = wave_sum(mi); mse = wave_sum(mse); rec = wave_sum(rec);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[wv] = ce; red[4 + wv] = mi; red[8 + wv] = mse; red[12 + wv] = rec; }
    __syncthreads();
    if (threadIdx.x < 4) {
      const float* p = red + 4 * threadIdx.x;
      a.parts[(int64_t)blk * 4 + threadIdx.x] = (p[0] + p[1]) + (p[2] + p[3]);
    }
    __syncthreads();
  }
  // weight / bias gradient partials: the block's row lanes summed in order through LDS, layer after layer
  float* prow = a.wpart + (int64_t)blk * (a.C * K + a.C + a.NR * K + a.NR);
  float* redb = red + 256 * 4 * HL_MAXC;
#pragma unroll
  for (int layer = 0; layer < 2; ++layer) {
    const int CC = layer ? a.NR : a.C;
#pragma unroll
    for (int c = 0; c < HL_MAXC; ++c) {
      const float4 g4 = layer ? gw2[c] : gw1[c];
      float* rc = red + c * 1024;
      rc[threadIdx.x * 4 + 0] = g4.x; rc[threadIdx.x * 4 + 1] = g4.y; rc[threadIdx.x * 4 + 2] = g4.z; rc[threadIdx.x * 4 + 3] = g4.w;
      redb[c * 256 + threadIdx.x] = (q == 0 && rl < rpb) ? (layer ? gb2[c] : gb1[c]) : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < CC * K; idx += 256) {
      const int c = idx / K, k = idx - c * K, qq = k / 4, j = k % 4;
      float t = 0.f;
      for (int l = 0; l < rpb; ++l) t += red[c * 1024 + (l * kq + qq) * 4 + j];
      prow[c * K + k] = t;
    }
    if (threadIdx.x < CC) {
      float t = 0.f;
      for (int l = 0; l < rpb; ++l) t += redb[threadIdx.x * 256 + l * kq];
      prow[CC * K + threadIdx.x] = t;
    }
    prow += CC * K + CC;
    __syncthreads();
  }
  if (blk == 0 && threadIdx.x == 0) {
    a.dgram[0] = a.w.lam[4] * 0.5f; a.dgram[1] = a.w.lam[5]; a.dgram[2] = a.w.lam[4] * 0.5f; a.dgram[3] = 0.f;
    a.dprob[0] = a.w.lam[2];
  }
}

__global__ void __launch_bounds__(256) k_head_loss_fwd(const HeadLossArgs a) {
  __shared__ float red[HL_RED_FLOATS];
  head_loss_body(blockIdx.x, a, red);
}

// The two launches that close a train step's forward — the output heads + loss head above and the Gram-form batch
// losses (k_gram_loss_fwd<true>, unit form) — read disjoint outputs of the launches in front of them and write disjoint
// buffers: as two ROLES of one grid (workgroups [0, n_head) the heads, the rest one Gram row each) they overlap instead of
// running back to back (9.9 + 11.1 us).
__global__ void __launch_bounds__(256)
k_head_loss_gram_fwd(const HeadLossArgs a, unsigned n_head, int B, int RD, int groups, const float* __restrict__ Gall,
                     float* __restrict__ partial, const float* __restrict__ t, int T, float gamma,
                     float* __restrict__ lap_out, GramUnit unit, float* __restrict__ Sall) {
  __shared__ float red[HL_RED_FLOATS];
  if (blockIdx.x < n_head) {
    head_loss_body(blockIdx.x, a, red);
  } else {
    const int idx = (int)(blockIdx.x - n_head);
    gram_loss_fwd_body<true>(idx % B, idx / B, groups, B, RD, Gall, nullptr, partial, t, T, gamma, lap_out, unit, Sall, red);
  }
}

__global__ void __launch_bounds__(256)
k_loss_final(const float* __restrict__ parts, int nparts, const float* __restrict__ gram, int gram_rows,
             const float* __restrict__ prob, int prob_rows, const float* __restrict__ wts, float* __restrict__ out) {
  __shared__ float lds[40];
  loss_final_body(parts, nparts, gram, gram_rows, prob, prob_rows, wts, out, lds);
}

static LossHeadW loss_head_w(const float* lam6, float hp_ce, float hp_mi);
static bool head_loss_ok(int K, int C, int NR) {
  const int kq = K / 4;
  return K % 4 == 0 && kq >= 1 && kq <= 64 && (kq & (kq - 1)) == 0 && C >= 1 && C <= HL_MAXC && NR >= 1 && NR <= HL_MAXC;
}
extern "C" int igcn_head_loss_supported(int K, int C, int NR) { return head_loss_ok(K, C, NR); }
extern "C" int igcn_head_loss_blocks(int B, int K) { return (int)igcn_cdiv((int64_t)2 * B, 256 / (K / 4)); }

// x1 / x2 [2B, K] (keep* [2B, K] or NULL), W1 [C, K] + b1, W2 [NR, K] + b2, y [B] int64, clin [B, NR], x_hat [2B, S],
// snps [B, S]; lam6, hp_*: HOST.  Outputs: logp_out [2B, C], reg_out [2B, NR], dx1 / dx2 [2B, K], dxhat [2B, S],
// parts [blocks, 4], wpart [blocks, C K + C + NR K + NR], dgram [4], dprob [1]  (blocks = igcn_head_loss_blocks(B, K)).
extern "C" int igcn_head_loss_fwd(int B, int K, int C, int NR, int S, const float* x1, const float* keep1, const float* W1,
                                  const float* b1, const float* x2, const float* keep2, const float* W2, const float* b2,
                                  const int64_t* y, const float* clin, const float* x_hat, const float* snps,
                                  const float* lam6 /*HOST [6]*/, float hp_ce, float hp_mi, float* logp_out, float* reg_out,
                                  float* dx1, float* dx2, float* dxhat, float* parts, float* wpart, float* dgram,
                                  float* dprob, void* stream) {
  IGCN_REQUIRE(B > 0 && S > 0 && head_loss_ok(K, C, NR), "head_loss_fwd: K/4 a power of two <= 64, 1 <= C, NR <= 4 (K=%d C=%d NR=%d)",
               K, C, NR);
  IGCN_REQUIRE(x1 && W1 && x2 && W2 && y && clin && x_hat && snps && logp_out && reg_out && dx1 && dx2 && dxhat && parts &&
                   wpart && dgram && dprob,
               "head_loss_fwd: null argument");
  IGCN_REQUIRE((((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)keep1 | (uintptr_t)keep2 | (uintptr_t)W1 | (uintptr_t)W2 |
                 (uintptr_t)dx1 | (uintptr_t)dx2 | (uintptr_t)parts) & 15) == 0,
               "head_loss_fwd: features, factors, weights and their gradients must be 16-byte aligned");
  const HeadLossArgs a = {B, K, C, NR, S, x1, keep1, W1, b1, x2, keep2, W2, b2, y, clin, x_hat, snps,
                          loss_head_w(lam6, hp_ce, hp_mi), logp_out, reg_out, dx1, dx2, dxhat, parts, wpart, dgram, dprob};
  hipLaunchKernelGGL(k_head_loss_fwd, dim3((unsigned)igcn_head_loss_blocks(B, K)), dim3(256), 0, (hipStream_t)stream, a);
  IGCN_CHECK_LAUNCH("head_loss_fwd");
  return IGCN_OK;
}

// igcn_head_loss_fwd and igcn_gram_loss_fwd_rbf_unit (out = NULL: the row partials stay in gscratch) as ONE launch.
extern "C" int igcn_head_loss_gram_fwd(int B, int K, int C, int NR, int S, const float* x1, const float* keep1,
                                       const float* W1, const float* b1, const float* x2, const float* keep2,
                                       const float* W2, const float* b2, const int64_t* y, const float* clin,
                                       const float* x_hat, const float* snps, const float* lam6 /*HOST [6]*/, float hp_ce,
                                       float hp_mi, float* logp_out, float* reg_out, float* dx1, float* dx2, float* dxhat,
                                       float* parts, float* wpart, float* dgram, float* dprob,
                                       int Bg, int RD, int groups, const float* G, const float* tsne, int T, float gamma,
                                       float* lap_out, float* gscratch, const float* gout /*HOST [2 groups]*/, float* Ssym,
                                       void* stream) {
  IGCN_REQUIRE(B > 0 && S > 0 && head_loss_ok(K, C, NR), "head_loss_gram_fwd: K/4 a power of two <= 64, 1 <= C, NR <= 4");
  IGCN_REQUIRE(x1 && W1 && x2 && W2 && y && clin && x_hat && snps && logp_out && reg_out && dx1 && dx2 && dxhat && parts &&
                   wpart && dgram && dprob,
               "head_loss_gram_fwd: null argument");
  IGCN_REQUIRE((((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)keep1 | (uintptr_t)keep2 | (uintptr_t)W1 | (uintptr_t)W2 |
                 (uintptr_t)dx1 | (uintptr_t)dx2 | (uintptr_t)parts) & 15) == 0,
               "head_loss_gram_fwd: features, factors, weights and their gradients must be 16-byte aligned");
  IGCN_REQUIRE(Bg > 0 && groups >= 1 && groups <= 4 && G && lap_out && gscratch && gout && Ssym && (tsne == nullptr || T > 0),
               "head_loss_gram_fwd: bad Gram-loss arguments (groups <= 4)");
  const HeadLossArgs a = {B, K, C, NR, S, x1, keep1, W1, b1, x2, keep2, W2, b2, y, clin, x_hat, snps,
                          loss_head_w(lam6, hp_ce, hp_mi), logp_out, reg_out, dx1, dx2, dxhat, parts, wpart, dgram, dprob};
  GramUnit u{};
  for (int g = 0; g < groups; ++g) {
    u.gc[g] = gout[2 * g];
    u.go[g] = gout[2 * g + 1];
  }
  const unsigned nh = (unsigned)igcn_head_loss_blocks(B, K);
  hipLaunchKernelGGL(k_head_loss_gram_fwd, dim3(nh + (unsigned)(Bg * groups)), dim3(256), 0, (hipStream_t)stream, a, nh, Bg,
                     RD, groups, G, gscratch, tsne, T, gamma, lap_out, u, Ssym);
  IGCN_CHECK_LAUNCH("head_loss_gram_fwd");
  return IGCN_OK;
}

// The loss value from its partial sums (loss_final.h): wts [10] DEVICE = {lam[0..5], hp_ce, hp_mi, B, NR}; out [8] = loss,
// terms[7].  While the stream defers its reductions (igcn_reduce_defer) the job joins the flush; else a launch of its own.
int igcn_queue_loss_final(const float* parts, int nparts, const float* gram, int gram_rows, const float* prob,
                          int prob_rows, const float* wts, float* out, hipStream_t st);      // plan.hip
extern "C" int igcn_loss_final(const float* parts, int nparts, const float* gram, int gram_rows, const float* prob,
                               int prob_rows, const float* wts, float* out, void* stream) {
  IGCN_REQUIRE(parts && nparts >= 1 && gram && gram_rows >= 1 && prob && prob_rows >= 1 && wts && out &&
                   ((uintptr_t)parts & 15) == 0,
               "loss_final: bad arguments");
  if (igcn_queue_loss_final(parts, nparts, gram, gram_rows, prob, prob_rows, wts, out, (hipStream_t)stream)) return IGCN_OK;
  hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(256), 0, (hipStream_t)stream, parts, nparts, gram, gram_rows, prob,
                     prob_rows, wts, out);
  IGCN_CHECK_LAUNCH("loss_final");
  return IGCN_OK;
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
