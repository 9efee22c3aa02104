// SGCN branch kernels: importance masks, GCN normalisation, scatter-aggregate (include/igcn.h).
// All per-node sums walk the plan's stable edge groups, so results are deterministic and the forward
// sums run in the same order as the reference's sequential scatter_add.
#include "common.h"

#define MAX_H0 8

// =================================================================================================
// Long edge lists (dense graphs: hundreds of edges per node): per-node sums over a list reached through a permutation
// (tgt_perm / src_perm) gather 4-byte values at the list's stride — one 64-byte sector per value.  In a dense or
// otherwise "transposed" structure the SAME list position of NEIGHBOURING nodes is adjacent in memory, so the walks
// below put the 64 lanes of a wave on 64 consecutive nodes and advance all of them one list position at a time: the
// permutation entries of a (64 nodes x TL_POS positions) tile are staged in LDS with coalesced reads, then every
// gather of a step is one contiguous run.  A lane owns its node's running sum (stored order: no cross-lane
// reduction); the waves of a workgroup take the tiles of a node group in turn and their partial sums meet in LDS in
// tile order — deterministic, no atomics.  For unstructured sparse graphs the gathers are no better than before, and
// no worse.  By-source lists of source-sorted edge lists are contiguous instead and keep the wave-per-list walk.
// =================================================================================================
#define TL_NODES 64
#define TL_POS 16
#define TL_WAVES 8
#define TL_T (64 * TL_WAVES)

// `fn(k)` is called by the lane that owns node n0 + lane for every edge id k of its list, in list order, for the
// tiles c = wave, wave + TL_WAVES, ...  tile: this wave's LDS scratch [TL_NODES][TL_POS + 1].
template <typename Fn>
__device__ __forceinline__ void tl_walk(int32_t (*tile)[TL_POS + 1], const int32_t* __restrict__ ptr,
                                        const int32_t* __restrict__ perm, int64_t n0, int64_t n_nodes, int wave,
                                        int lane, Fn fn) {
  const int64_t node = n0 + lane;
  const int32_t p0 = node < n_nodes ? ptr[node] : 0, p1 = node < n_nodes ? ptr[node + 1] : 0;
  int maxdeg = p1 - p0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o, 64));
  const int sub = lane >> 4, q = lane & 15;            // staging: 4 nodes x 16 positions per wave instruction
  for (int c = wave; c * TL_POS < maxdeg; c += TL_WAVES) {
#pragma unroll 4
    for (int r = 0; r < TL_NODES; r += 4) {
      const int tt = r + sub;
      const int32_t a0 = __shfl(p0, tt, 64), a1 = __shfl(p1, tt, 64);
      const int32_t pos = a0 + c * TL_POS + q;
      tile[tt][q] = pos < a1 ? perm[pos] : -1;
    }
    // (same wave wrote and reads the tile: LDS operations of a wave execute in order)
#pragma unroll 4
    for (int j = 0; j < TL_POS; ++j) {
      const int32_t k = tile[lane][j];
      if (k >= 0) fn(k);
    }
  }
}

// The other structure: lists that are CONTIGUOUS in the gathered arrays (the by-source lists of an edge list sorted by
// source — what coo_matrix / nonzero / dense_to_sparse produce — have src_perm = identity): there the lanes of a wave
// stride ONE node's list.  The waves of the workgroup take the 64 nodes in turn; out[node - n0] = sum_k fn(k, node).
template <typename Fn>
__device__ __forceinline__ void tl_rowwalk(float* out /*LDS [TL_NODES]*/, const int32_t* __restrict__ ptr,
                                           const int32_t* __restrict__ perm, int64_t n0, int64_t n_nodes, int wave,
                                           int lane, Fn fn) {
  for (int nn = wave; nn < TL_NODES; nn += TL_WAVES) {
    const int64_t node = n0 + nn;
    float acc = 0.f;
    if (node < n_nodes) {                               // wave-uniform
      const int32_t p1 = ptr[node + 1];
#pragma unroll 4
      for (int32_t pp = ptr[node] + lane; pp < p1; pp += 64) acc += fn(perm[pp], node);
      acc = wave_sum_all(acc);
    }
    if (lane == 0) out[nn] = acc;
  }
}

// sum of the waves' per-node partials in wave order; red: [TL_WAVES][TL_NODES]; result valid for tid < TL_NODES
__device__ __forceinline__ float tl_combine(float v, float (*red)[TL_NODES], int wave, int lane) {
  __syncthreads();
  red[wave][lane] = v;
  __syncthreads();
  float t = 0.f;
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < TL_WAVES; ++w) t += red[w][lane];
  }
  return t;
}

// gcn_norm forward, dense graphs: deg, dis, wl for 64 nodes per workgroup
__global__ void __launch_bounds__(TL_T)
k_gcn_norm_fwd_tiled(int64_t n_nodes, const float* __restrict__ ew, const int32_t* __restrict__ src32,
                     const int32_t* __restrict__ tgt_ptr, const int32_t* __restrict__ tgt_perm,
                     const int32_t* __restrict__ loop_edge, float* __restrict__ dis, float* __restrict__ wl) {
  __shared__ int32_t tiles[TL_WAVES][TL_NODES][TL_POS + 1];
  __shared__ float red[TL_WAVES][TL_NODES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * TL_NODES, node = n0 + lane;
  float deg = 0.f;
  tl_walk(tiles[wave], tgt_ptr, tgt_perm, n0, n_nodes, wave, lane, [&](int32_t k) {
    if (src32[k] != (int32_t)node) deg += ew[k];
  });
  deg = tl_combine(deg, red, wave, lane);
  if (wave == 0 && node < n_nodes) {
    const int32_t le = loop_edge[node];
    const float lw = le >= 0 ? ew[le] : 1.f;
    deg += lw;
    float d = 1.0f / sqrtf(deg);
    if (deg == 0.f) d = 0.f;
    dis[node] = d;
    wl[node] = lw;
  }
}

// gcn_norm backward phase 1, dense graphs (formula: k_gcn_norm_bwd_deg)
__global__ void __launch_bounds__(TL_T)
k_gcn_norm_bwd_deg_tiled(int64_t n_nodes, const float* __restrict__ ew, const float* __restrict__ dis,
                         const float* __restrict__ wl, const float* __restrict__ dwhat,
                         const float* __restrict__ dwhat_loop, const int32_t* __restrict__ src32,
                         const int32_t* __restrict__ dst32, const int32_t* __restrict__ tgt_ptr,
                         const int32_t* __restrict__ tgt_perm, const int32_t* __restrict__ src_ptr,
                         const int32_t* __restrict__ src_perm, float* __restrict__ ddeg) {
  __shared__ int32_t tiles[TL_WAVES][TL_NODES][TL_POS + 1];
  __shared__ float red[TL_WAVES][TL_NODES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * TL_NODES, node = n0 + lane;
  __shared__ float bysrc[TL_NODES];
  tl_rowwalk(bysrc, src_ptr, src_perm, n0, n_nodes, wave, lane, [&](int32_t k, int64_t nd) {
    const int32_t t = dst32[k];
    return t != (int32_t)nd ? dwhat[k] * ew[k] * dis[t] : 0.f;
  });
  float dd = 0.f;
  tl_walk(tiles[wave], tgt_ptr, tgt_perm, n0, n_nodes, wave, lane, [&](int32_t k) {
    const int32_t sn = src32[k];
    if (sn != (int32_t)node) dd += dwhat[k] * ew[k] * dis[sn];
  });
  dd = tl_combine(dd, red, wave, lane);                 // (its barriers also publish bysrc)
  if (wave == 0 && node < n_nodes) {
    const float di = dis[node];
    dd += bysrc[lane];
    dd += 2.f * dwhat_loop[node] * wl[node] * di;
    ddeg[node] = -0.5f * di * di * di * dd;
  }
}

#include "mask_reg.h"
struct EmReg {                    // greg == NULL: no regulariser in this launch
  const float* greg;              // backward: d loss / d (regulariser), device scalar
  const float* snps;              // SNP mask logits [n_snps] (may be NULL)
  float* dsnps;                   // backward: their gradient (regulariser part + mask part)
  int n_snps;
  float l1_x, ent_x, l1_e, ent_e, eps;
  // the SNP mask itself (cal_probability :147-151) for the stacked sweep: feat [B, n_snps] -> full [2B, n_snps] =
  // (feat | feat * sigmoid(logits)); backward: d_full [2B, n_snps] (its masked half is read)
  const float* feat;
  float* full;
  const float* d_full;
  int B;
};

// edge-mask backward node pass, dense graphs (formula: k_edge_mask_bwd_nodes); partial row layout identical
__global__ void __launch_bounds__(TL_T)
k_edge_mask_bwd_nodes_tiled(int64_t n_nodes, int rois, int h0, const float* __restrict__ x,
                            const float* __restrict__ prob, const float* __restrict__ pb,
                            const float* __restrict__ ew, const float* __restrict__ e,
                            const float* __restrict__ d_xm, const float* __restrict__ d_ewm,
                            const float* __restrict__ d_e, const float* __restrict__ d_x_plain,
                            const int32_t* __restrict__ tgt_ptr, const int32_t* __restrict__ tgt_perm,
                            const int32_t* __restrict__ src_ptr, const int32_t* __restrict__ src_perm,
                            float* __restrict__ dx, float* __restrict__ gx, float* __restrict__ pb_partial, EmReg rg,
                            float inv_ne) {
  __shared__ int32_t tiles[TL_WAVES][TL_NODES][TL_POS + 1];
  __shared__ float red[TL_WAVES][TL_NODES];
  __shared__ float pbred[TL_NODES][2 * MAX_H0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * TL_NODES, node = n0 + lane;
  const float gre = rg.greg ? rg.greg[0] * inv_ne : 0.f;
  auto dz = [&](int32_t k) {
    const float ek = e[k];
    float up = (d_ewm ? d_ewm[k] * ew[k] : 0.f) + (d_e ? d_e[k] : 0.f);
    if (rg.greg) up += gre * em_reg_grad(ek, rg.l1_e, rg.ent_e, rg.eps);
    return up * ek * (1.f - ek);
  };
  __shared__ float bysrc[TL_NODES];
  tl_rowwalk(bysrc, src_ptr, src_perm, n0, n_nodes, wave, lane, [&](int32_t k, int64_t) { return dz(k); });
  float T = 0.f;
  tl_walk(tiles[wave], tgt_ptr, tgt_perm, n0, n_nodes, wave, lane, [&](int32_t k) { T += dz(k); });
  T = tl_combine(T, red, wave, lane);                   // (its barriers also publish bysrc)
  const float S = bysrc[lane];
  if (wave == 0) {
    for (int j = 0; j < 2 * MAX_H0; ++j) pbred[lane][j] = 0.f;
    if (node < n_nodes) {
      const int64_t r = (node % rois) * h0;
      for (int h = 0; h < h0; ++h) {
        const float xv = x[node * h0 + h], pv = prob[r + h];
        const float g = (d_xm ? d_xm[node * h0 + h] : 0.f) + pb[h] * S + pb[h0 + h] * T;
        dx[node * h0 + h] = g * pv + (d_x_plain ? d_x_plain[node * h0 + h] : 0.f);
        gx[node * h0 + h] = g * xv;
        pbred[lane][h] = xv * pv * S;
        pbred[lane][MAX_H0 + h] = xv * pv * T;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * MAX_H0) {
    float t = 0.f;
    for (int l = 0; l < TL_NODES; ++l) t += pbred[l][threadIdx.x];
    pb_partial[(int64_t)blockIdx.x * 2 * MAX_H0 + threadIdx.x] = t;
  }
}

// =================================================================================================
// edge mask forward  (cal_probability, kernel/sgcn_img_snp.py:133-151)
// =================================================================================================
template <bool REG>
__global__ void __launch_bounds__(256)
k_edge_mask_fwd(int64_t n_nodes, int64_t n_edges, int rois, int h0, const float* __restrict__ x,
                const float* __restrict__ prob, const float* __restrict__ pb,
                const float* __restrict__ ew, const int32_t* __restrict__ src32,
                const int32_t* __restrict__ dst32, float* __restrict__ xm, float* __restrict__ e,
                float* __restrict__ ewm, float* __restrict__ x_plain, float* __restrict__ ew_plain, EmReg rg,
                float* __restrict__ reg_partial) {
  __shared__ float red[16];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  if (i < n_nodes * h0) {
    const int64_t node = i / h0;
    const int h = (int)(i - node * h0);
    const float xv = x[i];
    xm[i] = xv * prob[(node % rois) * h0 + h];
    if (x_plain) x_plain[i] = xv;                     // first half of the stacked (plain | masked) batch
  }
  if (i < n_edges) {
    const int64_t s = src32[i], d = dst32[i];
    const int64_t rs = (s % rois) * h0, rd = (d % rois) * h0;
    float z = 0.f;
    for (int h = 0; h < h0; ++h) z += (x[s * h0 + h] * prob[rs + h]) * pb[h];
    for (int h = 0; h < h0; ++h) z += (x[d * h0 + h] * prob[rd + h]) * pb[h0 + h];
    const float p = 1.f / (1.f + expf(-z));
    e[i] = p;
    const float wv = ew[i];
    ewm[i] = wv * p;
    if (ew_plain) ew_plain[i] = wv;
    if (REG) acc += em_reg_term(p, rg.l1_e, rg.ent_e, rg.eps) / (float)n_edges;
  }
  if (REG && rg.feat && i < (int64_t)rg.B * rg.n_snps) {   // SNP mask of the stacked sweep: plain | masked halves
    const float v = rg.feat[i];
    rg.full[i] = v;
    rg.full[(int64_t)rg.B * rg.n_snps + i] = v * (1.f / (1.f + __expf(-rg.snps[i % rg.n_snps])));   // = k_snps_mask_fwd, bit for bit
  }
  if (REG) {                                          // loss_probability's three means, one partial per workgroup
    const int64_t np = (int64_t)rois * h0;
    if (i < np) acc += em_reg_term(1.f / (1.f + expf(-prob[i])), rg.l1_x, rg.ent_x, rg.eps) / (float)np;
    if (rg.snps && i < rg.n_snps)
      acc += em_reg_term(1.f / (1.f + expf(-rg.snps[i])), rg.l1_x, rg.ent_x, rg.eps) / (float)rg.n_snps;
    acc = block_sum_all(acc, red);
    if (threadIdx.x == 0) reg_partial[blockIdx.x] = acc;
  }
}

static int64_t em_fwd_items(int64_t n_nodes, int64_t n_edges, int h0, int64_t n_snps) {   // n_snps: logits, or B * logits
  int64_t n = n_nodes * h0 > n_edges ? n_nodes * h0 : n_edges;
  return n > n_snps ? n : n_snps;
}

extern "C" int igcn_edge_mask_fwd(int64_t n_nodes, int64_t n_edges, int rois, int h0, const float* x,
                                  const float* prob, const float* prob_bias, const float* ew, const int32_t* src32,
                                  const int32_t* dst32, float* xm, float* e, float* ewm, float* x_plain,
                                  float* ew_plain, void* stream) {
  IGCN_REQUIRE(rois > 0 && h0 > 0 && h0 <= MAX_H0 && n_nodes % rois == 0,
               "edge_mask_fwd: need n_nodes %% rois == 0 and 0 < h0 <= %d (n_nodes=%lld rois=%d h0=%d)", MAX_H0,
               (long long)n_nodes, rois, h0);
  const int64_t n = em_fwd_items(n_nodes, n_edges, h0, 0);
  if (n == 0) return IGCN_OK;
  hipLaunchKernelGGL(k_edge_mask_fwd<false>, dim3((unsigned)igcn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     n_nodes, n_edges, rois, h0, x, prob, prob_bias, ew, src32, dst32, xm, e, ewm, x_plain, ew_plain,
                     EmReg{}, nullptr);
  IGCN_CHECK_LAUNCH("edge_mask_fwd");
  return IGCN_OK;
}

// The same launch also leaves loss_probability (kernel/sgcn_img_snp.py:153-181) as igcn_edge_mask_reg_blocks()
// workgroup partials whose SUM is the loss — mean r_x(sigmoid(prob)) + mean r_e(e) + mean r_x(sigmoid(snps_logits)),
// snps_logits NULL: without the last — so that a train step needs no regulariser launch of its own.
extern "C" int igcn_edge_mask_reg_blocks(int64_t n_nodes, int64_t n_edges, int h0, int n_snps, int snps_rows) {
  return (int)igcn_cdiv(em_fwd_items(n_nodes, n_edges, h0, (int64_t)(snps_rows > 0 ? snps_rows : 1) * n_snps), 256);
}
// snps_feat [snps_rows, n_snps] (or NULL): the launch also writes the SNP mask of the stacked sweep,
// snps_full [2 snps_rows, n_snps] = (snps_feat | snps_feat * sigmoid(snps_logits)).
extern "C" int igcn_edge_mask_fwd_reg(int64_t n_nodes, int64_t n_edges, int rois, int h0, const float* x,
                                      const float* prob, const float* prob_bias, const float* ew, const int32_t* src32,
                                      const int32_t* dst32, float* xm, float* e, float* ewm, float* x_plain,
                                      float* ew_plain, const float* snps_logits, int n_snps, float l1_x, float ent_x,
                                      float l1_e, float ent_e, float eps, float* reg_partial, const float* snps_feat,
                                      int snps_rows, float* snps_full, void* stream) {
  IGCN_REQUIRE(rois > 0 && h0 > 0 && h0 <= MAX_H0 && n_nodes % rois == 0 && n_nodes > 0 && n_edges > 0 && reg_partial &&
                   n_snps >= 0 && (snps_feat == nullptr || (snps_logits && snps_full && snps_rows > 0)),
               "edge_mask_fwd_reg: bad arguments");
  const int ns = snps_logits ? n_snps : 0;
  const int64_t n = em_fwd_items(n_nodes, n_edges, h0, (int64_t)(snps_feat ? snps_rows : 1) * ns);
  const EmReg rg = {nullptr, snps_logits, nullptr, ns, l1_x, ent_x, l1_e, ent_e, eps, snps_feat, snps_full, nullptr,
                    snps_feat ? snps_rows : 0};
  hipLaunchKernelGGL(k_edge_mask_fwd<true>, dim3((unsigned)igcn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     n_nodes, n_edges, rois, h0, x, prob, prob_bias, ew, src32, dst32, xm, e, ewm, x_plain, ew_plain, rg,
                     reg_partial);
  IGCN_CHECK_LAUNCH("edge_mask_fwd_reg");
  return IGCN_OK;
}

// =================================================================================================
// edge mask backward
//   dz_k = (d_ewm_k * ew_k + d_e_k) * e_k (1 - e_k)
//   g_i  = d_xm_i + pb[:h0] * S_i + pb[h0:] * T_i,   S_i = sum_{k: src=i} dz_k, T_i = sum_{k: dst=i} dz_k
//   dx_i = g_i * prob[roi(i)] ;  dprob[r] = sum_graphs g_i * x_i ;  dpb = (sum_i xm_i S_i , sum_i xm_i T_i)
// =================================================================================================
// LPN lanes share a node (1: low degree; 64: dense graphs, the wave strides the node's edge lists); a block covers
// 256/LPN consecutive nodes and writes one partial row.
template <int LPN>
__global__ void __launch_bounds__(256)
k_edge_mask_bwd_nodes(int64_t n_nodes, int rois, int h0, const float* __restrict__ x,
                      const float* __restrict__ prob, const float* __restrict__ pb,
                      const float* __restrict__ ew, const float* __restrict__ e,
                      const float* __restrict__ d_xm, const float* __restrict__ d_ewm,
                      const float* __restrict__ d_e, const float* __restrict__ d_x_plain,
                      const int32_t* __restrict__ tgt_ptr,
                      const int32_t* __restrict__ tgt_perm, const int32_t* __restrict__ src_ptr,
                      const int32_t* __restrict__ src_perm, float* __restrict__ dx,
                      float* __restrict__ gx /*[N,h0]*/, float* __restrict__ pb_partial /*[nblk,2h0]*/, EmReg rg,
                      float inv_ne) {
  __shared__ float red[4 * 2 * MAX_H0];
  const float gre = rg.greg ? rg.greg[0] * inv_ne : 0.f;
  float acc[2 * MAX_H0];
#pragma unroll
  for (int j = 0; j < 2 * MAX_H0; ++j) acc[j] = 0.f;
  const int sub = threadIdx.x % LPN;
  const int64_t i = (int64_t)blockIdx.x * (256 / LPN) + threadIdx.x / LPN;
  if (i < n_nodes) {                               // uniform over the LPN lanes of a node (the wave for LPN=64)
    // the first entry of BOTH lists per lane as one chain of three round trips (pointers; permutation entries; the
    // edge operands) — list after list it was six, in a kernel that is nothing but that chain (k = 3 graphs with four
    // lanes per node rarely have a second entry per lane; the loops behind take those)
    const int32_t s0 = src_ptr[i], s1 = src_ptr[i + 1], t0 = tgt_ptr[i], t1 = tgt_ptr[i + 1];
    // the node's own operands ride with the pointer loads (they are not needed before the list sums are done)
    const int64_t r = (i % rois) * h0;
    float nx[MAX_H0], npv[MAX_H0], ndxm[MAX_H0], ndxp[MAX_H0];
#pragma unroll
    for (int h = 0; h < MAX_H0; ++h) {
      const bool on = sub == 0 && h < h0;
      nx[h] = on ? x[i * h0 + h] : 0.f;
      npv[h] = on ? prob[r + h] : 0.f;
      ndxm[h] = on && d_xm ? d_xm[i * h0 + h] : 0.f;
      ndxp[h] = on && d_x_plain ? d_x_plain[i * h0 + h] : 0.f;
    }
    int32_t ps = s0 + sub, pt = t0 + sub;
    const bool hs = ps < s1, ht = pt < t1;
    int32_t ks = hs ? src_perm[ps] : -1, kt = ht ? tgt_perm[pt] : -1;
    if (ks < 0) ks = kt;                             // a lane with one entry reads it twice, a lane with none reads nothing
    if (kt < 0) kt = ks;
    auto term = [&](int32_t k) {
      const float ek = e[k];
      float up = (d_ewm ? d_ewm[k] * ew[k] : 0.f) + (d_e ? d_e[k] : 0.f);
      if (rg.greg) up += gre * em_reg_grad(ek, rg.l1_e, rg.ent_e, rg.eps);
      return up * ek * (1.f - ek);
    };
    float S = 0.f, T = 0.f;
    if (ks >= 0) {                                   // ONE region: both entries' operands are requested together
      const float fs = term(ks), ft = term(kt);
      S = hs ? fs : 0.f;
      T = ht ? ft : 0.f;
    }
    for (ps += LPN; ps < s1; ps += LPN) S += term(src_perm[ps]);
    for (pt += LPN; pt < t1; pt += LPN) T += term(tgt_perm[pt]);
    S = group_sum_all<LPN>(S);
    T = group_sum_all<LPN>(T);
    if (sub == 0) {
#pragma unroll
      for (int h = 0; h < MAX_H0; ++h)
        if (h < h0) {
          const float xv = nx[h], pv = npv[h];
          const float g = ndxm[h] + pb[h] * S + pb[h0 + h] * T;
          dx[i * h0 + h] = g * pv + ndxp[h];                                       // + the plain half of a stacked batch
          gx[i * h0 + h] = g * xv;
          acc[h] += xv * pv * S;
          acc[MAX_H0 + h] += xv * pv * T;
        }
    }
  }
  block_reduce_vec<2 * MAX_H0>(acc, red, pb_partial + (int64_t)blockIdx.x * 2 * MAX_H0);
}

// dprob[r,h] = sum_b gx[(b*rois + r), h]: one block per (roi, feature).  One MORE block sums the node kernel's block
// partials into the bias gradient dpb [2 h0] (16 row lanes x 16 column slots, lanes combined in order) — that sum used
// to be two further launches (column sums of the partial rows, then the pick-and-store).
__global__ void __launch_bounds__(256)
k_edge_mask_bwd_prob(int64_t n_graphs, int rois, int h0, const float* __restrict__ gx, float* __restrict__ dprob,
                     int64_t nblk, const float* __restrict__ partial, float* __restrict__ dpb,
                     const float* __restrict__ prob, EmReg rg) {
  __shared__ float red[16 * 16];
  const int j = blockIdx.x;
  if (j > rois * h0) {                               // one extra workgroup per SNP: the mask logit's gradient —
    const int k = j - (rois * h0 + 1);               // regulariser part + (stacked sweep) sum_b d_masked[b,k] feat[b,k]
    float acc = 0.f;
    if (rg.d_full && rg.feat) {
      const float* dm = rg.d_full + (int64_t)rg.B * rg.n_snps;
      for (int b = threadIdx.x; b < rg.B; b += 256) acc += dm[(int64_t)b * rg.n_snps + k] * rg.feat[(int64_t)b * rg.n_snps + k];
    }
    acc = block_sum_all(acc, red);
    if (threadIdx.x == 0) {
      const float p = 1.f / (1.f + expf(-rg.snps[k])), pm = 1.f / (1.f + __expf(-rg.snps[k]));   // (regulariser's | mask's sigmoid)
      rg.dsnps[k] = acc * pm * (1.f - pm) + rg.greg[0] / (float)rg.n_snps * em_reg_grad(p, rg.l1_x, rg.ent_x, rg.eps) * p * (1.f - p);
    }
    return;
  }
  if (j == rois * h0) {
    const int rl = threadIdx.x >> 4, cs = threadIdx.x & 15;
    float t = 0.f;
    if (cs < 2 * h0) {
      const int col = cs < h0 ? cs : MAX_H0 + (cs - h0);
#pragma unroll 4
      for (int64_t r = rl; r < nblk; r += 16) t += partial[r * 2 * MAX_H0 + col];
    }
    red[rl * 16 + cs] = t;
    __syncthreads();
    if (rl == 0 && cs < 2 * h0) {
      float a = 0.f;
      for (int l = 0; l < 16; ++l) a += red[l * 16 + cs];
      dpb[cs] = a;
    }
    return;
  }
  float t = 0.f;
  for (int64_t b = threadIdx.x; b < n_graphs; b += 256) t += gx[b * rois * h0 + j];
  t = block_sum_all(t, red);
  if (threadIdx.x == 0) {
    if (rg.greg) {                                   // + the regulariser's own term on sigmoid(prob)
      const float p = 1.f / (1.f + expf(-prob[j]));
      t += rg.greg[0] / (float)(rois * h0) * em_reg_grad(p, rg.l1_x, rg.ent_x, rg.eps) * p * (1.f - p);
    }
    dprob[j] = t;
  }
}

static int em_bwd_impl(int64_t n_nodes, int64_t n_edges, int rois, int h0, const float* x, const float* prob,
                       const float* prob_bias, const float* ew, const float* e, const float* d_xm, const float* d_ewm,
                       const float* d_e, const float* d_x_plain, const int32_t* tgt_ptr, const int32_t* tgt_perm,
                       const int32_t* src_ptr, const int32_t* src_perm, float* dx, float* dprob, float* dprob_bias,
                       float* scratch, EmReg rg, hipStream_t st) {
  const bool dense = n_edges >= 16 * n_nodes;
  const bool tiled = dense && !igcn_opt(IGCN_OPT_NO_TILED_LISTS);
  const int64_t nblk = tiled ? igcn_cdiv(n_nodes, TL_NODES) : igcn_cdiv(n_nodes, dense ? 4 : 64);
  const float inv_ne = n_edges > 0 ? 1.0f / (float)n_edges : 0.f;
  float* gx = scratch;
  float* part = scratch + n_nodes * h0;  // [nblk, 2*MAX_H0]
  if (tiled)                                       // dense graphs: lanes own nodes, permutation tiles through LDS
    hipLaunchKernelGGL(k_edge_mask_bwd_nodes_tiled, dim3((unsigned)nblk), dim3(TL_T), 0, st, n_nodes, rois, h0, x, prob,
                       prob_bias, ew, e, d_xm, d_ewm, d_e, d_x_plain, tgt_ptr, tgt_perm, src_ptr, src_perm, dx, gx, part, rg,
                       inv_ne);
  else if (dense)                                  // a wave per node strides its edge lists
    hipLaunchKernelGGL(k_edge_mask_bwd_nodes<64>, dim3((unsigned)nblk), dim3(256), 0, st, n_nodes, rois, h0, x, prob,
                       prob_bias, ew, e, d_xm, d_ewm, d_e, d_x_plain, tgt_ptr, tgt_perm, src_ptr, src_perm, dx, gx, part, rg,
                       inv_ne);
  else                                             // k = 3 graphs: four lanes share a node's six list entries
    hipLaunchKernelGGL(k_edge_mask_bwd_nodes<4>, dim3((unsigned)nblk), dim3(256), 0, st, n_nodes, rois, h0, x, prob,
                       prob_bias, ew, e, d_xm, d_ewm, d_e, d_x_plain, tgt_ptr, tgt_perm, src_ptr, src_perm, dx, gx, part, rg,
                       inv_ne);
  static_assert(2 * MAX_H0 <= 16, "k_edge_mask_bwd_prob: the bias-gradient block has 16 column slots");
  const bool snps_block = rg.greg && rg.snps && rg.dsnps && rg.n_snps > 0;
  hipLaunchKernelGGL(k_edge_mask_bwd_prob, dim3((unsigned)(rois * h0 + 1 + (snps_block ? rg.n_snps : 0))), dim3(256), 0, st,
                     n_nodes / rois, rois, h0, gx, dprob, nblk, part, dprob_bias, prob, rg);
  IGCN_CHECK_LAUNCH("edge_mask_bwd");
  return IGCN_OK;
}

extern "C" int igcn_edge_mask_bwd(int64_t n_nodes, int64_t n_edges, int rois, int h0, const float* x,
                                  const float* prob, const float* prob_bias, const float* ew, const float* e,
                                  const float* d_xm, const float* d_ewm, const float* d_e, const float* d_x_plain,
                                  const int32_t* tgt_ptr, const int32_t* tgt_perm, const int32_t* src_ptr,
                                  const int32_t* src_perm, float* dx, float* dprob, float* dprob_bias, float* scratch,
                                  void* stream) {
  IGCN_REQUIRE(rois > 0 && h0 > 0 && h0 <= MAX_H0 && n_nodes % rois == 0, "edge_mask_bwd: bad rois/h0");
  return em_bwd_impl(n_nodes, n_edges, rois, h0, x, prob, prob_bias, ew, e, d_xm, d_ewm, d_e, d_x_plain, tgt_ptr, tgt_perm,
                     src_ptr, src_perm, dx, dprob, dprob_bias, scratch, EmReg{}, (hipStream_t)stream);
}

// ... with the gradient of the regulariser of igcn_edge_mask_fwd_reg inside: d_reg [1] (device) = d loss / d (sum of
// the partials); its edge part joins d_e per edge, its sigmoid(prob) part dprob, and dsnps [n_snps] (may be NULL with
// snps_logits) receives the SNP mask logits' part.
extern "C" int igcn_edge_mask_bwd_reg(int64_t n_nodes, int64_t n_edges, int rois, int h0, const float* x,
                                      const float* prob, const float* prob_bias, const float* ew, const float* e,
                                      const float* d_xm, const float* d_ewm, const float* d_e, const float* d_x_plain,
                                      const int32_t* tgt_ptr, const int32_t* tgt_perm, const int32_t* src_ptr,
                                      const int32_t* src_perm, const float* d_reg, const float* snps_logits, int n_snps,
                                      float l1_x, float ent_x, float l1_e, float ent_e, float eps, float* dx, float* dprob,
                                      float* dprob_bias, float* dsnps, float* scratch, const float* snps_feat,
                                      int snps_rows, const float* d_snps_full, void* stream) {
  IGCN_REQUIRE(rois > 0 && h0 > 0 && h0 <= MAX_H0 && n_nodes % rois == 0 && d_reg, "edge_mask_bwd_reg: bad arguments");
  const EmReg rg = {d_reg, snps_logits, dsnps, snps_logits ? n_snps : 0, l1_x, ent_x, l1_e, ent_e, eps, snps_feat, nullptr,
                    d_snps_full, snps_feat ? snps_rows : 0};
  return em_bwd_impl(n_nodes, n_edges, rois, h0, x, prob, prob_bias, ew, e, d_xm, d_ewm, d_e, d_x_plain, tgt_ptr, tgt_perm,
                     src_ptr, src_perm, dx, dprob, dprob_bias, scratch, rg, (hipStream_t)stream);
}

// =================================================================================================
// gcn_norm forward: one thread per node
// =================================================================================================
template <int LPN>
__global__ void __launch_bounds__(256)
k_gcn_norm_fwd(int64_t n_nodes, const float* __restrict__ ew, const int32_t* __restrict__ src32,
               const int32_t* __restrict__ tgt_ptr, const int32_t* __restrict__ tgt_perm,
               const int32_t* __restrict__ loop_edge, float* __restrict__ dis, float* __restrict__ wl) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = gid / LPN;
  const int sub = (int)(gid % LPN);
  if (i >= n_nodes) return;
  float deg = 0.f;
  for (int32_t p = tgt_ptr[i] + sub; p < tgt_ptr[i + 1]; p += LPN) {
    const int32_t k = tgt_perm[p];
    if (src32[k] != (int32_t)i) deg += ew[k];
  }
  deg = group_sum_all<LPN>(deg);
  if (sub != 0) return;
  const int32_t le = loop_edge[i];
  const float lw = le >= 0 ? ew[le] : 1.f;
  deg += lw;
  float d = 1.0f / sqrtf(deg);      // deg^-1/2 ; deg==0 -> inf -> masked to 0 ; deg<0 -> NaN (as torch.pow)
  if (deg == 0.f) d = 0.f;
  dis[i] = d;
  wl[i] = lw;
}

struct EdgeRec {      // one 8-byte record per edge: neighbour index + normalised coefficient
  int32_t idx;
  float w;
};

// what[k] = dis[src]*ew[k]*dis[dst] (0 for stored loops: they are replaced) ; what_loop[i] = dis[i]*wl[i]*dis[i]
__global__ void k_gcn_norm_coef(int64_t n_nodes, int64_t n_edges, const float* __restrict__ ew,
                                const float* __restrict__ dis, const float* __restrict__ wl,
                                const int32_t* __restrict__ src32, const int32_t* __restrict__ dst32,
                                const int32_t* __restrict__ tgt_perm, const int32_t* __restrict__ src_perm,
                                float* __restrict__ what, float* __restrict__ what_loop,
                                EdgeRec* __restrict__ tstream, EdgeRec* __restrict__ sstream) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_edges) {
    const int32_t s = src32[i], t = dst32[i];
    what[i] = s != t ? dis[s] * ew[i] * dis[t] : 0.f;
    // the same coefficients as two streams in the order the aggregation kernels consume them: position p of the
    // by-target grouping holds (source node, coefficient), position p of the by-source grouping (target node, coef.)
    const int32_t kt = tgt_perm[i], ks = src_perm[i];
    const int32_t ts = src32[kt], tt = dst32[kt];
    tstream[i].idx = ts;
    tstream[i].w = ts != tt ? dis[ts] * ew[kt] * dis[tt] : 0.f;
    const int32_t ss = src32[ks], st = dst32[ks];
    sstream[i].idx = st;
    sstream[i].w = ss != st ? dis[ss] * ew[ks] * dis[st] : 0.f;
  }
  if (i < n_nodes) {
    const float d = dis[i];
    what_loop[i] = d * wl[i] * d;
  }
}

// Dense graphs: the same outputs as k_gcn_norm_coef in three coalesced passes instead of one pass of ~12 strided
// 4-byte gathers per edge: (1) what / what_loop in edge order; (2) the by-target record stream through the tiled
// lane-per-node walk (gathers of a step contiguous), its records transposed back through LDS so that a node's 16
// records leave as one 128-byte run; (3) the by-source stream position by position (contiguous lists).
__global__ void k_gcn_norm_what(int64_t n_nodes, int64_t n_edges, const float* __restrict__ ew,
                                const float* __restrict__ dis, const float* __restrict__ wl,
                                const int32_t* __restrict__ src32, const int32_t* __restrict__ dst32,
                                float* __restrict__ what, float* __restrict__ what_loop) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_edges) {
    const int32_t sn = src32[i], tn = dst32[i];
    what[i] = sn != tn ? dis[sn] * ew[i] * dis[tn] : 0.f;
  }
  if (i < n_nodes) {
    const float d = dis[i];
    what_loop[i] = d * wl[i] * d;
  }
}

__global__ void __launch_bounds__(TL_T)
k_stream_by_target_tiled(int64_t n_nodes, const float* __restrict__ what, const int32_t* __restrict__ src32,
                         const int32_t* __restrict__ tgt_ptr, const int32_t* __restrict__ tgt_perm,
                         EdgeRec* __restrict__ tstream) {
  __shared__ int32_t tiles[TL_WAVES][TL_NODES][TL_POS + 1];
  __shared__ EdgeRec recs[TL_WAVES][TL_NODES][TL_POS + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * TL_NODES, node = n0 + lane;
  const int32_t p0 = node < n_nodes ? tgt_ptr[node] : 0, p1 = node < n_nodes ? tgt_ptr[node + 1] : 0;
  int maxdeg = p1 - p0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o, 64));
  const int sub = lane >> 4, q = lane & 15;
  int32_t (*tile)[TL_POS + 1] = tiles[wave];
  EdgeRec (*rec)[TL_POS + 1] = recs[wave];
  for (int c = wave; c * TL_POS < maxdeg; c += TL_WAVES) {
#pragma unroll 4
    for (int r = 0; r < TL_NODES; r += 4) {
      const int tt = r + sub;
      const int32_t a0 = __shfl(p0, tt, 64), a1 = __shfl(p1, tt, 64);
      const int32_t pos = a0 + c * TL_POS + q;
      tile[tt][q] = pos < a1 ? tgt_perm[pos] : -1;
    }
#pragma unroll 4
    for (int j = 0; j < TL_POS; ++j) {                   // lanes = nodes: the gathers of a step are one run
      const int32_t k = tile[lane][j];
      EdgeRec e;
      e.idx = 0; e.w = 0.f;
      if (k >= 0) { e.idx = src32[k]; e.w = what[k]; }
      rec[lane][j] = e;
    }
#pragma unroll 4
    for (int r = 0; r < TL_NODES; r += 4) {              // lanes = (node, position): 128-byte runs out
      const int tt = r + sub;
      const int32_t a0 = __shfl(p0, tt, 64), a1 = __shfl(p1, tt, 64);
      const int32_t pos = a0 + c * TL_POS + q;
      if (pos < a1) tstream[pos] = rec[tt][q];
    }
  }
}

__global__ void k_stream_by_source(int64_t n_edges, const float* __restrict__ what, const int32_t* __restrict__ dst32,
                                   const int32_t* __restrict__ src_perm, EdgeRec* __restrict__ sstream) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_edges) return;
  const int32_t k = src_perm[i];
  EdgeRec e;
  e.idx = dst32[k];
  e.w = what[k];
  sstream[i] = e;
}

extern "C" int igcn_gcn_norm_fwd(int64_t n_nodes, int64_t n_edges, const float* ew, const int32_t* src32,
                                 const int32_t* dst32, const int32_t* tgt_ptr, const int32_t* tgt_perm,
                                 const int32_t* src_perm, const int32_t* loop_edge, float* dis, float* wl,
                                 float* what, float* what_loop, void* tstream, void* sstream, void* stream) {
  if (n_nodes == 0) return IGCN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (n_edges >= 16 * n_nodes && !igcn_opt(IGCN_OPT_NO_TILED_LISTS))      // dense graphs: lanes own nodes
    hipLaunchKernelGGL(k_gcn_norm_fwd_tiled, dim3((unsigned)igcn_cdiv(n_nodes, TL_NODES)), dim3(TL_T), 0, st, n_nodes,
                       ew, src32, tgt_ptr, tgt_perm, loop_edge, dis, wl);
  else if (n_edges >= 16 * n_nodes)                // (A/B: one wave per node)
    hipLaunchKernelGGL(k_gcn_norm_fwd<64>, dim3((unsigned)igcn_cdiv(n_nodes * 64, 256)), dim3(256), 0, st, n_nodes, ew,
                       src32, tgt_ptr, tgt_perm, loop_edge, dis, wl);
  else
    hipLaunchKernelGGL(k_gcn_norm_fwd<4>, dim3((unsigned)igcn_cdiv(n_nodes * 4, 256)), dim3(256), 0, st, n_nodes, ew,
                       src32, tgt_ptr, tgt_perm, loop_edge, dis, wl);
  const int64_t n = n_nodes > n_edges ? n_nodes : n_edges;
  if (n_edges >= 16 * n_nodes && !igcn_opt(IGCN_OPT_NO_TILED_LISTS)) {
    hipLaunchKernelGGL(k_gcn_norm_what, dim3((unsigned)igcn_cdiv(n, 256)), dim3(256), 0, st, n_nodes, n_edges, ew, dis,
                       wl, src32, dst32, what, what_loop);
    hipLaunchKernelGGL(k_stream_by_target_tiled, dim3((unsigned)igcn_cdiv(n_nodes, TL_NODES)), dim3(TL_T), 0, st,
                       n_nodes, what, src32, tgt_ptr, tgt_perm, (EdgeRec*)tstream);
    hipLaunchKernelGGL(k_stream_by_source, dim3((unsigned)igcn_cdiv(n_edges, 256)), dim3(256), 0, st, n_edges, what,
                       dst32, src_perm, (EdgeRec*)sstream);
  } else {
    hipLaunchKernelGGL(k_gcn_norm_coef, dim3((unsigned)igcn_cdiv(n, 256)), dim3(256), 0, st, n_nodes, n_edges, ew, dis,
                       wl, src32, dst32, tgt_perm, src_perm, what, what_loop, (EdgeRec*)tstream, (EdgeRec*)sstream);
  }
  IGCN_CHECK_LAUNCH("gcn_norm_fwd");
  return IGCN_OK;
}

// gcn_norm backward, phase 1 (per node): d(deg)
//   d dis_i = sum_{k out of i, non-loop} dwhat_k ew_k dis[dst_k] + sum_{k into i, non-loop} dwhat_k ew_k dis[src_k]
//             + 2 dwhat_loop_i wl_i dis_i ;   d deg_i = -1/2 dis_i^3 d dis_i
template <int LPN>
__global__ void __launch_bounds__(256)
k_gcn_norm_bwd_deg(int64_t n_nodes, const float* __restrict__ ew, const float* __restrict__ dis,
                   const float* __restrict__ wl, const float* __restrict__ dwhat,
                   const float* __restrict__ dwhat_loop, const int32_t* __restrict__ src32,
                   const int32_t* __restrict__ dst32, const int32_t* __restrict__ tgt_ptr,
                   const int32_t* __restrict__ tgt_perm, const int32_t* __restrict__ src_ptr,
                   const int32_t* __restrict__ src_perm, float* __restrict__ ddeg) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = gid / LPN;
  const int sub = (int)(gid % LPN);
  if (i >= n_nodes) return;
  float dd = 0.f;
  for (int32_t p = src_ptr[i] + sub; p < src_ptr[i + 1]; p += LPN) {
    const int32_t k = src_perm[p];
    const int32_t t = dst32[k];
    if (t != (int32_t)i) dd += dwhat[k] * ew[k] * dis[t];
  }
  for (int32_t p = tgt_ptr[i] + sub; p < tgt_ptr[i + 1]; p += LPN) {
    const int32_t k = tgt_perm[p];
    const int32_t s = src32[k];
    if (s != (int32_t)i) dd += dwhat[k] * ew[k] * dis[s];
  }
  dd = group_sum_all<LPN>(dd);
  if (sub != 0) return;
  const float di = dis[i];
  dd += 2.f * dwhat_loop[i] * wl[i] * di;
  ddeg[i] = -0.5f * di * di * di * dd;
}

// phase 2 (per edge)
__global__ void k_gcn_norm_bwd_edge(int64_t n_edges, const float* __restrict__ dis,
                                    const float* __restrict__ dwhat, const float* __restrict__ dwhat_loop,
                                    const float* __restrict__ ddeg, const int32_t* __restrict__ src32,
                                    const int32_t* __restrict__ dst32, const int32_t* __restrict__ loop_edge,
                                    float* __restrict__ dew) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_edges) return;
  const int32_t s = src32[k], t = dst32[k];
  float g;
  if (s != t) {
    g = dis[s] * dis[t] * dwhat[k] + ddeg[t];
  } else {
    g = (loop_edge[s] == (int32_t)k) ? ddeg[s] + dis[s] * dis[s] * dwhat_loop[s] : 0.f;
  }
  dew[k] = g;
}

extern "C" int igcn_gcn_norm_bwd(int64_t n_nodes, int64_t n_edges, const float* ew, const float* dis,
                                 const float* wl, const float* dwhat, const float* dwhat_loop,
                                 const int32_t* src32, const int32_t* dst32, const int32_t* tgt_ptr,
                                 const int32_t* tgt_perm, const int32_t* src_ptr, const int32_t* src_perm,
                                 const int32_t* loop_edge, float* dew, float* scratch, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_nodes == 0 || n_edges == 0) return IGCN_OK;
  if (n_edges >= 16 * n_nodes && !igcn_opt(IGCN_OPT_NO_TILED_LISTS))
    hipLaunchKernelGGL(k_gcn_norm_bwd_deg_tiled, dim3((unsigned)igcn_cdiv(n_nodes, TL_NODES)), dim3(TL_T), 0, st,
                       n_nodes, ew, dis, wl, dwhat, dwhat_loop, src32, dst32, tgt_ptr, tgt_perm, src_ptr, src_perm,
                       scratch);
  else if (n_edges >= 16 * n_nodes)
    hipLaunchKernelGGL(k_gcn_norm_bwd_deg<64>, dim3((unsigned)igcn_cdiv(n_nodes * 64, 256)), dim3(256), 0, st, n_nodes,
                       ew, dis, wl, dwhat, dwhat_loop, src32, dst32, tgt_ptr, tgt_perm, src_ptr, src_perm, scratch);
  else
    hipLaunchKernelGGL(k_gcn_norm_bwd_deg<4>, dim3((unsigned)igcn_cdiv(n_nodes * 4, 256)), dim3(256), 0, st, n_nodes,
                       ew, dis, wl, dwhat, dwhat_loop, src32, dst32, tgt_ptr, tgt_perm, src_ptr, src_perm, scratch);
  hipLaunchKernelGGL(k_gcn_norm_bwd_edge, dim3((unsigned)igcn_cdiv(n_edges, 256)), dim3(256), 0, st, n_edges, dis,
                     dwhat, dwhat_loop, scratch, src32, dst32, loop_edge, dew);
  IGCN_CHECK_LAUNCH("gcn_norm_bwd");
  return IGCN_OK;
}

// =================================================================================================
// scatter-aggregate forward.  Thread (node, f): FP = F rounded up to a power of two <= 64 lanes share a
// node, so the node's edge list is read once per FP-lane group (same-address loads broadcast) and the
// gathered source row h[src, 0:F] is one contiguous segment per group.
// =================================================================================================
template <int FP>
__global__ void __launch_bounds__(256)
k_gcn_propagate_fwd(int64_t n_nodes, int F, const float* __restrict__ h, int64_t ld_h,
                    const EdgeRec* __restrict__ tstream, const float* __restrict__ what_loop,
                    const float* __restrict__ bias, const int32_t* __restrict__ tgt_ptr,
                    float* __restrict__ out, int64_t ld_out, int relu) {
  constexpr int NPB = 256 / FP;  // nodes per block
  const int f = threadIdx.x % FP;
  const int64_t t = (int64_t)blockIdx.x * NPB + threadIdx.x / FP;
  if (t >= n_nodes || f >= F) return;
  float acc = 0.f;
  const int32_t p1 = tgt_ptr[t + 1];
#pragma unroll 4
  for (int32_t p = tgt_ptr[t]; p < p1; ++p) {      // coalesced 8-byte records; stored loops carry w = 0
    const EdgeRec e = tstream[p];
    acc += e.w * h[(int64_t)e.idx * ld_h + f];
  }
  acc += what_loop[t] * h[t * ld_h + f];
  acc += bias ? bias[f] : 0.f;
  if (relu) acc = fmaxf(acc, 0.f);
  out[t * ld_out + f] = acc;
}

// Low in-degree shape with 16 bytes per lane: thread = (target, feature quad).  F/4 lanes share a target, so a wave
// covers 64/(F/4) targets; per edge step it issues one 8-byte record load and one 16-byte row gather (a quarter of
// the vector-memory instructions of the thread-per-feature shape) and needs no cross-lane reduction.
template <int FQ>
__global__ void __launch_bounds__(256)
k_gcn_propagate_fwd_q(int64_t n_nodes, const float* __restrict__ h, int64_t ld_h,
                      const EdgeRec* __restrict__ tstream, const float* __restrict__ what_loop,
                      const float* __restrict__ bias, const int32_t* __restrict__ tgt_ptr,
                      float* __restrict__ out, int64_t ld_out, int relu) {
  constexpr int NPB = 256 / FQ;
  const int fq = threadIdx.x % FQ;
  const int64_t t = (int64_t)blockIdx.x * NPB + threadIdx.x / FQ;
  if (t >= n_nodes) return;
  const int32_t p0 = tgt_ptr[t], p1 = tgt_ptr[t + 1];
  // the self-loop operands do not depend on the edge walk: issue their loads first so that they are in flight
  // together with the pointer / record / row chain instead of adding a fourth dependent round trip after it
  const float wl = what_loop[t];
  const float4 hs = *reinterpret_cast<const float4*>(h + t * ld_h + fq * 4);
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) b4 = *reinterpret_cast<const float4*>(bias + fq * 4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // four edges per step, the tail predicated instead of peeled: a compiler-unrolled loop runs its remainder (ALL
  // of a 3-edge walk) one load-wait-gather-wait at a time, this one has up to four record loads and then up to
  // four row gathers in flight.  Same summation order as the plain loop.  (Clamping the tail to the last record
  // instead of masking it costs 8%: the launch is close enough to the L2 request rate that a wasted gather shows.)
  for (int32_t p = p0; p < p1; p += 4) {
    EdgeRec e[4];
    float4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      e[i].idx = 0; e[i].w = 0.f;
      if (p + i < p1) e[i] = tstream[p + i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p + i < p1) r[i] = *reinterpret_cast<const float4*>(h + (int64_t)e[i].idx * ld_h + fq * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (p + i < p1) {
        acc.x += e[i].w * r[i].x; acc.y += e[i].w * r[i].y; acc.z += e[i].w * r[i].z; acc.w += e[i].w * r[i].w;
      }
  }
  acc.x += wl * hs.x + b4.x; acc.y += wl * hs.y + b4.y; acc.z += wl * hs.z + b4.z; acc.w += wl * hs.w + b4.w;
  if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
  *reinterpret_cast<float4*>(out + t * ld_out + fq * 4) = acc;
}

// High in-degree variant (dense graphs: hundreds of edges per target): one wave per target node, every lane moves
// 16 bytes per memory instruction.  With FQ = F/4 lanes per feature row the wave covers 64/FQ edge slots: per step
// it issues ONE 8-byte record load (64/FQ consecutive records = up to 128 contiguous bytes) and ONE 16-byte row
// gather (64/FQ rows of F floats).  The vector-memory path retires one wave instruction per ~16 cycles whatever
// the width, so the (target, feature)-per-thread shape (4 B per lane) would be instruction-bound here.
template <int FQ>
__global__ void __launch_bounds__(256)
k_gcn_propagate_fwd_wide(int64_t n_nodes, const float* __restrict__ h, int64_t ld_h,
                         const EdgeRec* __restrict__ tstream, const float* __restrict__ what_loop,
                         const float* __restrict__ bias, const int32_t* __restrict__ tgt_ptr,
                         float* __restrict__ out, int64_t ld_out, int relu) {
  constexpr int SL = 64 / FQ;                     // edge slots per wave
  const int lane = threadIdx.x & 63;
  const int fq = lane % FQ, slot = lane / FQ;
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_nodes) return;                       // wave-uniform
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int32_t p1 = tgt_ptr[t + 1];
#pragma unroll 4
  for (int32_t p = tgt_ptr[t] + slot; p < p1; p += SL) {
    const EdgeRec e = tstream[p];
    const float4 r = *reinterpret_cast<const float4*>(h + (int64_t)e.idx * ld_h + fq * 4);
    acc.x += e.w * r.x; acc.y += e.w * r.y; acc.z += e.w * r.z; acc.w += e.w * r.w;
  }
#pragma unroll
  for (int o = FQ; o < 64; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
    acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
  }
  if (slot == 0) {
    const float wl = what_loop[t];
    const float4 hs = *reinterpret_cast<const float4*>(h + t * ld_h + fq * 4);
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b4 = *reinterpret_cast<const float4*>(bias + fq * 4);
    acc.x += wl * hs.x + b4.x; acc.y += wl * hs.y + b4.y; acc.z += wl * hs.z + b4.z; acc.w += wl * hs.w + b4.w;
    if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    *reinterpret_cast<float4*>(out + t * ld_out + fq * 4) = acc;
  }
}

// High in-degree variant for batches of UNIFORM graphs whose feature rows fit LDS (the dense 512-ROI graphs of the
// stress configuration: 512 x 16 floats = 32 KB per graph): the L2 gather of one 64-byte row per edge (8.4 M rows per
// launch, each row of a graph gathered 512 times) is what bounds the wave-per-target kernel above.  Here a workgroup
// (512 threads) owns LP_TG = 32 targets of ONE graph, stages that graph's rows in LDS once (coalesced 16-byte loads, row stride F + 4
// floats => conflict-free 16-byte row reads), and HBM only streams the 8-byte records: a lane loads TWO records per
// 16-byte load, reads their rows from LDS and keeps a whole F-wide accumulator; the 64 partial rows of a wave are
// summed with a halving butterfly (log2 F exchange steps, then the plain xor steps).  A record whose neighbour is
// not a node of this graph (not a block-diagonal batch) falls back to the global row: correct for any input.
#define LP_TG 32
#define LP_THREADS 512
// sum of `acc` over the 64 lanes: while a lane still holds more than one value, lanes m apart exchange the half the
// partner keeps (HALF shuffles instead of 2*HALF), then plain xor steps.  All indices are compile-time constants.
template <int F, int HALF, int M>
__device__ __forceinline__ void lp_butterfly(float (&acc)[F], int lane) {
  if constexpr (M >= 1) {
    if constexpr (HALF >= 1) {
      const bool up = (lane & M) != 0;
#pragma unroll
      for (int j = 0; j < HALF; ++j) {
        const float send = up ? acc[j] : acc[j + HALF];
        const float recv = __shfl_xor(send, M, 64);
        acc[j] = (up ? acc[j + HALF] : acc[j]) + recv;
      }
      lp_butterfly<F, HALF / 2, M / 2>(acc, lane);
    } else {
      acc[0] += __shfl_xor(acc[0], M, 64);
      lp_butterfly<F, 0, M / 2>(acc, lane);
    }
  }
}

__device__ __forceinline__ int lp_slot(int r, int R) { return (r >> 1) + (r & 1) * ((R + 1) >> 1); }

template <int FQ>
__global__ void __launch_bounds__(LP_THREADS)
k_gcn_propagate_fwd_lds(int64_t n_nodes, int R, const float* __restrict__ h, int64_t ld_h,
                        const EdgeRec* __restrict__ tstream, const float* __restrict__ what_loop,
                        const float* __restrict__ bias, const int32_t* __restrict__ tgt_ptr,
                        float* __restrict__ out, int64_t ld_out, int relu) {
  constexpr int F = 4 * FQ, LD = F + 4;
  extern __shared__ float lp_rows[];                 // [R][LD]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t g0 = (int64_t)blockIdx.y * R;        // first node of the graph
  for (int i = tid; i < R * FQ; i += LP_THREADS) {
    const int r = i / FQ, q = i - r * FQ;
    *reinterpret_cast<float4*>(lp_rows + lp_slot(r, R) * LD + q * 4) =
        *reinterpret_cast<const float4*>(h + (g0 + r) * ld_h + q * 4);
  }
  __syncthreads();
  constexpr int TPW = LP_TG / (LP_THREADS / 64);      // targets per wave
  const int t_lo = blockIdx.x * LP_TG + w * TPW;
  // (requesting the records of target i+1 before target i is summed — a software pipeline over the wave's targets —
  // measured SLOWER: 57 us against 40 us at the stress shape; the extra live registers cost more occupancy than the
  // overlap returns, the 32 resident waves of a CU already cover each other's chains)
  for (int ti = 0; ti < TPW; ++ti) {
    const int tl = t_lo + ti;
    if (tl >= R) break;                              // wave-uniform
    const int64_t t = g0 + tl;
    const int32_t p0 = tgt_ptr[t], p1 = tgt_ptr[t + 1];
    float acc[F];
#pragma unroll
    for (int j = 0; j < F; ++j) acc[j] = 0.f;
    // records two at a time from an even (16-byte aligned) position, FOUR 16-byte loads (512 records per wave) per
    // group; entries outside [p0, p1) carry weight 0 and point at the graph's row 0.
    // A lane holds records (2l, 2l+1): in a dense list those are rows (2l, 2l+1), so the lanes of one LDS row read
    // sit on every second row — the rows are therefore stored EVEN ROWS FIRST (slot(r) = r/2 + (r&1) ceil(R/2)), which
    // makes that access a run of consecutive slots (conflict-free) instead of a 2-way bank conflict.
    for (int32_t qb = (p0 & ~1); qb < p1; qb += 512) {
      int4 v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int32_t q = qb + c * 128 + 2 * lane;
        v[c] = make_int4((int)g0, 0, (int)g0, 0);
        if (q + 1 < p1) {
          v[c] = *reinterpret_cast<const int4*>(tstream + q);
        } else if (q < p1) {
          const EdgeRec e = tstream[q];
          v[c].x = e.idx;
          v[c].y = __float_as_int(e.w);
        }
        if (q < p0) { v[c].x = (int)g0; v[c].y = 0; }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int idx = k == 0 ? v[c].x : v[c].z;
          const float wgt = __int_as_float(k == 0 ? v[c].y : v[c].w);
          const int64_t loc = (int64_t)idx - g0;
          if (loc >= 0 && loc < R) {                 // two code paths: an LDS/global pointer select would turn the
            const float* row = lp_rows + lp_slot((int)loc, R) * LD;     // row reads into FLAT loads
#pragma unroll
            for (int cq = 0; cq < FQ; ++cq) {
              const float4 r4 = *reinterpret_cast<const float4*>(row + cq * 4);
              acc[4 * cq] += wgt * r4.x; acc[4 * cq + 1] += wgt * r4.y;
              acc[4 * cq + 2] += wgt * r4.z; acc[4 * cq + 3] += wgt * r4.w;
            }
          } else {
            const float* row = h + (int64_t)idx * ld_h;
#pragma unroll
            for (int cq = 0; cq < FQ; ++cq) {
              const float4 r4 = *reinterpret_cast<const float4*>(row + cq * 4);
              acc[4 * cq] += wgt * r4.x; acc[4 * cq + 1] += wgt * r4.y;
              acc[4 * cq + 2] += wgt * r4.z; acc[4 * cq + 3] += wgt * r4.w;
            }
          }
        }
      }
    }
    lp_butterfly<F, F / 2, 32>(acc, lane);
    // lane holds the sum of feature f = the log2(F) top bits of its id, read MSB = F/2 ... ; lanes with the low
    // (6 - log2 F) bits zero write
    constexpr int LOGF = FQ == 1 ? 2 : (FQ == 2 ? 3 : (FQ == 4 ? 4 : (FQ == 8 ? 5 : 6)));
    constexpr int LOW = 6 - LOGF;
    if ((lane & ((1 << LOW) - 1)) == 0) {
      int f = 0;
#pragma unroll
      for (int b2 = 0; b2 < LOGF; ++b2)                // step b2 used mask 32 >> b2 and kept half F >> (b2 + 1)
        if (lane & (32 >> b2)) f += F >> (b2 + 1);
      float v = acc[0] + what_loop[t] * lp_rows[lp_slot(tl, R) * LD + f] + (bias ? bias[f] : 0.f);
      if (relu) v = fmaxf(v, 0.f);
      out[t * ld_out + f] = v;
    }
  }
}

static bool propagate_lds_ok(int64_t n_nodes, int64_t n_edges, int F, int nodes_per_graph) {
  if (nodes_per_graph <= 0 || n_nodes % nodes_per_graph != 0) return false;
  if (!(F == 4 || F == 8 || F == 16 || F == 32 || F == 64)) return false;
  if ((size_t)nodes_per_graph * (F + 4) * sizeof(float) > 64 * 1024) return false;
  return n_edges >= 64 * n_nodes;                    // long lists: the staging pays for itself
}

template <int FQ>
static void launch_propagate_lds(int64_t n_nodes, int R, const float* h, int64_t ld_h, const EdgeRec* stream_,
                                 const float* what_loop, const float* bias, const int32_t* ptr, float* out,
                                 int64_t ld_out, int relu, hipStream_t st) {
  const size_t lds = (size_t)R * (4 * FQ + 4) * sizeof(float);
  dim3 grid((unsigned)igcn_cdiv(R, LP_TG), (unsigned)(n_nodes / R));
  hipLaunchKernelGGL((k_gcn_propagate_fwd_lds<FQ>), grid, dim3(LP_THREADS), lds, st, n_nodes, R, h, ld_h, stream_, what_loop,
                     bias, ptr, out, ld_out, relu);
}

static int pow2_ge(int F) {
  int p = 1;
  while (p < F) p <<= 1;
  return p;
}

extern "C" int igcn_gcn_propagate_fwd(int64_t n_nodes, int64_t n_edges, int F, int nodes_per_graph,
                                      const float* h, int64_t ld_h,
                                      const void* tstream, const float* what_loop, const float* bias,
                                      const int32_t* tgt_ptr,
                                      float* out, int64_t ld_out, int relu, void* stream) {
  IGCN_REQUIRE(F >= 1 && F <= 256, "gcn_propagate_fwd: F=%d unsupported (1..256)", F);
  if (n_nodes == 0) return IGCN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int FP = pow2_ge(F);
  const bool al16 = ((uintptr_t)h % 16 == 0) && ((uintptr_t)out % 16 == 0) && ld_h % 4 == 0 && ld_out % 4 == 0 &&
                    (bias == nullptr || (uintptr_t)bias % 16 == 0);
  if (al16 && ((uintptr_t)tstream % 16 == 0) && propagate_lds_ok(n_nodes, n_edges, F, nodes_per_graph) &&
      !igcn_opt(IGCN_OPT_PROPAGATE_NO_LDS)) {
    switch (F / 4) {
      case 1: launch_propagate_lds<1>(n_nodes, nodes_per_graph, h, ld_h, (const EdgeRec*)tstream, what_loop, bias, tgt_ptr, out, ld_out, relu, st); break;
      case 2: launch_propagate_lds<2>(n_nodes, nodes_per_graph, h, ld_h, (const EdgeRec*)tstream, what_loop, bias, tgt_ptr, out, ld_out, relu, st); break;
      case 4: launch_propagate_lds<4>(n_nodes, nodes_per_graph, h, ld_h, (const EdgeRec*)tstream, what_loop, bias, tgt_ptr, out, ld_out, relu, st); break;
      case 8: launch_propagate_lds<8>(n_nodes, nodes_per_graph, h, ld_h, (const EdgeRec*)tstream, what_loop, bias, tgt_ptr, out, ld_out, relu, st); break;
      default: launch_propagate_lds<16>(n_nodes, nodes_per_graph, h, ld_h, (const EdgeRec*)tstream, what_loop, bias, tgt_ptr, out, ld_out, relu, st); break;
    }
    IGCN_CHECK_LAUNCH("gcn_propagate_fwd_lds");
    return IGCN_OK;
  }
  if (n_edges >= 16 * n_nodes && al16 && (F == 4 || F == 8 || F == 16 || F == 32 || F == 64)) {
#define LAUNCH_WIDE(FQV)                                                                                          \
  hipLaunchKernelGGL((k_gcn_propagate_fwd_wide<FQV>), dim3((unsigned)igcn_cdiv(n_nodes, 4)), dim3(256), 0, st,     \
                     n_nodes, h, ld_h, (const EdgeRec*)tstream, what_loop, bias, tgt_ptr, out, ld_out, relu)
    switch (F / 4) {
      case 1: LAUNCH_WIDE(1); break;
      case 2: LAUNCH_WIDE(2); break;
      case 4: LAUNCH_WIDE(4); break;
      case 8: LAUNCH_WIDE(8); break;
      default: LAUNCH_WIDE(16); break;
    }
#undef LAUNCH_WIDE
    IGCN_CHECK_LAUNCH("gcn_propagate_fwd_wide");
    return IGCN_OK;
  }
  if (al16 && (F == 4 || F == 8 || F == 16 || F == 32 || F == 64)) {
#define LAUNCH_Q(FQV)                                                                                            \
  hipLaunchKernelGGL((k_gcn_propagate_fwd_q<FQV>), dim3((unsigned)igcn_cdiv(n_nodes, 256 / FQV)), dim3(256), 0,   \
                     st, n_nodes, h, ld_h, (const EdgeRec*)tstream, what_loop, bias, tgt_ptr, out, ld_out, relu)
    switch (F / 4) {
      case 1: LAUNCH_Q(1); break;
      case 2: LAUNCH_Q(2); break;
      case 4: LAUNCH_Q(4); break;
      case 8: LAUNCH_Q(8); break;
      default: LAUNCH_Q(16); break;
    }
#undef LAUNCH_Q
    IGCN_CHECK_LAUNCH("gcn_propagate_fwd_q");
    return IGCN_OK;
  }
#define LAUNCH_FWD(FPV)                                                                                         \
  hipLaunchKernelGGL((k_gcn_propagate_fwd<FPV>), dim3((unsigned)igcn_cdiv(n_nodes, 256 / FPV)), dim3(256), 0, st, \
                     n_nodes, F, h, ld_h, (const EdgeRec*)tstream, what_loop, bias, tgt_ptr, out, ld_out, relu)
  switch (FP) {
    case 1: LAUNCH_FWD(1); break;
    case 2: LAUNCH_FWD(2); break;
    case 4: LAUNCH_FWD(4); break;
    case 8: LAUNCH_FWD(8); break;
    case 16: LAUNCH_FWD(16); break;
    case 32: LAUNCH_FWD(32); break;
    case 64: LAUNCH_FWD(64); break;
    case 128: LAUNCH_FWD(128); break;
    default: LAUNCH_FWD(256); break;
  }
#undef LAUNCH_FWD
  IGCN_CHECK_LAUNCH("gcn_propagate_fwd");
  return IGCN_OK;
}

// Launch floor of the scatter-aggregate (measurement aid of bench.py's roofline): the grid
// igcn_gcn_propagate_fwd uses for (n_nodes, F) — thread = (target, feature quad), or one wave per target when
// `dense` — with the body removed (mode 0: nothing; mode 1: the 16-byte output store only).
__global__ void __launch_bounds__(256)
k_launch_floor(int64_t n_nodes, int fq_lanes, int dense, int mode, float* __restrict__ out, int64_t ld_out) {
  int64_t t;
  int fq;
  if (dense) {
    const int lane = threadIdx.x & 63;
    fq = lane % fq_lanes;
    t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lane / fq_lanes != 0) return;
  } else {
    fq = threadIdx.x % fq_lanes;
    t = (int64_t)blockIdx.x * (256 / fq_lanes) + threadIdx.x / fq_lanes;
  }
  if (t >= n_nodes || mode == 0) return;
  *reinterpret_cast<float4*>(out + t * ld_out + fq * 4) = make_float4(0.f, 1.f, 2.f, 3.f);
}

extern "C" int igcn_launch_floor(int64_t n_nodes, int F, int dense, int mode, float* out, void* stream) {
  IGCN_REQUIRE(n_nodes > 0 && (F == 4 || F == 8 || F == 16 || F == 32 || F == 64) && ((uintptr_t)out % 16 == 0),
               "launch_floor: F in {4,8,16,32,64}, 16-byte aligned out");
  const int fq = F / 4;
  const int64_t blocks = dense ? igcn_cdiv(n_nodes, 4) : igcn_cdiv(n_nodes, 256 / fq);
  hipLaunchKernelGGL(k_launch_floor, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_nodes, fq, dense,
                     mode, out, (int64_t)F);
  IGCN_CHECK_LAUNCH("launch_floor");
  return IGCN_OK;
}

// =================================================================================================
// scatter-aggregate backward
// =================================================================================================
// dh[s,f] = sum_{k out of s, non-loop} what_k g[dst_k,f] + what_loop_s g[s,f] ; g = dout*(out>0)
// also block partials of dbias[f] = sum_t g[t,f]
template <int FP>
__global__ void __launch_bounds__(256)
k_gcn_propagate_bwd_dh(int64_t n_nodes, int F, const float* __restrict__ dout, int64_t ld_dout,
                       const float* __restrict__ out, int64_t ld_out, int relu,
                       const EdgeRec* __restrict__ sstream, const float* __restrict__ what_loop,
                       const int32_t* __restrict__ src_ptr, float* __restrict__ dh, int64_t ld_dh,
                       float* __restrict__ dbias_partial /*[nblk, FP]*/) {
  constexpr int NPB = 256 / FP;
  __shared__ float red[256];
  const int f = threadIdx.x % FP;
  const int nl = threadIdx.x / FP;
  const int64_t s = (int64_t)blockIdx.x * NPB + nl;
  float gself = 0.f;
  if (s < n_nodes && f < F) {
    float acc = 0.f;
    const int32_t p1 = src_ptr[s + 1];
#pragma unroll 4
    for (int32_t p = src_ptr[s]; p < p1; ++p) {
      const EdgeRec e = sstream[p];
      float g = dout[(int64_t)e.idx * ld_dout + f];
      if (relu && !(out[(int64_t)e.idx * ld_out + f] > 0.f)) g = 0.f;
      acc += e.w * g;
    }
    gself = dout[s * ld_dout + f];
    if (relu && !(out[s * ld_out + f] > 0.f)) gself = 0.f;
    acc += what_loop[s] * gself;
    dh[s * ld_dh + f] = acc;
  }
  // dbias partial: sum over the block's nodes for each f
  red[threadIdx.x] = gself;
  __syncthreads();
  if (threadIdx.x < FP) {
    float t = 0.f;
    for (int j = 0; j < NPB; ++j) t += red[j * FP + threadIdx.x];
    dbias_partial[(int64_t)blockIdx.x * FP + threadIdx.x] = t;
  }
}

// 16-bytes-per-lane shape of the above: thread = (source node, feature quad)
template <int FQ>
__global__ void __launch_bounds__(256)
k_gcn_propagate_bwd_dh_q(int64_t n_nodes, const float* __restrict__ dout, int64_t ld_dout,
                         const float* __restrict__ out, int64_t ld_out, int relu,
                         const EdgeRec* __restrict__ sstream, const float* __restrict__ what_loop,
                         const int32_t* __restrict__ src_ptr, float* __restrict__ dh, int64_t ld_dh,
                         float* __restrict__ dbias_partial /*[nblk, 4*FQ]*/) {
  constexpr int NPB = 256 / FQ;
  __shared__ float4 red[256];
  const int fq = threadIdx.x % FQ, nl = threadIdx.x / FQ;
  const int64_t s = (int64_t)blockIdx.x * NPB + nl;
  float4 gself = make_float4(0.f, 0.f, 0.f, 0.f);
  if (s < n_nodes) {
    const int32_t p0 = src_ptr[s], p1 = src_ptr[s + 1];
    // self-loop operands first: in flight together with the pointer / record / row chain
    const float wl = what_loop[s];
    gself = *reinterpret_cast<const float4*>(dout + s * ld_dout + fq * 4);
    if (relu) {
      const float4 o = *reinterpret_cast<const float4*>(out + s * ld_out + fq * 4);
      gself.x = o.x > 0.f ? gself.x : 0.f; gself.y = o.y > 0.f ? gself.y : 0.f;
      gself.z = o.z > 0.f ? gself.z : 0.f; gself.w = o.w > 0.f ? gself.w : 0.f;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // four edges per step with a predicated tail (see k_gcn_propagate_fwd_q)
    for (int32_t p = p0; p < p1; p += 4) {
      EdgeRec e[4];
      float4 g[4], o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        e[i].idx = 0; e[i].w = 0.f;
        if (p + i < p1) e[i] = sstream[p + i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        o[i] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p + i < p1) {
          g[i] = *reinterpret_cast<const float4*>(dout + (int64_t)e[i].idx * ld_dout + fq * 4);
          if (relu) o[i] = *reinterpret_cast<const float4*>(out + (int64_t)e[i].idx * ld_out + fq * 4);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (p + i < p1) {
          if (relu) {
            g[i].x = o[i].x > 0.f ? g[i].x : 0.f; g[i].y = o[i].y > 0.f ? g[i].y : 0.f;
            g[i].z = o[i].z > 0.f ? g[i].z : 0.f; g[i].w = o[i].w > 0.f ? g[i].w : 0.f;
          }
          acc.x += e[i].w * g[i].x; acc.y += e[i].w * g[i].y; acc.z += e[i].w * g[i].z; acc.w += e[i].w * g[i].w;
        }
    }
    acc.x += wl * gself.x; acc.y += wl * gself.y; acc.z += wl * gself.z; acc.w += wl * gself.w;
    *reinterpret_cast<float4*>(dh + s * ld_dh + fq * 4) = acc;
  }
  red[threadIdx.x] = gself;
  __syncthreads();
  if (threadIdx.x < FQ) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < NPB; ++j) {
      const float4 v = red[j * FQ + threadIdx.x];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    float* dst = dbias_partial + ((int64_t)blockIdx.x * FQ + threadIdx.x) * 4;
    dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
  }
}

// dwhat_k (+)= g[dst_k,:].h[src_k,:] for non-loop edges ; dwhat_loop_i (+)= g[i,:].h[i,:]
__global__ void k_gcn_propagate_bwd_dw(int64_t n_nodes, int64_t n_edges, int F, const float* __restrict__ dout,
                                       int64_t ld_dout, const float* __restrict__ out, int64_t ld_out, int relu,
                                       const float* __restrict__ h, int64_t ld_h,
                                       const int32_t* __restrict__ src32, const int32_t* __restrict__ dst32,
                                       float* __restrict__ dwhat, float* __restrict__ dwhat_loop) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_edges + n_nodes) return;
  int64_t s, t;
  float* dstp;
  if (i < n_edges) {
    s = src32[i];
    t = dst32[i];
    dstp = dwhat + i;
    if (s == t) { *dstp = 0.f; return; }
  } else {
    s = t = i - n_edges;
    dstp = dwhat_loop + s;
  }
  float acc = 0.f;
  for (int f = 0; f < F; ++f) {
    float g = dout[t * ld_dout + f];
    if (relu && !(out[t * ld_out + f] > 0.f)) g = 0.f;
    acc += g * h[s * ld_h + f];
  }
  *dstp = acc;
}

// 16-byte rows: the scalar form above issues 3 F four-byte loads per edge, each lane on its own cache line
__global__ void __launch_bounds__(256)
k_gcn_propagate_bwd_dw_q(int64_t n_nodes, int64_t n_edges, int FQ, const float* __restrict__ dout, int64_t ld_dout,
                         const float* __restrict__ out, int64_t ld_out, int relu, const float* __restrict__ h,
                         int64_t ld_h, const int32_t* __restrict__ src32, const int32_t* __restrict__ dst32,
                         float* __restrict__ dwhat, float* __restrict__ dwhat_loop) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_edges + n_nodes) return;
  int64_t s, t;
  float* dstp;
  if (i < n_edges) {
    s = src32[i];
    t = dst32[i];
    dstp = dwhat + i;
    if (s == t) { *dstp = 0.f; return; }
  } else {
    s = t = i - n_edges;
    dstp = dwhat_loop + s;
  }
  const float4* g4 = reinterpret_cast<const float4*>(dout + t * ld_dout);
  const float4* o4 = reinterpret_cast<const float4*>(out + t * ld_out);
  const float4* h4 = reinterpret_cast<const float4*>(h + s * ld_h);
  float acc = 0.f;
  for (int f = 0; f < FQ; ++f) {
    float4 g = g4[f];
    const float4 hv = h4[f];
    if (relu) {
      const float4 o = o4[f];
      g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
    }
    acc += g.x * hv.x;                      // same summation order as the scalar kernel
    acc += g.y * hv.y;
    acc += g.z * hv.z;
    acc += g.w * hv.w;
  }
  *dstp = acc;
}

// dense graphs (hundreds of out-edges per source): one wave per source node, as k_gcn_propagate_fwd_wide; a block
// still covers 256/FQ source nodes so that the dbias partial layout equals the quad kernel's
template <int FQ>
__global__ void __launch_bounds__(256)
k_gcn_propagate_bwd_dh_wide(int64_t n_nodes, const float* __restrict__ dout, int64_t ld_dout,
                            const float* __restrict__ out, int64_t ld_out, int relu,
                            const EdgeRec* __restrict__ sstream, const float* __restrict__ what_loop,
                            const int32_t* __restrict__ src_ptr, float* __restrict__ dh, int64_t ld_dh,
                            float* __restrict__ dbias_partial /*[nblk, 4*FQ]*/) {
  constexpr int NPB = 256 / FQ, SL = 64 / FQ;
  __shared__ float4 red[4 * FQ];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fq = lane % FQ, slot = lane / FQ;
  float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < NPB / 4; ++it) {
    const int64_t s = (int64_t)blockIdx.x * NPB + it * 4 + w;
    if (s >= n_nodes) break;                        // wave-uniform
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int32_t p1 = src_ptr[s + 1];
#pragma unroll 4
    for (int32_t p = src_ptr[s] + slot; p < p1; p += SL) {
      const EdgeRec e = sstream[p];
      float4 g = *reinterpret_cast<const float4*>(dout + (int64_t)e.idx * ld_dout + fq * 4);
      if (relu) {
        const float4 o = *reinterpret_cast<const float4*>(out + (int64_t)e.idx * ld_out + fq * 4);
        g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
      }
      acc.x += e.w * g.x; acc.y += e.w * g.y; acc.z += e.w * g.z; acc.w += e.w * g.w;
    }
#pragma unroll
    for (int o2 = FQ; o2 < 64; o2 <<= 1) {
      acc.x += __shfl_xor(acc.x, o2, 64); acc.y += __shfl_xor(acc.y, o2, 64);
      acc.z += __shfl_xor(acc.z, o2, 64); acc.w += __shfl_xor(acc.w, o2, 64);
    }
    if (slot == 0) {
      float4 gself = *reinterpret_cast<const float4*>(dout + s * ld_dout + fq * 4);
      if (relu) {
        const float4 o = *reinterpret_cast<const float4*>(out + s * ld_out + fq * 4);
        gself.x = o.x > 0.f ? gself.x : 0.f; gself.y = o.y > 0.f ? gself.y : 0.f;
        gself.z = o.z > 0.f ? gself.z : 0.f; gself.w = o.w > 0.f ? gself.w : 0.f;
      }
      const float wl = what_loop[s];
      acc.x += wl * gself.x; acc.y += wl * gself.y; acc.z += wl * gself.z; acc.w += wl * gself.w;
      *reinterpret_cast<float4*>(dh + s * ld_dh + fq * 4) = acc;
      gsum.x += gself.x; gsum.y += gself.y; gsum.z += gself.z; gsum.w += gself.w;
    }
  }
  if (slot == 0) red[w * FQ + fq] = gsum;
  __syncthreads();
  if (threadIdx.x < FQ) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < 4; ++j) {
      const float4 v = red[j * FQ + threadIdx.x];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    float* dst = dbias_partial + ((int64_t)blockIdx.x * FQ + threadIdx.x) * 4;
    dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
  }
}

// Per-edge coefficient gradients for the same dense uniform batches: a workgroup takes a run of DW_CHUNK consecutive
// edges; when they all belong to ONE graph (block-diagonal batch in stored order) it stages that graph's g and h rows
// in LDS and every edge costs one coalesced (src, dst) read and two LDS row reads instead of two L2 row gathers.
// Edges outside the staged graph (a chunk across a graph boundary, or a batch that is not block diagonal) read the
// rows from global memory.
#define DW_CHUNK 32768
#define DW_THREADS 1024
template <int FQ>
__global__ void __launch_bounds__(DW_THREADS)
k_gcn_propagate_bwd_dw_lds(int64_t n_nodes, int64_t n_edges, int R, const float* __restrict__ g, int64_t ld_g,
                           const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ src32,
                           const int32_t* __restrict__ dst32, float* __restrict__ dwhat,
                           float* __restrict__ dwhat_loop) {
  constexpr int F = 4 * FQ, LD = F + 4;
  extern __shared__ float dw_rows[];                 // g rows [R][LD] then h rows [R][LD]
  float* gs = dw_rows;
  float* hs = dw_rows + (size_t)R * LD;
  const int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * DW_CHUNK;
  const int64_t i1 = i0 + DW_CHUNK < n_edges ? i0 + DW_CHUNK : n_edges;
  const int64_t graph = (int64_t)src32[i0] / R;      // the graph most of the chunk belongs to
  const int64_t g0 = graph * R;
  for (int i = tid; i < R * FQ; i += DW_THREADS) {
    const int r = i / FQ, q = i - r * FQ;
    *reinterpret_cast<float4*>(gs + r * LD + q * 4) = *reinterpret_cast<const float4*>(g + (g0 + r) * ld_g + q * 4);
    *reinterpret_cast<float4*>(hs + r * LD + q * 4) = *reinterpret_cast<const float4*>(h + (g0 + r) * ld_h + q * 4);
  }
  __syncthreads();
  for (int64_t i = i0 + tid; i < i1; i += DW_THREADS) {
    const int64_t sg = src32[i], tg = dst32[i];
    if (sg == tg) { dwhat[i] = 0.f; continue; }
    const int64_t sl = sg - g0, tl = tg - g0;
    const bool in = sl >= 0 && sl < R && tl >= 0 && tl < R;
    float acc = 0.f;
    if (in) {                                        // separate LDS / global code paths (no pointer select: FLAT loads)
      const float* hr = hs + (int)sl * LD;
      const float* gr = gs + (int)tl * LD;
#pragma unroll
      for (int c = 0; c < FQ; ++c) {
        const float4 gv = *reinterpret_cast<const float4*>(gr + c * 4);
        const float4 hv = *reinterpret_cast<const float4*>(hr + c * 4);
        acc += gv.x * hv.x;                          // same summation order as the global-memory kernels
        acc += gv.y * hv.y;
        acc += gv.z * hv.z;
        acc += gv.w * hv.w;
      }
    } else {
      const float* hr = h + sg * ld_h;
      const float* gr = g + tg * ld_g;
#pragma unroll
      for (int c = 0; c < FQ; ++c) {
        const float4 gv = *reinterpret_cast<const float4*>(gr + c * 4);
        const float4 hv = *reinterpret_cast<const float4*>(hr + c * 4);
        acc += gv.x * hv.x;
        acc += gv.y * hv.y;
        acc += gv.z * hv.z;
        acc += gv.w * hv.w;
      }
    }
    dwhat[i] = acc;
  }
  // self-loop coefficients of the staged graph's nodes: one chunk per graph does it (the one holding its first edge
  // ... which this workgroup cannot know cheaply), so node t is taken by the workgroup whose chunk index == t's
  // position modulo the grid: a plain grid-stride over nodes, rows from global memory
  for (int64_t t = (int64_t)blockIdx.x * DW_THREADS + tid; t < n_nodes; t += (int64_t)gridDim.x * DW_THREADS) {
    const float* gr = g + t * ld_g;
    const float* hr = h + t * ld_h;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < FQ; ++c) {
      const float4 gv = *reinterpret_cast<const float4*>(gr + c * 4);
      const float4 hv = *reinterpret_cast<const float4*>(hr + c * 4);
      acc += gv.x * hv.x;
      acc += gv.y * hv.y;
      acc += gv.z * hv.z;
      acc += gv.w * hv.w;
    }
    dwhat_loop[t] = acc;
  }
}

// block partials of the column sums of g [N, F] (bias gradient), layout [nblk, F] like the quad dh kernels
template <int FQ>
__global__ void __launch_bounds__(256)
k_colsum_partial_q(int64_t n_nodes, const float* __restrict__ g, int64_t ld_g, float* __restrict__ partial) {
  constexpr int NPB = 256 / FQ;
  __shared__ float4 red[256];
  const int fq = threadIdx.x % FQ, nl = threadIdx.x / FQ;
  const int64_t s = (int64_t)blockIdx.x * NPB + nl;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (s < n_nodes) v = *reinterpret_cast<const float4*>(g + s * ld_g + fq * 4);
  red[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x < FQ) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < NPB; ++j) {
      const float4 u = red[j * FQ + threadIdx.x];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    float* dst = partial + ((int64_t)blockIdx.x * FQ + threadIdx.x) * 4;
    dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
  }
}

// g = dout * [out > 0], once per call: the per-edge kernels then gather ONE row per edge end instead of two
__global__ void __launch_bounds__(256)
k_relu_mask_rows(int64_t n_nodes, int F, const float* __restrict__ dout, int64_t ld_dout,
                 const float* __restrict__ out, int64_t ld_out, float* __restrict__ g /*[N,F] dense*/) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_nodes * F) return;
  const int64_t r = i / F;
  const int c = (int)(i - r * F);
  g[i] = out[r * ld_out + c] > 0.f ? dout[r * ld_dout + c] : 0.f;
}

static size_t propagate_bwd_partial_floats(int64_t n_nodes, int F) {
  const int FP = pow2_ge(F > 0 ? F : 1);
  const int npb = FP >= 256 ? 1 : 256 / FP;
  return ((size_t)(igcn_cdiv(n_nodes, npb) * FP + 64) + 3) & ~(size_t)3;
}

extern "C" size_t igcn_gcn_propagate_bwd_scratch_floats(int64_t n_nodes, int F) {
  return propagate_bwd_partial_floats(n_nodes, F) + (size_t)n_nodes * (F > 0 ? F : 1);   // partials + masked dout
}

extern "C" int igcn_gcn_propagate_bwd(int64_t n_nodes, int64_t n_edges, int F, int nodes_per_graph,
                                      const float* dout, int64_t ld_dout,
                                      const float* out, int64_t ld_out, int relu, const float* h, int64_t ld_h,
                                      const void* sstream, const float* what_loop, const int32_t* src32,
                                      const int32_t* dst32, const int32_t* src_ptr,
                                      float* dh, int64_t ld_dh, float* dbias, int need_dw,
                                      float* dwhat, float* dwhat_loop, float* scratch, void* stream) {
  IGCN_REQUIRE(F >= 1 && F <= 256, "gcn_propagate_bwd: F=%d unsupported (1..256)", F);
  if (n_nodes == 0) return IGCN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (relu) {                                      // fold the ReLU mask into a dense copy of dout once
    float* gm = scratch + propagate_bwd_partial_floats(n_nodes, F);
    hipLaunchKernelGGL(k_relu_mask_rows, dim3((unsigned)igcn_cdiv(n_nodes * F, 256)), dim3(256), 0, st, n_nodes, F,
                       dout, ld_dout, out, ld_out, gm);
    dout = gm;
    ld_dout = F;
    relu = 0;
  }
  const int FP = pow2_ge(F);
  int64_t nblk = igcn_cdiv(n_nodes, 256 / FP);
  const bool al16 = ((uintptr_t)dout % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)dh % 16 == 0) &&
                    ld_dout % 4 == 0 && ld_out % 4 == 0 && ld_dh % 4 == 0;
  const bool quad = al16 && (F == 4 || F == 8 || F == 16 || F == 32 || F == 64);
  if (quad && relu == 0 && ((uintptr_t)sstream % 16 == 0) && (uintptr_t)h % 16 == 0 && ld_h % 4 == 0 &&
      propagate_lds_ok(n_nodes, n_edges, F, nodes_per_graph) && !igcn_opt(IGCN_OPT_PROPAGATE_NO_LDS)) {
    // dense uniform batch (stress shape): the graph's rows staged in LDS — dh is the forward kernel on the
    // by-source stream, the bias gradient a column-sum pass, the coefficient gradients an LDS-staged edge walk
    const int R = nodes_per_graph;
    nblk = igcn_cdiv(n_nodes, 256 / (F / 4));
#define LAUNCH_LDS_BWD(FQV)                                                                                       \
  do {                                                                                                            \
    launch_propagate_lds<FQV>(n_nodes, R, dout, ld_dout, (const EdgeRec*)sstream, what_loop, nullptr, src_ptr, dh, \
                              ld_dh, 0, st);                                                                       \
    if (dbias)                                                                                                    \
      hipLaunchKernelGGL((k_colsum_partial_q<FQV>), dim3((unsigned)nblk), dim3(256), 0, st, n_nodes, dout, ld_dout, \
                         scratch);                                                                                 \
    if (need_dw && 2 * (size_t)R * (4 * FQV + 4) * sizeof(float) <= 150 * 1024) {                                 \
      IGCN_ALLOW_BIG_LDS((k_gcn_propagate_bwd_dw_lds<FQV>));                                                      \
      hipLaunchKernelGGL((k_gcn_propagate_bwd_dw_lds<FQV>), dim3((unsigned)igcn_cdiv(n_edges, DW_CHUNK)),          \
                         dim3(DW_THREADS), 2 * (size_t)R * (4 * FQV + 4) * sizeof(float), st, n_nodes, n_edges, R, \
                         dout,                                                                                     \
                         ld_dout, h, ld_h, src32, dst32, dwhat, dwhat_loop);                                       \
      need_dw = 0;                                                                                                \
    }                                                                                                             \
  } while (0)
    switch (F / 4) {
      case 1: LAUNCH_LDS_BWD(1); break;
      case 2: LAUNCH_LDS_BWD(2); break;
      case 4: LAUNCH_LDS_BWD(4); break;
      case 8: LAUNCH_LDS_BWD(8); break;
      default: LAUNCH_LDS_BWD(16); break;
    }
#undef LAUNCH_LDS_BWD
  } else if (quad) {
    nblk = igcn_cdiv(n_nodes, 256 / (F / 4));
    const bool wide = n_edges >= 16 * n_nodes;
#define LAUNCH_BQ(FQV)                                                                                            \
  if (wide)                                                                                                       \
    hipLaunchKernelGGL((k_gcn_propagate_bwd_dh_wide<FQV>), dim3((unsigned)nblk), dim3(256), 0, st, n_nodes, dout,  \
                       ld_dout, out, ld_out, relu, (const EdgeRec*)sstream, what_loop, src_ptr, dh, ld_dh,         \
                       scratch);                                                                                   \
  else                                                                                                            \
    hipLaunchKernelGGL((k_gcn_propagate_bwd_dh_q<FQV>), dim3((unsigned)nblk), dim3(256), 0, st, n_nodes, dout,     \
                       ld_dout, out, ld_out, relu, (const EdgeRec*)sstream, what_loop, src_ptr, dh, ld_dh, scratch)
    switch (F / 4) {
      case 1: LAUNCH_BQ(1); break;
      case 2: LAUNCH_BQ(2); break;
      case 4: LAUNCH_BQ(4); break;
      case 8: LAUNCH_BQ(8); break;
      default: LAUNCH_BQ(16); break;
    }
#undef LAUNCH_BQ
  } else {
#define LAUNCH_BWD(FPV)                                                                                          \
  hipLaunchKernelGGL((k_gcn_propagate_bwd_dh<FPV>), dim3((unsigned)nblk), dim3(256), 0, st, n_nodes, F, dout,    \
                     ld_dout, out, ld_out, relu, (const EdgeRec*)sstream, what_loop, src_ptr, dh, ld_dh, scratch)
  switch (FP) {
    case 1: LAUNCH_BWD(1); break;
    case 2: LAUNCH_BWD(2); break;
    case 4: LAUNCH_BWD(4); break;
    case 8: LAUNCH_BWD(8); break;
    case 16: LAUNCH_BWD(16); break;
    case 32: LAUNCH_BWD(32); break;
    case 64: LAUNCH_BWD(64); break;
    case 128: LAUNCH_BWD(128); break;
    default: LAUNCH_BWD(256); break;
  }
#undef LAUNCH_BWD
  }
  IGCN_CHECK_LAUNCH("gcn_propagate_bwd_dh");
  if (dbias) {
    int rc = igcn_launch_reduce_rows_final(scratch, nblk, quad ? F : FP, F, dbias, st);
    if (rc) return rc;
  }
  if (need_dw) {
    if (quad && (uintptr_t)h % 16 == 0 && ld_h % 4 == 0)
      hipLaunchKernelGGL(k_gcn_propagate_bwd_dw_q, dim3((unsigned)igcn_cdiv(n_edges + n_nodes, 256)), dim3(256), 0,
                         st, n_nodes, n_edges, F / 4, dout, ld_dout, out, ld_out, relu, h, ld_h, src32, dst32, dwhat,
                         dwhat_loop);
    else
      hipLaunchKernelGGL(k_gcn_propagate_bwd_dw, dim3((unsigned)igcn_cdiv(n_edges + n_nodes, 256)), dim3(256), 0, st,
                         n_nodes, n_edges, F, dout, ld_dout, out, ld_out, relu, h, ld_h, src32, dst32, dwhat,
                         dwhat_loop);
    IGCN_CHECK_LAUNCH("gcn_propagate_bwd_dw");
  }
  return IGCN_OK;
}
