// SGCN message passing over DENSE brain graphs (BASELINE configs[4]: 512-ROI dense adjacency) — cal_probability,
// gcn_norm and every GCNConv of kernel/sgcn_img_snp.py:133-151,218-224 (+ the edge part of loss_probability :153-181)
// for batches whose graphs are COMPLETE: every graph stores all R x R (source, target) pairs in row-major order, which
// is what any dense adjacency matrix turned into COO looks like (numpy.nonzero / dense_to_sparse).
//
// For such a batch edge_index carries no information: edge k of graph g joins source k / R to target k % R, and
// edge_attr IS the dense matrix ew[g][src][dst].  igcn_dense_blocks_check verifies exactly that on the device (one pass
// over edge_index, every step, status word like the per-graph plan builders); everything else works on the dense
// blocks:
//   * no sorted plan, no (neighbour, coefficient) record streams, no per-edge intermediates in HBM at all.  The edge
//     mask e[s,d] = sigmoid(u[s] + v[d]) (u, v: two numbers per node), the masked weight ew * e and the normalised
//     coefficient dis[s] * w * dis[d] are RECOMPUTED from the 4-byte weight wherever they are needed, so every edge
//     pass reads ew once — 4 bytes per edge — and serves the plain AND the masked pass of a train step together;
//   * the aggregation out[d] = sum_s w[s,d] h'[s] (h' = dis * (X W^T)) is a 512 x 512 by 512 x 16 product per graph
//     whose left operand is that very matrix: it runs on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32) straight
//     from the registers the 16-byte loads land in — lanes (q = lane & 15, sub = lane >> 4) load ew[s0 + sub][d0 + 4 q ..
//     d0 + 4 q + 3], which are the A operands (row = target, k = source) of four 16-target tiles at once.  HBM-bound:
//     4 bytes per edge per launch;
//   * backward: dH' = sum_d w[s,d] g'[d] the same way on the transposed access; the degree gradient needs no edge pass
//     (ddeg[i] = -1/2 dis[i]^2 sum_l (g'_l[i] . AGG_l[i] + h'_l[i] . dH'_l[i]), node-level dot products); ONE more edge
//     pass yields the mask gradient: q[s,d] = sum_l g'_l[d] . h'_l[s] on the matrix cores (K = L * F), then
//     dz = ((q + ddeg[d]) ew + reg'(e)) e (1 - e) summed over rows (du) and columns (dv).
// Six edge passes of 4 B per edge per train step instead of ~30 passes of 4-20 B (sort, replicate, mask, norm, two
// record streams, four aggregations, three backward walks): DESIGN.md §4.
//
// Needs: uniform complete graphs, R % 64 == 0, R <= 1024, F == 16, L <= 4, H0 <= 8 (igcn_dense_sgcn_supported);
// anything else takes the general kernels (csrc/sgcn.hip).  Summation order: a target's incoming messages are added
// in matrix-core / tree order, not the reference's sequential scatter order (fp32 rounding ~1e-7, like the LDS-staged
// aggregation it replaces at this shape).
#include "common.h"
#include "dropout.h"

bool igcn_rider_dropout_take(hipStream_t st, DropJob& job);        // plan.hip

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DS_F 16
#define DS_MAXL 4
#define DS_MAXH0 8

struct DsParams {
  const float* W[DS_MAXL];
  const float* b[DS_MAXL];
};

struct DsReg {                                   // loss_probability's constants (sgcn_hyperparameters.py:18-21)
  float l1_x, ent_x, l1_e, ent_e, eps;
};

// The mask is evaluated once per edge in EVERY edge pass (six times per train step) instead of being stored: hardware
// exp / reciprocal / log (v_exp_f32, v_rcp_f32, v_log_f32: ~1e-6 relative) keep that at a handful of instructions.
// Every pass uses the same function, so forward and backward see the same mask.
// [ds_rcp: v_rcp_f32, 1 ulp.  __frcp_rn — what these functions used first — is the correctly rounded reciprocal: ten
// VALU instructions (v_div_scale x 2, v_rcp, four v_fma, v_div_fmas, v_div_fixup); four masks per aggregation step made
// k_ds_agg issue-bound at 3.1 us of VALU + MFMA per wave, two waves per SIMD: 9.1 us whether ew came from HBM or L2.]
__device__ __forceinline__ float ds_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float ds_sigmoid(float z) { return ds_rcp(1.f + __expf(-z)); }
// edge mask from the per-node factors a[s] = exp(-u[s]), b[d] = exp(-v[d]): sigmoid(u + v) = 1 / (1 + a b) — one
// multiply-add and one reciprocal per edge, no exponential (a b = inf -> 0, a b = 0 -> 1: the right limits)
__device__ __forceinline__ float ds_mask(float a, float b) { return ds_rcp(fmaf(a, b, 1.f)); }
__device__ __forceinline__ float ds_reg_term(float p, float l1, float ent, float eps) {
  return l1 * p - ent * (p * __logf(p + eps) + (1.f - p) * __logf((1.f - p) + eps));          // p in (0, 1)
}
__device__ __forceinline__ float ds_reg_grad(float p, float l1, float ent, float eps) {
  const float a = p + eps, b = (1.f - p) + eps;
  return l1 - ent * (__logf(a) + p * ds_rcp(a) - __logf(b) - (1.f - p) * ds_rcp(b));
}

// loss_probability's term of an EDGE mask p = sigmoid(z), from p and its logit z = u[s] + v[d]:
//   l1 p - ent (p log p + (1 - p) log(1 - p)) = l1 p + ent (L + (1 - p) z),  L = -log p = log(1 + e^-z)
// — one logarithm per edge instead of two.  The reference's eps (1e-6, kernel/sgcn_img_snp.py:153) inside its logarithms
// adds p log(1 + eps / p) + (1 - p) log(1 + eps / (1 - p)) = 2 eps to first order wherever p and 1 - p are >> eps: kept
// as the constant -2 ent eps; what is left is O(eps^2 / p), and at most eps on a saturated edge (tests: 1e-5 on the
// loss, 1e-3 on its gradients against the fp64 oracle WITH eps).
// Saturation: p = 0 (the factors' product overflowed) gives L clamped at 87 and z clamped to >= -L: p L = 0, as the
// reference's 0 * log(eps); p = 1 gives L = 0 and (1 - p) = 0.
__device__ __forceinline__ float ds_reg_term_logit(float p, float z, float l1, float ent, float eps) {
  const float L = -__logf(fmaxf(p, 1e-38f));
  return fmaf(l1, p, ent * (fmaf(1.f - p, fmaxf(z, -L), L) - 2.f * eps));
}
// u (or v) back from its stored factor exp(-u): the logit of an edge is u[s] + v[d].  Clamped so that a saturated
// factor (0 or inf) gives a finite logit: the mask's e (1 - e) is then 0 and kills the term.
__device__ __forceinline__ float ds_logit_part(float a) { return -__logf(fminf(fmaxf(a, 1e-37f), 1e37f)); }

template <int NC, bool M0>
__device__ __forceinline__ constexpr bool ds_masked(int c) { return NC == 2 ? c == 1 : M0; }

// The edge passes walk their rows in STEPS of one 16-byte load per lane.  Two forms of every edge kernel:
//  * pipelined (template flag PIPE; R = 256 or 512): the walk is software-pipelined over two register buffers of
//    chunk-many steps — the loads of chunk k + 1 are in flight while chunk k feeds the arithmetic — with NO branch or
//    predicate between a load and its use: with one, the compiler waits for every outstanding load (s_waitcnt vmcnt(0))
//    before the first multiply, i.e. "load everything, wait, compute" (measured on the aggregation: 19.1 against 14.6 us).
//    Hence a single-exit loop plus a peeled tail, and walks that are whole numbers of chunk pairs.  The transposed
//    aggregation additionally stages its node operands in LDS (it would otherwise issue eight 4-byte operand loads per
//    step); for the forward aggregation that did not pay (measurements at the call site).
//  * generic (any supported R): one step at a time out of global memory.
// Workgroup -> (graph, 64-node block) of the edge kernels, XCD-aware: the hardware deals consecutive workgroup ids
// round-robin over the 8 XCDs (each with its own L2).  With the plain (block, graph) grid the 8 blocks of a graph sit
// on 8 different XCDs: every XCD fetches every graph's node operands, and its ew reads are 256-byte pieces 2 KB apart.
// Here XCD c takes the work items [c T/8, (c+1) T/8) in order, so the blocks of a graph run on ONE XCD, back to back:
// the graph's h' / g' rows cross the fabric once, and the XCD's L2 sees whole 2 KB rows of ew.
__device__ __forceinline__ void ds_block_of(int nblk, int total, int id, int& g, int& xb) {
  int j = id;
  if ((total & 7) == 0) j = (id & 7) * (total >> 3) + (id >> 3);
  g = j / nblk;
  xb = j - g * nblk;
}
__device__ __forceinline__ void ds_block(int nblk, int& g, int& xb) {
  const int total = gridDim.x, id = blockIdx.x;
  int j = id;
  if ((total & 7) == 0) j = (id & 7) * (total >> 3) + (id >> 3);
  g = j / nblk;
  xb = j - g * nblk;
}

#ifdef DS_PROBE_ON
// phase stamps of the first workgroups of k_ds_agg (tools/dense_probe.py, IGCN_HIPCC_EXTRA=-DDS_PROBE_ON)
__device__ long long ds_probe_buf[8 * 8 * 16];                    // [workgroup][wave][stamp: wall clock (100 MHz) | 8 + stamp: shader clock]
#define DS_PROBE(i)                                                                                     \
  do {                                                                                                  \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 64 && (blockIdx.x & 7) == 0) {                          \
      ds_probe_buf[((blockIdx.x >> 3) * 8 + (threadIdx.x >> 6)) * 16 + (i)] = wall_clock64();           \
      ds_probe_buf[((blockIdx.x >> 3) * 8 + (threadIdx.x >> 6)) * 16 + 8 + (i)] = clock64();            \
    }                                                                                                   \
  } while (0)
extern "C" int igcn_debug_ds_probe(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ds_probe_buf), sizeof(long long) * 8 * 8 * 16);
}
#else
#define DS_PROBE(i)
#endif
#define DS_PCH 4
template <int V>
struct DsInt {
  static constexpr int value = V;
};
template <bool PIPE, int STEP, typename BufP, typename Buf1, typename LoadP, typename CompP, typename Load1, typename Comp1,
          int PCH = DS_PCH>
__device__ __forceinline__ void ds_walk(int begin, int end, LoadP loadp, CompP compp, Load1 load1, Comp1 comp1) {
  if constexpr (PIPE) {
    constexpr int CW = PCH * STEP;                          // rows (or columns) per chunk
    BufP b0, b1;
    loadp(b0, begin);
    int p = begin;
    for (; p + 2 * CW < end; p += 2 * CW) {
      loadp(b1, p + CW);
      compp(b0, p);
      loadp(b0, p + 2 * CW);
      compp(b1, p + CW);
    }
    loadp(b1, p + CW);
    compp(b0, p);
    compp(b1, p + CW);
  } else {
#pragma unroll 2
    for (int p = begin; p < end; p += STEP) {
      Buf1 b;
      load1(b, p);
      comp1(b, p);
    }
  }
}

// ---- forward workspace (kept for the backward): u, v [GR] | dis [NC][GR] | hp [L][NC][GR][F] | agg [L][NC][GR][F]
struct DsWs {
  int64_t u, v, dis, hp, agg, total;
};
__host__ __device__ inline DsWs ds_ws(int64_t GR, int L, int NC) {
  DsWs o;
  o.u = 0;
  o.v = GR;
  o.dis = 2 * GR;
  o.hp = o.dis + (int64_t)NC * GR;
  o.agg = o.hp + (int64_t)L * NC * GR * DS_F;
  o.total = o.agg + (int64_t)L * NC * GR * DS_F;
  return o;
}

extern "C" int igcn_dense_sgcn_supported(int R, int H0, int F, int L) {
  return R >= 64 && R <= 1024 && R % 64 == 0 && F == DS_F && L >= 1 && L <= DS_MAXL && H0 >= 1 && H0 <= DS_MAXH0;
}

extern "C" size_t igcn_dense_sgcn_ws_floats(int64_t n_graphs, int R, int L, int copies) {
  return (size_t)ds_ws(n_graphs * R, L, copies).total;
}

// number of loss_probability partials the forward leaves behind: one per (graph, 64-target block) + one for the
// node / SNP parts
extern "C" int igcn_dense_sgcn_reg_blocks(int64_t n_graphs, int R) { return (int)(n_graphs * (R / 64) + 1); }

// =================================================================================================
// structure check: edge k of graph g must be (g R + k / R, g R + k % R)
// =================================================================================================
// A pure read stream of 16 bytes per edge: one matrix ROW (R edges of one source) per trip of a workgroup, four rows'
// loads in flight per thread, the row's (graph, source) from ONE 32-bit division per row (round 3 divided 64-bit
// integers twice per edge pair: the check ran at 0.62 of the HBM rate on integer arithmetic).
// status[0] |= 4 (sticky: GraphPlan.check() reports it); status[1] = the LATEST check's verdict (cleared in front of
// every check, set to 1 on a mismatch; igcn_dense_sgcn_fwd turns a set verdict into NaN features — hence NaN outputs and
// loss: a batch that is not what the dense-block kernels assume cannot train silently).
// The check runs stand-alone (igcn_dense_blocks_check) or RIDES in igcn_dense_sgcn_fwd's first edge pass: extra
// workgroups of k_ds_deg's launch stream the index pairs beside its 4-byte weights (27.6 + 12.5 us as two launches).
#define DS_CHK_ROWS 4
// `halves` = blockDim.x / 256 groups of 256 threads, each on DS_CHK_ROWS rows of its own per trip; workgroup `blk` of `nblk`
__device__ __forceinline__ void ds_check_rows(int64_t blk, int64_t nblk, int64_t n_rows, int R,
                                              const int64_t* __restrict__ ei, int64_t n_edges,
                                              int32_t* __restrict__ status) {
  bool bad = false;
  const int halves = (int)blockDim.x >> 8, half = (int)threadIdx.x >> 8, t = (int)threadIdx.x & 255;
  const int64_t per = (int64_t)halves * DS_CHK_ROWS;
  for (int64_t r0 = blk * per + (int64_t)half * DS_CHK_ROWS; r0 < n_rows; r0 += nblk * per) {
    for (int d = t * 2; d < R; d += 512) {
      longlong2 s2[DS_CHK_ROWS], d2[DS_CHK_ROWS];
#pragma unroll
      for (int i = 0; i < DS_CHK_ROWS; ++i) {
        const int64_t row = r0 + i < n_rows ? r0 + i : r0, k = row * R + d;       // (n_rows = G R; R even: 16-B aligned)
        s2[i] = *reinterpret_cast<const longlong2*>(ei + k);
        d2[i] = *reinterpret_cast<const longlong2*>(ei + n_edges + k);
      }
#pragma unroll
      for (int i = 0; i < DS_CHK_ROWS; ++i) {
        const int64_t row = r0 + i < n_rows ? r0 + i : r0;
        const unsigned g = (unsigned)row / (unsigned)R;                            // (row < 2^31: checked by the caller)
        const int64_t src = row, dst = (int64_t)g * R + d;                        // source id == row index
        bad |= s2[i].x != src || s2[i].y != src || d2[i].x != dst || d2[i].y != dst + 1;
      }
    }
  }
  if (__syncthreads_or(bad) && threadIdx.x == 0) {
    atomicOr(status, 4);
    atomicOr(status + 1, 1);
  }
}
__global__ void __launch_bounds__(256)
k_ds_check(int64_t n_rows, int R, const int64_t* __restrict__ ei, int64_t n_edges, int32_t* __restrict__ status) {
  ds_check_rows(blockIdx.x, gridDim.x, n_rows, R, ei, n_edges, status);
}

extern "C" int igcn_dense_blocks_check(int64_t n_graphs, int R, const int64_t* edge_index, int32_t* status,
                                       void* stream) {
  IGCN_REQUIRE(n_graphs > 0 && R > 0 && R % 2 == 0 && edge_index && status && ((uintptr_t)edge_index & 15) == 0,
               "dense_blocks_check: bad arguments (R even, 16-byte aligned edge_index)");
  const int64_t rows = n_graphs * R, ne = rows * R;
  IGCN_REQUIRE(rows < ((int64_t)1 << 31), "dense_blocks_check: more than 2^31 nodes");
  int64_t blocks = igcn_cdiv(rows, DS_CHK_ROWS);
  blocks = blocks > 8192 ? 8192 : blocks;
  if (hipMemsetAsync(status + 1, 0, sizeof(int32_t), (hipStream_t)stream) != hipSuccess) {     // this check's own verdict
    igcn_set_error("dense_blocks_check: clearing the verdict failed");
    return IGCN_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(k_ds_check, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, R, edge_index, ne,
                     status);
  IGCN_CHECK_LAUNCH("dense_blocks_check");
  return IGCN_OK;
}

// =================================================================================================
// forward
// =================================================================================================
// u[i] = exp(-xm[i] . a[:H0]), v[i] = exp(-xm[i] . a[H0:]) with xm = x * prob (cal_probability :137-143: the pair logit
// of edge (s, d) is the sum of the two dot products, its sigmoid 1 / (1 + u[s] v[d])); workgroup 0 also leaves the node / SNP parts of loss_probability in reg_out[0]
__global__ void __launch_bounds__(256)
k_ds_prep(int64_t GR, int R, int H0, const float* __restrict__ x, const float* __restrict__ prob,
          const float* __restrict__ pb, float* __restrict__ u, float* __restrict__ v,
          const float* __restrict__ snps_prob, int n_snps, DsReg rg, float* __restrict__ reg_out,
          int32_t* __restrict__ clear_verdict /* status words: [1] = 0 in front of a check that rides in k_ds_deg */) {
  __shared__ float red[16];
  if (clear_verdict && blockIdx.x == 0 && threadIdx.x == 0) clear_verdict[1] = 0;
  const int64_t node = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (node < GR) {
    const int r = (int)(node % R);
    float uu = 0.f, vv = 0.f;
    for (int h = 0; h < H0; ++h) {
      const float xm = x[node * H0 + h] * prob[r * H0 + h];
      uu += xm * pb[h];
      vv += xm * pb[H0 + h];
    }
    u[node] = __expf(-uu);                              // the edge passes want exp(-u), exp(-v): ds_mask
    v[node] = __expf(-vv);
  }
  if (blockIdx.x == 0 && reg_out) {
    float acc = 0.f;
    const int np = R * H0;
    for (int i = threadIdx.x; i < np; i += 256) acc += ds_reg_term(ds_sigmoid(prob[i]), rg.l1_x, rg.ent_x, rg.eps) / (float)np;
    for (int i = threadIdx.x; i < n_snps; i += 256)
      acc += ds_reg_term(ds_sigmoid(snps_prob[i]), rg.l1_x, rg.ent_x, rg.eps) / (float)n_snps;
    acc = block_sum_all(acc, red);
    if (threadIdx.x == 0) reg_out[0] = acc;
  }
}

// deg[c][d] = sum_s w_c[s,d] -> dis = deg^-1/2 ; the edge part of loss_probability as one partial per workgroup.
// grid (R / 64, G), 512 threads: wave w walks source rows [w R/8, (w+1) R/8), lane (q, sub) loads 16 bytes of row
// s + sub at targets d0 + 4 q .. + 3.
template <int N>
struct DsDegBuf {
  float4 w4[N];
  float us[N];
};

template <int NC, bool M0, bool PIPE>
__global__ void __launch_bounds__(512)
k_ds_deg(int R, int64_t GR, const float* __restrict__ ew, const float* __restrict__ u, const float* __restrict__ v,
         float* __restrict__ dis, DsReg rg, float inv_ne, float* __restrict__ reg_partial, int n_deg_blocks,
         const int64_t* __restrict__ chk_ei, int64_t chk_rows, int64_t chk_edges, int32_t* __restrict__ status,
         int n_chk_blocks, int64_t d_total, const DropSegs d_segs, unsigned long long* __restrict__ d_state,
         float* __restrict__ d_out, const DropCounters d_cnt, unsigned d_blocks) {
  constexpr bool ANYM = NC == 2 || M0;
  if ((int)blockIdx.x >= n_deg_blocks + n_chk_blocks) {
    // the step's dropout rider (csrc/dropout.h; queued by the captured step, a launch of its own — the FIRST of the replay,
    // 7.7 us — where no per-graph plan build carries it): two of its 256-thread blocks per workgroup
    dropout_masks_body(2u * (blockIdx.x - (unsigned)(n_deg_blocks + n_chk_blocks)) + (threadIdx.x >> 8), d_blocks, d_total,
                       d_segs, d_state, d_out, d_cnt, threadIdx.x & 255u);
    return;
  }
  if ((int)blockIdx.x >= n_deg_blocks) {
    // the batch's structure check rides in this launch: these workgroups stream the index pairs (16 bytes per edge)
    // while the others stream the weights (4 bytes per edge); the verdict is read by the NEXT launch (k_ds_h0)
    ds_check_rows((int64_t)blockIdx.x - n_deg_blocks, (int64_t)n_chk_blocks, chk_rows, R, chk_ei, chk_edges, status);
    return;
  }
  __shared__ float red[8][NC][64];
  __shared__ float rsum[16];
  int g, xb;
  ds_block_of(R / 64, n_deg_blocks, (int)blockIdx.x, g, xb);
  const int d0 = xb * 64, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int q = lane & 15, sub = lane >> 4;
  const int rows = R / 8, sb = w * rows;
  float vd[4] = {0.f, 0.f, 0.f, 0.f};
  if (ANYM) {
    const float4 t = *reinterpret_cast<const float4*>(v + (int64_t)g * R + d0 + 4 * q);
    vd[0] = t.x; vd[1] = t.y; vd[2] = t.z; vd[3] = t.w;
  }
  float lvd[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) lvd[t] = ds_logit_part(vd[t]);
  float acc[NC][4];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[c][t] = 0.f;
  float racc = 0.f;
  const int64_t nb = (int64_t)g * R;
  const float* base = ew + nb * R + d0 + 4 * q;
  auto load = [&](auto& bf, int row) {
    constexpr int N = sizeof(bf.us) / sizeof(float);
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
      const int s = row + 4 * jj + sub;
      bf.w4[jj] = *reinterpret_cast<const float4*>(base + (int64_t)s * R);
      bf.us[jj] = ANYM ? u[nb + s] : 0.f;
    }
  };
  auto compute = [&](const auto& bf, int) {
    constexpr int N = sizeof(bf.us) / sizeof(float);
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
      const float wv[4] = {bf.w4[jj].x, bf.w4[jj].y, bf.w4[jj].z, bf.w4[jj].w};
      float ev[4] = {1.f, 1.f, 1.f, 1.f};
      if (ANYM) {
        const float lus = ds_logit_part(bf.us[jj]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          ev[t] = ds_mask(bf.us[jj], vd[t]);
          racc += ds_reg_term_logit(ev[t], lus + lvd[t], rg.l1_e, rg.ent_e, rg.eps);
        }
      }
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[c][t] += ds_masked<NC, M0>(c) ? wv[t] * ev[t] : wv[t];
    }
  };
  ds_walk<PIPE, 4, DsDegBuf<DS_PCH>, DsDegBuf<1>>(sb, sb + rows, load, compute, load, compute);
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float a = acc[c][t];
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      if (sub == 0) red[w][c][4 * q + t] = a;
    }
  racc = wave_sum(racc);
  if (lane == 0) rsum[w] = racc;
  __syncthreads();
  if (tid < NC * 64) {
    const int c = tid >> 6, dl = tid & 63;
    float deg = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) deg += red[ww][c][dl];
    dis[(int64_t)c * GR + nb + d0 + dl] = deg > 0.f ? 1.0f / sqrtf(deg) : 0.f;
  }
  if (ANYM && reg_partial && tid == 0) {
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) t += rsum[ww];
    reg_partial[(int64_t)g * (R / 64) + xb] = t * inv_ne;
  }
}

// hp_0[c][i] = dis_c[i] * (X_c[i] W_0^T), X = x (plain) or x * prob (masked)
template <int NC, bool M0>
__global__ void __launch_bounds__(256)
k_ds_h0(int64_t GR, int R, int H0, const float* __restrict__ x, const float* __restrict__ prob,
        const float* __restrict__ W0, const float* __restrict__ dis, float* __restrict__ hp0,
        const int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)NC * GR * DS_F) return;
  // the structure check of this batch (stand-alone earlier on the stream, or riding in k_ds_deg's launch) found edges that
  // are not the row-major complete graph these kernels assume: the first layer's features — and through them every output
  // and the loss — become NaN
  if (status != nullptr && status[1] != 0) {
    hp0[i] = __int_as_float(0x7fc00000);
    return;
  }
  const int f = (int)(i & (DS_F - 1));
  const int64_t cn = i >> 4, node = cn % GR;
  const int c = (int)(cn / GR), r = (int)(node % R);
  float a = 0.f;
  for (int h = 0; h < H0; ++h) {
    float xv = x[node * H0 + h];
    if (ds_masked<NC, M0>(c)) xv *= prob[r * H0 + h];
    a += xv * W0[f * H0 + h];
  }
  hp0[i] = dis[cn] * a;
}

// One GCNConv aggregation for every copy: AGG[c][d] = sum_s w_c[s,d] hp[c][s] on the matrix cores, then
// Y = relu(dis[d] AGG + b) into its columns of xcat and — when another layer follows — hp_next = dis * (Y W_next^T).
// grid (R / 64, G), 512 threads; dynamic LDS: 8 waves' partial tiles [8][NC][64][16] + the Y rows [NC][64][16].
template <int N, int NC, int VW = 4>
struct DsAggBuf {
  float w[N][VW];
  float bop[N][NC], us[N];
};

// TB = targets per workgroup; a lane carries VW = TB / 16 consecutive targets of a source row.  Only TB = 64 is
// instantiated: 32-target blocks (512 workgroups of 123 registers, two per CU, so that one workgroup's products and
// epilogue run under the other's stream — VERDICT r3's first route to 0.5 of HBM) were built and measured at 32 graphs x
// 512 nodes, both passes: 14.2 us against 12.7 us for the 64-target blocks in the same step (8-byte loads and 128-byte
// row segments per wave instruction, twice the fixed cost per workgroup).  The other way to two workgroups per CU — one
// workgroup per (block, COPY), plain and masked neighbours on one XCD so that the second reads ew from L2 — measured
// 15.9 us: each of them pays the whole stream's latency chain for half the products.
template <int NC, bool M0, bool PIPE, int TB = 64>
__global__ void __launch_bounds__(512)
k_ds_agg(int R, int64_t GR, const float* __restrict__ ew, const float* __restrict__ u, const float* __restrict__ v,
         const float* __restrict__ dis, const float* __restrict__ hp, const float* __restrict__ bias,
         const float* __restrict__ Wnext, float* __restrict__ agg, float* __restrict__ xcat, int ldx, int col0,
         float* __restrict__ hp_next) {
  constexpr int VW = TB / 16;
  extern __shared__ float ds_lds[];
  float* part = ds_lds;                                   // [8][NC][TB][16]
  float* ys = ds_lds + 8 * NC * TB * DS_F;                // [NC][TB][16]
  int g, xb;
  ds_block(R / TB, g, xb);
  const int d0 = xb * TB, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int q = lane & 15, sub = lane >> 4;
  const int rows = R / 8, sb = w * rows;
  constexpr bool ANYM = NC == 2 || M0;
  float vd[VW];
#pragma unroll
  for (int t = 0; t < VW; ++t) vd[t] = ANYM ? v[(int64_t)g * R + d0 + VW * q + t] : 0.f;
  f32x4 acc[NC][VW];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int t = 0; t < VW; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t nb = (int64_t)g * R;
  const float* base = ew + nb * R + d0 + VW * q;
  auto load = [&](auto& bf, int row) {
    constexpr int N = sizeof(bf.us) / sizeof(float);
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
      const int s = row + 4 * jj + sub;
      if constexpr (VW == 4) {
        const float4 t4 = *reinterpret_cast<const float4*>(base + (int64_t)s * R);
        bf.w[jj][0] = t4.x; bf.w[jj][1] = t4.y; bf.w[jj][2] = t4.z; bf.w[jj][3] = t4.w;
      } else {
        const float2 t2 = *reinterpret_cast<const float2*>(base + (int64_t)s * R);
        bf.w[jj][0] = t2.x; bf.w[jj][1] = t2.y;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) bf.bop[jj][c] = hp[((int64_t)c * GR + nb + s) * DS_F + q];   // B[k = sub][f = q]
      bf.us[jj] = ANYM ? u[nb + s] : 0.f;
    }
  };
  auto compute = [&](const auto& bf, int row) {
    constexpr int N = sizeof(bf.us) / sizeof(float);
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
      float ev[VW];
      // DS_ABL (tools/dense_ablate.sh; never in a shipped build — the results are wrong, only durations mean anything):
      // 1 = stream only (one VALU fma per product slot), 2 = masks kept, products on the VALU slot, 3 = matrix products
      // kept, masks dropped
#if defined(DS_ABL) && (DS_ABL == 1 || DS_ABL == 3)
#pragma unroll
      for (int t = 0; t < VW; ++t) ev[t] = 1.f + bf.us[jj] * 0.f;
#else
#pragma unroll
      for (int t = 0; t < VW; ++t) ev[t] = ANYM ? ds_mask(bf.us[jj], vd[t]) : 1.f;
#endif
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int t = 0; t < VW; ++t) {
          const float a = ds_masked<NC, M0>(c) ? bf.w[jj][t] * ev[t] : bf.w[jj][t];      // A[target VW q + t][k = sub]
#if defined(DS_ABL) && (DS_ABL == 1 || DS_ABL == 2)
          acc[c][t][0] = fmaf(a, bf.bop[jj][c], acc[c][t][0]);
#else
          acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bf.bop[jj][c], acc[c][t], 0, 0, 0);
#endif
        }
#ifdef DS_PROBE_ON
      if (row + 4 * jj == sb) DS_PROBE(1);            // the wave's first step has its data and its products are issued
      if (row + 4 * jj == sb + 28) DS_PROBE(2);       // ... its first 8 steps
#endif
    }
  };
  // epilogue operands of this thread (its (copy, target, feature) slots o = tid + 512 k and its row of W_next): requested
  // before the walk — behind the barrier each was a global round trip in front of the stores
  constexpr int EPI = NC * TB * DS_F / 512;
  float e_dis[EPI], e_bias = bias[tid & 15], e_w[DS_F];
#pragma unroll
  for (int k = 0; k < EPI; ++k) {
    const int o = tid + 512 * k, dl = (o >> 4) % TB, c = o / (DS_F * TB);
    e_dis[k] = dis[(int64_t)c * GR + nb + d0 + dl];
  }
#pragma unroll
  for (int f = 0; f < DS_F; ++f) e_w[f] = Wnext ? Wnext[(tid & 15) * DS_F + f] : 0.f;
  DS_PROBE(0);
  // Measured at 512-node graphs, both passes (us per launch; 42.4 MB of HBM traffic by the PMC counters = the bytes the
  // kernel has to move; 0.54 GFLOP of exact-fp32 MFMA = 3.4 us at peak, ~5 us of walk time by the phase probe):
  //   one step at a time                                                             19.7
  //   all 16 steps' loads up front, predicated (=> s_waitcnt vmcnt(0) before the first multiply) 19.1
  //   two-buffer pipeline, chunks of 4 steps, branch-free                             14.6
  //   the same, chunks of 8 (= every load of the wave in flight, second half arriving under the first half's products)  13.6  <- kept
  //   node operands staged in LDS per workgroup (64 KB + barrier) + all loads up front            17.3 / 21.4
  //   node operands staged in LDS per WAVE (8 KB, no barrier) + all loads up front                17.0
  // The single-pass instance (NC = 1: 4 instead of 8 matrix instructions per step) runs in 9.1 us.
  // Round 5, where the 13.0 us of the launch inside the captured configs[4] step go (tools/dense_ablate.sh: kernel durations
  // with parts compiled out; tools/dense_probe.py step: stamps of the first workgroups, shader clock 2.37 GHz in the step):
  //   products compiled out (DS_ABL=1)   8.97 us   workgroup: first data +2.6 us, all 16 steps' data by +3.4, done +4.4
  //   masks kept, no products (2)        8.95 us   (the masks cost nothing)
  //   products kept, no masks (3)       13.67 us
  //   as shipped                        12.99 us   workgroup: first data +2.6, walk done +7.7, done +8.6
  // i.e. ~4.5 us of every one of these launches lie outside its workgroups (dispatch of 256 x 512 threads with 72 KB of
  // LDS, end-of-kernel write-back: the in-step floor of any launch), the data of a workgroup arrives as ONE burst 2.6 us
  // after its waves start (1.5 us stand-alone, warm or evicted alike), and the 256 matrix instructions of a SIMD
  // (8192 cycles = 3.46 us at 2.37 GHz) then run in 4.2-5.1 us with nothing left to overlap them with: latency + products +
  // epilogue in series.  Tried on that picture, all within +-0.1 us of 13.0 and not kept: every load pinned in front of the
  // first product (__builtin_amdgcn_sched_barrier; the scheduler otherwise sinks half of them in between the products),
  // u as ONE load per wave + lane permutes (48 instead of 64 loads in flight), the 64 masks of a wave computed during the
  // data wait (219 registers).  Worse: loads issued in 2 / 4 groups with a fence-less workgroup barrier between the groups
  // so that the two waves of a SIMD get their data alternately (13.6 / 14.0).  With exact-fp32 products the launch is
  // bounded near 4.5 + 2.6 + 3.5 + 0.9 = 11.5 us (0.46 of HBM by its bytes); the 10.6 us that 0.5 asks for needs fewer
  // matrix cycles, i.e. split-bf16 products on a transposed operand (LDS transpose + 2.5 conversion instructions per edge).
  if (R == 512)
    ds_walk<PIPE, 4, DsAggBuf<8, NC, VW>, DsAggBuf<1, NC, VW>, decltype(load), decltype(compute), decltype(load),
            decltype(compute), 8>(sb, sb + rows, load, compute, load, compute);
  else
    ds_walk<PIPE, 4, DsAggBuf<DS_PCH, NC, VW>, DsAggBuf<1, NC, VW>>(sb, sb + rows, load, compute, load, compute);
  DS_PROBE(3);
  // accumulator lane (q, sub), tile t, register r = (target d0 + VW (4 sub + r) + t, feature q)
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int t = 0; t < VW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        part[((w * NC + c) * TB + VW * (4 * sub + r) + t) * DS_F + q] = acc[c][t][r];
  __syncthreads();
  DS_PROBE(4);
#pragma unroll
  for (int k = 0; k < EPI; ++k) {
    const int o = tid + 512 * k, f = o & 15, dl = (o >> 4) % TB, c = o / (DS_F * TB);
    float a = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) a += part[((ww * NC + c) * TB + dl) * DS_F + f];
    const int64_t node = (int64_t)c * GR + nb + d0 + dl;
    agg[node * DS_F + f] = a;
    const float t_ = e_dis[k] * a + e_bias;
    const float y = t_ < 0.f ? 0.f : t_;                 // ReLU that lets NaN through (fmaxf would swallow the poisoned degrees)
    xcat[node * ldx + col0 + f] = y;
    ys[(c * TB + dl) * DS_F + f] = y;
  }
  DS_PROBE(5);
  if (Wnext == nullptr) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < EPI; ++k) {
    const int o = tid + 512 * k, dl = (o >> 4) % TB, c = o / (DS_F * TB);     // output feature = tid & 15
    float a = 0.f;
#pragma unroll
    for (int f = 0; f < DS_F; ++f) a += ys[(c * TB + dl) * DS_F + f] * e_w[f];
    const int64_t node = (int64_t)c * GR + nb + d0 + dl;
    hp_next[node * DS_F + (tid & 15)] = e_dis[k] * a;
  }
  DS_PROBE(6);
}

// =================================================================================================
// backward
// =================================================================================================
// backward workspace: gp [L][NC][F][GR] (feature-major: the matrix-core operands of the edge passes are 16-byte loads
//                     along the node axis) | dhp [NC][GR][F] | T [GR] | ddeg [GR] | dup [R/64 .. R/16][GR] | dv [GR] |
//                     dxin [NC][GR][H0] | dxm [GR][H0] | db partials [L][nblk][F] | dW partials [L][nblk][F*F] |
//                     da partials [nblk2][2 H0]
struct DsBws {
  int64_t gp, dhp, T, ddeg, dup, dv, dxin, dxm, pdb, pdw, pda, total;
  int64_t nblk, nblk2;
};
__host__ __device__ inline DsBws ds_bws(int64_t GR, int R, int H0, int L, int NC) {
  DsBws o;
  o.nblk = (int64_t)NC * GR / 16;
  o.nblk2 = (GR + 63) / 64;
  int64_t p = 0;
  o.gp = p; p += (int64_t)L * NC * GR * DS_F;
  o.dhp = p; p += (int64_t)NC * GR * DS_F;
  o.T = p; p += GR;
  o.ddeg = p; p += GR;
  o.dup = p; p += (int64_t)(R / 16) * GR;
  o.dv = p; p += GR;
  o.dxin = p; p += (int64_t)NC * GR * H0;
  o.dxm = p; p += GR * H0;
  o.pdb = p; p += (int64_t)L * o.nblk * DS_F;
  o.pdw = p; p += (int64_t)L * o.nblk * DS_F * DS_F;
  o.pda = p; p += o.nblk2 * 2 * H0;
  o.total = p;
  return o;
}

extern "C" size_t igcn_dense_sgcn_bwd_ws_floats(int64_t n_graphs, int R, int H0, int L, int copies) {
  return (size_t)ds_bws(n_graphs * R, R, H0, L, copies).total;
}

// g'_{L-1} = dis * dY * (Y > 0) for the last layer + its bias-gradient partials.  256 threads = 16 nodes x 16 features
// of ONE copy; grid NC GR / 16.
__global__ void __launch_bounds__(256)
k_ds_node_top(int64_t GR, int L, const float* __restrict__ xcat, const float* __restrict__ dxcat,
              const float* __restrict__ dxcat2 /* a second consumer's gradient of xcat, or NULL */, int ldx,
              const float* __restrict__ dis, float* __restrict__ gp_last, float* __restrict__ db_partial) {
  __shared__ float gs[16][DS_F];
  const int tid = threadIdx.x, f = tid & 15, nl = tid >> 4;
  const int64_t cn = (int64_t)blockIdx.x * 16 + nl;                   // copy * GR + node
  const int64_t off = cn * ldx + (int64_t)(L - 1) * DS_F + f;
  const float g = xcat[off] > 0.f ? dxcat[off] + (dxcat2 ? dxcat2[off] : 0.f) : 0.f;
  gp_last[((cn / GR) * DS_F + f) * GR + cn % GR] = dis[cn] * g;       // feature-major: [copy][f][node]
  gs[nl][f] = g;
  __syncthreads();
  if (tid < DS_F) {
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) s += gs[n][tid];
    db_partial[(int64_t)blockIdx.x * DS_F + tid] = s;
  }
}

// dH'[c][s] = sum_d w_c[s,d] g'[c][d] (the transposed aggregation) on the matrix cores.  grid (R / 64, G), 512 threads:
// wave w owns source tile (w & 3) of the block's 64 sources and one half (w >> 2) of the targets.
template <int N, int NC>
struct DsAggTBuf {
  float4 w4[N], v4[N], b4[N][NC];
};

template <int NC, bool M0, bool PIPE>
__global__ void __launch_bounds__(512)
k_ds_aggT(int R, int64_t GR, const float* __restrict__ ew, const float* __restrict__ u, const float* __restrict__ v,
          const float* __restrict__ gp, float* __restrict__ dhp) {
  // dynamic LDS only (a kernel with static LDS cannot raise its dynamic limit to the full 160 KB):
  //   part [2][4][NC][16][16] | FAST: lgp [NC][16][R + 4] | lv [R]
  extern __shared__ float dt_lds[];
  float (*part)[4][NC][16][DS_F] = reinterpret_cast<float (*)[4][NC][16][DS_F]>(dt_lds);
  int g, xb;
  ds_block(R / 64, g, xb);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int q = lane & 15, sub = lane >> 4, st = w & 3, dh = w >> 2;
  const int s0 = xb * 64 + 16 * st;
  const int64_t nb = (int64_t)g * R;
  constexpr bool ANYM = NC == 2 || M0;
  const float us = ANYM ? u[nb + s0 + q] : 0.f;
  f32x4 acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* row = ew + (nb + s0 + q) * (int64_t)R + 4 * sub;
  const int dbeg = dh * (R / 2), dend = dbeg + R / 2;
  auto load = [&](auto& bf, int col) {
    constexpr int N = sizeof(bf.w4) / sizeof(float4);
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
      const int ds = col + 16 * jj;
      bf.w4[jj] = *reinterpret_cast<const float4*>(row + ds);
      if (ANYM) bf.v4[jj] = *reinterpret_cast<const float4*>(v + nb + ds + 4 * sub);
#pragma unroll
      for (int c = 0; c < NC; ++c)            // B[k = sub][f = q] of the four products: targets ds + 4 sub + (0..3)
        bf.b4[jj][c] = *reinterpret_cast<const float4*>(gp + ((int64_t)c * DS_F + q) * GR + nb + ds + 4 * sub);
    }
  };
  auto compute = [&](const auto& bf, int) {
    constexpr int N = sizeof(bf.w4) / sizeof(float4);
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
      const float wv[4] = {bf.w4[jj].x, bf.w4[jj].y, bf.w4[jj].z, bf.w4[jj].w};
      float ev[4] = {1.f, 1.f, 1.f, 1.f};
      if (ANYM) {
        ev[0] = ds_mask(us, bf.v4[jj].x); ev[1] = ds_mask(us, bf.v4[jj].y);
        ev[2] = ds_mask(us, bf.v4[jj].z); ev[3] = ds_mask(us, bf.v4[jj].w);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float bv[4] = {bf.b4[jj][c].x, bf.b4[jj][c].y, bf.b4[jj][c].z, bf.b4[jj][c].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float a = ds_masked<NC, M0>(c) ? wv[t] * ev[t] : wv[t];                     // A[source q][k = sub]
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[t], acc[c], 0, 0, 0);
        }
      }
    }
  };
  if constexpr (PIPE) {
    const int ldg = R + 4;                                // padded row: the 16 features of a lane group hit other banks
    float* lgp = dt_lds + 2 * 4 * NC * 16 * DS_F;         // [NC][16][R + 4], feature-major like gp
    float* lv = lgp + NC * DS_F * ldg;                    // [R]
    for (int i = tid; i < NC * DS_F * (R / 4); i += 512) {
      const int rowi = i / (R / 4), c4 = i - rowi * (R / 4);          // rowi = c * 16 + f
      *reinterpret_cast<float4*>(lgp + rowi * ldg + 4 * c4) =
          *reinterpret_cast<const float4*>(gp + (int64_t)rowi * GR + nb + 4 * c4);
    }
    if (ANYM)
      for (int i = tid; i < R; i += 512) lv[i] = v[nb + i];
    __syncthreads();
    // [Round 5: every adjacency load of the walk issued BEFORE the staging and pinned there (sched_barrier) — 16 float4 per
    // lane in flight instead of the two the scheduler leaves ahead of each s_waitcnt — ran 21.2 us against 12.4 in the
    // stress step (17.9 without the pin): a wave-load here is 16 rows x 64 bytes, HALF of each 128-byte line, and steps jj,
    // jj + 1 share the lines; issued far apart, the second half has left the 32 KB L1 before it is asked for.]
    auto run = [&](auto nsc) {
      constexpr int NS = decltype(nsc)::value;
      float4 w4[NS];
#pragma unroll
      for (int jj = 0; jj < NS; ++jj) w4[jj] = *reinterpret_cast<const float4*>(row + dbeg + 16 * jj);
#pragma unroll
      for (int jj = 0; jj < NS; ++jj) {
        const int ds = dbeg + 16 * jj;
        DsAggTBuf<1, NC> bf;
        bf.w4[0] = w4[jj];
        if (ANYM) bf.v4[0] = *reinterpret_cast<const float4*>(lv + ds + 4 * sub);
#pragma unroll
        for (int c = 0; c < NC; ++c) bf.b4[0][c] = *reinterpret_cast<const float4*>(lgp + (c * DS_F + q) * ldg + ds + 4 * sub);
        compute(bf, 0);
      }
    };
    if (R == 512) run(DsInt<16>{}); else run(DsInt<8>{});
  } else {
    ds_walk<false, 16, DsAggTBuf<DS_PCH, NC>, DsAggTBuf<1, NC>>(dbeg, dend, load, compute, load, compute);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[dh][st][c][4 * sub + r][q] = acc[c][r];                // (source 4 sub + r, feature q)
  __syncthreads();
  for (int o = tid; o < 4 * NC * 16 * DS_F; o += 512) {
    const int f = o & 15, i = (o >> 4) & 15, c = (o >> 8) % NC, t4 = o / (256 * NC);
    const float a = part[0][t4][c][i][f] + part[1][t4][c][i][f];
    dhp[((int64_t)c * GR + nb + xb * 64 + 16 * t4 + i) * DS_F + f] = a;
  }
}

// Node-level step between two transposed aggregations, for layer l (256 threads = 16 nodes x 16 features of one copy):
//   dH = dis * dH' ; dW_l partial = dH^T X_l ; dX = dH W_l ;
//   masked copy: T[i] (+)= g'_l[i] . AGG_l[i] + h'_l[i] . dH'_l[i], and at l == 0: ddeg[i] = -1/2 dis[i]^2 T[i]
//   l > 0: g'_{l-1} = dis * (dY_{l-1} + dX) * (Y_{l-1} > 0) + bias-gradient partials of layer l-1 ; l == 0: dxin = dX
template <int NC, bool M0>
__global__ void __launch_bounds__(256)
k_ds_node_mid(int64_t GR, int R, int H0, int L, int l, const float* __restrict__ x, const float* __restrict__ prob,
              const float* __restrict__ xcat, const float* __restrict__ dxcat, const float* __restrict__ dxcat2, int ldx,
              const float* __restrict__ dis, const float* __restrict__ Wl, const float* __restrict__ hp_l,
              const float* __restrict__ agg_l, const float* __restrict__ gp_l, const float* __restrict__ dhp,
              float* __restrict__ gp_prev, float* __restrict__ dxin, float* __restrict__ T,
              float* __restrict__ ddeg, float* __restrict__ dw_partial, float* __restrict__ db_prev_partial) {
  __shared__ float dHs[16][DS_F + 1], Xs[16][DS_F + 1], Ws[DS_F][DS_F + 1], gs[16][DS_F];
  const int tid = threadIdx.x, f = tid & 15, nl = tid >> 4;
  const int fin = l == 0 ? H0 : DS_F;
  const int64_t cn = (int64_t)blockIdx.x * 16 + nl, node = cn % GR;
  const int c = (int)(cn / GR);
  const bool masked = ds_masked<NC, M0>(c);
  const float di = dis[cn];
  const float dh = dhp[cn * DS_F + f];
  dHs[nl][f] = di * dh;
  if (f < fin) {
    float xv;
    if (l == 0) {
      xv = x[node * H0 + f];
      if (masked) xv *= prob[(node % R) * H0 + f];
    } else {
      xv = xcat[cn * ldx + (int64_t)(l - 1) * DS_F + f];
    }
    Xs[nl][f] = xv;
    Ws[nl][f] = Wl[nl * fin + f];                              // W_l [F][fin]: row nl (an output feature), column f
  }
  if (masked) {
    float t = gp_l[((int64_t)c * DS_F + f) * GR + node] * agg_l[cn * DS_F + f] + hp_l[cn * DS_F + f] * dh;
    t = group_sum_all<16>(t);
    if (f == 0) {
      const float tot = (l == L - 1 ? 0.f : T[node]) + t;
      T[node] = tot;
      if (l == 0) ddeg[node] = -0.5f * di * di * tot;
    }
  }
  __syncthreads();
  {                                                           // dW partial: thread (fo = nl, fi = f)
    if (f < fin) {
      float s = 0.f;
#pragma unroll
      for (int n = 0; n < 16; ++n) s += dHs[n][nl] * Xs[n][f];
      dw_partial[(int64_t)blockIdx.x * (DS_F * fin) + nl * fin + f] = s;
    }
  }
  float dxv = 0.f;
  if (f < fin) {
#pragma unroll
    for (int fo = 0; fo < DS_F; ++fo) dxv += dHs[nl][fo] * Ws[fo][f];
  }
  if (l == 0) {
    if (f < fin) dxin[cn * H0 + f] = dxv;
    return;
  }
  const int64_t off = cn * ldx + (int64_t)(l - 1) * DS_F + f;
  const float gval = xcat[off] > 0.f ? dxcat[off] + (dxcat2 ? dxcat2[off] : 0.f) + dxv : 0.f;
  gp_prev[((int64_t)c * DS_F + f) * GR + node] = di * gval;
  gs[nl][f] = gval;
  __syncthreads();
  if (tid < DS_F) {
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) s += gs[n][tid];
    db_prev_partial[(int64_t)blockIdx.x * DS_F + tid] = s;
  }
}

// The mask gradient's edge pass (masked copy): q[s,d] = sum_l g'_l[d] . h'_l[s] on the matrix cores (K = L F),
//   dz = ((q + ddeg[d]) ew + greg reg'(e) / nE) e (1 - e);  du[s] = sum_d dz (one partial per 64-target block), dv[d] = sum_s dz.
// grid (R / 64, G), 512 threads: the workgroup owns 64 targets, wave w the sources [w R / 8, (w + 1) R / 8).
// Lane (q = lane & 15, sub = lane >> 4) loads 16 bytes ew[s0 + 4 sub + r][d0 + 4 q ..] for r = 0..3 — a wave's load
// instruction covers four rows x 256 contiguous bytes, as in k_ds_agg (the first form of this kernel gave a wave 16
// targets: 16 rows x 64 bytes per instruction, 29 us for the 33.5 MB) — and these are exactly the elements of four
// accumulator tiles t: D_t[i = source 4 sub + r][j = q] <-> (source s0 + 4 sub + r, target d0 + 4 q + t), with
// A = h'_l[s0 + q][features] (one 16-byte load per layer: k index (kk, sub) <-> feature 4 sub + kk) and
// B_t = g'_l[d0 + 4 q + t][features] (feature-major g': 16-byte loads over t, once per wave).
template <int N, int L>
struct DsMbBuf {
  float4 w4[N][4], h4[N][L];
  float us[N][4];
};

__device__ __forceinline__ float ds_row16_sum(float v) {        // sum over the 16 lanes of a DPP row, in every lane
  auto dpp = [](float a, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});             // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});             // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});            // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});            // row_mirror
  return v;
}

template <int L, bool PIPE>
__global__ void __launch_bounds__(512)
k_ds_mask_bwd(int R, int64_t GR, const float* __restrict__ ew, const float* __restrict__ u, const float* __restrict__ v,
              const float* __restrict__ gpm /*[L][..][F][GR] at the masked copy*/, const float* __restrict__ hpm,
              int64_t lstride, const float* __restrict__ ddeg, const float* __restrict__ greg, DsReg rg, float inv_ne,
              float* __restrict__ dup, float* __restrict__ dv) {
  __shared__ float dvs[8][64];
  int g, xb;
  ds_block(R / 64, g, xb);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int q = lane & 15, sub = lane >> 4;
  const int d0 = xb * 64;
  const int64_t nb = (int64_t)g * R;
  const float gr = greg ? greg[0] * inv_ne : 0.f;
  float bop[L][4][4];                                                       // [layer][kk][tile t]
#pragma unroll
  for (int l = 0; l < L; ++l)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float4 g4 = *reinterpret_cast<const float4*>(gpm + l * lstride + (int64_t)(4 * sub + kk) * GR + nb + d0 + 4 * q);
      bop[l][kk][0] = g4.x; bop[l][kk][1] = g4.y; bop[l][kk][2] = g4.z; bop[l][kk][3] = g4.w;
    }
  float vd[4], dd[4], dvacc[4] = {0.f, 0.f, 0.f, 0.f};
  {
    const float4 v4 = *reinterpret_cast<const float4*>(v + nb + d0 + 4 * q);
    const float4 d4 = *reinterpret_cast<const float4*>(ddeg + nb + d0 + 4 * q);
    vd[0] = v4.x; vd[1] = v4.y; vd[2] = v4.z; vd[3] = v4.w;
    dd[0] = d4.x; dd[1] = d4.y; dd[2] = d4.z; dd[3] = d4.w;
  }
  float lvd[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) lvd[t] = ds_logit_part(vd[t]);
  const float c0 = gr * rg.l1_e, c1 = -gr * rg.ent_e;
  const int nblk = R / 16, sbeg = 16 * ((w * nblk) / 8), send = 16 * (((w + 1) * nblk) / 8);   // whole 16-source blocks
  auto load = [&](auto& bf, int srow) {
    constexpr int N = sizeof(bf.h4) / sizeof(bf.h4[0]);
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
      const int s0 = srow + 16 * jj;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bf.w4[jj][r] = *reinterpret_cast<const float4*>(ew + (nb + s0 + 4 * sub + r) * (int64_t)R + d0 + 4 * q);
        bf.us[jj][r] = u[nb + s0 + 4 * sub + r];
      }
#pragma unroll
      for (int l = 0; l < L; ++l)
        bf.h4[jj][l] = *reinterpret_cast<const float4*>(hpm + l * lstride + (nb + s0 + q) * DS_F + 4 * sub);   // A[source q][k]
    }
  };
  auto compute = [&](const auto& bf, int srow) {
    constexpr int N = sizeof(bf.h4) / sizeof(bf.h4[0]);
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
      const int s0 = srow + 16 * jj;
      f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const float hv[4] = {bf.h4[jj][l].x, bf.h4[jj][l].y, bf.h4[jj][l].z, bf.h4[jj][l].w};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[kk], bop[l][kk][t], acc[t], 0, 0, 0);
      }
      float rs[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float wv[4] = {bf.w4[jj][r].x, bf.w4[jj][r].y, bf.w4[jj][r].z, bf.w4[jj][r].w};
        const float lus = ds_logit_part(bf.us[jj][r]);
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // reg'(e) e (1 - e) with e = sigmoid(z): log(e / (1 - e)) IS the logit z = u[s] + v[d], so the entropy term's
          // gradient l1 - ent (log e - log(1 - e)) costs one add per edge instead of two logarithms and two reciprocals
          // (the reference's eps = 1e-6 inside the logarithms changes the product by O(eps): see ds_reg_term_logit)
          const float e = ds_mask(bf.us[jj][r], vd[t]);
          const float de = fmaf(acc[t][r] + dd[t], wv[t], fmaf(c1, lus + lvd[t], c0));
          const float dz = de * fmaf(-e, e, e);
          dvacc[t] += dz;
          a += dz;
        }
        rs[r] = ds_row16_sum(a);                                              // over the block's 64 targets
      }
      if (q < 4) {
        const float out = q == 0 ? rs[0] : q == 1 ? rs[1] : q == 2 ? rs[2] : rs[3];
        dup[(int64_t)xb * GR + nb + s0 + 4 * sub + q] = out;
      }
    }
  };
  // two-buffer pipeline of single 16-source blocks (two blocks' 20 loads per lane in flight; chunks of two blocks: 256
  // registers and spills)
  ds_walk<PIPE, 16, DsMbBuf<1, L>, DsMbBuf<1, L>, decltype(load), decltype(compute), decltype(load), decltype(compute), 1>(
      sbeg, send, load, compute, load, compute);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float a = dvacc[t];
    a += __shfl_xor(a, 16, 64);
    a += __shfl_xor(a, 32, 64);
    if (sub == 0) dvs[w][4 * q + t] = a;
  }
  __syncthreads();
  if (tid < 64) {
    float a = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) a += dvs[ww][tid];
    dv[nb + d0 + tid] = a;
  }
}

// node-level end of the mask backward: du = sum of the tile partials, dxm = du a[:H0] + dv a[H0:] + dX_0(masked copy),
// dx = dX_0(plain copy) + dxm * prob, d prob_bias partials (sum_i du xm, sum_i dv xm)
template <int NC, bool M0>
__global__ void __launch_bounds__(64)
k_ds_mask_nodes(int64_t GR, int R, int H0, const float* __restrict__ x, const float* __restrict__ prob,
                const float* __restrict__ pb, const float* __restrict__ dup, const float* __restrict__ dv,
                const float* __restrict__ dxin, float* __restrict__ dxm, float* __restrict__ dx,
                float* __restrict__ da_partial) {
  // one wave per workgroup (GR / 64 of them: 16 K nodes are 256 workgroups, not 64)
  constexpr bool ANYM = NC == 2 || M0;
  constexpr int CM = NC - 1;                                     // index of the masked copy when there is one
  const int64_t node = (int64_t)blockIdx.x * 64 + threadIdx.x;
  float da[2 * DS_MAXH0];
#pragma unroll
  for (int h = 0; h < 2 * DS_MAXH0; ++h) da[h] = 0.f;
  if (node < GR) {
    const int r = (int)(node % R);
    float du = 0.f, dvv = 0.f;
    if (ANYM) {
      // R / 64 partials (one per 64-target block of k_ds_mask_bwd), independent loads
#pragma unroll 8
      for (int p = 0; p < R / 64; ++p) du += dup[(int64_t)p * GR + node];
      dvv = dv[node];
    }
#pragma unroll
    for (int h = 0; h < DS_MAXH0; ++h)
      if (h < H0) {
        const float xv = x[node * H0 + h], pv = prob[r * H0 + h];
        float g = 0.f;
        if (NC == 2 || !M0) g = dxin[node * H0 + h];                              // the plain copy's dX_0 (copy 0)
        if (ANYM) {
          const float m = du * pb[h] + dvv * pb[H0 + h] + dxin[((int64_t)CM * GR + node) * H0 + h];
          dxm[node * H0 + h] = m;
          g += m * pv;
          da[h] = du * xv * pv;
          da[DS_MAXH0 + h] = dvv * xv * pv;
        }
        dx[node * H0 + h] = g;
      }
  }
  if (!ANYM) return;
#pragma unroll
  for (int h = 0; h < 2 * DS_MAXH0; ++h) {
    const float s = wave_sum(da[h]);                             // valid in lane 0
    if (threadIdx.x == 0 && (h < H0 || (h >= DS_MAXH0 && h - DS_MAXH0 < H0)))
      da_partial[(int64_t)blockIdx.x * 2 * H0 + (h < DS_MAXH0 ? h : H0 + h - DS_MAXH0)] = s;
  }
}

// d prob[r,h] = sum_g dxm[g,r,h] x[g,r,h] + the regulariser's gradient ; d snps_prob = the regulariser's gradient.
// 8 lanes share an item and split the graphs (a thread per item walked 32 graphs serially on 7 workgroups: 10 us).
__global__ void __launch_bounds__(256)
k_ds_dprob(int64_t G, int R, int H0, const float* __restrict__ x, const float* __restrict__ prob,
           const float* __restrict__ dxm, const float* __restrict__ snps_prob, int n_snps,
           const float* __restrict__ greg, DsReg rg, float* __restrict__ dprob, float* __restrict__ dsnps) {
  const int i = blockIdx.x * 32 + (threadIdx.x >> 3), gs = threadIdx.x & 7;
  const int np = R * H0;
  const float gr = greg ? greg[0] : 0.f;
  if (i < np) {
    float s = 0.f;
    for (int64_t g = gs; g < G; g += 8) s += dxm[g * np + i] * x[g * np + i];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (gs == 0) {
      const float p = ds_sigmoid(prob[i]);
      dprob[i] = s + gr * ds_reg_grad(p, rg.l1_x, rg.ent_x, rg.eps) * p * (1.f - p) / (float)np;
    }
  } else if (i < np + n_snps && dsnps && gs == 0) {
    const int k = i - np;
    const float p = ds_sigmoid(snps_prob[k]);
    dsnps[k] = gr * ds_reg_grad(p, rg.l1_x, rg.ent_x, rg.eps) * p * (1.f - p) / (float)n_snps;
  }
}

// =================================================================================================
// entry points
// =================================================================================================
static int ds_check_args(const char* nm, int64_t G, int R, int H0, int F, int L, int copies) {
  IGCN_REQUIRE(G > 0 && (copies == 1 || copies == 2), "%s: bad sizes", nm);
  if (!igcn_dense_sgcn_supported(R, H0, F, L)) {
    igcn_set_error("%s: needs 64 <= R <= 1024, R %% 64 == 0, F == 16, L <= %d, H0 <= %d (R=%d, H0=%d, F=%d, L=%d)", nm,
                   DS_MAXL, DS_MAXH0, R, H0, F, L);
    return IGCN_ERR_UNSUPPORTED;
  }
  return IGCN_OK;
}

#define DS_DISPATCH(...)                                                                \
  if (copies == 2) { constexpr int NC = 2; constexpr bool M0 = false; __VA_ARGS__; }       \
  else if (first_masked) { constexpr int NC = 1; constexpr bool M0 = true; __VA_ARGS__; }  \
  else { constexpr int NC = 1; constexpr bool M0 = false; __VA_ARGS__; }

extern "C" int igcn_dense_sgcn_fwd(int64_t n_graphs, int R, int H0, int F, int L, int copies, int first_masked,
                                   const float* x, const float* prob, const float* prob_bias, const float* ew,
                                   const float* const* W /*HOST [L]*/, const float* const* b /*HOST [L]*/,
                                   const float* snps_prob, int n_snps, float l1_x, float ent_x, float l1_e, float ent_e,
                                   float eps, float* xcat, float* reg_partials, float* ws, int32_t* status,
                                   const int64_t* check_edge_index, void* stream) {
  int rc = ds_check_args("dense_sgcn_fwd", n_graphs, R, H0, F, L, copies);
  if (rc) return rc;
  IGCN_REQUIRE(((uintptr_t)ew & 15) == 0 && ((uintptr_t)ws & 15) == 0, "dense_sgcn_fwd: ew / ws must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t GR = n_graphs * R;
  const bool anym = copies == 2 || first_masked;
  const DsWs o = ds_ws(GR, L, copies);
  const DsReg rg = {l1_x, ent_x, l1_e, ent_e, eps};
  const int nreg = igcn_dense_sgcn_reg_blocks(n_graphs, R);
  float* regp = anym ? reg_partials : nullptr;
  const bool ride = check_edge_index != nullptr;           // the batch's structure check rides in k_ds_deg's launch
  IGCN_REQUIRE(!ride || (status != nullptr && ((uintptr_t)check_edge_index & 15) == 0 && GR < ((int64_t)1 << 31)),
               "dense_sgcn_fwd: a riding check needs the status words and a 16-byte aligned edge_index");
  if (anym) {
    hipLaunchKernelGGL(k_ds_prep, dim3((unsigned)igcn_cdiv(GR, 256)), dim3(256), 0, st, GR, R, H0, x, prob, prob_bias,
                       ws + o.u, ws + o.v, snps_prob, snps_prob ? n_snps : 0, rg, regp ? regp + (nreg - 1) : nullptr,
                       ride ? status : nullptr);
  } else if (ride && hipMemsetAsync(status + 1, 0, sizeof(int32_t), st) != hipSuccess) {
    igcn_set_error("dense_sgcn_fwd: clearing the verdict failed");
    return IGCN_ERR_LAUNCH;
  }
  const dim3 eg((unsigned)(n_graphs * (R / 64)));            // 1-D: ds_block maps ids to (graph, block), XCD-aware
  const float inv_ne = 1.0f / (float)((double)n_graphs * R * R);
  const bool pipe = R == 256 || R == 512;              // pipelined walks need whole chunk pairs (R % 256 == 0); the LDS-
                                                       // staged transposed aggregation additionally R <= 512
#define DS_PIPE(...) if (pipe) { constexpr bool PIPE = true; __VA_ARGS__; } else { constexpr bool PIPE = false; __VA_ARGS__; }
  int64_t chk = ride ? igcn_cdiv(GR, 2 * DS_CHK_ROWS) : 0;         // one trip of 8 rows per 512-thread workgroup
  chk = chk > 4096 ? 4096 : chk;
  // a dropout rider waiting on the stream (igcn_rider_dropout) is carried by this launch
  DropJob job = {};
  if (!igcn_rider_dropout_take(st, job)) job.blocks = 0;
  const dim3 dg((unsigned)(eg.x + chk + (job.blocks + 1u) / 2u));
  DS_DISPATCH(DS_PIPE(hipLaunchKernelGGL((k_ds_deg<NC, M0, PIPE>), dg, dim3(512), 0, st, R, GR, ew, ws + o.u, ws + o.v,
                                         ws + o.dis, rg, inv_ne, regp, (int)eg.x, check_edge_index, GR, GR * R, status,
                                         (int)chk, job.total, job.sg, job.state, job.out, job.cnt, job.blocks)));
  DS_DISPATCH(hipLaunchKernelGGL((k_ds_h0<NC, M0>), dim3((unsigned)igcn_cdiv((int64_t)copies * GR * DS_F, 256)),
                                 dim3(256), 0, st, GR, R, H0, x, prob, W[0], ws + o.dis, ws + o.hp,
                                 (const int32_t*)status));
  const size_t lds = (size_t)(8 * copies * 64 * DS_F + copies * 64 * DS_F) * sizeof(float);
  for (int l = 0; l < L; ++l) {
    const float* hp = ws + o.hp + (int64_t)l * copies * GR * DS_F;
    float* hpn = l + 1 < L ? ws + o.hp + (int64_t)(l + 1) * copies * GR * DS_F : nullptr;
    float* ag = ws + o.agg + (int64_t)l * copies * GR * DS_F;
    DS_DISPATCH(DS_PIPE(if (lds > 64 * 1024) IGCN_ALLOW_BIG_LDS((k_ds_agg<NC, M0, PIPE>));
                        hipLaunchKernelGGL((k_ds_agg<NC, M0, PIPE>), eg, dim3(512), lds, st, R, GR, ew, ws + o.u, ws + o.v,
                                           ws + o.dis, hp, b[l], l + 1 < L ? W[l + 1] : nullptr, ag, xcat, L * DS_F,
                                           l * DS_F, hpn)));
  }
  IGCN_CHECK_LAUNCH("dense_sgcn_fwd");
  return IGCN_OK;
}

extern "C" int igcn_dense_sgcn_bwd(int64_t n_graphs, int R, int H0, int F, int L, int copies, int first_masked,
                                   const float* x, const float* prob, const float* prob_bias, const float* ew,
                                   const float* const* W, const float* snps_prob, int n_snps, float l1_x, float ent_x,
                                   float l1_e, float ent_e, float eps, const float* xcat, const float* dxcat,
                                   const float* dxcat2 /* NULL: xcat had one consumer */,
                                   const float* d_reg /*device [1] or NULL*/, const float* ws, float* bws, float* dx,
                                   float* dprob, float* dprob_bias, float* dsnps_prob, float* dparams, void* stream) {
  int rc = ds_check_args("dense_sgcn_bwd", n_graphs, R, H0, F, L, copies);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t GR = n_graphs * R;
  const bool anym = copies == 2 || first_masked;
  const DsWs o = ds_ws(GR, L, copies);
  const DsBws q = ds_bws(GR, R, H0, L, copies);
  const DsReg rg = {l1_x, ent_x, l1_e, ent_e, eps};
  const int ldx = L * DS_F;
  const bool pipe = R == 256 || R == 512;
  const dim3 eg((unsigned)(n_graphs * (R / 64)));            // 1-D: ds_block maps ids to (graph, block), XCD-aware
  const int64_t lsz = (int64_t)copies * GR * DS_F;                    // one layer of hp / agg / gp
  const float* dis = ws + o.dis;
  hipLaunchKernelGGL(k_ds_node_top, dim3((unsigned)q.nblk), dim3(256), 0, st, GR, L, xcat, dxcat, dxcat2, ldx, dis,
                     bws + q.gp + (int64_t)(L - 1) * lsz, bws + q.pdb + (int64_t)(L - 1) * q.nblk * DS_F);
  for (int l = L - 1; l >= 0; --l) {
    const size_t ldt = (size_t)(2 * 4 * copies * 16 * DS_F + (pipe ? copies * DS_F * (R + 4) + R : 0)) * sizeof(float);
    DS_DISPATCH(DS_PIPE(if (ldt > 64 * 1024) IGCN_ALLOW_BIG_LDS((k_ds_aggT<NC, M0, PIPE>));
                        hipLaunchKernelGGL((k_ds_aggT<NC, M0, PIPE>), eg, dim3(512), ldt, st, R, GR, ew, ws + o.u, ws + o.v,
                                           bws + q.gp + (int64_t)l * lsz, bws + q.dhp)));
    DS_DISPATCH(hipLaunchKernelGGL(
        (k_ds_node_mid<NC, M0>), dim3((unsigned)q.nblk), dim3(256), 0, st, GR, R, H0, L, l, x, prob, xcat, dxcat, dxcat2, ldx, dis,
        W[l], ws + o.hp + (int64_t)l * lsz, ws + o.agg + (int64_t)l * lsz, bws + q.gp + (int64_t)l * lsz, bws + q.dhp,
        l > 0 ? bws + q.gp + (int64_t)(l - 1) * lsz : nullptr, bws + q.dxin, bws + q.T, bws + q.ddeg,
        bws + q.pdw + (int64_t)l * q.nblk * DS_F * DS_F, l > 0 ? bws + q.pdb + (int64_t)(l - 1) * q.nblk * DS_F : nullptr));
  }
  if (anym) {
    const int cm = copies - 1;
    const float inv_ne = 1.0f / (float)((double)n_graphs * R * R);
    const float* gpm = bws + q.gp + (int64_t)cm * GR * DS_F;
    const float* hpm = ws + o.hp + (int64_t)cm * GR * DS_F;
    // (pipelined walk: every wave needs at least two 16-source blocks, i.e. R >= 256)
#define DS_MB(LV)                                                                                                       \
  if (R % 256 == 0) hipLaunchKernelGGL((k_ds_mask_bwd<LV, true>), eg, dim3(512), 0, st, R, GR, ew, ws + o.u, ws + o.v, gpm, \
                                       hpm, lsz, bws + q.ddeg, d_reg, rg, inv_ne, bws + q.dup, bws + q.dv);                \
  else hipLaunchKernelGGL((k_ds_mask_bwd<LV, false>), eg, dim3(512), 0, st, R, GR, ew, ws + o.u, ws + o.v, gpm, hpm, lsz,    \
                          bws + q.ddeg, d_reg, rg, inv_ne, bws + q.dup, bws + q.dv)
    switch (L) {
      case 1: DS_MB(1); break;
      case 2: DS_MB(2); break;
      case 3: DS_MB(3); break;
      default: DS_MB(4); break;
    }
#undef DS_MB
  }
  DS_DISPATCH(hipLaunchKernelGGL((k_ds_mask_nodes<NC, M0>), dim3((unsigned)q.nblk2), dim3(64), 0, st, GR, R, H0, x,
                                 prob, prob_bias, bws + q.dup, bws + q.dv, bws + q.dxin, bws + q.dxm, dx, bws + q.pda));
  if (anym) {
    const int np = R * H0 + (dsnps_prob ? n_snps : 0);
    hipLaunchKernelGGL(k_ds_dprob, dim3((unsigned)igcn_cdiv(np, 32)), dim3(256), 0, st, n_graphs, R, H0, x, prob,
                       bws + q.dxm, snps_prob, dsnps_prob ? n_snps : 0, d_reg, rg, dprob, dsnps_prob);
  }
  IGCN_CHECK_LAUNCH("dense_sgcn_bwd");
  // parameter gradients: block partials -> [dW_0 | db_0 | dW_1 | db_1 | ...] (final reductions: deferrable)
  int64_t off = 0;
  for (int l = 0; l < L; ++l) {
    const int fin = l == 0 ? H0 : DS_F;
    rc = igcn_launch_reduce_rows_final(bws + q.pdw + (int64_t)l * q.nblk * DS_F * DS_F, q.nblk, DS_F * fin, DS_F * fin,
                                       dparams + off, st);
    if (rc) return rc;
    off += DS_F * fin;
    rc = igcn_launch_reduce_rows_final(bws + q.pdb + (int64_t)l * q.nblk * DS_F, q.nblk, DS_F, DS_F, dparams + off, st);
    if (rc) return rc;
    off += DS_F;
  }
  if (anym) {
    rc = igcn_launch_reduce_rows_final(bws + q.pda, q.nblk2, 2 * H0, 2 * H0, dprob_bias, st);
    if (rc) return rc;
  }
  return IGCN_OK;
}
