// Node-wise read-outs of the GO network: per-node linear transform + BatchNorm1d(#nodes) + ReLU, fused.
//   pre[b,n,:] = W x[b,:,n]             x [B,F,N] channel-major, W [D,F]
//   out[b,n,:] = relu( (pre - mean_n) * rstd_n * gamma[n] + beta[n] )         out [B,N,D]
// BatchNorm1d(N) on a [B,N,D] tensor normalises every NODE over (batch, feature) — go_model.py:117-121
// (conc_for_attention, D = dim_snps_atten), :123-128 (conc + B, D = 1), :130-136 (conc_D + B_D, D = 1).
// Replaces per call: a permute copy, a K=5 rocBLAS GEMM, ~6 BatchNorm kernels, a clamp, and in the backward the
// K = B*N weight-gradient GEMMs that rocBLAS runs at ~190 us each.
#include "common.h"
#include <type_traits>

#define RO_T 256
#define RO_NL 64      // node lanes per block
#define RO_SG 4       // sample groups per block

// W is read through wave-uniform addresses: the compiler keeps it on the scalar path (s_load + SGPR operands),
// so no vector registers are spent on the D*F weights.
template <int F, int D>
__device__ __forceinline__ void ro_pre(const float* __restrict__ w, const float (&x)[F], float (&o)[D]) {
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < F; ++c) t += w[d * F + c] * x[c];
    o[d] = t;
  }
}

// Samples are split into `groups` equal groups that are normalised independently (the two forward passes of a
// train step batched into one launch keep their own BatchNorm statistics).  gridDim.y = groups * cpg chunks;
// chunk cy covers samples [b0,b1) of group g = cy / cpg.
struct RoChunk {
  int g, ck, b0, b1, first;
};
// A kernel body sees its block through RoBlk: its own launch (blockIdx / gridDim), or its half of a PAIRED launch in
// which two read-outs of the same input share one grid (k_nlbn_pair_*).
struct RoBlk { int bx, by, gx, gy; };
#define RO_BLK_OF_LAUNCH RoBlk{(int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y}
__device__ __forceinline__ RoChunk ro_chunk(int B, int groups, const RoBlk rb) {
  const int cpg = rb.gy / groups, bg = B / groups;
  RoChunk c;
  c.g = rb.by / cpg;
  c.ck = rb.by % cpg;
  const int per = (bg + cpg - 1) / cpg;
  c.first = c.g * bg;
  c.b0 = c.first + c.ck * per;
  c.b1 = min(c.first + bg, c.b0 + per);
  return c;
}

// ---- forward pass 1: shifted sums per (node, sample chunk) ---------------------------------------
// partial[chunk][0][n] = sum (pre - pivot_n), partial[chunk][1][n] = sum (pre - pivot_n)^2 ; pivot from sample 0
template <int F, int D>
__device__ __forceinline__ void
nlbn_stats_body(const RoBlk rb, int B, int N, int groups, const float* __restrict__ x, const float* __restrict__ W,
             float* __restrict__ partial) {
  __shared__ float s1[RO_SG][RO_NL], s2[RO_SG][RO_NL];
  const float* __restrict__ w = W;
  const int nl = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int n = rb.bx * RO_NL + nl;
  const RoChunk ch = ro_chunk(B, groups, rb);
  const int b0 = ch.b0, b1 = ch.b1;
  float a1 = 0.f, a2 = 0.f;
  if (n < N) {
    float xv[F], pre[D];
#pragma unroll
    for (int c = 0; c < F; ++c) xv[c] = x[((int64_t)ch.first * F + c) * N + n];   // pivot: the group's first sample
    ro_pre<F, D>(w, xv, pre);
    float piv = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) piv += pre[d];
    piv *= (1.f / D);
    for (int b = b0 + sg; b < b1; b += RO_SG) {
#pragma unroll
      for (int c = 0; c < F; ++c) xv[c] = x[((int64_t)b * F + c) * N + n];
      ro_pre<F, D>(w, xv, pre);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const float v = pre[d] - piv;
        a1 += v;
        a2 += v * v;
      }
    }
  }
  s1[sg][nl] = a1;
  s2[sg][nl] = a2;
  __syncthreads();
  if (sg == 0 && n < N) {
    float* p = partial + (int64_t)rb.by * 2 * N;
    p[n] = (s1[0][nl] + s1[1][nl]) + (s1[2][nl] + s1[3][nl]);
    p[N + n] = (s2[0][nl] + s2[1][nl]) + (s2[2][nl] + s2[3][nl]);
  }
}
template <int F, int D>
__global__ void __launch_bounds__(RO_T)
k_nlbn_stats(int B, int N, int groups, const float* __restrict__ x, const float* __restrict__ W,
             float* __restrict__ partial) {
  nlbn_stats_body<F, D>(RO_BLK_OF_LAUNCH, B, N, groups, x, W, partial);
}

// ---- forward pass 2: finalise statistics (training) or take the running ones (eval) ----------------
// mean_out / rstd_out are [groups, N]; the running statistics are updated group after group, exactly as two
// successive forward calls would.
// 64 nodes x 4 chunk slices per workgroup: the cpg partials of a node are summed by four lanes (one batch of
// independent loads each) and combined through LDS — with a thread per node and ceil(N/64) waves on the whole chip
// the serial walk over the chunks was a 10-us latency chain for a few KB of data.
#define RO_FIN_MAXG 8
template <int F, int D>
__device__ __forceinline__ void
nlbn_finalize_body(const RoBlk rb, int B, int N, int groups, int cpg, int training, float eps, float momentum,
                const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ partial,
                float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ mean_out,
                float* __restrict__ rstd_out) {
  __shared__ float r1[RO_FIN_MAXG][4][64], r2[RO_FIN_MAXG][4][64];
  const int nl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int n = rb.bx * 64 + nl;
  const bool live = n < N;
  const int bg = B / groups;
  if (training && live) {
    for (int g = 0; g < groups; ++g) {
      float a1 = 0.f, a2 = 0.f;
      for (int k0 = sl; k0 < cpg; k0 += 32) {           // eight chunk partials per batch, all loads before the first add
        float v1[8], v2[8];                             // (a `#pragma unroll 8` loop leaves 2-7 trips to a serial remainder)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + 4 * u, kc = k < cpg ? k : sl;
          v1[u] = partial[(int64_t)(g * cpg + kc) * 2 * N + n];
          v2[u] = partial[(int64_t)(g * cpg + kc) * 2 * N + N + n];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + 4 * u < cpg) {
            a1 += v1[u];
            a2 += v2[u];
          }
      }
      r1[g][sl][nl] = a1;
      r2[g][sl][nl] = a2;
    }
  }
  __syncthreads();
  if (sl != 0 || !live) return;
  float rm = running_mean[n], rv = running_var[n];
  for (int g = 0; g < groups; ++g) {
    float mean, var;
    if (training) {
      const float* __restrict__ w = W;
      float xv[F], pre[D];
#pragma unroll
      for (int c = 0; c < F; ++c) xv[c] = x[((int64_t)g * bg * F + c) * N + n];
      ro_pre<F, D>(w, xv, pre);
      float piv = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) piv += pre[d];
      piv *= (1.f / D);
      const float a1 = (r1[g][0][nl] + r1[g][1][nl]) + (r1[g][2][nl] + r1[g][3][nl]);
      const float a2 = (r2[g][0][nl] + r2[g][1][nl]) + (r2[g][2][nl] + r2[g][3][nl]);
      const float cnt = (float)bg * D;
      const float m = a1 / cnt;
      mean = piv + m;
      var = fmaxf(a2 / cnt - m * m, 0.f);
      rm = (1.f - momentum) * rm + momentum * mean;
      rv = (1.f - momentum) * rv + momentum * var * (cnt / (cnt - 1.f));
    } else {
      mean = rm;
      var = rv;
    }
    mean_out[(int64_t)g * N + n] = mean;
    rstd_out[(int64_t)g * N + n] = 1.0f / sqrtf(var + eps);
  }
  if (training) {
    running_mean[n] = rm;
    running_var[n] = rv;
  }
}
template <int F, int D>
__global__ void __launch_bounds__(256)
k_nlbn_finalize(int B, int N, int groups, int cpg, int training, float eps, float momentum,
                const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ partial,
                float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ mean_out,
                float* __restrict__ rstd_out) {
  nlbn_finalize_body<F, D>(RO_BLK_OF_LAUNCH, B, N, groups, cpg, training, eps, momentum, x, W, partial, running_mean, running_var, mean_out, rstd_out);
}

// ---- forward pass 3: normalise + ReLU, write [B,N,D] ---------------------------------------------
template <int F, int D>
__device__ __forceinline__ void
nlbn_apply_body(const RoBlk rb, int B, int N, int groups, const float* __restrict__ x, const float* __restrict__ W,
             const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
             const float* __restrict__ rstd, const float* __restrict__ keep /*[B,N] or NULL (D == 1)*/,
             float* __restrict__ out) {
  const float* __restrict__ w = W;
  const int nl = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int n = rb.bx * RO_NL + nl;
  if (n >= N) return;
  const RoChunk ch = ro_chunk(B, groups, rb);
  const int b0 = ch.b0, b1 = ch.b1;
  const float sc = rstd[(int64_t)ch.g * N + n] * gamma[n], sh = beta[n] - mean[(int64_t)ch.g * N + n] * sc;
  for (int b = b0 + sg; b < b1; b += RO_SG) {
    float xv[F], pre[D];
#pragma unroll
    for (int c = 0; c < F; ++c) xv[c] = x[((int64_t)b * F + c) * N + n];
    ro_pre<F, D>(w, xv, pre);
    float* o = out + ((int64_t)b * N + n) * D;
    if constexpr (D % 4 == 0) {
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        float4 v;
        v.x = fmaxf(pre[d] * sc + sh, 0.f);
        v.y = fmaxf(pre[d + 1] * sc + sh, 0.f);
        v.z = fmaxf(pre[d + 2] * sc + sh, 0.f);
        v.w = fmaxf(pre[d + 3] * sc + sh, 0.f);
        *reinterpret_cast<float4*>(o + d) = v;
      }
    } else {
      const float kp = keep ? keep[(int64_t)b * N + n] : 1.f;          // fused dropout of the read-out (D == 1)
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] = fmaxf(pre[d] * sc + sh, 0.f) * kp;
    }
  }
}
template <int F, int D>
__global__ void __launch_bounds__(RO_T)
k_nlbn_apply(int B, int N, int groups, const float* __restrict__ x, const float* __restrict__ W,
             const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
             const float* __restrict__ rstd, const float* __restrict__ keep /*[B,N] or NULL (D == 1)*/,
             float* __restrict__ out) {
  nlbn_apply_body<F, D>(RO_BLK_OF_LAUNCH, B, N, groups, x, W, gamma, beta, mean, rstd, keep, out);
}

// Row-coalesced form of the pass above for D % 4 == 0, D <= 64: lane = (node, output quad), so a wave's store
// instruction covers 64/(D/4) whole rows of `out` = contiguous memory (the thread-per-node form writes D/4 float4
// per lane at a stride of D floats: every store instruction touches 64 different 128-byte lines).  The F inputs of
// a node are loaded by its first F lanes and passed around with shuffles.
template <int F, int D>
__device__ __forceinline__ void
nlbn_apply_q_body(const RoBlk rb, int B, int N, int groups, const float* __restrict__ x, const float* __restrict__ W,
               const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
               const float* __restrict__ rstd, float* __restrict__ out) {
  constexpr int DQ = D / 4, NPW = 64 / DQ;             // quads per row, nodes per wave
  static_assert(D % 4 == 0 && 64 % DQ == 0 && F <= DQ, "k_nlbn_apply_q: unsupported shape");
  const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int q = lane % DQ, nw = lane / DQ;
  const int n = rb.bx * NPW + nw;
  const bool live = n < N;
  const int nc = live ? n : N - 1;
  const RoChunk ch = ro_chunk(B, groups, rb);
  float w[4][F];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < F; ++c) w[r][c] = W[(q * 4 + r) * F + c];
  const float sc = rstd[(int64_t)ch.g * N + nc] * gamma[nc], sh = beta[nc] - mean[(int64_t)ch.g * N + nc] * sc;
  const int base = lane - q;                            // first lane of this node
  for (int b = ch.b0 + sg; b < ch.b1; b += RO_SG) {
    const float mine = q < F ? x[((int64_t)b * F + q) * N + nc] : 0.f;
    float4 v;
    float* vv = reinterpret_cast<float*>(&v);
    float xv[F];
#pragma unroll
    for (int c = 0; c < F; ++c) xv[c] = __shfl(mine, base + c, 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < F; ++c) t += w[r][c] * xv[c];
      vv[r] = fmaxf(t * sc + sh, 0.f);
    }
    if (live) *reinterpret_cast<float4*>(out + ((int64_t)b * N + n) * D + q * 4) = v;
  }
}
template <int F, int D>
__global__ void __launch_bounds__(RO_T)
k_nlbn_apply_q(int B, int N, int groups, const float* __restrict__ x, const float* __restrict__ W,
               const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
               const float* __restrict__ rstd, float* __restrict__ out) {
  nlbn_apply_q_body<F, D>(RO_BLK_OF_LAUNCH, B, N, groups, x, W, gamma, beta, mean, rstd, out);
}

static int ro_quad_chunks(unsigned grid_y, int groups) {
  const int cpg = (int)grid_y / groups;
  return cpg > 8 ? 8 : cpg;
}

template <int F, int D>
static void ro_launch_apply(dim3 grid, hipStream_t st, int B, int N, int groups, const float* x, const float* W,
                            const float* gamma, const float* beta, const float* mean, const float* rstd,
                            const float* keep, float* out) {
  if constexpr (D % 4 == 0 && D <= 64 && D >= 16 && 64 % (D / 4) == 0 && F <= D / 4) {
    // 64/(D/4) nodes per wave give 8x the workgroups of the thread-per-node grid: fewer, longer sample chunks
    dim3 gq((unsigned)igcn_cdiv(N, 64 / (D / 4)), ro_quad_chunks(grid.y, groups) * groups);
    hipLaunchKernelGGL((k_nlbn_apply_q<F, D>), gq, dim3(RO_T), 0, st, B, N, groups, x, W, gamma, beta, mean, rstd, out);
  } else {
    hipLaunchKernelGGL((k_nlbn_apply<F, D>), grid, dim3(RO_T), 0, st, B, N, groups, x, W, gamma, beta, mean, rstd, keep,
                       out);
  }
}

// a thread's D consecutive floats with 16-byte loads / stores when D allows it (rows of [.., D] tensors start on
// 16-byte boundaries then): the scalar form issues D vector-memory instructions per row
template <int D>
__device__ __forceinline__ void ro_load_row(const float* __restrict__ p, float (&v)[D]) {
  if constexpr (D % 4 == 0) {
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      const float4 t = *reinterpret_cast<const float4*>(p + d);
      v[d] = t.x; v[d + 1] = t.y; v[d + 2] = t.z; v[d + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = p[d];
  }
}

template <int D>
__device__ __forceinline__ void ro_store_row(float* __restrict__ p, const float (&v)[D]) {
  if constexpr (D % 4 == 0) {
#pragma unroll
    for (int d = 0; d < D; d += 4) *reinterpret_cast<float4*>(p + d) = make_float4(v[d], v[d + 1], v[d + 2], v[d + 3]);
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d) p[d] = v[d];
  }
}

// ---- backward pass 1: per node sum(dy), sum(dy*xhat) over (b,d), dy = dout * [out > 0] ---------------
template <int F, int D>
__device__ __forceinline__ void
nlbn_bwd_stats_body(const RoBlk rb, int B, int N, int groups, const float* __restrict__ x, const float* __restrict__ W,
                 const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                 const float* __restrict__ rstd, const float* __restrict__ dout, const float* __restrict__ keep,
                 float* __restrict__ partial) {
  __shared__ float s1[RO_SG][RO_NL], s2[RO_SG][RO_NL];
  const float* __restrict__ w = W;
  const int nl = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int n = rb.bx * RO_NL + nl;
  const RoChunk ch = ro_chunk(B, groups, rb);
  const int b0 = ch.b0, b1 = ch.b1;
  float a1 = 0.f, a2 = 0.f;
  if (n < N) {
    const float mu = mean[(int64_t)ch.g * N + n], rs = rstd[(int64_t)ch.g * N + n], ga = gamma[n], be = beta[n];
    for (int b = b0 + sg; b < b1; b += RO_SG) {
      float xv[F], pre[D];
#pragma unroll
      for (int c = 0; c < F; ++c) xv[c] = x[((int64_t)b * F + c) * N + n];
      ro_pre<F, D>(w, xv, pre);
      float g[D];
      ro_load_row<D>(dout + ((int64_t)b * N + n) * D, g);
      const float kp = keep ? keep[(int64_t)b * N + n] : 1.f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const float xh = (pre[d] - mu) * rs;
        const float dy = (xh * ga + be > 0.f) ? g[d] * kp : 0.f;
        a1 += dy * xh;      // -> dgamma
        a2 += dy;           // -> dbeta
      }
    }
  }
  s1[sg][nl] = a1;
  s2[sg][nl] = a2;
  __syncthreads();
  if (sg == 0 && n < N) {
    float* p = partial + (int64_t)(ch.ck * groups + ch.g) * 2 * N;      // [chunk][group][2][N]
    p[n] = (s1[0][nl] + s1[1][nl]) + (s1[2][nl] + s1[3][nl]);
    p[N + n] = (s2[0][nl] + s2[1][nl]) + (s2[2][nl] + s2[3][nl]);
  }
}
template <int F, int D>
__global__ void __launch_bounds__(RO_T)
k_nlbn_bwd_stats(int B, int N, int groups, const float* __restrict__ x, const float* __restrict__ W,
                 const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                 const float* __restrict__ rstd, const float* __restrict__ dout, const float* __restrict__ keep,
                 float* __restrict__ partial) {
  nlbn_bwd_stats_body<F, D>(RO_BLK_OF_LAUNCH, B, N, groups, x, W, gamma, beta, mean, rstd, dout, keep, partial);
}

// ---- backward pass 2: dx [B,F,N] and the weight gradient -------------------------------------------
//   training: dpre = gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat)) ;  eval: dpre = gamma*rstd*dy
//   D*F <= 16 : dW accumulated in registers, block-reduced, one partial row per block (wpartial)
//   otherwise : dpre [B,N,D] is written out and dW = sum_b dpre_b^T x_b runs on the MFMA batched-sum GEMM
template <int F, int D>
__device__ __forceinline__ void
nlbn_bwd_apply_body(const RoBlk rb, int B, int N, int groups, int training, const float* __restrict__ x, const float* __restrict__ W,
                 const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                 const float* __restrict__ rstd, const float* __restrict__ dout, const float* __restrict__ keep,
                 const float* __restrict__ partial, int nchunks, float* __restrict__ dpre_out,
                 float* __restrict__ dx, float* __restrict__ wpartial) {
  constexpr bool SMALL = (D * F <= 16);
  constexpr int NW = SMALL ? D * F : 1;
  __shared__ float red[(RO_T / 64) * NW];
  const float* __restrict__ w = W;
  const int nl = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int n = rb.bx * RO_NL + nl;
  float gw[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) gw[j] = 0.f;
  if (n < N) {
    const RoChunk ch = ro_chunk(B, groups, rb);
    const int b0 = ch.b0, b1 = ch.b1;
    const float mu = mean[(int64_t)ch.g * N + n], rs = rstd[(int64_t)ch.g * N + n], ga = gamma[n], be = beta[n];
    const float cnt = (float)(B / groups) * D;
    // this group's (sum dy*xhat, sum dy): the chunk partials of pass 1 [chunk][group][2][N], summed here in chunk
    // order — the launch that used to do it between the two passes is gone
    float t1 = 0.f, t2 = 0.f;
    if (training)
      for (int c = 0; c < nchunks; ++c) {
        const float* p = partial + (int64_t)(c * groups + ch.g) * 2 * N;
        t2 += p[n];
        t1 += p[N + n];
      }
    const float m1 = t1 / cnt;                              // mean(dy)
    const float m2 = t2 / cnt;                              // mean(dy*xhat)
    for (int b = b0 + sg; b < b1; b += RO_SG) {
      float xv[F], pre[D], dxv[F];
#pragma unroll
      for (int c = 0; c < F; ++c) {
        xv[c] = x[((int64_t)b * F + c) * N + n];
        dxv[c] = 0.f;
      }
      ro_pre<F, D>(w, xv, pre);
      float g[D];
      ro_load_row<D>(dout + ((int64_t)b * N + n) * D, g);
      const float kp = keep ? keep[(int64_t)b * N + n] : 1.f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const float xh = (pre[d] - mu) * rs;
        const float dy = (xh * ga + be > 0.f) ? g[d] * kp : 0.f;
        const float t = ga * rs * (dy - m1 - xh * m2);
        if constexpr (SMALL) {
#pragma unroll
          for (int c = 0; c < F; ++c) gw[d * F + c] += t * xv[c];
        } else {
          g[d] = t;                                             // dpre, stored as a row below
        }
#pragma unroll
        for (int c = 0; c < F; ++c) dxv[c] += w[d * F + c] * t;
      }
      if constexpr (!SMALL) ro_store_row<D>(dpre_out + ((int64_t)b * N + n) * D, g);
#pragma unroll
      for (int c = 0; c < F; ++c) dx[((int64_t)b * F + c) * N + n] = dxv[c];
    }
  }
  if constexpr (SMALL)
    block_reduce_vec<NW>(gw, red, wpartial + ((int64_t)rb.by * rb.gx + rb.bx) * NW);
}
template <int F, int D>
__global__ void __launch_bounds__(RO_T)
k_nlbn_bwd_apply(int B, int N, int groups, int training, const float* __restrict__ x, const float* __restrict__ W,
                 const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                 const float* __restrict__ rstd, const float* __restrict__ dout, const float* __restrict__ keep,
                 const float* __restrict__ partial, int nchunks, float* __restrict__ dpre_out,
                 float* __restrict__ dx, float* __restrict__ wpartial) {
  nlbn_bwd_apply_body<F, D>(RO_BLK_OF_LAUNCH, B, N, groups, training, x, W, gamma, beta, mean, rstd, dout, keep, partial, nchunks, dpre_out, dx, wpartial);
}

// Sum over the G (= 2, 4, 8, 16) consecutive lanes of a group with DPP cross-lane moves (VALU; __shfl_xor is an LDS-pipe
// ds_bpermute): xor 1 and xor 2 as quad permutations, then the 8- and 16-lane mirrors (a mirror pairs every lane with
// one of the other half, which is all a sum needs).  Result in every lane of the group.
#define RO_QU 4        // samples per trip of the statistics pass (loads only)
template <int G>
__device__ __forceinline__ float ro_group_sum_dpp(float v) {
  static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "ro_group_sum_dpp: group size");
  auto dpp = [](float a, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  if constexpr (G >= 2) v += dpp(v, std::integral_constant<int, 0xB1>{});     // quad_perm [1,0,3,2]
  if constexpr (G >= 4) v += dpp(v, std::integral_constant<int, 0x4E>{});     // quad_perm [2,3,0,1]
  if constexpr (G >= 8) v += dpp(v, std::integral_constant<int, 0x141>{});    // row_half_mirror
  if constexpr (G >= 16) v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
  return v;
}

// ---- row-coalesced backward (D % 4 == 0, 16 <= D <= 64): lane = (node, quad of outputs) -------------------------
// dout rows are read (and nothing but dx / block partials written) with one 16-byte access per lane over contiguous
// memory; a node's F inputs are loaded by its first F lanes and shuffled around; sums over a node's outputs are
// xor-shuffles inside its D/4 lanes.  The weight gradient is accumulated in registers (4 x F per lane) and leaves the
// workgroup as one [D, F] partial: no dpre [B, N, D] round trip through HBM and no batched GEMM behind it.
template <int F, int D>
__device__ __forceinline__ void
nlbn_bwd_stats_q_body(const RoBlk rb, int B, int N, int groups, const float* __restrict__ x, const float* __restrict__ W,
                   const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                   const float* __restrict__ rstd, const float* __restrict__ dout, float* __restrict__ partial) {
  constexpr int DQ = D / 4, NPW = 64 / DQ;
  __shared__ float s1[RO_SG][NPW], s2[RO_SG][NPW];
  const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int q = lane % DQ, nw = lane / DQ, base = lane - q;
  const int n = rb.bx * NPW + nw;
  const bool live = n < N;
  const int nc = live ? n : N - 1;
  const RoChunk ch = ro_chunk(B, groups, rb);
  float w[4][F];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < F; ++c) w[r][c] = W[(q * 4 + r) * F + c];
  const float mu = mean[(int64_t)ch.g * N + nc], rs = rstd[(int64_t)ch.g * N + nc], ga = gamma[nc], be = beta[nc];
  float a1 = 0.f, a2 = 0.f;
  // RO_QU samples per trip, their loads issued together (clamped, branch-free): one sample per trip was a chain of
  // eight dependent round trips per thread (14.4 us for 30 MB)
  for (int b0 = ch.b0 + sg; b0 < ch.b1; b0 += RO_QU * RO_SG) {
    float mine[RO_QU];
    float4 g4[RO_QU];
#pragma unroll
    for (int u = 0; u < RO_QU; ++u) {
      const int b = min(b0 + u * RO_SG, ch.b1 - 1);
      mine[u] = q < F ? x[((int64_t)b * F + q) * N + nc] : 0.f;
      g4[u] = *reinterpret_cast<const float4*>(dout + ((int64_t)b * N + nc) * D + q * 4);
    }
#pragma unroll
    for (int u = 0; u < RO_QU; ++u) {
      if (b0 + u * RO_SG >= ch.b1) break;                       // (uniform over the wave)
      const float g[4] = {g4[u].x, g4[u].y, g4[u].z, g4[u].w};
      float xv[F];
#pragma unroll
      for (int c = 0; c < F; ++c) xv[c] = __shfl(mine[u], base + c, 64);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pre = 0.f;
#pragma unroll
        for (int c = 0; c < F; ++c) pre += w[r][c] * xv[c];
        const float xh = (pre - mu) * rs;
        const float dy = (xh * ga + be > 0.f) ? g[r] : 0.f;
        a1 += dy * xh;
        a2 += dy;
      }
    }
  }
  a1 = ro_group_sum_dpp<DQ>(a1);
  a2 = ro_group_sum_dpp<DQ>(a2);
  if (q == 0) {
    s1[sg][nw] = a1;
    s2[sg][nw] = a2;
  }
  __syncthreads();
  if (sg == 0 && q == 0 && live) {
    float* p = partial + (int64_t)(ch.ck * groups + ch.g) * 2 * N;      // [chunk][group][2][N]
    p[n] = (s1[0][nw] + s1[1][nw]) + (s1[2][nw] + s1[3][nw]);
    p[N + n] = (s2[0][nw] + s2[1][nw]) + (s2[2][nw] + s2[3][nw]);
  }
}
template <int F, int D>
__global__ void __launch_bounds__(RO_T)
k_nlbn_bwd_stats_q(int B, int N, int groups, const float* __restrict__ x, const float* __restrict__ W,
                   const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                   const float* __restrict__ rstd, const float* __restrict__ dout, float* __restrict__ partial) {
  nlbn_bwd_stats_q_body<F, D>(RO_BLK_OF_LAUNCH, B, N, groups, x, W, gamma, beta, mean, rstd, dout, partial);
}

template <int F, int D>
__device__ __forceinline__ void
nlbn_bwd_apply_q_body(const RoBlk rb, int B, int N, int groups, int training, const float* __restrict__ x, const float* __restrict__ W,
                   const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                   const float* __restrict__ rstd, const float* __restrict__ dout,
                   const float* __restrict__ partial, int nchunks, float* __restrict__ dx,
                   float* __restrict__ wpartial) {
  constexpr int DQ = D / 4, NPW = 64 / DQ;
  __shared__ float red[RO_T / 64][D * F];
  const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int q = lane % DQ, nw = lane / DQ, base = lane - q;
  const int n = rb.bx * NPW + nw;
  const bool live = n < N;
  const int nc = live ? n : N - 1;
  const RoChunk ch = ro_chunk(B, groups, rb);
  float w[4][F], gw[4][F];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < F; ++c) {
      w[r][c] = W[(q * 4 + r) * F + c];
      gw[r][c] = 0.f;
    }
  const float mu = mean[(int64_t)ch.g * N + nc], rs = rstd[(int64_t)ch.g * N + nc], ga = gamma[nc], be = beta[nc];
  const float cnt = (float)(B / groups) * D;
  float t1 = 0.f, t2 = 0.f;                                  // this group's (sum dy*xhat, sum dy): chunk partials of
  if (training)                                              // pass 1, summed here in chunk order
    for (int c = 0; c < nchunks; ++c) {
      const float* p = partial + (int64_t)(c * groups + ch.g) * 2 * N;
      t2 += p[nc];
      t1 += p[N + nc];
    }
  const float m1 = t1 / cnt;                                 // mean(dy)
  const float m2 = t2 / cnt;                                 // mean(dy*xhat)
  // (one sample per trip: batching the loads as in pass 1 did not pay here — 17.4 us against 18.7 with two samples per
  // trip and 23.3 with four: the pass is bound by its per-sample arithmetic and register footprint, not its load chain)
  for (int b = ch.b0 + sg; b < ch.b1; b += RO_SG) {
    const float mine = q < F ? x[((int64_t)b * F + q) * N + nc] : 0.f;
    const float4 g4 = *reinterpret_cast<const float4*>(dout + ((int64_t)b * N + nc) * D + q * 4);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    float xv[F], dxv[F];
#pragma unroll
    for (int c = 0; c < F; ++c) {
      xv[c] = __shfl(mine, base + c, 64);
      dxv[c] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float pre = 0.f;
#pragma unroll
      for (int c = 0; c < F; ++c) pre += w[r][c] * xv[c];
      const float xh = (pre - mu) * rs;
      const float dy = (xh * ga + be > 0.f) ? g[r] : 0.f;
      const float t = live ? ga * rs * (dy - m1 - xh * m2) : 0.f;       // shadow lanes add nothing to dW
#pragma unroll
      for (int c = 0; c < F; ++c) {
        gw[r][c] += t * xv[c];
        dxv[c] += w[r][c] * t;
      }
    }
    float mydx = 0.f;                                       // lane q < F ends up with channel q of dx
#pragma unroll
    for (int c = 0; c < F; ++c) {
      const float t = ro_group_sum_dpp<DQ>(dxv[c]);
      if (q == c) mydx = t;
    }
    if (live && q < F) dx[((int64_t)b * F + q) * N + n] = mydx;
  }
  // dW partial of the workgroup: sum over the wave's nodes (lanes with equal q), then over the four waves
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < F; ++c) {
      float t = gw[r][c];
#pragma unroll
      for (int o = DQ; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
      if (nw == 0) red[sg][(q * 4 + r) * F + c] = t;
    }
  __syncthreads();
  float* prow = wpartial + ((int64_t)rb.by * rb.gx + rb.bx) * (D * F);
  for (int j = threadIdx.x; j < D * F; j += RO_T) prow[j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
}
template <int F, int D>
__global__ void __launch_bounds__(RO_T)
k_nlbn_bwd_apply_q(int B, int N, int groups, int training, const float* __restrict__ x, const float* __restrict__ W,
                   const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                   const float* __restrict__ rstd, const float* __restrict__ dout,
                   const float* __restrict__ partial, int nchunks, float* __restrict__ dx,
                   float* __restrict__ wpartial) {
  nlbn_bwd_apply_q_body<F, D>(RO_BLK_OF_LAUNCH, B, N, groups, training, x, W, gamma, beta, mean, rstd, dout, partial, nchunks, dx, wpartial);
}

template <int F, int D>
static constexpr bool ro_quad_ok() {
  return D % 4 == 0 && D >= 16 && D <= 64 && 64 % (D / 4) == 0 && F <= D / 4;
}

#define RO_DISPATCH(F, D, CALL)                                    \
  if (F == 5 && D == 32) { CALL(5, 32); }                          \
  else if (F == 5 && D == 48) { CALL(5, 48); }                     \
  else if (F == 5 && D == 30) { CALL(5, 30); }                     \
  else if (F == 5 && D == 20) { CALL(5, 20); }                     \
  else if (F == 5 && D == 16) { CALL(5, 16); }                     \
  else if (F == 5 && D == 12) { CALL(5, 12); }                     \
  else if (F == 5 && D == 24) { CALL(5, 24); }                     \
  else if (F == 5 && D == 10) { CALL(5, 10); }                     \
  else if (F == 5 && D == 8) { CALL(5, 8); }                       \
  else if (F == 5 && D == 6) { CALL(5, 6); }                       \
  else if (F == 5 && D == 4) { CALL(5, 4); }                       \
  else if (F == 5 && D == 3) { CALL(5, 3); }                       \
  else if (F == 5 && D == 2) { CALL(5, 2); }                       \
  else if (F == 5 && D == 5) { CALL(5, 5); }                       \
  else if (F == 5 && D == 1) { CALL(5, 1); }                       \
  else if (F == 2 && D == 1) { CALL(2, 1); }                       \
  else {                                                           \
    igcn_set_error("node_linear_bn: unsupported (F=%d, D=%d)", F, D); \
    return IGCN_ERR_UNSUPPORTED;                                   \
  }

static int ro_cpg(int B, int groups) {
  // sample chunks per group: up to 32 (enough workgroups for small N), but never fewer than RO_SG samples per chunk — a
  // workgroup's four waves take a chunk's samples in turn, so one-sample chunks (configs[4]: 32 samples per group) ran
  // three of four waves idle in 4x the workgroups (k_nlbn_pair_bwd_apply: 27 us for 28 MB)
  const int bg = B / groups, c = bg / RO_SG;
  return c >= 32 ? 32 : (c > 0 ? c : 1);
}

extern "C" size_t igcn_node_linear_bn_scratch_floats(int B, int N, int groups) {
  return (size_t)groups * ro_cpg(B, groups) * 2 * N + 64;
}

extern "C" int igcn_node_linear_bn_fwd(int B, int F, int N, int D, int groups, const float* x, const float* W,
                                       const float* gamma, const float* beta, float* running_mean,
                                       float* running_var, int training, float momentum, float eps,
                                       const float* keep /*[B,N] dropout factors, D == 1 only, or NULL*/, float* out,
                                       float* save_mean, float* save_rstd, float* scratch, void* stream) {
  IGCN_REQUIRE(B > 0 && N > 0 && groups >= 1 && groups <= RO_FIN_MAXG && B % groups == 0,
               "node_linear_bn_fwd: bad sizes (1 <= groups <= 8, B divisible by groups)");
  IGCN_REQUIRE(keep == nullptr || D == 1, "node_linear_bn_fwd: fused dropout (keep) needs D == 1");
  IGCN_REQUIRE(!training || (int64_t)(B / groups) * D > 1,
               "node_linear_bn_fwd: need more than one value per node and group to train");
  hipStream_t st = (hipStream_t)stream;
  const int cpg = ro_cpg(B, groups);
  dim3 grid((unsigned)igcn_cdiv(N, RO_NL), groups * cpg);
#define CALL(FV, DV)                                                                                             \
  if (training) hipLaunchKernelGGL((k_nlbn_stats<FV, DV>), grid, dim3(RO_T), 0, st, B, N, groups, x, W, scratch); \
  hipLaunchKernelGGL((k_nlbn_finalize<FV, DV>), dim3((unsigned)igcn_cdiv(N, 64)), dim3(256), 0, st, B, N, groups, \
                     cpg, training, eps, momentum, x, W, scratch, running_mean, running_var, save_mean,           \
                     save_rstd);                                                                                  \
  ro_launch_apply<FV, DV>(grid, st, B, N, groups, x, W, gamma, beta, save_mean, save_rstd, keep, out)
  RO_DISPATCH(F, D, CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("node_linear_bn_fwd");
  return IGCN_OK;
}

int igcn_gemm_f32_batched_sum_impl(int64_t M, int64_t N, int64_t K, int batch, const float* A, int64_t sam,
                                   int64_t sak, int64_t a_batch, const float* B, int64_t sbn, int64_t sbk,
                                   int64_t b_batch, float* C, int64_t ldc, float* scratch, hipStream_t st);

extern "C" size_t igcn_node_linear_bn_bwd_scratch_floats(int B, int F, int N, int D, int groups) {
  const size_t cpg = ro_cpg(B, groups);
  const size_t stats = (size_t)groups * cpg * 2 * N + (size_t)groups * 2 * N;
  const size_t blocks = (size_t)igcn_cdiv(N, RO_NL) * groups * cpg;
  if (D * F <= 16) return stats + blocks * D * F + 68;
  // dpre rows + batched-GEMM slabs, or (row-coalesced path) one [D, F] partial per workgroup of 64/(D/4) nodes
  const size_t gemm_path = (size_t)B * N * D + (size_t)16 * B * D * F;
  const size_t quad_path = D % 4 == 0 ? (size_t)igcn_cdiv(N, 64 / (D / 4 > 64 ? 64 : D / 4)) * groups * cpg * D * F : 0;
  return stats + (gemm_path > quad_path ? gemm_path : quad_path) + 68;
}

template <int F, int D>
static int ro_bwd(dim3 grid, int cpg, hipStream_t st, int B, int N, int groups, int training, const float* x,
                  const float* W, const float* gamma, const float* beta, const float* save_mean,
                  const float* save_rstd, const float* dout, const float* keep, float* stats, float* dgg, float* aux,
                  float* dx, float* dW, float* dgb) {
  int rc;
  if constexpr (ro_quad_ok<F, D>()) {
    const int cq = ro_quad_chunks(grid.y, groups);           // chunks per group of the row-coalesced kernels
    dim3 gq((unsigned)igcn_cdiv(N, 64 / (D / 4)), cq * groups);
    hipLaunchKernelGGL((k_nlbn_bwd_stats_q<F, D>), gq, dim3(RO_T), 0, st, B, N, groups, x, W, gamma, beta, save_mean,
                       save_rstd, dout, stats);
    hipLaunchKernelGGL((k_nlbn_bwd_apply_q<F, D>), gq, dim3(RO_T), 0, st, B, N, groups, training, x, W, gamma, beta,
                       save_mean, save_rstd, dout, stats, cq, dx, aux);
    IGCN_CHECK_LAUNCH("node_linear_bn_bwd(q)");
    // dgamma | dbeta = the chunk partials [chunk][group][2N] summed over chunks AND groups: rows of 2N
    if ((rc = igcn_launch_reduce_rows_final(stats, (int64_t)cq * groups, 2 * (int64_t)N, 2 * N, dgb, st))) return rc;
    return igcn_launch_reduce_rows_final(aux, (int64_t)gq.x * gq.y, D * F, D * F, dW, st);
  } else {
    hipLaunchKernelGGL((k_nlbn_bwd_stats<F, D>), grid, dim3(RO_T), 0, st, B, N, groups, x, W, gamma, beta, save_mean,
                       save_rstd, dout, keep, stats);
    // thin rows (D = 1): a thread's own work is a handful of loads, so summing `cpg` chunk partials per thread costs
    // more than the launch it saves (measured: +17 us against -4.5) — the chunk sums keep their own launch here
    if ((rc = igcn_launch_reduce_rows(stats, cpg, (int64_t)groups * 2 * N, groups * 2 * N, dgg, 0, st))) return rc;
    hipLaunchKernelGGL((k_nlbn_bwd_apply<F, D>), grid, dim3(RO_T), 0, st, B, N, groups, training, x, W, gamma, beta,
                       save_mean, save_rstd, dout, keep, dgg, 1, aux, dx, aux);
    IGCN_CHECK_LAUNCH("node_linear_bn_bwd");
    if ((rc = igcn_launch_reduce_rows_final(dgg, groups, 2 * (int64_t)N, 2 * N, dgb, st))) return rc;
    if (D * F <= 16) return igcn_launch_reduce_rows_final(aux, (int64_t)grid.x * grid.y, D * F, D * F, dW, st);
    // dW[d,c] = sum_b sum_n dpre[b,n,d] * x[b,c,n]
    return igcn_gemm_f32_batched_sum_impl(D, F, N, B, aux, 1, D, (int64_t)N * D, x, N, 1, (int64_t)F * N, dW, F,
                                          aux + (size_t)B * N * D, st);
  }
}

extern "C" int igcn_node_linear_bn_bwd(int B, int F, int N, int D, int groups, int training, const float* x,
                                       const float* W, const float* gamma, const float* beta,
                                       const float* save_mean, const float* save_rstd, const float* dout,
                                       const float* keep, float* dx, float* dW,
                                       float* dgb /*[2,N]: dgamma, dbeta*/, float* scratch, void* stream) {
  IGCN_REQUIRE(B > 0 && N > 0 && groups >= 1 && B % groups == 0, "node_linear_bn_bwd: bad sizes");
  IGCN_REQUIRE(keep == nullptr || D == 1, "node_linear_bn_bwd: fused dropout (keep) needs D == 1");
  hipStream_t st = (hipStream_t)stream;
  const int cpg = ro_cpg(B, groups);
  dim3 grid((unsigned)igcn_cdiv(N, RO_NL), groups * cpg);
  float* stats = scratch;                                    // [cpg][groups][2][N]
  float* dgg = stats + (size_t)groups * cpg * 2 * N;         // [groups][2][N]
  float* aux = scratch + (((size_t)(dgg - scratch) + (size_t)groups * 2 * N + 3) & ~(size_t)3);   // 16-B aligned:
                                                             // wpartial (small) or dpre rows, then slabs (large)
#define CALL(FV, DV)                                                                                           \
  return ro_bwd<FV, DV>(grid, cpg, st, B, N, groups, training, x, W, gamma, beta, save_mean, save_rstd, dout,   \
                        keep, stats, dgg, aux, dx, dW, dgb)
  RO_DISPATCH(F, D, CALL)
#undef CALL
}

// =================================================================================================
// Two read-outs of the SAME input in paired launches (go_model.py:254-255: conc_for_attention [D1 = dim_snps_atten,
// row-coalesced kernels] and conc [D2 = 1, thread-per-node kernels, fused dropout] both read the encoder output): the
// kernel bodies above, selected by blockIdx.z, in one grid per pass — three launches forward and three backward instead
// of six and five, and the two small grids share the chip.
// =================================================================================================
struct RoSide {
  const float *W, *gamma, *beta, *keep, *mean_c, *rstd_c, *dout;
  float *running_mean, *running_var, *mean, *rstd, *partial, *out, *dgg, *aux, *dx;
  float eps, momentum;
};
__device__ __forceinline__ bool ro_in(const RoBlk& rb) { return rb.bx < rb.gx && rb.by < rb.gy; }
#define RO_SIDE_BLK(gxv, gyv) RoBlk{(int)blockIdx.x, (int)blockIdx.y, (int)(gxv), (int)(gyv)}

template <int F, int D1, int D2>
__global__ void __launch_bounds__(RO_T)
k_nlbn_pair_stats(int B, int N, int groups, const float* __restrict__ x, RoSide a, RoSide b) {
  const RoBlk rb = RO_BLK_OF_LAUNCH;                            // both sides use the same (node block, chunk) grid
  if (blockIdx.z == 0) nlbn_stats_body<F, D1>(rb, B, N, groups, x, a.W, a.partial);
  else nlbn_stats_body<F, D2>(rb, B, N, groups, x, b.W, b.partial);
}

template <int F, int D1, int D2>
__global__ void __launch_bounds__(256)
k_nlbn_pair_finalize(int B, int N, int groups, int cpg, int training, const float* __restrict__ x, RoSide a, RoSide b) {
  const RoBlk rb = RO_BLK_OF_LAUNCH;
  if (blockIdx.z == 0)
    nlbn_finalize_body<F, D1>(rb, B, N, groups, cpg, training, a.eps, a.momentum, x, a.W, a.partial, a.running_mean,
                              a.running_var, a.mean, a.rstd);
  else
    nlbn_finalize_body<F, D2>(rb, B, N, groups, cpg, training, b.eps, b.momentum, x, b.W, b.partial, b.running_mean,
                              b.running_var, b.mean, b.rstd);
}

// side a: row-coalesced kernels on grid (ax, ay); side b: thread-per-node kernels on grid (bx, by)
template <int F, int D1, int D2>
__global__ void __launch_bounds__(RO_T)
k_nlbn_pair_apply(int B, int N, int groups, int ax, int ay, int bx, int by, const float* __restrict__ x, RoSide a,
                  RoSide b) {
  if (blockIdx.z == 0) {
    const RoBlk rb = RO_SIDE_BLK(ax, ay);
    if (ro_in(rb)) nlbn_apply_q_body<F, D1>(rb, B, N, groups, x, a.W, a.gamma, a.beta, a.mean_c, a.rstd_c, a.out);
  } else {
    const RoBlk rb = RO_SIDE_BLK(bx, by);
    if (ro_in(rb)) nlbn_apply_body<F, D2>(rb, B, N, groups, x, b.W, b.gamma, b.beta, b.mean_c, b.rstd_c, b.keep, b.out);
  }
}

template <int F, int D1, int D2>
__global__ void __launch_bounds__(RO_T)
k_nlbn_pair_bwd_stats(int B, int N, int groups, int ax, int ay, int bx, int by, const float* __restrict__ x, RoSide a,
                      RoSide b) {
  if (blockIdx.z == 0) {
    const RoBlk rb = RO_SIDE_BLK(ax, ay);
    if (ro_in(rb))
      nlbn_bwd_stats_q_body<F, D1>(rb, B, N, groups, x, a.W, a.gamma, a.beta, a.mean_c, a.rstd_c, a.dout, a.partial);
  } else {
    const RoBlk rb = RO_SIDE_BLK(bx, by);
    if (ro_in(rb))
      nlbn_bwd_stats_body<F, D2>(rb, B, N, groups, x, b.W, b.gamma, b.beta, b.mean_c, b.rstd_c, b.dout, b.keep, b.partial);
  }
}

template <int F, int D1, int D2>
__global__ void __launch_bounds__(RO_T)
k_nlbn_pair_bwd_apply(int B, int N, int groups, int training, int ax, int ay, int a_chunks, int bx, int by,
                      const float* __restrict__ x, RoSide a, RoSide b) {
  if (blockIdx.z == 0) {
    const RoBlk rb = RO_SIDE_BLK(ax, ay);
    if (ro_in(rb))
      nlbn_bwd_apply_q_body<F, D1>(rb, B, N, groups, training, x, a.W, a.gamma, a.beta, a.mean_c, a.rstd_c, a.dout,
                                   a.partial, a_chunks, a.dx, a.aux);
  } else {
    const RoBlk rb = RO_SIDE_BLK(bx, by);
    if (ro_in(rb))
      nlbn_bwd_apply_body<F, D2>(rb, B, N, groups, training, x, b.W, b.gamma, b.beta, b.mean_c, b.rstd_c, b.dout, b.keep,
                                 b.dgg, 1, b.aux, b.dx, b.aux);
  }
}

// the (F, D1, D2) combinations of the paired launches: D1 on the row-coalesced kernels, D2 = 1
// (of the D values of the reference's sweep only 32 = 2 layers x 16 hidden meets the row-coalesced kernels' shape rule)
#define RO_PAIR_DISPATCH(F, D1, D2, CALL)                                  \
  if (F == 5 && D2 == 1 && D1 == 32) { CALL(5, 32, 1); }                   \
  else {                                                                   \
    igcn_set_error("node_linear_bn_pair: unsupported (F=%d, D1=%d, D2=%d)", F, D1, D2); \
    return IGCN_ERR_UNSUPPORTED;                                           \
  }

extern "C" int igcn_node_linear_bn_pair_supported(int F, int D1, int D2) {
  return F == 5 && D2 == 1 && D1 == 32;
}

// Arguments per side as igcn_node_linear_bn_fwd (scratch: igcn_node_linear_bn_scratch_floats each); keep2 [B,N] or NULL.
extern "C" int igcn_node_linear_bn_pair_fwd(int B, int F, int N, int groups, const float* x, int training,
                                            int D1, const float* W1, const float* gamma1, const float* beta1,
                                            float* running_mean1, float* running_var1, float momentum1, float eps1,
                                            float* out1, float* save_mean1, float* save_rstd1, float* scratch1,
                                            int D2, const float* W2, const float* gamma2, const float* beta2,
                                            float* running_mean2, float* running_var2, float momentum2, float eps2,
                                            const float* keep2, float* out2, float* save_mean2, float* save_rstd2,
                                            float* scratch2, void* stream) {
  IGCN_REQUIRE(B > 0 && N > 0 && groups >= 1 && groups <= RO_FIN_MAXG && B % groups == 0,
               "node_linear_bn_pair_fwd: bad sizes (1 <= groups <= 8, B divisible by groups)");
  IGCN_REQUIRE(!training || (int64_t)(B / groups) > 1, "node_linear_bn_pair_fwd: need more than one sample per group to train");
  hipStream_t st = (hipStream_t)stream;
  const int cpg = ro_cpg(B, groups);
  dim3 grid((unsigned)igcn_cdiv(N, RO_NL), groups * cpg);
  RoSide a = {}, b = {};
  a.W = W1; a.gamma = gamma1; a.beta = beta1; a.running_mean = running_mean1; a.running_var = running_var1;
  a.mean = save_mean1; a.rstd = save_rstd1; a.mean_c = save_mean1; a.rstd_c = save_rstd1; a.partial = scratch1;
  a.out = out1; a.eps = eps1; a.momentum = momentum1;
  b.W = W2; b.gamma = gamma2; b.beta = beta2; b.running_mean = running_mean2; b.running_var = running_var2;
  b.mean = save_mean2; b.rstd = save_rstd2; b.mean_c = save_mean2; b.rstd_c = save_rstd2; b.partial = scratch2;
  b.out = out2; b.keep = keep2; b.eps = eps2; b.momentum = momentum2;
#define CALL(FV, D1V, D2V)                                                                                        \
  {                                                                                                               \
    if (training)                                                                                                 \
      hipLaunchKernelGGL((k_nlbn_pair_stats<FV, D1V, D2V>), dim3(grid.x, grid.y, 2), dim3(RO_T), 0, st, B, N, groups, \
                         x, a, b);                                                                                \
    hipLaunchKernelGGL((k_nlbn_pair_finalize<FV, D1V, D2V>), dim3((unsigned)igcn_cdiv(N, 64), 1, 2), dim3(256), 0, st, \
                       B, N, groups, cpg, training, x, a, b);                                                     \
    const int ax = (int)igcn_cdiv(N, 64 / (D1V / 4)), ay = ro_quad_chunks(grid.y, groups) * groups;               \
    const unsigned mx = (unsigned)(ax > (int)grid.x ? ax : (int)grid.x), my = (unsigned)(ay > (int)grid.y ? ay : (int)grid.y); \
    hipLaunchKernelGGL((k_nlbn_pair_apply<FV, D1V, D2V>), dim3(mx, my, 2), dim3(RO_T), 0, st, B, N, groups, ax, ay,  \
                       (int)grid.x, (int)grid.y, x, a, b);                                                        \
  }
  RO_PAIR_DISPATCH(F, D1, D2, CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("node_linear_bn_pair_fwd");
  return IGCN_OK;
}

// Arguments per side as igcn_node_linear_bn_bwd (scratch: igcn_node_linear_bn_bwd_scratch_floats each).
extern "C" int igcn_node_linear_bn_pair_bwd(int B, int F, int N, int groups, int training, const float* x,
                                            int D1, const float* W1, const float* gamma1, const float* beta1,
                                            const float* save_mean1, const float* save_rstd1, const float* dout1,
                                            float* dx1, float* dW1, float* dgb1, float* scratch1,
                                            int D2, const float* W2, const float* gamma2, const float* beta2,
                                            const float* save_mean2, const float* save_rstd2, const float* dout2,
                                            const float* keep2, float* dx2, float* dW2, float* dgb2, float* scratch2,
                                            void* stream) {
  IGCN_REQUIRE(B > 0 && N > 0 && groups >= 1 && B % groups == 0, "node_linear_bn_pair_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const int cpg = ro_cpg(B, groups);
  dim3 grid((unsigned)igcn_cdiv(N, RO_NL), groups * cpg);
  auto split = [&](float* scratch, float*& stats, float*& dgg, float*& aux) {
    stats = scratch;
    dgg = stats + (size_t)groups * cpg * 2 * N;
    aux = scratch + (((size_t)(dgg - scratch) + (size_t)groups * 2 * N + 3) & ~(size_t)3);
  };
  RoSide a = {}, b = {};
  float *st1, *dg1, *ax1, *st2, *dg2, *ax2;
  split(scratch1, st1, dg1, ax1);
  split(scratch2, st2, dg2, ax2);
  a.W = W1; a.gamma = gamma1; a.beta = beta1; a.mean_c = save_mean1; a.rstd_c = save_rstd1; a.dout = dout1;
  a.partial = st1; a.dgg = dg1; a.aux = ax1; a.dx = dx1;
  b.W = W2; b.gamma = gamma2; b.beta = beta2; b.mean_c = save_mean2; b.rstd_c = save_rstd2; b.dout = dout2;
  b.keep = keep2; b.partial = st2; b.dgg = dg2; b.aux = ax2; b.dx = dx2;
  int rc;
#define CALL(FV, D1V, D2V)                                                                                        \
  {                                                                                                               \
    const int cq = ro_quad_chunks(grid.y, groups);                                                                \
    const int ax = (int)igcn_cdiv(N, 64 / (D1V / 4)), ay = cq * groups;                                           \
    const unsigned mx = (unsigned)(ax > (int)grid.x ? ax : (int)grid.x), my = (unsigned)(ay > (int)grid.y ? ay : (int)grid.y); \
    hipLaunchKernelGGL((k_nlbn_pair_bwd_stats<FV, D1V, D2V>), dim3(mx, my, 2), dim3(RO_T), 0, st, B, N, groups, ax, ay, \
                       (int)grid.x, (int)grid.y, x, a, b);                                                        \
    if ((rc = igcn_launch_reduce_rows(st2, cpg, (int64_t)groups * 2 * N, groups * 2 * N, dg2, 0, st))) return rc;  \
    hipLaunchKernelGGL((k_nlbn_pair_bwd_apply<FV, D1V, D2V>), dim3(mx, my, 2), dim3(RO_T), 0, st, B, N, groups,    \
                       training, ax, ay, cq, (int)grid.x, (int)grid.y, x, a, b);                                  \
    IGCN_CHECK_LAUNCH("node_linear_bn_pair_bwd");                                                                 \
    if ((rc = igcn_launch_reduce_rows_final(st1, (int64_t)cq * groups, 2 * (int64_t)N, 2 * N, dgb1, st))) return rc; \
    if ((rc = igcn_launch_reduce_rows_final(ax1, (int64_t)ax * ay, D1V * FV, D1V * FV, dW1, st))) return rc;       \
    if ((rc = igcn_launch_reduce_rows_final(dg2, groups, 2 * (int64_t)N, 2 * N, dgb2, st))) return rc;             \
    return igcn_launch_reduce_rows_final(ax2, (int64_t)grid.x * grid.y, D2V * FV, D2V * FV, dW2, st);              \
  }
  RO_PAIR_DISPATCH(F, D1, D2, CALL)
#undef CALL
}

// =================================================================================================
// BatchNorm1d(C) (+ optional ReLU) on a 2-D input [B, C] with grouped statistics — the latent MLP of the GO
// network (go_model.py:138-146).  Tiny tensors ([512,32]): one workgroup per channel, groups handled in turn.
// =================================================================================================
__global__ void __launch_bounds__(256)
k_bn1d_fwd(int B, int C, int groups, int training, float momentum, float eps, int relu,
           const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
           const float* __restrict__ keep, float* __restrict__ running_mean, float* __restrict__ running_var,
           float* __restrict__ y,
           float* __restrict__ save_mean, float* __restrict__ save_rstd) {
  __shared__ float red[16];
  const int c = blockIdx.x, bg = B / groups;
  float rm = running_mean[c], rv = running_var[c];
  const float ga = gamma[c], be = beta[c];
  for (int g = 0; g < groups; ++g) {
    const float* xg = x + (int64_t)g * bg * C + c;
    float mean, var;
    if (training) {
      float s = 0.f;
      for (int b = threadIdx.x; b < bg; b += 256) s += xg[(int64_t)b * C];
      mean = block_sum_all(s, red) / (float)bg;
      float v = 0.f;
      for (int b = threadIdx.x; b < bg; b += 256) {
        const float d = xg[(int64_t)b * C] - mean;
        v += d * d;
      }
      var = block_sum_all(v, red) / (float)bg;
      rm = (1.f - momentum) * rm + momentum * mean;
      rv = (1.f - momentum) * rv + momentum * var * ((float)bg / (float)(bg - 1));
    } else {
      mean = rm;
      var = rv;
    }
    const float rstd = 1.0f / sqrtf(var + eps);
    if (threadIdx.x == 0) {
      save_mean[g * C + c] = mean;
      save_rstd[g * C + c] = rstd;
    }
    float* yg = y + (int64_t)g * bg * C + c;
    const float* kg = keep ? keep + (int64_t)g * bg * C + c : nullptr;
    for (int b = threadIdx.x; b < bg; b += 256) {
      float t = (xg[(int64_t)b * C] - mean) * rstd * ga + be;
      t = relu ? fmaxf(t, 0.f) : t;
      yg[(int64_t)b * C] = kg ? t * kg[(int64_t)b * C] : t;          // fused dropout of the activation
    }
  }
  if (training && threadIdx.x == 0) {
    running_mean[c] = rm;
    running_var[c] = rv;
  }
}

__global__ void __launch_bounds__(256)
k_bn1d_bwd(int B, int C, int groups, int training, int relu, const float* __restrict__ x,
           const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ save_mean,
           const float* __restrict__ save_rstd, const float* __restrict__ dy, const float* __restrict__ keep,
           float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[16];
  const int c = blockIdx.x, bg = B / groups;
  const float ga = gamma[c], be = beta[c];
  float dg_tot = 0.f, db_tot = 0.f;
  for (int g = 0; g < groups; ++g) {
    const float mean = save_mean[g * C + c], rstd = save_rstd[g * C + c];
    const float* xg = x + (int64_t)g * bg * C + c;
    const float* dyg = dy + (int64_t)g * bg * C + c;
    const float* kg = keep ? keep + (int64_t)g * bg * C + c : nullptr;
    float s1 = 0.f, s2 = 0.f;
    for (int b = threadIdx.x; b < bg; b += 256) {
      const float xh = (xg[(int64_t)b * C] - mean) * rstd;
      const float up = kg ? dyg[(int64_t)b * C] * kg[(int64_t)b * C] : dyg[(int64_t)b * C];
      const float d = (!relu || xh * ga + be > 0.f) ? up : 0.f;
      s1 += d;
      s2 += d * xh;
    }
    s1 = block_sum_all(s1, red);
    s2 = block_sum_all(s2, red);
    dg_tot += s2;
    db_tot += s1;
    const float m1 = training ? s1 / (float)bg : 0.f, m2 = training ? s2 / (float)bg : 0.f;
    float* dxg = dx + (int64_t)g * bg * C + c;
    for (int b = threadIdx.x; b < bg; b += 256) {
      const float xh = (xg[(int64_t)b * C] - mean) * rstd;
      const float up = kg ? dyg[(int64_t)b * C] * kg[(int64_t)b * C] : dyg[(int64_t)b * C];
      const float d = (!relu || xh * ga + be > 0.f) ? up : 0.f;
      dxg[(int64_t)b * C] = ga * rstd * (d - m1 - xh * m2);
    }
  }
  if (threadIdx.x == 0) {
    dgamma[c] = dg_tot;
    dbeta[c] = db_tot;
  }
}

// The same two kernels for the shapes the model has (<= 2 groups of <= 1024 samples): a channel's values are read ONCE
// into registers — both groups' loads in flight together — and the statistics of the two groups share their barrier
// pairs.  [The general form walks the column three times per group, one group after the other: eight dependent
// barrier-separated trips over 64 KB, 6-7 us on 32 workgroups.]  Same per-thread order and reduction tree: same bits.
#define BN1_VP 4
// PRODUCER (template): where the column comes from.  0 = x as given.  1 = the sum of `n_slabs` split-K slabs
// [n_slabs][B*C] of the product in front (igcn_gemm_f32 with act | 0x200 leaves them un-summed): the slab-sum launch
// between the product and this kernel disappears, the sum is taken in slab order (k_gemm_splitk_reduce's) and written to
// `x_out` for the backward — 8.6 us against 5.0 + 5.1 for the two launches (round 5).  [A third form that computed a
// narrow product itself — the latent MLP's 32 -> 32 layer, every channel's workgroup reading all of h — was measured at
// 12.0 us against 4.7 + 4.9 for product + BatchNorm and removed: 32 workgroups cannot read 64 KB each, a row per lane,
// in less than the 8-workgroup product and its launch take.]
template <int PRODUCER>
__global__ void __launch_bounds__(256)
k_bn1d_fwd_reg(int B, int C, int groups, int training, float momentum, float eps, int relu,
               const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
               const float* __restrict__ keep, float* __restrict__ running_mean, float* __restrict__ running_var,
               float* __restrict__ y, float* __restrict__ save_mean, float* __restrict__ save_rstd, int n_slabs,
               float* __restrict__ x_out) {
  __shared__ float red[32];
  const int c = blockIdx.x, bg = B / groups;
  float rm = running_mean[c], rv = running_var[c];
  const float ga = gamma[c], be = beta[c];
  float xv[2][BN1_VP], kv[2][BN1_VP];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int j = 0; j < BN1_VP; ++j) {
      const int b = threadIdx.x + 256 * j;
      const bool ok = g < groups && b < bg;
      const int64_t o = ((int64_t)g * bg + b) * C + c;
      if constexpr (PRODUCER == 1) {
        float t = 0.f;
        if (ok) {
          for (int z0 = 0; z0 < n_slabs; z0 += 8) {          // slab order, eight loads in flight (= slab_sum, gemm.hip)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const float xs = x[(int64_t)(z0 + u < n_slabs ? z0 + u : 0) * B * C + o];
              v[u] = z0 + u < n_slabs ? xs : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
          }
          x_out[o] = t;
        }
        xv[g][j] = t;
      } else {
        xv[g][j] = ok ? x[o] : 0.f;
      }
      kv[g][j] = (ok && keep) ? keep[o] : 1.f;
    }
  float mean[2] = {rm, rm}, var[2] = {rv, rv};
  if (training) {
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < BN1_VP; ++j)
        if (threadIdx.x + 256 * j < bg) s[g] += xv[g][j];
    block_sum_all2(s[0], s[1], red);
    mean[0] = s[0] / (float)bg;
    mean[1] = s[1] / (float)bg;
    float v[2] = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < BN1_VP; ++j)
        if (threadIdx.x + 256 * j < bg) {
          const float d = xv[g][j] - mean[g];
          v[g] += d * d;
        }
    block_sum_all2(v[0], v[1], red);
    var[0] = v[0] / (float)bg;
    var[1] = v[1] / (float)bg;
    for (int g = 0; g < groups; ++g) {                 // running statistics: group after group
      rm = (1.f - momentum) * rm + momentum * mean[g];
      rv = (1.f - momentum) * rv + momentum * var[g] * ((float)bg / (float)(bg - 1));
    }
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    if (g >= groups) break;
    const float rstd = 1.0f / sqrtf(var[g] + eps);
    if (threadIdx.x == 0) {
      save_mean[g * C + c] = mean[g];
      save_rstd[g * C + c] = rstd;
    }
#pragma unroll
    for (int j = 0; j < BN1_VP; ++j) {
      const int b = threadIdx.x + 256 * j;
      if (b < bg) {
        float t = (xv[g][j] - mean[g]) * rstd * ga + be;
        t = relu ? fmaxf(t, 0.f) : t;
        y[((int64_t)g * bg + b) * C + c] = keep ? t * kv[g][j] : t;
      }
    }
  }
  if (training && threadIdx.x == 0) {
    running_mean[c] = rm;
    running_var[c] = rv;
  }
}

__global__ void __launch_bounds__(256)
k_bn1d_bwd_reg(int B, int C, int groups, int training, int relu, const float* __restrict__ x,
               const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ save_mean,
               const float* __restrict__ save_rstd, const float* __restrict__ dy, const float* __restrict__ keep,
               float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[32];
  const int c = blockIdx.x, bg = B / groups;
  const float ga = gamma[c], be = beta[c];
  float xh[2][BN1_VP], dv[2][BN1_VP], mean[2], rstd[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    mean[g] = g < groups ? save_mean[g * C + c] : 0.f;
    rstd[g] = g < groups ? save_rstd[g * C + c] : 0.f;
  }
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int j = 0; j < BN1_VP; ++j) {
      const int b = threadIdx.x + 256 * j;
      const bool ok = g < groups && b < bg;
      const int64_t o = ((int64_t)g * bg + b) * C + c;
      const float xr = ok ? x[o] : 0.f, up0 = ok ? dy[o] : 0.f, k = (ok && keep) ? keep[o] : 1.f;
      xh[g][j] = (xr - mean[g]) * rstd[g];
      const float up = keep ? up0 * k : up0;
      dv[g][j] = (ok && (!relu || xh[g][j] * ga + be > 0.f)) ? up : 0.f;
    }
  float dg_tot = 0.f, db_tot = 0.f, m1[2], m2[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < BN1_VP; ++j)
      if (threadIdx.x + 256 * j < bg) {
        s1 += dv[g][j];
        s2 += dv[g][j] * xh[g][j];
      }
    block_sum_all2(s1, s2, red);
    if (g < groups) {
      dg_tot += s2;
      db_tot += s1;
    }
    m1[g] = training ? s1 / (float)bg : 0.f;
    m2[g] = training ? s2 / (float)bg : 0.f;
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    if (g >= groups) break;
#pragma unroll
    for (int j = 0; j < BN1_VP; ++j) {
      const int b = threadIdx.x + 256 * j;
      if (b < bg) dx[((int64_t)g * bg + b) * C + c] = ga * rstd[g] * (dv[g][j] - m1[g] - xh[g][j] * m2[g]);
    }
  }
  if (threadIdx.x == 0) {
    dgamma[c] = dg_tot;
    dbeta[c] = db_tot;
  }
}

extern "C" int igcn_bn1d_fwd(int B, int C, int groups, const float* x, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, int training, float momentum, float eps,
                             int relu, const float* keep /*[B,C] dropout factors or NULL*/, float* y,
                             float* save_mean, float* save_rstd, void* stream) {
  IGCN_REQUIRE(B > 0 && C > 0 && groups >= 1 && B % groups == 0 && (!training || B / groups > 1),
               "bn1d_fwd: bad sizes");
  if (groups <= 2 && B / groups <= 256 * BN1_VP)
    hipLaunchKernelGGL(k_bn1d_fwd_reg<0>, dim3(C), dim3(256), 0, (hipStream_t)stream, B, C, groups, training, momentum,
                       eps, relu, x, gamma, beta, keep, running_mean, running_var, y, save_mean, save_rstd, 0,
                       (float*)nullptr);
  else
    hipLaunchKernelGGL(k_bn1d_fwd, dim3(C), dim3(256), 0, (hipStream_t)stream, B, C, groups, training, momentum, eps,
                       relu, x, gamma, beta, keep, running_mean, running_var, y, save_mean, save_rstd);
  IGCN_CHECK_LAUNCH("bn1d_fwd");
  return IGCN_OK;
}

// BatchNorm1d(C) (+ ReLU, + dropout factors) of a column whose split-K slabs the product in front left un-summed
// (igcn_gemm_f32 with act | 0x200): `slabs` [n_slabs][B, C]; x_out [B, C] receives the sum (the backward's operand).
// Shapes: groups <= 2, B / groups <= 1024 (igcn_bn1d_fwd_supported).
extern "C" int igcn_bn1d_fwd_supported(int B, int groups) {
  return groups >= 1 && groups <= 2 && B > 0 && B % groups == 0 && B / groups <= 256 * BN1_VP;
}
extern "C" int igcn_bn1d_fwd_slabs(int B, int C, int groups, const float* slabs, int n_slabs, float* x_out,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   int training, float momentum, float eps, int relu, const float* keep, float* y,
                                   float* save_mean, float* save_rstd, void* stream) {
  IGCN_REQUIRE(C > 0 && igcn_bn1d_fwd_supported(B, groups) && (!training || B / groups > 1) && slabs && x_out &&
                   n_slabs >= 1,
               "bn1d_fwd_slabs: unsupported shape (B=%d groups=%d n_slabs=%d)", B, groups, n_slabs);
  hipLaunchKernelGGL(k_bn1d_fwd_reg<1>, dim3(C), dim3(256), 0, (hipStream_t)stream, B, C, groups, training, momentum,
                     eps, relu, slabs, gamma, beta, keep, running_mean, running_var, y, save_mean, save_rstd, n_slabs,
                     x_out);
  IGCN_CHECK_LAUNCH("bn1d_fwd_slabs");
  return IGCN_OK;
}

extern "C" int igcn_bn1d_bwd(int B, int C, int groups, int training, int relu, const float* x, const float* gamma,
                             const float* beta, const float* save_mean, const float* save_rstd, const float* dy,
                             const float* keep, float* dx, float* dgamma, float* dbeta, void* stream) {
  IGCN_REQUIRE(B > 0 && C > 0 && groups >= 1 && B % groups == 0, "bn1d_bwd: bad sizes");
  if (groups <= 2 && B / groups <= 256 * BN1_VP)
    hipLaunchKernelGGL(k_bn1d_bwd_reg, dim3(C), dim3(256), 0, (hipStream_t)stream, B, C, groups, training, relu, x,
                       gamma, beta, save_mean, save_rstd, dy, keep, dx, dgamma, dbeta);
  else
    hipLaunchKernelGGL(k_bn1d_bwd, dim3(C), dim3(256), 0, (hipStream_t)stream, B, C, groups, training, relu, x, gamma,
                       beta, save_mean, save_rstd, dy, keep, dx, dgamma, dbeta);
  IGCN_CHECK_LAUNCH("bn1d_bwd");
  return IGCN_OK;
}
