// Fused cross-attention fusion block — kernel/sgcn_img_snp.py:239-242:
//   out = relu( MultiheadAttention(D, H, batch_first)(query = batch_x [B,Lq,D], key = value = atten_out [B,Lk,D]) )
// One workgroup (512 threads) per sample; everything between the two inputs and the ReLU'd output stays on chip:
// per head the K/V/Q projections are computed straight into LDS (K, V [B,Lk,D] never touch HBM), scores/softmax/PV
// run with lanes = query rows (K/V rows are LDS broadcasts), the out-projection reads the head outputs from LDS.
// The backward recomputes the projections and the probabilities from the saved log-sum-exp (flash style), forms
// dQ with lanes = query rows and dK/dV with lanes = keys (no atomics), and emits per-sample partial parameter
// gradients that a row reduction sums over the batch.
#include "common.h"

#define XA_T 512

struct XaDims {
  int H, Lq, Lk;
};

template <int HD, int H>
struct XaSmem {
  static constexpr int D = HD * H, ldk = HD + 1, ldd = D + 1;
  static __host__ __device__ int rp(int Lq) { return (Lq + 63) / 64 * 64; }
  // forward: Ks, Vs, Qs, Os, Wo, merge[G][Lq][HD+2]
  static __host__ __device__ size_t fwd_floats(int Lq, int Lk) {
    const int G = XA_T / rp(Lq);
    return wst() + (size_t)2 * Lk * ldk + (size_t)Lq * ldk + (size_t)Lq * ldd + (size_t)D * ldd +
           (size_t)G * Lq * (HD + 2);
  }
  // per-head stage of the q/k/v projection rows: [3][HD][D] weights + [3][HD] biases (16-byte multiple)
  static __host__ __device__ size_t wst() { return ((size_t)3 * HD * D + 3 * HD + 3) / 4 * 4; }
  // backward: Ks, Vs, Qs, dOs, Os(->dXq), scratch = max(dYs, merge), lse, delta, dQs
  static __host__ __device__ size_t bwd_floats(int Lq, int Lk) {
    const int G = XA_T / rp(Lq);
    const size_t mrg = (size_t)G * Lq * HD, dys = (size_t)Lq * ldd;
    return wst() + (size_t)2 * Lk * ldk + (size_t)Lq * ldk + 2 * (size_t)Lq * ldd + (mrg > dys ? mrg : dys) +
           2 * (size_t)Lq + (size_t)Lq * ldk;
  }
};

// stage head h's projection rows of in_proj_weight/bias into LDS: Ws[which][c][d], bs[which][c]
template <int HD, int D>
__device__ __forceinline__ void xa_stage_w(const float* __restrict__ w_in, const float* __restrict__ b_in, int h,
                                           float* __restrict__ Ws) {
  float* bs = Ws + 3 * HD * D;
  for (int t = threadIdx.x; t < 3 * HD * D; t += XA_T) {
    const int which = t / (HD * D), rem = t % (HD * D);
    Ws[t] = w_in[(which * D + h * HD) * D + rem];
  }
  for (int t = threadIdx.x; t < 3 * HD; t += XA_T) bs[t] = b_in[(t / HD) * D + h * HD + (t % HD)];
}

// row j of `src` [L,D] x staged weight block `which` (0 q, 1 k, 2 v) -> dst[j][0:HD]; weights are LDS broadcasts
template <int HD, int D>
__device__ __forceinline__ void xa_project_row(const float* __restrict__ src, int j, const float* __restrict__ Ws,
                                               int which, float* __restrict__ dst) {
  constexpr int ldk = HD + 1;
  const float* w = Ws + which * HD * D;
  const float* bs = Ws + 3 * HD * D + which * HD;
  float m[D];
#pragma unroll
  for (int d = 0; d < D; ++d) m[d] = src[(int64_t)j * D + d];
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    float acc = bs[c];
#pragma unroll
    for (int d = 0; d < D; ++d) acc += m[d] * w[c * D + d];
    dst[j * ldk + c] = acc;
  }
}

template <int HD, int D>
__device__ __forceinline__ void xa_project_all(const float* __restrict__ xqb, int Lq, const float* __restrict__ memb,
                                               int Lk, const float* __restrict__ Ws, float* __restrict__ Qs,
                                               float* __restrict__ Ks, float* __restrict__ Vs) {
  for (int t = threadIdx.x; t < 2 * Lk + Lq; t += XA_T) {
    if (t < Lk) xa_project_row<HD, D>(memb, t, Ws, 1, Ks);
    else if (t < 2 * Lk) xa_project_row<HD, D>(memb, t - Lk, Ws, 2, Vs);
    else xa_project_row<HD, D>(xqb, t - 2 * Lk, Ws, 0, Qs);
  }
}

template <int HD, int H>
__global__ void __launch_bounds__(XA_T)
k_xattn_fwd(int Lq, int Lk, const float* __restrict__ xq, const float* __restrict__ mem,
            const float* __restrict__ w_in, const float* __restrict__ b_in, const float* __restrict__ w_out,
            const float* __restrict__ b_out, float* __restrict__ out, float* __restrict__ o_save,
            float* __restrict__ lse) {
  constexpr int D = HD * H, ldk = HD + 1, ldd = D + 1;
  extern __shared__ float smem[];
  const int rp = (Lq + 63) / 64 * 64, G = XA_T / rp;
  float* Ws = smem;
  float* Ks = Ws + XaSmem<HD, H>::wst();
  float* Vs = Ks + (size_t)Lk * ldk;
  float* Qs = Vs + (size_t)Lk * ldk;
  float* Os = Qs + (size_t)Lq * ldk;
  float* Wo = Os + (size_t)Lq * ldd;
  float* Mg = Wo + (size_t)D * ldd;          // [G][Lq][HD+2] : (m, l, o[HD])
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xqb = xq + (int64_t)b * Lq * D;
  const float* memb = mem + (int64_t)b * Lk * D;
  const float scale = rsqrtf((float)HD);
  for (int t = tid; t < D * D; t += XA_T) Wo[(t / D) * ldd + (t % D)] = w_out[t];

  const int i = tid % rp, g = tid / rp;
  const bool active = i < Lq && g < G;
  const int kper = (Lk + G - 1) / G;
  const int j0 = g * kper, j1 = min(Lk, j0 + kper);

  for (int h = 0; h < H; ++h) {
    __syncthreads();                          // previous head's readers are done with Ws/Ks/Vs/Qs/Mg
    xa_stage_w<HD, D>(w_in, b_in, h, Ws);
    __syncthreads();
    xa_project_all<HD, D>(xqb, Lq, memb, Lk, Ws, Qs, Ks, Vs);
    __syncthreads();
    if (active) {
      float q[HD], o[HD];
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        q[c] = Qs[i * ldk + c] * scale;
        o[c] = 0.f;
      }
      float m = -INFINITY;
      for (int j = j0; j < j1; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) s += q[c] * Ks[j * ldk + c];
        m = fmaxf(m, s);
      }
      float l = 0.f;
      for (int j = j0; j < j1; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) s += q[c] * Ks[j * ldk + c];
        const float p = __expf(s - m);
        l += p;
#pragma unroll
        for (int c = 0; c < HD; ++c) o[c] += p * Vs[j * ldk + c];
      }
      float* mg = Mg + ((size_t)g * Lq + i) * (HD + 2);
      mg[0] = m;
      mg[1] = l;
#pragma unroll
      for (int c = 0; c < HD; ++c) mg[2 + c] = o[c];
    }
    __syncthreads();
    if (i < Lq && g == 0) {                   // merge the G key groups of row i
      float M = -INFINITY;
      for (int gg = 0; gg < G; ++gg) M = fmaxf(M, Mg[((size_t)gg * Lq + i) * (HD + 2)]);
      float L = 0.f, o[HD];
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] = 0.f;
      for (int gg = 0; gg < G; ++gg) {
        const float* mg = Mg + ((size_t)gg * Lq + i) * (HD + 2);
        if (mg[1] > 0.f) {
          const float f = __expf(mg[0] - M);
          L += mg[1] * f;
#pragma unroll
          for (int c = 0; c < HD; ++c) o[c] += mg[2 + c] * f;
        }
      }
      const float inv = 1.f / L;
#pragma unroll
      for (int c = 0; c < HD; ++c) Os[i * ldd + h * HD + c] = o[c] * inv;
      lse[((int64_t)b * H + h) * Lq + i] = M + __logf(L);
    }
  }
  __syncthreads();
  for (int t = tid; t < Lq * D; t += XA_T) {
    const int r = t / D, oc = t % D;
    float y = b_out[oc];
#pragma unroll
    for (int c = 0; c < D; ++c) y += Os[r * ldd + c] * Wo[oc * ldd + c];
    out[((int64_t)b * Lq + r) * D + oc] = fmaxf(y, 0.f);
    o_save[((int64_t)b * Lq + r) * D + oc] = Os[r * ldd + oc];
  }
}

// partial layout per sample (== dparams layout): dW_in [3D,D], db_in [3D], dW_out [D,D], db_out [D]
template <int HD, int H>
__global__ void __launch_bounds__(XA_T)
k_xattn_bwd(int Lq, int Lk, const float* __restrict__ xq, const float* __restrict__ mem,
            const float* __restrict__ w_in, const float* __restrict__ b_in, const float* __restrict__ w_out,
            const float* __restrict__ out, const float* __restrict__ o_save, const float* __restrict__ lse,
            const float* __restrict__ dout, float* __restrict__ dxq, float* __restrict__ dmem,
            float* __restrict__ partial) {
  constexpr int D = HD * H, ldk = HD + 1, ldd = D + 1;
  constexpr int PW = 3 * D * D + 3 * D + D * D + D;
  extern __shared__ float smem[];
  const int rp = (Lq + 63) / 64 * 64, G = XA_T / rp;
  const size_t mrg = (size_t)G * Lq * HD, dys = (size_t)Lq * ldd;
  float* Ws = smem;
  float* Ks = Ws + XaSmem<HD, H>::wst();
  float* Vs = Ks + (size_t)Lk * ldk;
  float* Qs = Vs + (size_t)Lk * ldk;
  float* dOs = Qs + (size_t)Lq * ldk;
  float* Os = dOs + (size_t)Lq * ldd;         // O, later the dXq accumulator
  float* Sc = Os + (size_t)Lq * ldd;          // dY first, then the dQ merge buffer [G][Lq][HD]
  float* lse_s = Sc + (mrg > dys ? mrg : dys);
  float* del_s = lse_s + Lq;
  float* dQs = del_s + Lq;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xqb = xq + (int64_t)b * Lq * D;
  const float* memb = mem + (int64_t)b * Lk * D;
  float* pb = partial + (int64_t)b * PW;
  const float scale = rsqrtf((float)HD);

  // ---- dY = dout * [out > 0] ; O ----
  for (int t = tid; t < Lq * D; t += XA_T) {
    const int r = t / D, c = t % D;
    const int64_t gi = ((int64_t)b * Lq + r) * D + c;
    Sc[r * ldd + c] = out[gi] > 0.f ? dout[gi] : 0.f;
    Os[r * ldd + c] = o_save[gi];
  }
  __syncthreads();
  // dW_out[o][c] = sum_r dY[r][o] O[r][c] ; db_out[o] = sum_r dY[r][o] ; dO[r][c] = sum_o dY[r][o] W_out[o][c]
  for (int t = tid; t < D * D + D; t += XA_T) {
    float acc = 0.f;
    if (t < D * D) {
      const int o = t / D, c = t % D;
      for (int r = 0; r < Lq; ++r) acc += Sc[r * ldd + o] * Os[r * ldd + c];
      pb[3 * D * D + 3 * D + t] = acc;
    } else {
      const int o = t - D * D;
      for (int r = 0; r < Lq; ++r) acc += Sc[r * ldd + o];
      pb[3 * D * D + 3 * D + D * D + o] = acc;
    }
  }
  for (int t = tid; t < Lq * D; t += XA_T) {
    const int r = t / D, c = t % D;
    float acc = 0.f;
#pragma unroll
    for (int o = 0; o < D; ++o) acc += Sc[r * ldd + o] * w_out[o * D + c];
    dOs[r * ldd + c] = acc;
  }
  __syncthreads();                            // dOs complete; Sc (dY) free; Os still O

  const int i = tid % rp, g = tid / rp;
  const bool active = i < Lq && g < G;
  const int kper = (Lk + G - 1) / G;
  const int j0 = g * kper, j1 = min(Lk, j0 + kper);
  // per-key accumulator of dmem over heads (keys handled: tid and tid + XA_T ...): Lk <= 2*XA_T supported
  float dm0[D], dm1[D];
#pragma unroll
  for (int d = 0; d < D; ++d) dm0[d] = dm1[d] = 0.f;

  for (int h = 0; h < H; ++h) {
    if (h > 0) __syncthreads();
    // delta_i = sum_{c in head} dO[i][c] O[i][c]   (O must still be intact: the dXq accumulator takes Os over
    // only after the LAST head's delta -> keep O and dXq separate by accumulating dXq into dxq (global) instead)
    for (int r = tid; r < Lq; r += XA_T) {
      float dl = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) dl += dOs[r * ldd + h * HD + c] * Os[r * ldd + h * HD + c];
      del_s[r] = dl;
      lse_s[r] = lse[((int64_t)b * H + h) * Lq + r];
    }
    xa_stage_w<HD, D>(w_in, b_in, h, Ws);
    __syncthreads();
    xa_project_all<HD, D>(xqb, Lq, memb, Lk, Ws, Qs, Ks, Vs);
    __syncthreads();
    // ---- dQ: lanes = query rows, each key group accumulates a partial ----
    if (active) {
      float q[HD], dov[HD], dq[HD];
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        q[c] = Qs[i * ldk + c] * scale;
        dov[c] = dOs[i * ldd + h * HD + c];
        dq[c] = 0.f;
      }
      const float ls = lse_s[i], dl = del_s[i];
      for (int j = j0; j < j1; ++j) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) {
          s += q[c] * Ks[j * ldk + c];
          dp += dov[c] * Vs[j * ldk + c];
        }
        const float ds = __expf(s - ls) * (dp - dl) * scale;
#pragma unroll
        for (int c = 0; c < HD; ++c) dq[c] += ds * Ks[j * ldk + c];
      }
#pragma unroll
      for (int c = 0; c < HD; ++c) Sc[((size_t)g * Lq + i) * HD + c] = dq[c];
    }
    __syncthreads();
    for (int t = tid; t < Lq * HD; t += XA_T) {
      float acc = 0.f;
      for (int gg = 0; gg < G; ++gg) acc += Sc[(size_t)gg * Lq * HD + t];
      dQs[(t / HD) * ldk + (t % HD)] = acc;
    }
    // ---- dK, dV: lanes = keys ----
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int j = tid + rep * XA_T;
      if (j < Lk) {
        float kj[HD], vj[HD], dk[HD], dv[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) {
          kj[c] = Ks[j * ldk + c];
          vj[c] = Vs[j * ldk + c];
          dk[c] = dv[c] = 0.f;
        }
        for (int r = 0; r < Lq; ++r) {
          float s = 0.f, dp = 0.f;
#pragma unroll
          for (int c = 0; c < HD; ++c) {
            s += Qs[r * ldk + c] * kj[c];
            dp += dOs[r * ldd + h * HD + c] * vj[c];
          }
          const float p = __expf(s * scale - lse_s[r]);
          const float ds = p * (dp - del_s[r]) * scale;
#pragma unroll
          for (int c = 0; c < HD; ++c) {
            dk[c] += ds * Qs[r * ldk + c];
            dv[c] += p * dOs[r * ldd + h * HD + c];
          }
        }
        // own rows of Ks/Vs now hold dK/dV (only this thread reads row j in this phase)
#pragma unroll
        for (int c = 0; c < HD; ++c) {
          Ks[j * ldk + c] = dk[c];
          Vs[j * ldk + c] = dv[c];
        }
        // dmem[j][d] += dK_j Wk_h + dV_j Wv_h
#pragma unroll
        for (int d = 0; d < D; ++d) {
          float acc = 0.f;
#pragma unroll
          for (int c = 0; c < HD; ++c)
            acc += dk[c] * Ws[(HD + c) * D + d] + dv[c] * Ws[(2 * HD + c) * D + d];
          if (rep == 0) dm0[d] += acc; else dm1[d] += acc;
        }
      }
    }
    __syncthreads();                          // dQs, dK (Ks), dV (Vs) complete
    // ---- partial parameter gradients of this head ----
    for (int t = tid; t < HD * D + 3 * HD; t += XA_T) {
      if (t < HD * D) {
        const int c = t / D, d = t % D;
        float ak = 0.f, av = 0.f, aq = 0.f;
        for (int j = 0; j < Lk; ++j) {
          const float mv = memb[(int64_t)j * D + d];
          ak += Ks[j * ldk + c] * mv;
          av += Vs[j * ldk + c] * mv;
        }
        for (int r = 0; r < Lq; ++r) aq += dQs[r * ldk + c] * xqb[(int64_t)r * D + d];
        pb[(0 * D + h * HD + c) * D + d] = aq;
        pb[(1 * D + h * HD + c) * D + d] = ak;
        pb[(2 * D + h * HD + c) * D + d] = av;
      } else {
        const int u = t - HD * D, which = u / HD, c = u % HD;
        float acc = 0.f;
        if (which == 0) { for (int r = 0; r < Lq; ++r) acc += dQs[r * ldk + c]; }
        else if (which == 1) { for (int j = 0; j < Lk; ++j) acc += Ks[j * ldk + c]; }
        else { for (int j = 0; j < Lk; ++j) acc += Vs[j * ldk + c]; }
        pb[3 * D * D + which * D + h * HD + c] = acc;
      }
    }
    // dxq[r][d] (+)= sum_c dQ[r][c] Wq_h[c][d]
    for (int t = tid; t < Lq * D; t += XA_T) {
      const int r = t / D, d = t % D;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) acc += dQs[r * ldk + c] * Ws[c * D + d];
      float* p = dxq + ((int64_t)b * Lq + r) * D + d;
      *p = (h == 0 ? 0.f : *p) + acc;
    }
  }
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int j = tid + rep * XA_T;
    if (j < Lk) {
      float* p = dmem + ((int64_t)b * Lk + j) * D;
#pragma unroll
      for (int d = 0; d < D; ++d) p[d] = rep == 0 ? dm0[d] : dm1[d];
    }
  }
}

#define XA_DISPATCH(hd, h, CALL)                              \
  if (hd == 16 && h == 2) { CALL(16, 2); }                    \
  else if (hd == 4 && h == 2) { CALL(4, 2); }                 \
  else if (hd == 6 && h == 2) { CALL(6, 2); }                 \
  else if (hd == 10 && h == 2) { CALL(10, 2); }               \
  else if (hd == 15 && h == 2) { CALL(15, 2); }               \
  else if (hd == 24 && h == 2) { CALL(24, 2); }               \
  else { return 0; }

// bytes of dynamic LDS, or 0 when (head_dim, heads, Lq, Lk) is not supported by the fused kernel
extern "C" size_t igcn_xattn_lds_bytes(int D, int H, int Lq, int Lk, int backward) {
  if (H <= 0 || D % H || Lq <= 0 || Lk <= 0 || Lq > XA_T || Lk > 2 * XA_T) return 0;
  const int hd = D / H;
  size_t fl = 0;
#define CALL(HDV, HV) fl = backward ? XaSmem<HDV, HV>::bwd_floats(Lq, Lk) : XaSmem<HDV, HV>::fwd_floats(Lq, Lk)
  XA_DISPATCH(hd, H, CALL)
#undef CALL
  const size_t bytes = fl * sizeof(float);
  return bytes <= 160 * 1024 ? bytes : 0;
}

extern "C" int igcn_xattn_fwd(int B, int D, int H, int Lq, int Lk, const float* xq, const float* mem,
                              const float* w_in, const float* b_in, const float* w_out, const float* b_out,
                              float* out, float* o_save, float* lse, void* stream) {
  const size_t lds = igcn_xattn_lds_bytes(D, H, Lq, Lk, 0);
  if (lds == 0) {
    igcn_set_error("xattn_fwd: unsupported shape D=%d H=%d Lq=%d Lk=%d", D, H, Lq, Lk);
    return IGCN_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  const int hd = D / H;
#define CALL(HDV, HV)                                                                                           \
  {                                                                                                             \
    IGCN_ALLOW_BIG_LDS((k_xattn_fwd<HDV, HV>));                                        \
    hipLaunchKernelGGL((k_xattn_fwd<HDV, HV>), dim3(B), dim3(XA_T), lds, st, Lq, Lk, xq, mem, w_in, b_in, w_out,  \
                       b_out, out, o_save, lse);                                                                \
  }
  XA_DISPATCH(hd, H, CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("xattn_fwd");
  return IGCN_OK;
}

extern "C" size_t igcn_xattn_param_floats(int D) { return (size_t)(3 * D * D + 3 * D + D * D + D); }

extern "C" int igcn_xattn_bwd(int B, int D, int H, int Lq, int Lk, const float* xq, const float* mem,
                              const float* w_in, const float* b_in, const float* w_out, const float* out,
                              const float* o_save, const float* lse, const float* dout, float* dxq, float* dmem,
                              float* dparams, float* scratch /*[B * param_floats]*/, void* stream) {
  const size_t lds = igcn_xattn_lds_bytes(D, H, Lq, Lk, 1);
  if (lds == 0) {
    igcn_set_error("xattn_bwd: unsupported shape D=%d H=%d Lq=%d Lk=%d", D, H, Lq, Lk);
    return IGCN_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  const int hd = D / H;
#define CALL(HDV, HV)                                                                                           \
  {                                                                                                             \
    IGCN_ALLOW_BIG_LDS((k_xattn_bwd<HDV, HV>));                                        \
    hipLaunchKernelGGL((k_xattn_bwd<HDV, HV>), dim3(B), dim3(XA_T), lds, st, Lq, Lk, xq, mem, w_in, b_in, w_out,  \
                       out, o_save, lse, dout, dxq, dmem, scratch);                                             \
  }
  XA_DISPATCH(hd, H, CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("xattn_bwd");
  const int pw = (int)igcn_xattn_param_floats(D);
  return igcn_launch_reduce_rows(scratch, B, pw, pw, dparams, 0, st);
}
