// Cross-attention core for head_dim 16 on the CDNA4 matrix cores (exact-fp32 v_mfma_f32_16x16x4_f32), in place on
// the projection outputs like attn_core.hip:   q [B,Lq,H*16]   kv [B,Lk,2,H*16]   o [B,Lq,H*16]
// (kernel/sgcn_img_snp.py:240, the nn.MultiheadAttention core; softmax(q k^T / 4) v per head).
//
// Layout trick (no cross-lane transposes, no LDS round trip for the probabilities): a 16x16 score tile is computed
// TRANSPOSED, S^T = K_tile Q_tile^T, so that in the MFMA accumulator layout lane (g = l>>4, n = l&15) holds
// S^T[key 4g+r][query n] in register r.  That is exactly the B-operand layout (B[k = g][n]) of a 16x16x4 MFMA whose
// reduction index k runs over the keys {4g+r : g = 0..3} for a fixed register r — so register r of every lane feeds
// the r-th of four MFMAs O^T += V_r^T P_r^T directly, with the A operand (V) simply read in the matching key order.
// All softmax statistics are per query = per lane column n: running max / sum live in one register each.
// The backward applies the same trick twice: wave tasks that own a 16-query tile (S^T orientation: dQ) and wave tasks
// that own a 16-key tile (S orientation: dK, dV); no atomics, every output element has exactly one writer.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AM_LD 17              // padded LDS row (16 floats + 1): conflict-free column walks
#define AM_HD 16
#define AM_MAX_WAVES 16

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// rows [0, rows_valid) of a [*, D]-strided source (16 floats at column offset col0) -> padded LDS rows; rows up to
// rows_pad are zero-filled
__device__ __forceinline__ void am_stage(const float* __restrict__ src, int64_t row_stride, int rows_valid,
                                         int rows_pad, float* __restrict__ dst) {
  for (int t = threadIdx.x; t < rows_pad * 4; t += blockDim.x) {
    const int j = t >> 2, c = (t & 3) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < rows_valid) v = *reinterpret_cast<const float4*>(src + (int64_t)j * row_stride + c);
    float* d = dst + j * AM_LD + c;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

__global__ void __launch_bounds__(64 * AM_MAX_WAVES)
k_attn_mfma_fwd(int H, int Lq, int Lk, const float* __restrict__ q, const float* __restrict__ kv,
                float* __restrict__ o, float* __restrict__ lse) {
  extern __shared__ float smem[];
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * AM_HD;
  const int Lkp = (Lk + 15) & ~15, nkt = Lkp >> 4, nqt = (Lq + 15) >> 4;
  float* Ks = smem;
  float* Vs = Ks + (size_t)Lkp * AM_LD;
  const float* kbase = kv + (int64_t)b * Lk * 2 * D + h * AM_HD;
  am_stage(kbase, 2 * D, Lk, Lkp, Ks);
  am_stage(kbase + D, 2 * D, Lk, Lkp, Vs);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6, n = lane & 15, g = lane >> 4;
  const float scale = 0.25f;                                    // 1/sqrt(16)
  for (int qt = w; qt < nqt; qt += nw) {
    const int qi = qt * 16 + n;
    float qb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      qb[c] = (qi < Lq) ? q[(int64_t)(b * Lq + qi) * D + h * AM_HD + 4 * c + g] * scale : 0.f;
    float m = -INFINITY, l = 0.f;
    f32x4 oacc = {0.f, 0.f, 0.f, 0.f};
    for (int kt0 = 0; kt0 < nkt; kt0 += 4) {
      f32x4 s[4];
      float tmax = -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s[u] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (kt0 + u < nkt) {                                    // wave-uniform
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          const float* kr = Ks + ((kt0 + u) * 16 + n) * AM_LD + g;
#pragma unroll
          for (int c = 0; c < 4; ++c) acc = mfma4(kr[4 * c], qb[c], acc);
          const int key0 = (kt0 + u) * 16 + 4 * g;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s[u][r] = (key0 + r < Lk) ? acc[r] : -INFINITY;
            tmax = fmaxf(tmax, s[u][r]);
          }
        }
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float mn = fmaxf(m, tmax);
      const float f = __expf(m - mn);                           // exp(-inf) = 0 on the first group
      m = mn;
      l *= f;
      oacc *= f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (kt0 + u < nkt) {
          const float* vr = Vs + ((kt0 + u) * 16 + 4 * g) * AM_LD + n;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __expf(s[u][r] - m);
            l += p;
            oacc = mfma4(vr[r * AM_LD], p, oacc);
          }
        }
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (qi < Lq) {
      const float inv = 1.f / l;
      *reinterpret_cast<float4*>(o + (int64_t)(b * Lq + qi) * D + h * AM_HD + 4 * g) =
          make_float4(oacc[0] * inv, oacc[1] * inv, oacc[2] * inv, oacc[3] * inv);
      if (g == 0) lse[((int64_t)b * H + h) * Lq + qi] = m + __logf(l);
    }
  }
}

// backward: dq [B,Lq,D], dkv [B,Lk,2,D]
__global__ void __launch_bounds__(64 * AM_MAX_WAVES)
k_attn_mfma_bwd(int H, int Lq, int Lk, const float* __restrict__ q, const float* __restrict__ kv,
                const float* __restrict__ o, const float* __restrict__ lse, const float* __restrict__ dout,
                float* __restrict__ dq, float* __restrict__ dkv) {
  extern __shared__ float smem[];
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * AM_HD;
  const int Lkp = (Lk + 15) & ~15, Lqp = (Lq + 15) & ~15, nkt = Lkp >> 4, nqt = Lqp >> 4;
  float* Ks = smem;
  float* Vs = Ks + (size_t)Lkp * AM_LD;
  float* Qs = Vs + (size_t)Lkp * AM_LD;
  float* dOs = Qs + (size_t)Lqp * AM_LD;
  float* ls = dOs + (size_t)Lqp * AM_LD;                        // lse [Lqp]   (+inf on padding rows: p = 0)
  float* dl = ls + Lqp;                                         // delta [Lqp]
  int* next_task = (int*)(dl + Lqp);                            // wave task counter
  const float* kbase = kv + (int64_t)b * Lk * 2 * D + h * AM_HD;
  const float* qbase = q + (int64_t)b * Lq * D + h * AM_HD;
  const float* dobase = dout + (int64_t)b * Lq * D + h * AM_HD;
  am_stage(kbase, 2 * D, Lk, Lkp, Ks);
  am_stage(kbase + D, 2 * D, Lk, Lkp, Vs);
  am_stage(qbase, D, Lq, Lqp, Qs);
  am_stage(dobase, D, Lq, Lqp, dOs);
  for (int r = threadIdx.x; r < Lqp; r += blockDim.x) {
    float d = 0.f, lv = INFINITY;
    if (r < Lq) {
      const float* op = o + (int64_t)(b * Lq + r) * D + h * AM_HD;
      const float* dp = dobase + (int64_t)r * D;
#pragma unroll
      for (int c = 0; c < AM_HD; ++c) d += op[c] * dp[c];
      lv = lse[((int64_t)b * H + h) * Lq + r];
    }
    dl[r] = d;
    ls[r] = lv;
  }
  if (threadIdx.x == 0) *next_task = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  const float scale = 0.25f;
  // wave tasks, long ones first: [0, nqt) own a query tile (25-tile loops), [nqt, nqt+nkt) own a key tile
  for (;;) {
    int task = 0;
    if (lane == 0) task = atomicAdd(next_task, 1);
    task = __shfl(task, 0, 64);
    if (task >= nqt + nkt) break;
    if (task < nqt) {
      // ---- dQ for queries [16*task, 16*task+16): S^T orientation, lane column n = query ----
      const int qrow = task * 16 + n;
      float qb[4], dob[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        qb[c] = Qs[qrow * AM_LD + 4 * c + g] * scale;
        dob[c] = dOs[qrow * AM_LD + 4 * c + g];
      }
      const float lsn = ls[qrow], dln = dl[qrow];
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};                         // dQ^T[hd 4g+r][query n]
      for (int kt = 0; kt < nkt; ++kt) {
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
        const float* kr = Ks + (kt * 16 + n) * AM_LD + g;
        const float* vr = Vs + (kt * 16 + n) * AM_LD + g;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          st = mfma4(kr[4 * c], qb[c], st);                     // S^T = K Q^T (scaled)
          dpt = mfma4(vr[4 * c], dob[c], dpt);                  // dP^T = V dO^T
        }
        const float* kc = Ks + (kt * 16 + 4 * g) * AM_LD + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ds = __expf(st[r] - lsn) * (dpt[r] - dln) * scale;   // padded keys: K row = 0 kills the term
          acc = mfma4(kc[r * AM_LD], ds, acc);                  // dQ^T += K^T dS^T
        }
      }
      if (qrow < Lq)
        *reinterpret_cast<float4*>(dq + (int64_t)(b * Lq + qrow) * D + h * AM_HD + 4 * g) =
            make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
      // ---- dK, dV for keys [16*kt, 16*kt+16): S orientation, lane column n = key ----
      const int kt = task - nqt, krow = kt * 16 + n;
      float kb[4], vb[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        kb[c] = Ks[krow * AM_LD + 4 * c + g] * scale;
        vb[c] = Vs[krow * AM_LD + 4 * c + g];
      }
      f32x4 dka = {0.f, 0.f, 0.f, 0.f}, dva = {0.f, 0.f, 0.f, 0.f};     // dK^T / dV^T [hd 4g+r][key n]
      for (int qt = 0; qt < nqt; ++qt) {
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
        const float* qr = Qs + (qt * 16 + n) * AM_LD + g;
        const float* dr = dOs + (qt * 16 + n) * AM_LD + g;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          st = mfma4(qr[4 * c], kb[c], st);                     // S = Q K^T (scaled): rows = queries 4g+r
          dpt = mfma4(dr[4 * c], vb[c], dpt);                   // dP = dO V^T
        }
        const float* qc = Qs + (qt * 16 + 4 * g) * AM_LD + n;
        const float* dc = dOs + (qt * 16 + 4 * g) * AM_LD + n;
        const float* lsr = ls + qt * 16 + 4 * g;
        const float* dlr = dl + qt * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(st[r] - lsr[r]);               // padding queries: lse = +inf -> p = 0
          const float ds = p * (dpt[r] - dlr[r]) * scale;
          dva = mfma4(dc[r * AM_LD], p, dva);                   // dV^T += dO^T P
          dka = mfma4(qc[r * AM_LD], ds, dka);                  // dK^T += Q^T dS
        }
      }
      if (krow < Lk) {
        float* base = dkv + ((int64_t)(b * Lk + krow) * 2) * D + h * AM_HD + 4 * g;
        *reinterpret_cast<float4*>(base) = make_float4(dka[0], dka[1], dka[2], dka[3]);
        *reinterpret_cast<float4*>(base + D) = make_float4(dva[0], dva[1], dva[2], dva[3]);
      }
    }
  }
}

static size_t am_lds_bytes(int Lq, int Lk, int backward) {
  const size_t Lkp = (size_t)((Lk + 15) & ~15), Lqp = (size_t)((Lq + 15) & ~15);
  size_t fl = 2 * Lkp * AM_LD;
  if (backward) fl += 2 * Lqp * AM_LD + 2 * Lqp + 4;
  return fl * sizeof(float);
}

// 0 when the MFMA path does not cover the shape
size_t igcn_attn_mfma_lds_bytes(int D, int H, int Lq, int Lk, int backward) {
  if (H <= 0 || D != H * AM_HD || Lq <= 0 || Lk <= 0) return 0;
  const size_t bytes = am_lds_bytes(Lq, Lk, backward);
  return bytes <= 150 * 1024 ? bytes : 0;
}

int igcn_attn_mfma_fwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, float* o, float* lse,
                       hipStream_t st) {
  const size_t lds = am_lds_bytes(Lq, Lk, 0);
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute((const void*)k_attn_mfma_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  int waves = (Lq + 15) / 16;
  if (waves > AM_MAX_WAVES) waves = AM_MAX_WAVES;
  if (waves < 4) waves = 4;                       // enough threads to stage K/V quickly
  hipLaunchKernelGGL(k_attn_mfma_fwd, dim3(B * H), dim3(64 * waves), lds, st, H, Lq, Lk, q, kv, o, lse);
  IGCN_CHECK_LAUNCH("attn_mfma_fwd");
  return IGCN_OK;
}

int igcn_attn_mfma_bwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, const float* o,
                       const float* lse, const float* dout, float* dq, float* dkv, hipStream_t st) {
  const size_t lds = am_lds_bytes(Lq, Lk, 1);
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute((const void*)k_attn_mfma_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  const int tasks = (Lq + 15) / 16 + (Lk + 15) / 16;
  int waves = tasks < 8 ? (tasks < 4 ? 4 : tasks) : 8;
  hipLaunchKernelGGL(k_attn_mfma_bwd, dim3(B * H), dim3(64 * waves), lds, st, H, Lq, Lk, q, kv, o, lse, dout, dq, dkv);
  IGCN_CHECK_LAUNCH("attn_mfma_bwd");
  return IGCN_OK;
}
