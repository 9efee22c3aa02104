// Cross-attention core on the projection outputs IN PLACE (no head transposes, no contiguous copies):
//   q  [B, Lq, H*HD]        = query projection          kv [B, Lk, 2, H*HD] = key|value projection (one GEMM)
//   o  [B, Lq, H*HD]        = softmax(q k^T / sqrt(HD)) v  per head, heads concatenated (input of out_proj)
// kernel/sgcn_img_snp.py:240 (nn.MultiheadAttention core).  One workgroup per (sample, head); K and V of the head
// live in LDS un-padded so every K/V row is read as four 16-byte LDS broadcasts; lanes = query rows (forward, dQ) or
// keys (dK/dV), 48-64 FMAs per 8 LDS reads.  Backward recomputes the probabilities from the saved log-sum-exp.
#include <stdlib.h>

#include "common.h"

#define AC_T 1024     // 16 waves = 4 per SIMD: one workgroup per CU (K,V of the head fill LDS) hides its own latencies

template <int HD>
__device__ __forceinline__ void ld_row(const float* __restrict__ p, float (&r)[HD]) {
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    const float4 v = *reinterpret_cast<const float4*>(p + c);
    r[c] = v.x; r[c + 1] = v.y; r[c + 2] = v.z; r[c + 3] = v.w;
  }
}

template <int HD>
__device__ __forceinline__ float dot_row(const float (&a)[HD], const float (&b)[HD]) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < HD; ++c) s += a[c] * b[c];
  return s;
}

// stage K and V rows of (b,h) and `rows` query-side rows of up to two sources into LDS with 16-byte copies
template <int HD>
__device__ __forceinline__ void stage_kv(const float* __restrict__ kv, int b, int h, int H, int Lk,
                                         float* __restrict__ Ks, float* __restrict__ Vs) {
  const int D = H * HD, nq = HD / 4;
  for (int t = threadIdx.x; t < Lk * nq; t += AC_T) {
    const int j = t / nq, c = (t % nq) * 4;
    const float* base = kv + ((int64_t)(b * Lk + j) * 2) * D + h * HD + c;
    *reinterpret_cast<float4*>(Ks + j * HD + c) = *reinterpret_cast<const float4*>(base);
    *reinterpret_cast<float4*>(Vs + j * HD + c) = *reinterpret_cast<const float4*>(base + D);
  }
}

template <int HD>
__device__ __forceinline__ void stage_q(const float* __restrict__ src, int b, int h, int H, int Lq,
                                        float* __restrict__ dst) {
  const int D = H * HD, nq = HD / 4;
  for (int t = threadIdx.x; t < Lq * nq; t += AC_T) {
    const int i = t / nq, c = (t % nq) * 4;
    *reinterpret_cast<float4*>(dst + i * HD + c) =
        *reinterpret_cast<const float4*>(src + (int64_t)(b * Lq + i) * D + h * HD + c);
  }
}

template <int HD>
__global__ void __launch_bounds__(AC_T)
k_attn_core_fwd(int H, int Lq, int Lk, const float* __restrict__ q, const float* __restrict__ kv,
                float* __restrict__ o, float* __restrict__ lse) {
  extern __shared__ float smem[];
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * HD;
  const int rp = (Lq + 31) / 32 * 32, G = AC_T / rp;       // key groups per query row
  float* Ks = smem;
  float* Vs = Ks + (size_t)Lk * HD;
  float* Qs = Vs + (size_t)Lk * HD;
  float* Mg = Qs + (size_t)Lq * HD;                        // [G][Lq][HD+2]
  stage_kv<HD>(kv, b, h, H, Lk, Ks, Vs);
  stage_q<HD>(q, b, h, H, Lq, Qs);
  __syncthreads();
  const float scale = rsqrtf((float)HD);
  const int i = threadIdx.x % rp, g = threadIdx.x / rp;
  const int kper = (Lk + G - 1) / G, j0 = g * kper, j1 = min(Lk, j0 + kper);
  if (i < Lq && g < G) {
    float qi[HD], acc[HD];
    ld_row<HD>(Qs + i * HD, qi);
#pragma unroll
    for (int c = 0; c < HD; ++c) { qi[c] *= scale; acc[c] = 0.f; }
    // single pass, keys in chunks of 8: chunk scores -> chunk max -> one rescale of the running sums per chunk
    float m = -INFINITY, l = 0.f;
    for (int jb = j0; jb < j1; jb += 8) {
      float sc[8];
      float cm = m;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        sc[u] = -INFINITY;
        if (jb + u < j1) {
          float kj[HD];
          ld_row<HD>(Ks + (jb + u) * HD, kj);
          sc[u] = dot_row<HD>(qi, kj);
        }
        cm = fmaxf(cm, sc[u]);
      }
      const float f = __expf(m - cm);           // exp(-inf) = 0 on the first chunk
      l *= f;
#pragma unroll
      for (int c = 0; c < HD; ++c) acc[c] *= f;
      m = cm;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (jb + u < j1) {
          float vj[HD];
          ld_row<HD>(Vs + (jb + u) * HD, vj);
          const float p = __expf(sc[u] - m);
          l += p;
#pragma unroll
          for (int c = 0; c < HD; ++c) acc[c] += p * vj[c];
        }
      }
    }
    float* mg = Mg + ((size_t)g * Lq + i) * (HD + 2);
    mg[0] = m;
    mg[1] = l;
#pragma unroll
    for (int c = 0; c < HD; ++c) mg[2 + c] = acc[c];
  }
  __syncthreads();
  if (i < Lq && g == 0) {
    float M = -INFINITY;
    for (int gg = 0; gg < G; ++gg) M = fmaxf(M, Mg[((size_t)gg * Lq + i) * (HD + 2)]);
    float L = 0.f, acc[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    for (int gg = 0; gg < G; ++gg) {
      const float* mg = Mg + ((size_t)gg * Lq + i) * (HD + 2);
      if (mg[1] > 0.f) {
        const float f = __expf(mg[0] - M);
        L += mg[1] * f;
#pragma unroll
        for (int c = 0; c < HD; ++c) acc[c] += mg[2 + c] * f;
      }
    }
    const float inv = 1.f / L;
    float* op = o + (int64_t)(b * Lq + i) * D + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4)
      *reinterpret_cast<float4*>(op + c) =
          make_float4(acc[c] * inv, acc[c + 1] * inv, acc[c + 2] * inv, acc[c + 3] * inv);
    lse[((int64_t)b * H + h) * Lq + i] = M + __logf(L);
  }
}

// backward: dq [B,Lq,D], dkv [B,Lk,2,D]
template <int HD>
__global__ void __launch_bounds__(AC_T)
k_attn_core_bwd(int H, int Lq, int Lk, const float* __restrict__ q, const float* __restrict__ kv,
                const float* __restrict__ o, const float* __restrict__ lse, const float* __restrict__ dout,
                float* __restrict__ dq, float* __restrict__ dkv) {
  extern __shared__ float smem[];
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * HD;
  const int rp = (Lq + 31) / 32 * 32, G = AC_T / rp;
  float* Ks = smem;
  float* Vs = Ks + (size_t)Lk * HD;
  float* Qs = Vs + (size_t)Lk * HD;
  float* dOs = Qs + (size_t)Lq * HD;
  float* ls = dOs + (size_t)Lq * HD;                        // lse [Lq]
  float* dl = ls + Lq;                                      // delta [Lq]
  float* Mg = dl + Lq;                                      // dq partials [G][Lq][HD]
  stage_kv<HD>(kv, b, h, H, Lk, Ks, Vs);
  stage_q<HD>(q, b, h, H, Lq, Qs);
  stage_q<HD>(dout, b, h, H, Lq, dOs);
  for (int r = threadIdx.x; r < Lq; r += AC_T) {
    const float* op = o + (int64_t)(b * Lq + r) * D + h * HD;
    const float* dp = dout + (int64_t)(b * Lq + r) * D + h * HD;
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) d += op[c] * dp[c];
    dl[r] = d;
    ls[r] = lse[((int64_t)b * H + h) * Lq + r];
  }
  __syncthreads();
  const float scale = rsqrtf((float)HD);
  // ---- dQ: lanes = query rows, G key groups ----
  const int i = threadIdx.x % rp, g = threadIdx.x / rp;
  const int kper = (Lk + G - 1) / G, j0 = g * kper, j1 = min(Lk, j0 + kper);
  if (i < Lq && g < G) {
    float qi[HD], doi[HD], acc[HD];
    ld_row<HD>(Qs + i * HD, qi);
    ld_row<HD>(dOs + i * HD, doi);
#pragma unroll
    for (int c = 0; c < HD; ++c) { qi[c] *= scale; acc[c] = 0.f; }
    const float lsi = ls[i], dli = dl[i];
    for (int j = j0; j < j1; ++j) {
      float kj[HD], vj[HD];
      ld_row<HD>(Ks + j * HD, kj);
      ld_row<HD>(Vs + j * HD, vj);
      const float ds = __expf(dot_row<HD>(qi, kj) - lsi) * (dot_row<HD>(doi, vj) - dli) * scale;
#pragma unroll
      for (int c = 0; c < HD; ++c) acc[c] += ds * kj[c];
    }
    float* mg = Mg + ((size_t)g * Lq + i) * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) mg[c] = acc[c];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < Lq * HD; t += AC_T) {
    float acc = 0.f;
    for (int gg = 0; gg < G; ++gg) acc += Mg[(size_t)gg * Lq * HD + t];
    dq[(int64_t)(b * Lq + t / HD) * D + h * HD + (t % HD)] = acc;
  }
  // ---- dK, dV: lanes = (key, row part); the rs parts of a key sit on adjacent lanes and are summed by shuffles ----
  int rs = 1;
  while (rs < 16 && Lk * rs * 2 <= AC_T) rs *= 2;
  const int rper = (Lq + rs - 1) / rs;
  for (int t0 = 0; t0 < Lk * rs; t0 += AC_T) {
    const int t = t0 + threadIdx.x;
    const bool live = t < Lk * rs;
    const int j = live ? t / rs : 0, part = t % rs;
    float kj[HD], vj[HD], dk[HD], dv[HD];
    ld_row<HD>(Ks + j * HD, kj);
    ld_row<HD>(Vs + j * HD, vj);
#pragma unroll
    for (int c = 0; c < HD; ++c) dk[c] = dv[c] = 0.f;
    const int r1 = live ? min(Lq, (part + 1) * rper) : 0;
    for (int r = part * rper; r < r1; ++r) {
      float qr[HD], dor[HD];
      ld_row<HD>(Qs + r * HD, qr);
      ld_row<HD>(dOs + r * HD, dor);
      const float p = __expf(dot_row<HD>(qr, kj) * scale - ls[r]);
      const float ds = p * (dot_row<HD>(dor, vj) - dl[r]) * scale;
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        dk[c] += ds * qr[c];
        dv[c] += p * dor[c];
      }
    }
    for (int o2 = 1; o2 < rs; o2 <<= 1) {
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        dk[c] += __shfl_xor(dk[c], o2, 64);
        dv[c] += __shfl_xor(dv[c], o2, 64);
      }
    }
    if (live && part == 0) {
      float* base = dkv + ((int64_t)(b * Lk + j) * 2) * D + h * HD;
#pragma unroll
      for (int c = 0; c < HD; c += 4) {
        *reinterpret_cast<float4*>(base + c) = make_float4(dk[c], dk[c + 1], dk[c + 2], dk[c + 3]);
        *reinterpret_cast<float4*>(base + D + c) = make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]);
      }
    }
  }
}

static size_t ac_lds_floats(int HD, int Lq, int Lk, int backward) {
  const int rp = (Lq + 31) / 32 * 32, G = AC_T / rp;
  if (backward) return (size_t)2 * Lk * HD + 2 * (size_t)Lq * HD + 2 * (size_t)Lq + (size_t)G * Lq * HD;
  return (size_t)2 * Lk * HD + (size_t)Lq * HD + (size_t)G * Lq * (HD + 2);
}

#define AC_DISPATCH(hd, CALL)                 \
  if (hd == 16) { CALL(16); }                 \
  else if (hd == 4) { CALL(4); }              \
  else if (hd == 8) { CALL(8); }              \
  else if (hd == 12) { CALL(12); }            \
  else if (hd == 20) { CALL(20); }            \
  else if (hd == 24) { CALL(24); }            \
  else { return 0; }

// attn_mfma.hip: head_dim <= 32 on the matrix cores (any Lq; K, V (and Q, dO) of one head must fit LDS)
size_t igcn_attn_mfma_lds_bytes(int D, int H, int Lq, int Lk, int backward);
int igcn_attn_mfma_fwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, float* o, float* lse,
                       hipStream_t st);
int igcn_attn_mfma_bwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, const float* o,
                       const float* lse, const float* dout, float* dq, float* dkv, hipStream_t st);
size_t igcn_attn_mfma_chunked_scratch_floats(int B, int H, int Lq);
int igcn_attn_mfma_fwd_chunked(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, float* o,
                               float* lse, hipStream_t st);
int igcn_attn_mfma_bwd_chunked(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, const float* o,
                               const float* lse, const float* dout, float* dq, float* dkv, float* delta,
                               hipStream_t st);

static bool use_mfma(int D, int H, int Lq, int Lk) {
  static int valu_only = -1;                       // IGCN_ATTN_VALU=1: force the VALU kernels (A/B comparisons)
  if (valu_only < 0) {
    const char* e = getenv("IGCN_ATTN_VALU");
    valu_only = (e && e[0] == '1') ? 1 : 0;
  }
  return !valu_only && igcn_attn_mfma_lds_bytes(D, H, Lq, Lk, 1) != 0;
}

// K, V (and Q, dO) of a head too large for LDS: the chunked matrix-core kernels stream them (head_dim <= 32)
static bool use_mfma_chunked(int D, int H, int Lq, int Lk) {
  return H > 0 && D > 0 && D % H == 0 && D / H <= 32 && Lq > 0 && Lk > 0 && !use_mfma(D, H, Lq, Lk) &&
         igcn_attn_mfma_lds_bytes(D, H, 16, 16, 1) != 0 && getenv("IGCN_ATTN_VALU") == nullptr;
}

// dynamic LDS bytes needed, or 0 when the shape is not covered (matrix-core path: head_dim <= 32, any Lq; the VALU
// kernels of this file — head_dim in {4,8,12,16,20,24}, Lq <= 256 — serve IGCN_ATTN_VALU=1 A/B runs and shapes whose
// K/V do not fit the matrix-core kernel's LDS budget)
extern "C" size_t igcn_attn_core_bwd_scratch_floats(int B, int H, int Lq) {
  return igcn_attn_mfma_chunked_scratch_floats(B, H, Lq);
}

extern "C" size_t igcn_attn_core_lds_bytes(int D, int H, int Lq, int Lk, int backward) {
  if (H > 0 && Lq > 0 && Lk > 0 && use_mfma(D, H, Lq, Lk)) return igcn_attn_mfma_lds_bytes(D, H, Lq, Lk, backward);
  if (use_mfma_chunked(D, H, Lq, Lk)) return 96 * 1024;            // streamed in ~96 KB chunks
  if (H <= 0 || D % H || Lq <= 0 || Lq > 256 || Lk <= 0) return 0;
  const int hd = D / H;
  if (!(hd == 4 || hd == 8 || hd == 12 || hd == 16 || hd == 20 || hd == 24)) return 0;
  const size_t bytes = ac_lds_floats(hd, Lq, Lk, backward) * sizeof(float);
  return bytes <= 160 * 1024 ? bytes : 0;
}

extern "C" int igcn_attn_core_fwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, float* o,
                                  float* lse, void* stream) {
  if (H > 0 && Lq > 0 && Lk > 0 && use_mfma(D, H, Lq, Lk))
    return igcn_attn_mfma_fwd(B, D, H, Lq, Lk, q, kv, o, lse, (hipStream_t)stream);
  if (use_mfma_chunked(D, H, Lq, Lk))
    return igcn_attn_mfma_fwd_chunked(B, D, H, Lq, Lk, q, kv, o, lse, (hipStream_t)stream);
  const size_t lds = igcn_attn_core_lds_bytes(D, H, Lq, Lk, 0);
  if (lds == 0) { igcn_set_error("attn_core_fwd: unsupported shape D=%d H=%d Lq=%d Lk=%d", D, H, Lq, Lk); return IGCN_ERR_UNSUPPORTED; }
  const int hd = D / H;
#define CALL(HDV)                                                                                                  \
  {                                                                                                                \
    IGCN_ALLOW_BIG_LDS((k_attn_core_fwd<HDV>));                                                            \
    hipLaunchKernelGGL((k_attn_core_fwd<HDV>), dim3(B * H), dim3(AC_T), lds, (hipStream_t)stream, H, Lq, Lk, q, kv, o, lse); \
  }
  AC_DISPATCH(hd, CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("attn_core_fwd");
  return IGCN_OK;
}

extern "C" int igcn_attn_core_bwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, const float* o,
                                  const float* lse, const float* dout, float* dq, float* dkv, float* scratch,
                                  void* stream) {
  if (H > 0 && Lq > 0 && Lk > 0 && use_mfma(D, H, Lq, Lk))
    return igcn_attn_mfma_bwd(B, D, H, Lq, Lk, q, kv, o, lse, dout, dq, dkv, (hipStream_t)stream);
  if (use_mfma_chunked(D, H, Lq, Lk)) {
    IGCN_REQUIRE(scratch != nullptr, "attn_core_bwd: this shape needs igcn_attn_core_bwd_scratch_floats() of scratch");
    return igcn_attn_mfma_bwd_chunked(B, D, H, Lq, Lk, q, kv, o, lse, dout, dq, dkv, scratch, (hipStream_t)stream);
  }
  const size_t lds = igcn_attn_core_lds_bytes(D, H, Lq, Lk, 1);
  if (lds == 0) { igcn_set_error("attn_core_bwd: unsupported shape D=%d H=%d Lq=%d Lk=%d", D, H, Lq, Lk); return IGCN_ERR_UNSUPPORTED; }
  const int hd = D / H;
#define CALL(HDV)                                                                                                  \
  {                                                                                                                \
    IGCN_ALLOW_BIG_LDS((k_attn_core_bwd<HDV>));                                                            \
    hipLaunchKernelGGL((k_attn_core_bwd<HDV>), dim3(B * H), dim3(AC_T), lds, (hipStream_t)stream, H, Lq, Lk, q, kv, o, lse, dout, dq, dkv); \
  }
  AC_DISPATCH(hd, CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("attn_core_bwd");
  return IGCN_OK;
}
