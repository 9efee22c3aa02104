// Shared host/device helpers for libigcn.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/igcn.h"

#define IGCN_WAVE 64

void igcn_set_error(const char* fmt, ...);

// A/B switches of the library (IGCN_* environment variables of earlier rounds).  They are read ONCE, by the binding when it
// loads the library (igcn_configure, include/igcn.h) — no getenv in any launch path, so a variable flipped in the
// middle of a process cannot silently change which kernel a captured graph's next eager twin runs.
#define IGCN_OPT_NO_TILED_LISTS 1u        /* dense graphs: wave-per-list walks instead of the tiled node-lane kernels */
#define IGCN_OPT_PROPAGATE_NO_LDS 2u      /* dense graphs: the wave-per-target aggregation instead of the LDS-staged one */
#define IGCN_OPT_SPMM_NO_LDS 4u           /* SNP <-> GO maps: the first (untiled) CSR kernels */
#define IGCN_OPT_GO_ATTN_CM 8u            /* GO attention backward: the global-memory kernels even when a sample fits LDS */
#define IGCN_OPT_DEBUG_REDUCE 16u         /* print every deferred reduction at the flush */
#define IGCN_OPT_ATTN_FP32_CORE 32u       /* bf16 feature transforms: keep the fp32 attention core (igcn_attn_core_bf16_supported() = 0) */
#define IGCN_OPT_ATTN_BWD_TWICE 64u       /* fp32 attention backward: the two-orientation kernel (every score tile computed twice) even where the shared-tile kernel applies */
#define IGCN_OPT_ATTN_EXACT_FP32 128u     /* fp32 attention core, head_dim 16: the exact-fp32 MFMA kernels (csrc/attn_mfma.hip) instead of the split-bf16 ones (csrc/attn_split.hip) */
extern unsigned g_igcn_options;
extern int g_igcn_gemm_bn_cap;            /* > 0: cap of the GEMM tile width (sweeps only) */
extern int g_igcn_attn_chunk_rows;        /* > 0: rows per LDS chunk of the streamed attention kernels (sweeps only) */
static inline bool igcn_opt(unsigned bit) { return (g_igcn_options & bit) != 0; }

#define IGCN_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      igcn_set_error(__VA_ARGS__);              \
      return IGCN_ERR_BADARG;                   \
    }                                           \
  } while (0)

// Kernels that ask for more than 64 KB of dynamic LDS need the attribute raised once PER DEVICE (one process may
// drive several GPUs); legal during stream capture.
struct IgcnPerDevice { bool done[64]; };
static inline bool igcn_first_on_device(IgcnPerDevice& f) {
  int d = 0;
  (void)hipGetDevice(&d);
  d &= 63;
  if (f.done[d]) return false;
  f.done[d] = true;
  return true;
}
#define IGCN_ALLOW_BIG_LDS(kernel)                                                                              \
  do {                                                                                                          \
    static IgcnPerDevice f_ = {};                                                                               \
    if (igcn_first_on_device(f_))                                                                               \
      (void)hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  } while (0)

#define IGCN_CHECK_LAUNCH(name)                                          \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) {                                              \
      igcn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return IGCN_ERR_LAUNCH;                                            \
    }                                                                    \
  } while (0)

static inline int64_t igcn_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- wave / block reductions (deterministic: fixed tree) ---------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;  // valid in lane 0
}

__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// GO attention backward, output j of dparams [2 FOUT FIN + 3 FOUT] from the block partials gpart [(2 FOUT + 3) FIN][parts]
// (go.hip k_go_attn_bwd_finish and, deferred, plan.hip k_multi_reduce: ONE body so that both give the same bits).  A
// weight-gradient output is one entry's sum, an attention-vector output the dot product of a weight row with FIN entry
// sums; every entry is summed by ONE wave (lanes stride the partials, fixed tree), waves take entries d = w, w + nw, ...
// G: >= FIN floats of LDS.  All threads of the workgroup must call (barrier inside).
__device__ __forceinline__ void go_attn_finish_output(const float* __restrict__ gpart, int64_t parts, int FIN, int FOUT,
                                                      const float* __restrict__ w_inc, const float* __restrict__ w_s,
                                                      float* __restrict__ dparams, int j, float* G) {
  const int KW = FOUT * FIN, lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int first = j < 2 * KW ? j : (2 * FOUT + (j - 2 * KW) / FOUT) * FIN, cnt = j < 2 * KW ? 1 : FIN;
  for (int d = w; d < cnt; d += nw) {
    const float* src = gpart + (int64_t)(first + d) * parts;
    float t = 0.f;
#pragma unroll 4
    for (int64_t i = lane; i < parts; i += 64) t += src[i];
    t = wave_sum_all(t);
    if (lane == 0) G[d] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (j < 2 * KW) {
      dparams[j] = G[0];
    } else {
      const int q = j - 2 * KW, which = q / FOUT, c = q % FOUT;
      const float* wm = which == 2 ? w_s : w_inc;
      float t = 0.f;
      for (int d = 0; d < FIN; ++d) t = __fmaf_rn(wm[c * FIN + d], G[d], t);
      dparams[j] = t;
    }
  }
}

// Sum over groups of G consecutive lanes (G = 1, 2, 4, ..., 64), result in every lane of the group; fixed tree.
template <int G>
__device__ __forceinline__ float group_sum_all(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blockDim.x <= 1024; `red` must hold >= 16 floats of LDS. Result valid in all threads.
__device__ __forceinline__ float block_sum_all(float v, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();  // protect `red` from a previous use
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// Two block-wide sums behind ONE barrier pair (same arithmetic as two block_sum_all calls); `red` >= 32 floats.
__device__ __forceinline__ void block_sum_all2(float& a, float& b, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  a = wave_sum(a);
  b = wave_sum(b);
  __syncthreads();  // protect `red` from a previous use
  if (lane == 0) {
    red[w] = a;
    red[16 + w] = b;
  }
  __syncthreads();
  float ta = 0.f, tb = 0.f;
  for (int i = 0; i < nw; ++i) {
    ta += red[i];
    tb += red[16 + i];
  }
  a = ta;
  b = tb;
}

// Reduce NV per-thread values over the block; thread j < NV of the block ends up holding total j in
// out[j] (written to global partial row).  red: LDS float[(blockDim/64) * NV].
template <int NV>
__device__ __forceinline__ void block_reduce_vec(const float (&v)[NV], float* red, float* partial_row) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float s = wave_sum(v[j]);
    if (lane == 0) red[w * NV + j] = s;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < NV; j += blockDim.x) {
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i * NV + j];
    partial_row[j] = t;
  }
}

// out[j] = sum_{r<rows} partial[r*ld + j], j < n.  One thread per j, rows summed in order.
__global__ void k_reduce_rows(const float* __restrict__ partial, int64_t rows, int64_t ld, int n,
                              float* __restrict__ out, int accumulate);

int igcn_launch_reduce_contig(const float* partial, int64_t rows, int n, float* out, hipStream_t st);
int igcn_launch_reduce_rows(const float* partial, int64_t rows, int64_t ld, int n, float* out, int accumulate,
                            hipStream_t st);
// The same column sums for an output that is a FINAL parameter gradient (read by nothing before the optimiser):
// while igcn_reduce_defer(1) is in force the reduction is queued instead of launched, and igcn_reduce_flush runs
// every queued reduction of the backward pass in ONE launch (include/igcn.h).
int igcn_launch_reduce_rows_batched(const float* partial, int64_t rows, int64_t ld, int n, float* out, int batch,
                                    int64_t p_batch, int64_t o_batch, hipStream_t st);
int igcn_launch_reduce_rows_final(const float* partial, int64_t rows, int64_t ld, int n, float* out,
                                  hipStream_t st);
