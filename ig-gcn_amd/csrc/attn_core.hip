// Cross-attention core on the projection outputs IN PLACE (no head transposes, no contiguous copies):
//   q  [B, Lq, H*HD]        = query projection          kv [B, Lk, 2, H*HD] = key|value projection (one GEMM)
//   o  [B, Lq, H*HD]        = softmax(q k^T / sqrt(HD)) v  per head, heads concatenated (input of out_proj)
// kernel/sgcn_img_snp.py:240 (nn.MultiheadAttention core).  This file is the C-ABI entry and the shape dispatch; the
// kernels are in attn_mfma.hip (matrix cores, any head_dim <= 32): one workgroup per (sample, head) with K, V (and, in
// the backward, Q and dO) of the head in LDS, or — when those do not fit — chunked variants that stream the other side
// of the attention through LDS.  [Round 1's VALU kernels, reachable only through an A/B switch since the matrix-core
// path covered every shape they did, are removed.]
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

// attn_mfma.hip
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

// attn_split.hip: head_dim 16 on split bf16 operands (fp32-grade results, bf16 matrix rate) — the default for that width;
// IGCN_ATTN_EXACT_FP32=1 keeps the exact-fp32 kernels
extern "C" int igcn_attn_core_split_supported(int D, int H, int Lq, int Lk);
extern "C" int igcn_attn_core_split_bwd_supported(int D, int H, int Lq, int Lk);
extern "C" int igcn_attn_core_split_fwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, float* o,
                                        float* lse, void* stream);
extern "C" int igcn_attn_core_split_bwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv,
                                        const float* o, const float* lse, const float* dout, float* dq, float* dkv,
                                        float* scratch, void* stream);
static bool aligned16(const void* a, const void* b, const void* c, const void* d) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0;
}
static bool use_split(int D, int H, int Lq, int Lk) {
  return !igcn_opt(IGCN_OPT_ATTN_EXACT_FP32) && g_igcn_attn_chunk_rows == 0 && igcn_attn_core_split_supported(D, H, Lq, Lk);
}

// K, V (and Q, dO) of one head fit LDS
static bool use_resident(int D, int H, int Lq, int Lk) {
  if (g_igcn_attn_chunk_rows > 0 && g_igcn_attn_chunk_rows < Lk) return false;      // sweeps: force the streamed kernels
  return H > 0 && Lq > 0 && Lk > 0 && igcn_attn_mfma_lds_bytes(D, H, Lq, Lk, 1) != 0;
}

// ... or do not: the chunked kernels stream them (head_dim <= 32)
static bool use_chunked(int D, int H, int Lq, int Lk) {
  return H > 0 && D > 0 && D % H == 0 && D / H <= 32 && Lq > 0 && Lk > 0 && !use_resident(D, H, Lq, Lk) &&
         igcn_attn_mfma_lds_bytes(D, H, 16, 16, 1) != 0;
}

extern "C" size_t igcn_attn_core_bwd_scratch_floats(int B, int H, int Lq) {
  return igcn_attn_mfma_chunked_scratch_floats(B, H, Lq);
}

// dynamic LDS bytes needed, or 0 when the shape is not covered (head_dim > 32)
extern "C" size_t igcn_attn_core_lds_bytes(int D, int H, int Lq, int Lk, int backward) {
  if (use_resident(D, H, Lq, Lk)) return igcn_attn_mfma_lds_bytes(D, H, Lq, Lk, backward);
  if (use_chunked(D, H, Lq, Lk)) return 96 * 1024;               // streamed in ~96 KB chunks
  return 0;
}

extern "C" int igcn_attn_core_fwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, float* o,
                                  float* lse, void* stream) {
  if (use_split(D, H, Lq, Lk) && aligned16(q, kv, o, o)) return igcn_attn_core_split_fwd(B, D, H, Lq, Lk, q, kv, o, lse, stream);
  if (use_resident(D, H, Lq, Lk)) return igcn_attn_mfma_fwd(B, D, H, Lq, Lk, q, kv, o, lse, (hipStream_t)stream);
  if (use_chunked(D, H, Lq, Lk))
    return igcn_attn_mfma_fwd_chunked(B, D, H, Lq, Lk, q, kv, o, lse, (hipStream_t)stream);
  igcn_set_error("attn_core_fwd: unsupported shape D=%d H=%d Lq=%d Lk=%d (head_dim <= 32)", D, H, Lq, Lk);
  return IGCN_ERR_UNSUPPORTED;
}

extern "C" int igcn_attn_core_bwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, const float* o,
                                  const float* lse, const float* dout, float* dq, float* dkv, float* scratch,
                                  void* stream) {
  if (use_split(D, H, Lq, Lk) && igcn_attn_core_split_bwd_supported(D, H, Lq, Lk) && aligned16(q, kv, o, dout) &&
      aligned16(dq, dkv, dq, dkv))
    return igcn_attn_core_split_bwd(B, D, H, Lq, Lk, q, kv, o, lse, dout, dq, dkv, scratch, stream);
  if (use_resident(D, H, Lq, Lk))
    return igcn_attn_mfma_bwd(B, D, H, Lq, Lk, q, kv, o, lse, dout, dq, dkv, (hipStream_t)stream);
  if (use_chunked(D, H, Lq, Lk)) {
    IGCN_REQUIRE(scratch != nullptr, "attn_core_bwd: this shape needs igcn_attn_core_bwd_scratch_floats() of scratch");
    return igcn_attn_mfma_bwd_chunked(B, D, H, Lq, Lk, q, kv, o, lse, dout, dq, dkv, scratch, (hipStream_t)stream);
  }
  igcn_set_error("attn_core_bwd: unsupported shape D=%d H=%d Lq=%d Lk=%d (head_dim <= 32)", D, H, Lq, Lk);
  return IGCN_ERR_UNSUPPORTED;
}
