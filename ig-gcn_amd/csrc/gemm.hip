// Dense feature transforms on the CDNA4 matrix cores: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).
//   C[m,n] = act( sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk] + bias[n] )
// One 256-thread workgroup = 4 waves; wave w owns rows [16w,16w+16) of a 64 x BN tile and keeps BN/16
// 16x16 accumulators.  A/B tiles are staged through LDS in BK=16 slices (row pad +1 => conflict-free
// operand reads).  The staging thread->element map follows whichever
// operand axis is contiguous in memory, so NT (forward), NN (input gradient) and TN (weight gradient)
// all read global memory coalesced.  Long-K / few-tile problems are split over gridDim.z into partial
// slabs that a second kernel sums in slab order (deterministic).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define G_BM 64
#define G_BK 32
#define G_LD (G_BK + 2)   // row stride 34: MFMA operand reads [m][4ks + (lane>>4)] hit 32 distinct banks per half-wave

// One operand tile (ROWS x G_BK) global -> registers -> LDS.  `rfast` = the row (m or n) axis is the contiguous
// one in memory, else the k axis is.  VEC: 16-byte loads along the contiguous axis (host guarantees alignment and
// that no float4 straddles the matrix edge); otherwise scalar loads with per-element bounds checks.
template <int ROWS, bool VEC>
struct TileLoader {
  static constexpr int NV = (ROWS * G_BK / 4 + 255) / 256;   // float4 per thread
  static constexpr int NS = (ROWS * G_BK + 255) / 256;       // scalars per thread
  float4 v[VEC ? NV : 1];
  float s[VEC ? 1 : NS];

  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t srow, int64_t sk, bool rfast, int64_t r0,
                                       int64_t rows, int64_t kb, int64_t k_end) {
    const int tid = threadIdx.x;
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int f = tid + i * 256;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < ROWS * G_BK / 4) {
          const int r = rfast ? (f % (ROWS / 4)) * 4 : f / (G_BK / 4);
          const int k = rfast ? f / (ROWS / 4) : (f % (G_BK / 4)) * 4;
          const int64_t gr = r0 + r, gk = kb + k;
          if (gr < rows && gk < k_end) v[i] = *reinterpret_cast<const float4*>(P + gr * srow + gk * sk);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int idx = tid + i * 256;
        s[i] = 0.f;
        if (idx < ROWS * G_BK) {
          const int r = rfast ? idx % ROWS : idx / G_BK;
          const int k = rfast ? idx / ROWS : idx % G_BK;
          const int64_t gr = r0 + r, gk = kb + k;
          if (gr < rows && gk < k_end) s[i] = P[gr * srow + gk * sk];
        }
      }
    }
  }

  __device__ __forceinline__ void store(float (*T)[G_LD], bool rfast) const {
    const int tid = threadIdx.x;
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int f = tid + i * 256;
        if (f < ROWS * G_BK / 4) {
          if (rfast) {
            const int r = (f % (ROWS / 4)) * 4, k = f / (ROWS / 4);
            T[r][k] = v[i].x; T[r + 1][k] = v[i].y; T[r + 2][k] = v[i].z; T[r + 3][k] = v[i].w;
          } else {
            const int r = f / (G_BK / 4), k = (f % (G_BK / 4)) * 4;
            T[r][k] = v[i].x; T[r][k + 1] = v[i].y; T[r][k + 2] = v[i].z; T[r][k + 3] = v[i].w;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int idx = tid + i * 256;
        if (idx < ROWS * G_BK) {
          const int r = rfast ? idx % ROWS : idx / G_BK;
          const int k = rfast ? idx / ROWS : idx % G_BK;
          T[r][k] = s[i];
        }
      }
    }
  }
};

template <int BN, bool VEC>
__global__ void __launch_bounds__(256)
k_gemm_f32(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t sam, int64_t sak,
           const float* __restrict__ B, int64_t sbn, int64_t sbk, const float* __restrict__ bias,
           float* __restrict__ C, int64_t ldc, int act, int64_t k_per_split, int64_t slab_stride,
           int64_t a_zs, int64_t b_zs, int zsplit) {
  __shared__ float As[2][G_BM][G_LD];
  __shared__ float Bs[2][BN][G_LD];
  constexpr int NT = BN / 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * G_BM, n0 = (int64_t)blockIdx.y * BN;
  // blockIdx.z = (batch index) * zsplit + (K slice): K slices of one product (a_zs == b_zs == 0) and/or a batch
  // of independent products (a_zs/b_zs = element offsets per batch) whose slabs are all summed afterwards
  const int64_t bidx = blockIdx.z / zsplit, ks_id = blockIdx.z % zsplit;
  const int64_t k_begin = ks_id * k_per_split;
  const int64_t k_end = k_begin + k_per_split < K ? k_begin + k_per_split : K;
  A += bidx * a_zs;
  B += bidx * b_zs;
  const bool a_rfast = (sam == 1 && sak != 1), b_rfast = (sbn == 1 && sbk != 1);

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // software pipeline: the global loads of K-tile i+1 are in flight while the matrix cores work on tile i
  TileLoader<G_BM, VEC> la;
  TileLoader<BN, VEC> lb;
  if (k_begin < k_end) {
    la.load(A, sam, sak, a_rfast, m0, M, k_begin, k_end);
    lb.load(B, sbn, sbk, b_rfast, n0, N, k_begin, k_end);
    la.store(As[0], a_rfast);
    lb.store(Bs[0], b_rfast);
  }
  __syncthreads();
  int buf = 0;
  for (int64_t kb = k_begin; kb < k_end; kb += G_BK) {
    const bool more = kb + G_BK < k_end;
    if (more) {
      la.load(A, sam, sak, a_rfast, m0, M, kb + G_BK, k_end);
      lb.load(B, sbn, sbk, b_rfast, n0, N, kb + G_BK, k_end);
    }
#pragma unroll
    for (int ks = 0; ks < G_BK / 4; ++ks) {
      const float a = As[buf][w * 16 + (lane & 15)][ks * 4 + (lane >> 4)];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float b = Bs[buf][t * 16 + (lane & 15)][ks * 4 + (lane >> 4)];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
      }
    }
    if (more) {
      la.store(As[buf ^ 1], a_rfast);
      lb.store(Bs[buf ^ 1], b_rfast);
    }
    __syncthreads();
    buf ^= 1;
  }
  // epilogue: C/D layout col = lane&15, row = (lane>>4)*4 + r
  float* Cz = C + (int64_t)blockIdx.z * slab_stride;
  const bool final_out = (gridDim.z == 1);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t gn = n0 + t * 16 + (lane & 15);
    if (gn >= N) continue;
    const float bv = (final_out && bias) ? bias[gn] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t gm = m0 + w * 16 + (lane >> 4) * 4 + r;
      if (gm < M) {
        float v = acc[t][r] + bv;
        if (final_out && act == 1) v = fmaxf(v, 0.f);
        Cz[gm * ldc + gn] = v;
      }
    }
  }
}

// can operand (ptr, row stride, k stride, rows, K) be read with aligned float4 along its contiguous axis?
static bool vec_ok(const float* p, int64_t srow, int64_t sk, int64_t rows, int64_t K) {
  if (((uintptr_t)p & 15) != 0) return false;
  if (sk == 1) return (srow % 4 == 0) && (K % 4 == 0);                 // k contiguous
  if (srow == 1) return (sk % 4 == 0) && (rows % 4 == 0);              // row contiguous
  return false;
}

__global__ void k_gemm_splitk_reduce(int64_t M, int64_t N, int split_k, const float* __restrict__ slabs,
                                     const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int act) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i - m * N;
  float t = 0.f;
#pragma unroll 8
  for (int z = 0; z < split_k; ++z) t += slabs[(int64_t)z * M * N + i];      // independent loads in flight
  if (bias) t += bias[n];
  if (act == 1) t = fmaxf(t, 0.f);
  C[m * ldc + n] = t;
}

extern "C" int igcn_gemm_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t sam, int64_t sak,
                             const float* B, int64_t sbn, int64_t sbk, const float* bias, float* C, int64_t ldc,
                             int act, int split_k, float* scratch, void* stream) {
  IGCN_REQUIRE(M > 0 && N > 0 && K >= 0 && split_k >= 1, "gemm_f32: bad sizes M=%lld N=%lld K=%lld split=%d",
               (long long)M, (long long)N, (long long)K, split_k);
  IGCN_REQUIRE(split_k == 1 || scratch != nullptr, "gemm_f32: split_k>1 needs scratch");
  hipStream_t st = (hipStream_t)stream;
  if (split_k > K / G_BK) split_k = (int)(K / G_BK > 0 ? K / G_BK : 1);
  int64_t kps = igcn_cdiv(igcn_cdiv(K, split_k), G_BK) * G_BK;
  if (kps == 0) kps = G_BK;
  split_k = (int)igcn_cdiv(K > 0 ? K : 1, kps);
  const bool split = split_k > 1;
  float* out = split ? scratch : C;
  const int64_t ld = split ? N : ldc;
  const int64_t slab = split ? M * N : 0;
  const int bn = N <= 16 ? 16 : (N <= 32 ? 32 : 64);
  dim3 grid((unsigned)igcn_cdiv(M, G_BM), (unsigned)igcn_cdiv(N, bn), (unsigned)split_k);
  const bool vec = vec_ok(A, sam, sak, M, K) && vec_ok(B, sbn, sbk, N, K);
#define LAUNCH_G(BNV, VECV)                                                                                    \
  hipLaunchKernelGGL((k_gemm_f32<BNV, VECV>), grid, dim3(256), 0, st, M, N, K, A, sam, sak, B, sbn, sbk, bias, \
                     out, ld, act, kps, slab, (int64_t)0, (int64_t)0, split_k)
  if (vec) {
    if (bn == 16) { LAUNCH_G(16, true); } else if (bn == 32) { LAUNCH_G(32, true); } else { LAUNCH_G(64, true); }
  } else {
    if (bn == 16) { LAUNCH_G(16, false); } else if (bn == 32) { LAUNCH_G(32, false); } else { LAUNCH_G(64, false); }
  }
#undef LAUNCH_G
  IGCN_CHECK_LAUNCH("gemm_f32");
  if (split && ldc == N && M * N <= 4096 && split_k > 32 && bias == nullptr && act == 0)
    return igcn_launch_reduce_rows(scratch, split_k, M * N, (int)(M * N), C, 0, st);   // block-per-output tree
  if (split) {
    hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3((unsigned)igcn_cdiv(M * N, 256)), dim3(256), 0, st, M, N, split_k,
                       scratch, bias, C, ldc, act);
    IGCN_CHECK_LAUNCH("gemm_splitk_reduce");
  }
  return IGCN_OK;
}


// sum_z A_z . B_z^T : gridDim.z = batch, every z covers the whole K; slabs reduced in z order.
// Used for weight gradients whose reduction index is (sample, node) with a non-affine address (channel-major x).
int igcn_gemm_f32_batched_sum_impl(int64_t M, int64_t N, int64_t K, int batch, const float* A, int64_t sam,
                                   int64_t sak, int64_t a_batch, const float* B, int64_t sbn, int64_t sbk,
                                   int64_t b_batch, float* C, int64_t ldc, float* scratch, hipStream_t st);

extern "C" int igcn_gemm_f32_batched_sum(int64_t M, int64_t N, int64_t K, int batch, const float* A, int64_t sam,
                                         int64_t sak, int64_t a_batch, const float* B, int64_t sbn, int64_t sbk,
                                         int64_t b_batch, float* C, int64_t ldc, float* scratch, void* stream) {
  return igcn_gemm_f32_batched_sum_impl(M, N, K, batch, A, sam, sak, a_batch, B, sbn, sbk, b_batch, C, ldc, scratch,
                                        (hipStream_t)stream);
}

int igcn_gemm_f32_batched_sum_impl(int64_t M, int64_t N, int64_t K, int batch, const float* A, int64_t sam,
                                   int64_t sak, int64_t a_batch, const float* B, int64_t sbn, int64_t sbk,
                                   int64_t b_batch, float* C, int64_t ldc, float* scratch, hipStream_t st) {
  IGCN_REQUIRE(M > 0 && N > 0 && K > 0 && batch >= 1 && scratch != nullptr && (a_batch != 0 || b_batch != 0),
               "gemm_f32_batched_sum: bad arguments");
  const int bn = N <= 16 ? 16 : (N <= 32 ? 32 : 64);
  // long K per product: also slice K so that ~2k workgroups are in flight (slabs = batch * ksplit)
  int ksplit = 1;
  while (ksplit < 16 && (int64_t)batch * ksplit * 2 <= 2048 && K / (ksplit * 2) >= 128) ksplit *= 2;
  const int64_t kps = igcn_cdiv(igcn_cdiv(K, ksplit), G_BK) * G_BK;
  ksplit = (int)igcn_cdiv(K, kps);
  const int slabs = batch * ksplit;
  dim3 grid((unsigned)igcn_cdiv(M, G_BM), (unsigned)igcn_cdiv(N, bn), (unsigned)slabs);
  const bool vec = vec_ok(A, sam, sak, M, K) && vec_ok(B, sbn, sbk, N, K) && a_batch % 4 == 0 && b_batch % 4 == 0;
#define LAUNCH_B(BNV, VECV)                                                                                     \
  hipLaunchKernelGGL((k_gemm_f32<BNV, VECV>), grid, dim3(256), 0, st, M, N, K, A, sam, sak, B, sbn, sbk,         \
                     (const float*)nullptr, scratch, N, 0, kps, M * N, a_batch, b_batch, ksplit)
  if (vec) {
    if (bn == 16) { LAUNCH_B(16, true); } else if (bn == 32) { LAUNCH_B(32, true); } else { LAUNCH_B(64, true); }
  } else {
    if (bn == 16) { LAUNCH_B(16, false); } else if (bn == 32) { LAUNCH_B(32, false); } else { LAUNCH_B(64, false); }
  }
#undef LAUNCH_B
  IGCN_CHECK_LAUNCH("gemm_f32_batched_sum");
  if (ldc == N && M * N <= 4096 && slabs > 32)     // few outputs, many slabs: block-per-output tree reduce
    return igcn_launch_reduce_rows(scratch, slabs, M * N, (int)(M * N), C, 0, st);
  hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3((unsigned)igcn_cdiv(M * N, 256)), dim3(256), 0, st, M, N, slabs,
                     scratch, (const float*)nullptr, C, ldc, 0);
  IGCN_CHECK_LAUNCH("gemm_batched_reduce");
  return IGCN_OK;
}
