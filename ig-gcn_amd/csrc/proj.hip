// Backward of a bias-free projection y = x W^T (x [M, K], W [N, K], y [M, N]) in ONE pass over the gradient:
//   dX [M, K] = G W          dW [N, K] = G^T X            (G = dL/dy [M, N])
// for the packed input projection of the cross-attention (kernel/sgcn_img_snp.py:240-241: nn.MultiheadAttention's
// in_proj; K = embed dim = 32, N = 32 for the queries, 64 for key | value).  As two GEMMs (igcn_gemm_f32_grouped) G is
// read twice — and for key | value G is the 52 MB gradient of a [512 x 400, 64] tensor: 180 MB of traffic for 48 us, all
// of it bandwidth.  Here a workgroup walks 64-row blocks: G and X tiles staged in LDS once, dX tile out of the matrix
// cores straight to HBM, dW accumulated in registers over the workgroup's blocks and left as ONE partial per workgroup
// (summed by a final, deferrable reduction).  122 MB instead of 180 MB.
// Exact fp32 (v_mfma_f32_16x16x4_f32).  K = 32 or 48 (the embed dims of two- and three-layer models at hidden 16: round 5,
// the (3, 16) entry of the reference's sweep left these kernels for grouped GEMMs that cost 253 us instead of 48), N = K
// or 2 K; everything else takes the grouped GEMM.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PJ_KMAX 48
#define PJ_ROWS 64

extern "C" int igcn_proj_bwd_supported(int64_t M, int N, int K) { return (K == 32 || K == 48) && (N == K || N == 2 * K) && M > 0; }

// workgroups: enough to fill the chip a few times, never more than row blocks
extern "C" int igcn_proj_bwd_blocks(int64_t M) {
  const int64_t nb = igcn_cdiv(M, PJ_ROWS);
  return (int)(nb < 512 ? nb : 512);          // every workgroup leaves an N x K partial for the final reduction (1024 / 512 / 320 workgroups: kernel 27.6 / 29.1 / 38.3 us, reduction 28.4 / 20.8 / 17.1)
}

// Row stride of a workgroup's dW partial.  [Padding the 4 / 8 KB rows by 192 bytes, against a suspected channel-conflict
// pattern in the final reduction's 16-column strips, changed nothing: k_multi_reduce 36.7 us either way.]
#define PJ_PAD 0
__host__ __device__ static inline int64_t pj_ps(int N, int K) { return (int64_t)N * K + PJ_PAD; }
// scratch floats of one projection: igcn_proj_bwd_blocks(M) partial rows of dW (padded) and, behind them, of db
// (sized for the deepest K the kernels take, so that the helper keeps its two arguments)
extern "C" size_t igcn_proj_bwd_scratch_floats(int64_t M, int N) {
  return (size_t)igcn_proj_bwd_blocks(M) * (size_t)(pj_ps(N, PJ_KMAX) + N) + 16;
}

struct PjArgs {
  int64_t M;
  const float *G, *X, *W;
  float *dX, *dWp;
  int blocks;
  float* dbp;                 // [blocks][N] partials of the bias gradient = column sums of G (or NULL)
  int db_zero;                // leading columns whose bias gradient is structurally zero (written as exact zeros)
};

#define PJ_LDS_FLOATS (PJ_ROWS * (2 * PJ_KMAX + 4) + PJ_ROWS * (PJ_KMAX + 4) + 2 * PJ_KMAX * (PJ_KMAX + 4))

template <int N, int K>
__device__ __forceinline__ void proj_bwd_body(float* lds, const PjArgs& a, int bid) {
  constexpr int LG = N + 4, LX = K + 4;               // padded rows: operand reads of 16 lanes spread over the banks
  constexpr int KT = K / 16, NRT = N / 16;            // 16-wide column tiles of dX / dW, row tiles of dW
  float (*Gs)[LG] = reinterpret_cast<float (*)[LG]>(lds);
  float (*Xs)[LX] = reinterpret_cast<float (*)[LX]>(lds + PJ_ROWS * LG);
  float (*Ws)[LX] = reinterpret_cast<float (*)[LX]>(lds + PJ_ROWS * LG + PJ_ROWS * LX);
  const int64_t M = a.M;
  const float* __restrict__ G = a.G;
  const float* __restrict__ X = a.X;
  const float* __restrict__ W = a.W;
  float* __restrict__ dX = a.dX;
  float* __restrict__ dW_partial = a.dWp;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
  for (int i = tid; i < N * K / 4; i += 256) {
    const int r = i / (K / 4), c4 = i - r * (K / 4);
    *reinterpret_cast<float4*>(&Ws[r][4 * c4]) = *reinterpret_cast<const float4*>(W + r * K + 4 * c4);
  }
  // dW tiles of this wave: the NRT x KT tiles of 16 x 16 dealt round-robin (tile j = w + 4 i: row tile j / KT, column
  // tile j % KT) — 1 (32 x 32), 2 (64 x 32), 2-3 (48 x 48) or 4-5 (96 x 48) per wave
  constexpr int TILES = NRT * KT, TPW = (TILES + 3) / 4;
  f32x4 dw[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) dw[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc = 0.f;                                    // (column tid % N, row slice tid / N) of G over this workgroup's tiles
  // bias gradient: NP row slices of a tile, a power of two that divides 64 with NP * N <= 256 threads
  constexpr int NP = N <= 32 ? 8 : N <= 64 ? 4 : 2, RP = PJ_ROWS / NP;
  constexpr int XQ = K / 16, GQ = (N + 15) / 16;       // 16-byte loads per lane of the X / G tile (64 x K / 64 x N floats)
  const int64_t nblk = (M + PJ_ROWS - 1) / PJ_ROWS;
  for (int64_t blk = bid; blk < nblk; blk += a.blocks) {
    const int64_t r0 = blk * PJ_ROWS;
    __syncthreads();                                    // the previous block's tiles are fully consumed (and Ws is in)
    // stage: G tile [64, N] and X tile [64, K], 16 bytes per lane, rows past M zero-filled
    float4 gq[GQ], xq[XQ];
#pragma unroll
    for (int k = 0; k < GQ; ++k) {
      const int i = tid + 256 * k, r = i / (N / 4), c4 = i - r * (N / 4);
      gq[k] = (r < PJ_ROWS && r0 + r < M) ? *reinterpret_cast<const float4*>(G + (r0 + r) * N + 4 * c4)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
      const int i = tid + 256 * k, r = i / (K / 4), c4 = i - r * (K / 4);
      xq[k] = r0 + r < M ? *reinterpret_cast<const float4*>(X + (r0 + r) * K + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < GQ; ++k) {
      const int i = tid + 256 * k, r = i / (N / 4), c4 = i - r * (N / 4);
      if (r < PJ_ROWS) *reinterpret_cast<float4*>(&Gs[r][4 * c4]) = gq[k];
    }
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
      const int i = tid + 256 * k, r = i / (K / 4), c4 = i - r * (K / 4);
      *reinterpret_cast<float4*>(&Xs[r][4 * c4]) = xq[k];
    }
    __syncthreads();
    if (a.dbp && tid < NP * N) {                        // bias gradient: thread = (column, row slice), its slice of every
      const int col = tid % N, rr = (tid / N) * RP;     // tile summed in registers; slices meet after the loop
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < RP; ++r) t += Gs[rr + r][col];
      dbacc += t;
    }
    // dX tile, TRANSPOSED accumulator: dX^T[k][row] = sum_n W^T[k][n] G^T[n][row] — lane (g, n) then owns four
    // consecutive columns 4 g .. 4 g + 3 (+ 16 t) of row 16 w + n: one 16-byte store per tile
    f32x4 dx[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) dx[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < N / 4; ++kk) {
      const float b = Gs[16 * w + n][4 * kk + g];       // B[k = g][col = row n] = G[row][n-index 4 kk + g]
#pragma unroll
      for (int t = 0; t < KT; ++t) dx[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ws[4 * kk + g][16 * t + n], b, dx[t], 0, 0, 0);
    }
    // A above: lane (g, n) supplies A[i = n][k = g] = W^T[column 16 t + n][n-index 4 kk + g] = Ws[4 kk + g][16 t + n]
    if (r0 + 16 * w + n < M) {
#pragma unroll
      for (int t = 0; t < KT; ++t)
        *reinterpret_cast<float4*>(dX + (r0 + 16 * w + n) * K + 16 * t + 4 * g) =
            make_float4(dx[t][0], dx[t][1], dx[t][2], dx[t][3]);
    }
    // dW += G_tile^T X_tile over the 64 rows: A[i = n-index 16 rt + n][k = row 4 kk + g], B[k][col 16 ct + n]
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int j = w + 4 * t;
      if (j < TILES) {                                  // (uniform over the wave)
        const int rt = j / KT, ct = j - rt * KT;
#pragma unroll
        for (int kk = 0; kk < PJ_ROWS / 4; ++kk)
          dw[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Gs[4 * kk + g][16 * rt + n], Xs[4 * kk + g][16 * ct + n], dw[t], 0, 0, 0);
      }
    }
  }
  // the workgroup's partial of dW [N, K]: accumulator lane (g, n), register r = (row 16 rt + 4 g + r, column 16 ct + n)
  float* out = dW_partial + (int64_t)bid * pj_ps(N, K);
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int j = w + 4 * t;
    if (j < TILES) {
      const int rt = j / KT, ct = j - rt * KT;
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(16 * rt + 4 * g + r) * K + 16 * ct + n] = dw[t][r];
    }
  }
  if (a.dbp) {
    __syncthreads();                                    // the tiles are consumed: LDS is free
    if (tid < NP * N) lds[tid] = dbacc;                 // [slice][column]
    __syncthreads();
    if (tid < N) {
      float t = 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) t += lds[p * N + tid];
      a.dbp[(int64_t)bid * N + tid] = tid < a.db_zero ? 0.f : t;
    }
  }
}

// One launch for up to two projections (the query block and the key | value block of the packed in-projection): the
// first `a.blocks` workgroups take `a`, the rest `b` (b.blocks == 0: single).
template <int K>
__global__ void __launch_bounds__(256) k_proj_bwd(PjArgs a, int na, PjArgs b, int nbn) {
  __shared__ __attribute__((aligned(16))) float lds[PJ_ROWS * (2 * K + 4) + PJ_ROWS * (K + 4) + 2 * K * (K + 4)];
  const bool first = (int)blockIdx.x < a.blocks;
  const PjArgs& p = first ? a : b;
  const int bid = first ? blockIdx.x : blockIdx.x - a.blocks;
  if ((first ? na : nbn) == 2 * K)
    proj_bwd_body<2 * K, K>(lds, p, bid);
  else
    proj_bwd_body<K, K>(lds, p, bid);
}
#define PJ_BWD_LAUNCH(grid, ...)                                                                        \
  do {                                                                                                  \
    if (K == 48) hipLaunchKernelGGL(k_proj_bwd<48>, dim3(grid), dim3(256), 0, st, __VA_ARGS__);          \
    else hipLaunchKernelGGL(k_proj_bwd<32>, dim3(grid), dim3(256), 0, st, __VA_ARGS__);                  \
  } while (0)

static int pj_check(int64_t M, int N, int K, const float* G, const float* X, const float* W, float* dX, float* dW,
                    float* scratch) {
  if (!igcn_proj_bwd_supported(M, N, K)) {
    igcn_set_error("proj_bwd: needs K in {32, 48} and N in {K, 2 K} (M=%lld N=%d K=%d)", (long long)M, N, K);
    return IGCN_ERR_UNSUPPORTED;
  }
  IGCN_REQUIRE((((uintptr_t)G | (uintptr_t)X | (uintptr_t)W | (uintptr_t)dX) & 15) == 0 && dW && scratch,
               "proj_bwd: operands must be 16-byte aligned");
  return IGCN_OK;
}

// scratch: igcn_proj_bwd_scratch_floats(M, N) floats.  dW is a FINAL reduction (igcn_reduce_defer).
extern "C" int igcn_proj_bwd(int64_t M, int N, int K, const float* G, const float* X, const float* W, float* dX,
                             float* dW, float* scratch, void* stream) {
  int rc = pj_check(M, N, K, G, X, W, dX, dW, scratch);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const PjArgs a = {M, G, X, W, dX, scratch, igcn_proj_bwd_blocks(M), nullptr, 0}, none = {};
  PJ_BWD_LAUNCH(a.blocks, a, N, none, 0);
  IGCN_CHECK_LAUNCH("proj_bwd");
  return igcn_launch_reduce_rows_final(scratch, a.blocks, pj_ps(N, K), N * K, dW, st);
}

extern "C" int igcn_proj_bwd_pair_bias(int64_t M1, int N1, const float* G1, const float* X1, const float* W1, float* dX1,
                                       float* dW1, float* scratch1, float* db1, int db_zero1, int64_t M2, int N2,
                                       const float* G2, const float* X2, const float* W2, float* dX2, float* dW2,
                                       float* scratch2, float* db2, int db_zero2, int K, void* stream);

// two projections (different M / N, same K) in one launch
extern "C" int igcn_proj_bwd_pair(int64_t M1, int N1, const float* G1, const float* X1, const float* W1, float* dX1,
                                  float* dW1, float* scratch1, int64_t M2, int N2, const float* G2, const float* X2,
                                  const float* W2, float* dX2, float* dW2, float* scratch2, int K, void* stream) {
  int rc = pj_check(M1, N1, K, G1, X1, W1, dX1, dW1, scratch1);
  if (rc) return rc;
  rc = pj_check(M2, N2, K, G2, X2, W2, dX2, dW2, scratch2);
  if (rc) return rc;
  return igcn_proj_bwd_pair_bias(M1, N1, G1, X1, W1, dX1, dW1, scratch1, nullptr, 0, M2, N2, G2, X2, W2, dX2, dW2, scratch2,
                                 nullptr, 0, K, stream);
}

// ... with the BIAS gradients db_i [N_i] = column sums of G_i from the same pass (NULL: not wanted); the first
// db_zero_i entries are written as exact zeros (the key bias of the attention's key | value projection: a softmax over
// keys cannot see it).  scratch_i: igcn_proj_bwd_scratch_floats(M_i, N_i) floats; db_i are final reductions.
extern "C" int igcn_proj_bwd_pair_bias(int64_t M1, int N1, const float* G1, const float* X1, const float* W1, float* dX1,
                                       float* dW1, float* scratch1, float* db1, int db_zero1, int64_t M2, int N2,
                                       const float* G2, const float* X2, const float* W2, float* dX2, float* dW2,
                                       float* scratch2, float* db2, int db_zero2, int K, void* stream) {
  int rc = pj_check(M1, N1, K, G1, X1, W1, dX1, dW1, scratch1);
  if (rc) return rc;
  rc = pj_check(M2, N2, K, G2, X2, W2, dX2, dW2, scratch2);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int nb1 = igcn_proj_bwd_blocks(M1), nb2 = igcn_proj_bwd_blocks(M2);
  const PjArgs a = {M1, G1, X1, W1, dX1, scratch1, nb1, db1 ? scratch1 + (size_t)nb1 * pj_ps(N1, K) : nullptr, db_zero1};
  const PjArgs b = {M2, G2, X2, W2, dX2, scratch2, nb2, db2 ? scratch2 + (size_t)nb2 * pj_ps(N2, K) : nullptr, db_zero2};
  PJ_BWD_LAUNCH(a.blocks + b.blocks, a, N1, b, N2);
  IGCN_CHECK_LAUNCH("proj_bwd_pair");
  if ((rc = igcn_launch_reduce_rows_final(scratch1, a.blocks, pj_ps(N1, K), N1 * K, dW1, st))) return rc;
  if ((rc = igcn_launch_reduce_rows_final(scratch2, b.blocks, pj_ps(N2, K), N2 * K, dW2, st))) return rc;
  if (db1 && (rc = igcn_launch_reduce_rows_final(a.dbp, a.blocks, N1, N1, db1, st))) return rc;
  if (db2 && (rc = igcn_launch_reduce_rows_final(b.dbp, b.blocks, N2, N2, db2, st))) return rc;
  return IGCN_OK;
}

// -------------------------------------------------------------------------------------------------------------
// Forward of the same projections, y = x W^T + b, for K = 32 or 48 (kernel/sgcn_img_snp.py:240: the packed in-projection of the
// cross-attention — queries [B*rois, 32] -> 32, key | value [B*snps, 32] -> 64).  With a reduction depth of 32 the product
// is pure streaming (per 64-row tile: 8 KB in, 8 / 16 KB out, 32 / 64 matrix instructions per wave): the general tiled
// GEMM spends 32 us on 45 MB here (it walks K in steps with a barrier each, and reads x once per 32-column tile).  This
// kernel keeps W and b in LDS for the workgroup's lifetime, stages one 64-row tile of x per step, and writes y from
// TRANSPOSED accumulators (lane (g, n) owns columns 16 t + 4 g .. + 3 of row 16 w + n) through LDS as contiguous runs.
// In-step 31.7 -> 19.0 us (stores straight from the accumulators, 16 rows x 64 bytes per instruction: 20-22 us).
// Exact fp32 (v_mfma_f32_16x16x4_f32).
// -------------------------------------------------------------------------------------------------------------
struct PfArgs {
  int64_t M;
  const float *X, *W, *bias;                 // bias may be NULL
  float* Y;
  int blocks;
};

extern "C" int igcn_proj_fwd_blocks(int64_t M) {
  const int64_t nb = igcn_cdiv(M, PJ_ROWS);
  return (int)(nb < 1024 ? nb : 1024);        // in-step at the bench shape (45 MB): 2048 / 1024 / 768 / 512 workgroups -> 20.2 / 19.0 / 20.5 / 19.0 us
}

template <int N, int K>
__device__ __forceinline__ void proj_fwd_body(float* lds, const PfArgs& a, int bid) {
  constexpr int LX = K + 4, NT = N / 16, XQ = K / 16;
  float (*Ws)[LX] = reinterpret_cast<float (*)[LX]>(lds);                              // [N][K + 4]
  float (*Xs)[LX] = reinterpret_cast<float (*)[LX]>(lds + 2 * K * LX);                 // [64][K + 4]
  float (*Ys)[N + 4] = reinterpret_cast<float (*)[N + 4]>(lds + 2 * K * LX + 64 * LX); // [64][N + 4]
  const int64_t M = a.M;
  const float* __restrict__ X = a.X;
  float* __restrict__ Y = a.Y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
  const int64_t nblk = (M + PJ_ROWS - 1) / PJ_ROWS;
  // the next tile's rows are loaded before the current one is multiplied (K / 16 16-byte loads per lane in flight); the
  // first tile's loads go out together with W's
  float4 xq[XQ];
  auto load = [&](int64_t blk) {
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
      const int i = tid + 256 * k, r = i / (K / 4), c4 = i - r * (K / 4);
      const int64_t row = blk * PJ_ROWS + r;
      xq[k] = (blk < nblk && row < M) ? *reinterpret_cast<const float4*>(X + row * K + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  load(bid);
  constexpr int WQ = (N * K / 4 + 255) / 256;
  float4 wq[WQ];
#pragma unroll
  for (int k = 0; k < WQ; ++k)
    wq[k] = tid + 256 * k < N * K / 4 ? *reinterpret_cast<const float4*>(a.W + (tid + 256 * k) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
    bv[t] = a.bias ? *reinterpret_cast<const float4*>(a.bias + 16 * t + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < WQ; ++k) {
    const int i = tid + 256 * k, r = i / (K / 4), c4 = i - r * (K / 4);
    if (i < N * K / 4) *reinterpret_cast<float4*>(&Ws[r][4 * c4]) = wq[k];
  }
  for (int64_t blk = bid; blk < nblk; blk += a.blocks) {
    __syncthreads();                                    // the previous tile is fully consumed (and Ws is in)
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
      const int i = tid + 256 * k, r = i / (K / 4), c4 = i - r * (K / 4);
      *reinterpret_cast<float4*>(&Xs[r][4 * c4]) = xq[k];
    }
    __syncthreads();
    load(blk + a.blocks);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{bv[t].x, bv[t].y, bv[t].z, bv[t].w};
#pragma unroll
    for (int kk = 0; kk < K / 4; ++kk) {
      const float b = Xs[16 * w + n][4 * kk + g];       // B[k = g][col = row n of the tile]
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ws[16 * t + n][4 * kk + g], b, acc[t], 0, 0, 0);
    }
    // out through LDS: the 64 x N tile is one contiguous piece of y, written as whole 1 KB runs per wave instruction
    // (straight from the accumulators a store instruction covers 16 rows x 64 bytes, 256 bytes apart)
#pragma unroll
    for (int t = 0; t < NT; ++t)
      *reinterpret_cast<float4*>(&Ys[16 * w + n][16 * t + 4 * g]) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (N + 15) / 16; ++k) {
      const int i = tid + 256 * k, r = i / (N / 4), c4 = i - r * (N / 4);
      const int64_t row = blk * PJ_ROWS + r;
      if (r < PJ_ROWS && row < M) *reinterpret_cast<float4*>(Y + row * N + 4 * c4) = *reinterpret_cast<const float4*>(&Ys[r][4 * c4]);
    }
  }
}

template <int K>
__global__ void __launch_bounds__(256) k_proj_fwd(PfArgs a, int na, PfArgs b, int nbn) {
  __shared__ __attribute__((aligned(16))) float lds[2 * K * (K + 4) + 64 * (K + 4) + 64 * (2 * K + 4)];
  const bool first = (int)blockIdx.x < a.blocks;
  const PfArgs& p = first ? a : b;
  const int bid = first ? blockIdx.x : blockIdx.x - a.blocks;
  if ((first ? na : nbn) == 2 * K)
    proj_fwd_body<2 * K, K>(lds, p, bid);
  else
    proj_fwd_body<K, K>(lds, p, bid);
}

static int pf_check(int64_t M, int N, int K, const float* X, const float* W, const float* bias, float* Y) {
  if (!igcn_proj_bwd_supported(M, N, K)) {
    igcn_set_error("proj_fwd: needs K in {32, 48} and N in {K, 2 K} (M=%lld N=%d K=%d)", (long long)M, N, K);
    return IGCN_ERR_UNSUPPORTED;
  }
  IGCN_REQUIRE((((uintptr_t)X | (uintptr_t)W | (uintptr_t)bias | (uintptr_t)Y) & 15) == 0 && X && W && Y,
               "proj_fwd: operands must be 16-byte aligned");
  return IGCN_OK;
}

// y1 [M1, N1] = x1 [M1, K] W1^T + b1 and y2 [M2, N2] = x2 W2^T + b2 in one launch (M2 = 0: the first alone)
extern "C" int igcn_proj_fwd_pair(int64_t M1, int N1, const float* X1, const float* W1, const float* b1, float* Y1,
                                  int64_t M2, int N2, const float* X2, const float* W2, const float* b2, float* Y2, int K,
                                  void* stream) {
  int rc = pf_check(M1, N1, K, X1, W1, b1, Y1);
  if (rc) return rc;
  PfArgs a = {M1, X1, W1, b1, Y1, igcn_proj_fwd_blocks(M1)}, b = {};
  if (M2 > 0) {
    rc = pf_check(M2, N2, K, X2, W2, b2, Y2);
    if (rc) return rc;
    b = PfArgs{M2, X2, W2, b2, Y2, igcn_proj_fwd_blocks(M2)};
  }
  if (K == 48)
    hipLaunchKernelGGL(k_proj_fwd<48>, dim3(a.blocks + b.blocks), dim3(256), 0, (hipStream_t)stream, a, N1, b, N2);
  else
    hipLaunchKernelGGL(k_proj_fwd<32>, dim3(a.blocks + b.blocks), dim3(256), 0, (hipStream_t)stream, a, N1, b, N2);
  IGCN_CHECK_LAUNCH("proj_fwd_pair");
  return IGCN_OK;
}
