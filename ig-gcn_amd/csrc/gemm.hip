// Dense feature transforms on the CDNA4 matrix cores: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).
//   C[m,n] = act( sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk] + bias[n] )
// One 256-thread workgroup = 4 waves; wave w owns rows [16w,16w+16) of a 64 x BN tile and keeps BN/16
// 16x16 accumulators.  A/B tiles are staged through LDS in 32-deep K slices (padded rows => conflict-free
// operand reads), one or two slices ahead in registers.  The staging thread->element map follows whichever
// operand axis is contiguous in memory (a compile-time property of the instantiation), so NT (forward), NN (input
// gradient) and TN (weight gradient) all read global memory coalesced.  Long-K / few-tile problems are split over gridDim.z into partial
// slabs that a second kernel sums in slab order (deterministic).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define G_BM 64
#define G_BK 32

// One operand tile (ROWS x G_BK) of k_gemm_bf16, global -> registers (-> bf16 -> LDS).  `rfast` = the row (m or n) axis is the contiguous
// one in memory, else the k axis is.  VW = floats per load along the contiguous axis: 4 (16-byte loads), 2 (8-byte
// loads: row strides such as 54 or 3182 floats) — the host guarantees alignment and that no vector straddles the
// matrix edge — or 1 (scalar loads with per-element bounds checks).
template <int ROWS, int VW>
struct TileLoader {
  static constexpr int NU = (ROWS * G_BK / VW + 255) / 256;   // vectors per thread
  float v[NU][VW];
  unsigned ok;                                                // bit i: vector i lies inside the matrix and the K slice

  // Branch-free on purpose: every thread issues every load (out-of-range vectors read the matrix origin, a broadcast
  // hit) and the zeroing happens at store time.  With the loads under divergent or uniform branches the compiler can no
  // longer count how many are outstanding and waits for ALL of them before the first LDS store — which serialises a
  // multi-stage pipeline back into one round trip per K step.
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t srow, int64_t sk, bool rfast, int64_t r0,
                                       int64_t rows, int64_t kb, int64_t k_end) {
    const int tid = threadIdx.x;
    ok = 0;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int f = tid + i * 256;
      const int r = rfast ? (f % (ROWS / VW)) * VW : f / (G_BK / VW);
      const int k = rfast ? f / (ROWS / VW) : (f % (G_BK / VW)) * VW;
      const int64_t gr = r0 + r, gk = kb + k;
      const bool in = f < ROWS * G_BK / VW && gr < rows && gk < k_end;
      ok |= in ? 1u << i : 0u;
      const float* src = in ? P + gr * srow + gk * sk : P;
      if constexpr (VW == 4) {
        const float4 t = *reinterpret_cast<const float4*>(src);
        v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
      } else if constexpr (VW == 2) {
        const float2 t = *reinterpret_cast<const float2*>(src);
        v[i][0] = t.x; v[i][1] = t.y;
      } else {
        v[i][0] = *src;
      }
    }
  }
};

#ifdef G_PROBE_ON
__device__ long long g_probe_buf[8 * 8];
#define G_PROBE(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z < 8 ) { g_probe_buf[blockIdx.z * 8 + (i)] = wall_clock64(); g_probe_buf[blockIdx.z * 8 + 4 + (i)] = clock64(); } } while (0)
extern "C" int igcn_debug_gemm_probe(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe_buf), sizeof(long long) * 64);
}
#define G_ACC(j) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = clock64(); __builtin_amdgcn_sched_barrier(0); g_acc[j] += t_ - g_t; g_t = t_; } while (0)
#else
#define G_PROBE(i)
#define G_ACC(j)
#endif

// 16 zero bytes: the source of every out-of-range vector (beyond the matrix edge or the K slice).  Loading zeros
// instead of masking afterwards keeps the loads unconditional AND the LDS stores free of selects.
__device__ float4 g_zero_src = {0.f, 0.f, 0.f, 0.f};

// One operand tile (ROWS x G_BK) of k_gemm_f32, global -> registers -> LDS, layout known at compile time.
// RF = the row (m or n) axis is the contiguous one in memory, else the k axis is.  LDS row stride: 36 floats for a
// k-fast operand (16-byte aligned rows: one ds_write_b128 per vector; the operand reads [row][4 ks + (lane >> 4)] stay
// conflict-free: 36 r mod 64 visits every multiple of 4 once over 16 rows), 34 for a row-fast one (scalar stores).
template <int ROWS, int VW, bool RF>
struct Stage {
  static constexpr int NU = (ROWS * G_BK / VW + 255) / 256;   // vectors per thread
  static constexpr int LD = RF ? G_BK + 2 : G_BK + 4;
  float v[NU][VW];

  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t srow, int64_t sk, int64_t r0, int64_t rows,
                                       int64_t kb, int64_t k_end) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int f = tid + i * 256;
      const int r = RF ? (f % (ROWS / VW)) * VW : f / (G_BK / VW);
      const int k = RF ? f / (ROWS / VW) : (f % (G_BK / VW)) * VW;
      const int64_t gr = r0 + r, gk = kb + k;
      const bool in = f < ROWS * G_BK / VW && gr < rows && gk < k_end;
      const float* src = in ? P + gr * srow + gk * sk : reinterpret_cast<const float*>(&g_zero_src);
      if constexpr (VW == 4) {
        const float4 t = *reinterpret_cast<const float4*>(src);
        v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
      } else if constexpr (VW == 2) {
        const float2 t = *reinterpret_cast<const float2*>(src);
        v[i][0] = t.x; v[i][1] = t.y;
      } else {
        v[i][0] = *src;
      }
    }
  }

  __device__ __forceinline__ void store(float (*T)[LD]) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int f = tid + i * 256;
      if (f < ROWS * G_BK / VW) {
        if constexpr (RF) {
          const int r = (f % (ROWS / VW)) * VW, k = f / (ROWS / VW);
#pragma unroll
          for (int j = 0; j < VW; ++j) T[r + j][k] = v[i][j];
        } else {
          const int r = f / (G_BK / VW), k = (f % (G_BK / VW)) * VW;
          if constexpr (VW == 4) {
            *reinterpret_cast<float4*>(&T[r][k]) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
          } else {
#pragma unroll
            for (int j = 0; j < VW; ++j) T[r][k + j] = v[i][j];
          }
        }
      }
    }
  }
};

// PF = K-tiles in flight in registers ahead of the one in LDS (1 for products of one or two K steps — the short-K
// streaming products, where registers are better spent on more workgroups per CU — 2 otherwise).
template <int BN, int VW, int PF, bool ARF, bool BRF>
__device__ __forceinline__ void
gemm_f32_tile(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t sam, int64_t sak,
              const float* __restrict__ B, int64_t sbn, int64_t sbk, const float* __restrict__ bias,
              float* __restrict__ C, int64_t ldc, int act, int64_t k_per_split, int64_t slab_stride,
              int64_t a_zs, int64_t b_zs, int zsplit, const dim3 tile, const bool final_out) {
  // dynamic LDS: one buffer per operand when the workgroup has a single K step (short-K streaming products: more
  // workgroups per CU hide each other's load latency), two otherwise
  extern __shared__ __attribute__((aligned(16))) float g_lds[];
  typedef Stage<G_BM, VW, ARF> StA;
  typedef Stage<BN, VW, BRF> StB;
  const int nbuf = k_per_split <= G_BK ? 1 : 2;
  float (*As)[G_BM][StA::LD] = reinterpret_cast<float (*)[G_BM][StA::LD]>(g_lds);
  float (*Bs)[BN][StB::LD] = reinterpret_cast<float (*)[BN][StB::LD]>(g_lds + (size_t)nbuf * G_BM * StA::LD);
  constexpr int NT = BN / 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // (an XCD-aware remap of the launch order — each L2 serving one K slice / one run of N tiles — measured 0.5 % slower
  // in the step: these operands are a few MB and the Infinity Cache already absorbs the re-reads)
  const int64_t m0 = (int64_t)tile.x * G_BM, n0 = (int64_t)tile.y * BN;
  // tile.z = (batch index) * zsplit + (K slice): K slices of one product (a_zs == b_zs == 0) and/or a batch
  // of independent products (a_zs/b_zs = element offsets per batch) whose slabs are all summed afterwards
  const int64_t bidx = tile.z / zsplit, ks_id = tile.z % zsplit;
  const int64_t k_begin = ks_id * k_per_split;
  const int64_t k_end = k_begin + k_per_split < K ? k_begin + k_per_split : K;
  A += bidx * a_zs;
  B += bidx * b_zs;

  G_PROBE(0);
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // software pipeline, PF tiles deep: while the matrix cores work on K-tile i (in LDS), the global loads of tiles
  // i+1 .. i+PF are in flight in registers.  Every load is unconditional (see g_zero_src): with loads under divergent
  // or uniform branches the compiler can no longer count how many are outstanding and waits for ALL of them before
  // the first LDS store, which turns a multi-stage pipeline back into one round trip per K step.
  StA la[PF];
  StB lb[PF];
  la[0].load(A, sam, sak, m0, M, k_begin, k_end);
  lb[0].load(B, sbn, sbk, n0, N, k_begin, k_end);
  la[0].store(As[0]);
  lb[0].store(Bs[0]);
  const int nsteps = (int)((k_end - k_begin + G_BK - 1) / G_BK);
  if (nsteps > 1) {                                    // block-uniform; ahead of the loop, so no path skips loads in it
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int64_t kk = k_begin + (int64_t)(s + 1) * G_BK;
      la[s].load(A, sam, sak, m0, M, kk, k_end);
      lb[s].load(B, sbn, sbk, n0, N, kk, k_end);
    }
  }
  __syncthreads();
  G_PROBE(1);
  int buf = 0;
  // one K step of the tile product.  All LDS reads of the step are issued first and the matrix instructions follow
  // back to back: with one wave per SIMD there is nobody else to cover a read -> multiply -> read -> multiply chain.
  auto mma_tile = [&](int bsel) {
    float af[G_BK / 4], bf[NT][G_BK / 4];
#pragma unroll
    for (int ks = 0; ks < G_BK / 4; ++ks) {
      af[ks] = As[bsel][w * 16 + (lane & 15)][ks * 4 + (lane >> 4)];
#pragma unroll
      for (int t = 0; t < NT; ++t) bf[t][ks] = Bs[bsel][t * 16 + (lane & 15)][ks * 4 + (lane >> 4)];
    }
    __builtin_amdgcn_sched_barrier(0);                 // the scheduler would otherwise re-interleave reads and multiplies
#pragma unroll
    for (int ks = 0; ks < G_BK / 4; ++ks) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        // operands swapped: the accumulator holds the TRANSPOSED tile, i.e. lane (g = lane>>4, j = lane&15) owns
        // C[m0 + 16 w + j][n0 + 16 t + 4 g + r], r = 0..3 — four consecutive columns of one row = one 16-byte store
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t][ks], af[ks], acc[t], 0, 0, 0);
      }
    }
  };
  // The K loop runs PF steps per trip so that every register stage has a compile-time index (moving an in-flight
  // stage into another register would wait for its loads), and it has ONE exit, at the top: with an exit per step the
  // compiler routes them all through the loop latch, sees a path from step s straight back to step s, and again
  // waits for every outstanding load.  The < PF left-over steps run after the loop as straight-line code.
  int i = 0;
#ifdef G_PROBE_ON
  long long g_acc[4] = {0, 0, 0, 0}, g_t = clock64();
#endif
  for (; i + PF <= nsteps && nsteps > 1; i += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      mma_tile(buf);
      G_ACC(0);
      la[s].store(As[buf ^ 1]);                        // stage s holds tile i+s+1, requested PF steps ago (all zero
      lb[s].store(Bs[buf ^ 1]);                        // past the end of the slice) ...
      G_ACC(1);
      const int64_t kk = k_begin + (int64_t)(i + s + 1 + PF) * G_BK;        // ... and is free for tile i+s+1+PF
      la[s].load(A, sam, sak, m0, M, kk, k_end);
      lb[s].load(B, sbn, sbk, n0, N, kk, k_end);
      G_ACC(2);
      __syncthreads();
      G_ACC(3);
      buf ^= 1;
    }
  }
#ifdef G_PROBE_ON
  if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int j = 0; j < 4; ++j) g_probe_buf[16 + j] = g_acc[j];
#endif
#pragma unroll
  for (int s = 0; s < PF; ++s) {                       // (s < PF - 1 steps when the loop ran; one when nsteps == 1)
    if (i + s >= nsteps) break;                        // block-uniform
    mma_tile(buf);
    if (i + s + 1 < nsteps) {                          // the tile this step stages is already in flight
      la[s].store(As[buf ^ 1]);
      lb[s].store(Bs[buf ^ 1]);
      __syncthreads();
      buf ^= 1;
    }
  }
  G_PROBE(2);
  // epilogue: 16 bytes per lane when the output rows allow it
  float* Cz = C + (int64_t)tile.z * slab_stride;
  const int64_t gm = m0 + w * 16 + (lane & 15);
  const bool c_vec = (ldc % 4 == 0) && (((uintptr_t)Cz & 15) == 0);
  if (gm < M) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int64_t gn = n0 + t * 16 + (lane >> 4) * 4;
      if (gn >= N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[t][r] + ((final_out && bias && gn + r < N) ? bias[gn + r] : 0.f);
        if (final_out && act == 1) v[r] = fmaxf(v[r], 0.f);
      }
      float* dst = Cz + gm * ldc + gn;
      if (c_vec && gn + 3 < N) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (gn + r < N) dst[r] = v[r];
      }
    }
  }
}

// [Tried: XCD-aware tile order — XCD c taking the tiles [c T/8, (c+1) T/8) of a product in order, so that the tiles
// sharing an operand block meet in one L2 (the two 256 x 256 x 2880 Gram products read 29.6 MB for 5.9 MB of operands by
// the PMC counters).  Every product of the two benchmarked steps got SLOWER by 0.3-1 us (Gram 15.3 -> 16.3 us): these
// launches are latency chains of a dozen K steps, not traffic-bound, and the re-reads come out of the Infinity Cache.]
template <int BN, int VW, int PF, bool ARF, bool BRF>
__global__ void __launch_bounds__(256)
k_gemm_f32(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t sam, int64_t sak,
           const float* __restrict__ B, int64_t sbn, int64_t sbk, const float* __restrict__ bias,
           float* __restrict__ C, int64_t ldc, int act, int64_t k_per_split, int64_t slab_stride,
           int64_t a_zs, int64_t b_zs, int zsplit) {
  gemm_f32_tile<BN, VW, PF, ARF, BRF>(M, N, K, A, sam, sak, B, sbn, sbk, bias, C, ldc, act, k_per_split, slab_stride,
                                      a_zs, b_zs, zsplit, blockIdx, gridDim.z == 1);
}

// Several products of DIFFERENT shapes in one launch (the dX and dW products of a linear layer's backward, which share
// the upstream gradient and nothing else): the workgroups of all problems form one flat grid, so the small grids of a
// backward pass fill the chip together instead of one after the other.  Tile width, vector width and pipeline depth are
// common to the group (host: the narrowest / safest of the members); the operand layouts are per problem.
#define GG_MAX 4
struct GemmProb {
  int64_t M, N, K;
  const float* A; int64_t sam, sak;
  const float* B; int64_t sbn, sbk;
  const float* bias; float* C; int64_t ldc; int act;
  int64_t kps, slab; int zsplit;
  int gx, gy, gz, wg0;                                  // tiles per axis; first workgroup of the problem in the flat grid
  int arf, brf;
};
struct GemmGroup { int n; GemmProb p[GG_MAX]; };

template <int BN, int VW, int PF>
__global__ void __launch_bounds__(256) k_gemm_f32_grouped(const GemmGroup G) {
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GG_MAX; ++i)
    if (i < G.n && (int)blockIdx.x >= G.p[i].wg0) pi = i;
  const GemmProb& p = G.p[pi];
  const int l = (int)blockIdx.x - p.wg0;
  const dim3 tile((unsigned)(l % p.gx), (unsigned)((l / p.gx) % p.gy), (unsigned)(l / (p.gx * p.gy)));
  const bool fin = p.gz == 1;
#define GG_BODY(ARFV, BRFV)                                                                                          \
  gemm_f32_tile<BN, VW, PF, ARFV, BRFV>(p.M, p.N, p.K, p.A, p.sam, p.sak, p.B, p.sbn, p.sbk, p.bias, p.C, p.ldc, p.act,  \
                                        p.kps, p.slab, 0, 0, p.zsplit, tile, fin)
  if (p.arf) { if (p.brf) GG_BODY(true, true); else GG_BODY(true, false); }
  else       { if (p.brf) GG_BODY(false, true); else GG_BODY(false, false); }
#undef GG_BODY
}

// -------------------------------------------------------------------------------------------------------------
// bf16 feature transforms (BASELINE configs[4]: "bf16 feature transforms on CDNA4 MFMA"): the same tiling with the
// operands rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) while they are staged into LDS, products on
// v_mfma_f32_16x16x32_bf16 (16x the rate of the fp32 matrix instruction), accumulation and output in fp32.
// Activations and weights stay fp32 in HBM, so the entry point has the signature of igcn_gemm_f32 and a model
// switches between the two with a flag.  LDS rows hold 32 bf16 + 8 pad (80 B): an operand fragment
// (row = lane & 15, k = 8 (lane >> 4) .. +7) is ONE 16-byte LDS read.
// -------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define G_LDB (G_BK + 8)

template <int ROWS, int VW>
__device__ __forceinline__ void store_tile_bf16(const TileLoader<ROWS, VW>& l, __bf16 (*T)[G_LDB], bool rfast) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < TileLoader<ROWS, VW>::NU; ++i) {
    const int f = tid + i * 256;
    if (f < ROWS * G_BK / VW) {
      const bool in = (l.ok >> i) & 1u;
      if (rfast) {
        const int r = (f % (ROWS / VW)) * VW, k = f / (ROWS / VW);
#pragma unroll
        for (int j = 0; j < VW; ++j) T[r + j][k] = (__bf16)(in ? l.v[i][j] : 0.f);
      } else {
        const int r = f / (G_BK / VW), k = (f % (G_BK / VW)) * VW;
#pragma unroll
        for (int j = 0; j < VW; ++j) T[r][k + j] = (__bf16)(in ? l.v[i][j] : 0.f);
      }
    }
  }
}

template <int BN, int VW>
__device__ __forceinline__ void
gemm_bf16_tile(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t sam, int64_t sak,
            const float* __restrict__ B, int64_t sbn, int64_t sbk, const float* __restrict__ bias,
            float* __restrict__ C, int64_t ldc, int act, int64_t k_per_split, int64_t slab_stride,
            int64_t a_zs, int64_t b_zs, int zsplit, const dim3 tile, const bool final_out) {
  extern __shared__ float g_lds[];
  const int nbuf = k_per_split <= G_BK ? 1 : 2;
  __bf16 (*As)[G_BM][G_LDB] = reinterpret_cast<__bf16 (*)[G_BM][G_LDB]>(g_lds);
  __bf16 (*Bs)[BN][G_LDB] = reinterpret_cast<__bf16 (*)[BN][G_LDB]>(reinterpret_cast<__bf16*>(g_lds) +
                                                                      (size_t)nbuf * G_BM * G_LDB);
  constexpr int NT = BN / 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t m0 = (int64_t)tile.x * G_BM, n0 = (int64_t)tile.y * BN;
  const int64_t bidx = tile.z / zsplit, ks_id = tile.z % zsplit;
  const int64_t k_begin = ks_id * k_per_split;
  const int64_t k_end = k_begin + k_per_split < K ? k_begin + k_per_split : K;
  A += bidx * a_zs;
  B += bidx * b_zs;
  const bool a_rfast = (sam == 1 && sak != 1), b_rfast = (sbn == 1 && sbk != 1);

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  TileLoader<G_BM, VW> la;
  TileLoader<BN, VW> lb;
  if (k_begin < k_end) {
    la.load(A, sam, sak, a_rfast, m0, M, k_begin, k_end);
    lb.load(B, sbn, sbk, b_rfast, n0, N, k_begin, k_end);
    store_tile_bf16<G_BM, VW>(la, As[0], a_rfast);
    store_tile_bf16<BN, VW>(lb, Bs[0], b_rfast);
  }
  __syncthreads();
  int buf = 0;
  for (int64_t kb = k_begin; kb < k_end; kb += G_BK) {
    const bool more = kb + G_BK < k_end;
    if (more) {
      la.load(A, sam, sak, a_rfast, m0, M, kb + G_BK, k_end);
      lb.load(B, sbn, sbk, b_rfast, n0, N, kb + G_BK, k_end);
    }
    // one 32-deep matrix instruction per accumulator tile; operands swapped as in k_gemm_f32 (transposed accumulator:
    // lane (g, j) owns C[m0 + 16 w + j][n0 + 16 t + 4 g + r], four consecutive columns of one row)
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(&As[buf][w * 16 + (lane & 15)][(lane >> 4) * 8]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Bs[buf][t * 16 + (lane & 15)][(lane >> 4) * 8]);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[t], 0, 0, 0);
    }
    if (more) {
      store_tile_bf16<G_BM, VW>(la, As[buf ^ 1], a_rfast);
      store_tile_bf16<BN, VW>(lb, Bs[buf ^ 1], b_rfast);
    }
    __syncthreads();
    buf ^= 1;
  }
  float* Cz = C + (int64_t)tile.z * slab_stride;
  const int64_t gm = m0 + w * 16 + (lane & 15);
  const bool c_vec = (ldc % 4 == 0) && (((uintptr_t)Cz & 15) == 0);
  if (gm < M) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int64_t gn = n0 + t * 16 + (lane >> 4) * 4;
      if (gn >= N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[t][r] + ((final_out && bias && gn + r < N) ? bias[gn + r] : 0.f);
        if (final_out && act == 1) v[r] = fmaxf(v[r], 0.f);
      }
      float* dst = Cz + gm * ldc + gn;
      if (c_vec && gn + 3 < N) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (gn + r < N) dst[r] = v[r];
      }
    }
  }
}

template <int BN, int VW>
__global__ void __launch_bounds__(256)
k_gemm_bf16(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t sam, int64_t sak,
            const float* __restrict__ B, int64_t sbn, int64_t sbk, const float* __restrict__ bias,
            float* __restrict__ C, int64_t ldc, int act, int64_t k_per_split, int64_t slab_stride,
            int64_t a_zs, int64_t b_zs, int zsplit) {
  gemm_bf16_tile<BN, VW>(M, N, K, A, sam, sak, B, sbn, sbk, bias, C, ldc, act, k_per_split, slab_stride, a_zs, b_zs, zsplit,
                         blockIdx, gridDim.z == 1);
}

// bf16-operand form of k_gemm_f32_grouped (the operand layouts are runtime properties of this kernel anyway)
template <int BN, int VW>
__global__ void __launch_bounds__(256) k_gemm_bf16_grouped(const GemmGroup G) {
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GG_MAX; ++i)
    if (i < G.n && (int)blockIdx.x >= G.p[i].wg0) pi = i;
  const GemmProb& p = G.p[pi];
  const int l = (int)blockIdx.x - p.wg0;
  const dim3 tile((unsigned)(l % p.gx), (unsigned)((l / p.gx) % p.gy), (unsigned)(l / (p.gx * p.gy)));
  gemm_bf16_tile<BN, VW>(p.M, p.N, p.K, p.A, p.sam, p.sak, p.B, p.sbn, p.sbk, p.bias, p.C, p.ldc, p.act, p.kps, p.slab, 0, 0,
                         p.zsplit, tile, p.gz == 1);
}

static size_t gemm_bf16_lds_bytes(int bn, int64_t k_per_split) {
  return (size_t)(k_per_split <= G_BK ? 1 : 2) * (G_BM + bn) * G_LDB * sizeof(__bf16);
}

// widest aligned vector (4, 2 or 1 floats) operand (ptr, row stride, k stride, rows, K) can be read with along its
// contiguous axis
static int vec_width(const float* p, int64_t srow, int64_t sk, int64_t rows, int64_t K) {
  for (int w = 4; w >= 2; w >>= 1) {
    if (((uintptr_t)p & (uintptr_t)(4 * w - 1)) != 0) continue;
    if (sk == 1 && srow % w == 0 && K % w == 0) return w;               // k contiguous
    if (srow == 1 && sk % w == 0 && rows % w == 0) return w;            // row contiguous
  }
  return 1;
}
static int min_int(int a, int b) { return a < b ? a : b; }
static size_t gemm_lds_bytes(int bn, int64_t k_per_split, bool a_rfast, bool b_rfast) {
  return (size_t)(k_per_split <= G_BK ? 1 : 2) * sizeof(float) *
         ((size_t)G_BM * (a_rfast ? G_BK + 2 : G_BK + 4) + (size_t)bn * (b_rfast ? G_BK + 2 : G_BK + 4));
}

// k_gemm_f32 instantiation table: tile width x vector width x pipeline depth x operand layouts.
struct GemmArgs {
  int64_t M, N, K;
  const float* A; int64_t sam, sak;
  const float* B; int64_t sbn, sbk;
  const float* bias; float* C; int64_t ldc; int act;
  int64_t kps, slab, a_zs, b_zs; int zsplit;
};
template <int BN, int VW, int PF, bool ARF, bool BRF>
static void launch_f32_t(dim3 grid, hipStream_t st, const GemmArgs& g) {
  hipLaunchKernelGGL((k_gemm_f32<BN, VW, PF, ARF, BRF>), grid, dim3(256), gemm_lds_bytes(BN, g.kps, ARF, BRF), st, g.M,
                     g.N, g.K, g.A, g.sam, g.sak, g.B, g.sbn, g.sbk, g.bias, g.C, g.ldc, g.act, g.kps, g.slab, g.a_zs,
                     g.b_zs, g.zsplit);
}
template <int BN, int VW, int PF>
static void launch_f32_l(dim3 grid, hipStream_t st, const GemmArgs& g) {
  const bool arf = (g.sam == 1 && g.sak != 1), brf = (g.sbn == 1 && g.sbk != 1);
  if (arf) { if (brf) launch_f32_t<BN, VW, PF, true, true>(grid, st, g); else launch_f32_t<BN, VW, PF, true, false>(grid, st, g); }
  else     { if (brf) launch_f32_t<BN, VW, PF, false, true>(grid, st, g); else launch_f32_t<BN, VW, PF, false, false>(grid, st, g); }
}
template <int BN, int VW>
static void launch_f32_p(dim3 grid, hipStream_t st, const GemmArgs& g) {
  const int64_t span = g.kps < g.K ? g.kps : g.K;
  if (igcn_cdiv(span, G_BK) <= 2) launch_f32_l<BN, VW, 1>(grid, st, g); else launch_f32_l<BN, VW, 2>(grid, st, g);
}
template <int BN>
static void launch_f32_v(int vw, dim3 grid, hipStream_t st, const GemmArgs& g) {
  if (vw == 4) launch_f32_p<BN, 4>(grid, st, g); else if (vw == 2) launch_f32_p<BN, 2>(grid, st, g); else launch_f32_p<BN, 1>(grid, st, g);
}
static void launch_f32(int bn, int vw, dim3 grid, hipStream_t st, const GemmArgs& g) {
  if (bn == 16) launch_f32_v<16>(vw, grid, st, g); else if (bn == 32) launch_f32_v<32>(vw, grid, st, g); else launch_f32_v<64>(vw, grid, st, g);
}

// sum over the split_k slabs of one output element, in slab order, eight loads in flight per batch.  [`#pragma unroll 8`
// on the plain loop: a 2- or 4-way split runs entirely in the remainder loop the compiler adds — one load per trip, each
// waiting out its own round trip.]  Slots past split_k re-read slab 0 and count as +0.
__device__ __forceinline__ float slab_sum(const float* __restrict__ p, int split_k, int64_t stride) {
  float t = 0.f;
  for (int z0 = 0; z0 < split_k; z0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float x = p[(int64_t)(z0 + u < split_k ? z0 + u : 0) * stride];
      v[u] = z0 + u < split_k ? x : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) t += v[u];
  }
  return t;
}

// (gridDim.y = independent products of a batch: slabs [batch][split_k][M*N], C + blockIdx.y * c_batch)
__global__ void k_gemm_splitk_reduce(int64_t M, int64_t N, int split_k, const float* __restrict__ slabs,
                                     const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int act,
                                     int64_t c_batch) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  slabs += (int64_t)blockIdx.y * split_k * M * N;
  C += (int64_t)blockIdx.y * c_batch;
  const int64_t m = i / N, n = i - m * N;
  const float t0 = slab_sum(slabs + i, split_k, M * N);
  float t = t0;
  if (bias) t += bias[n];
  if (act == 1) t = fmaxf(t, 0.f);
  C[m * ldc + n] = t;
}

// the slab sums of up to four split products of the SAME output shape in one launch (blockIdx.y = product)
// (round 5: the products may differ in shape — the heads' two [512, 64] sums and the Gram riders' two [256, 256] sums of
// one grouped launch were two launches of this kernel; gridDim.x covers the largest, the others' spare workgroups leave)
struct SlabSums { const float* slabs[4]; const float* bias[4]; float* C[4]; int split[4]; int64_t M[4], N[4], ldc[4]; int act[4]; };
__global__ void k_gemm_splitk_reduce_multi(SlabSums q) {
  const int64_t M = q.M[blockIdx.y], N = q.N[blockIdx.y], ldc = q.ldc[blockIdx.y];
  const int act = q.act[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const float* __restrict__ slabs = q.slabs[blockIdx.y];
  const float* __restrict__ bias = q.bias[blockIdx.y];
  float* __restrict__ C = q.C[blockIdx.y];
  const int split_k = q.split[blockIdx.y];
  const int64_t m = i / N, n = i - m * N;
  float t = slab_sum(slabs + i, split_k, M * N);
  if (bias) t += bias[n];
  if (act == 1) t = fmaxf(t, 0.f);
  C[m * ldc + n] = t;
}

// Launch shape, fitted to a sweep over the products of the train step (tools/gemm_sweep.py): the tile is 64 x BN.
// A wide tile only pays when there are tens of thousands of rows to stream; otherwise narrower tiles put more
// workgroups on the chip.  K is split so that ~512 workgroups exist, each keeping at least one 32-deep K step.
static int gemm_tile_n(int64_t M, int64_t N) {
  if (N <= 16) return 16;
  if (N <= 32) return 32;
  if (M >= 32768) return 64;
  return igcn_cdiv(M, G_BM) * igcn_cdiv(N, 32) < 128 ? 16 : 32;
}

extern "C" int igcn_gemm_f32_split_k(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K < 128) return 1;
  const int64_t tiles = igcn_cdiv(M, G_BM) * igcn_cdiv(N, gemm_tile_n(M, N));
  // a split costs a second pass over the slabs: it pays up to ~128 tiles (r02 sweep: 182-199 tiles ran 10-15 % faster
  // whole than in two slices), and ~512 workgroups — two per CU, covering each other's stalls — is where it stops
  // paying; beyond 256 slices the slab sum outweighs the shorter K loops.
  if (tiles > 128) return 1;
  int64_t sk = (512 + tiles / 2) / tiles;
  if (sk > 256) sk = 256;
  if (sk > K / G_BK) sk = K / G_BK;
  return (int)(sk < 1 ? 1 : sk);
}

// the slab sum behind a split product: deferred to the backward's flush (parameter gradients), a block-per-output tree
// (few outputs, many slabs), or the plain column walk with bias / activation
static int gemm_sum_slabs(int split_k, bool final_grad, int64_t M, int64_t N, float* scratch, const float* bias,
                          float* C, int64_t ldc, int act, hipStream_t st) {
  if (split_k <= 1) return IGCN_OK;
  if (final_grad && ldc == N && bias == nullptr && act == 0 && M * N < ((int64_t)1 << 31))
    return igcn_launch_reduce_rows_final(scratch, split_k, M * N, (int)(M * N), C, st);
  if (ldc == N && M * N <= 4096 && split_k > 32 && bias == nullptr && act == 0)
    return igcn_launch_reduce_rows(scratch, split_k, M * N, (int)(M * N), C, 0, st);   // block-per-output tree
  hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3((unsigned)igcn_cdiv(M * N, 256)), dim3(256), 0, st, M, N, split_k,
                     scratch, bias, C, ldc, act, (int64_t)0);
  IGCN_CHECK_LAUNCH("gemm_splitk_reduce");
  return IGCN_OK;
}

static int gemm_launch(bool bf16, int64_t M, int64_t N, int64_t K, const float* A, int64_t sam, int64_t sak,
                       const float* B, int64_t sbn, int64_t sbk, const float* bias, float* C, int64_t ldc, int act,
                       int split_k, float* scratch, void* stream) {
  const char* nm = bf16 ? "gemm_bf16" : "gemm_f32";
  const bool final_grad = (act & 0x100) != 0;     // the output is a parameter gradient: its split-K sum may be deferred
  const bool leave_slabs = (act & 0x200) != 0;    // a consumer sums the slabs itself (igcn_bn1d_fwd_slabs): no sum launch
  act &= 0xff;
  IGCN_REQUIRE(M > 0 && N > 0 && K >= 0 && split_k >= 1, "%s: bad sizes M=%lld N=%lld K=%lld split=%d", nm,
               (long long)M, (long long)N, (long long)K, split_k);
  IGCN_REQUIRE(split_k == 1 || scratch != nullptr, "%s: split_k>1 needs scratch", nm);
  IGCN_REQUIRE(!leave_slabs || (bias == nullptr && act == 0 && !final_grad && ldc == N),
               "%s: act | 0x200 (leave the slabs) takes no bias / activation / deferral and a dense C", nm);
  hipStream_t st = (hipStream_t)stream;
  if (split_k > K / G_BK) split_k = (int)(K / G_BK > 0 ? K / G_BK : 1);
  int64_t kps = igcn_cdiv(igcn_cdiv(K, split_k), G_BK) * G_BK;
  if (kps == 0) kps = G_BK;
  split_k = (int)igcn_cdiv(K > 0 ? K : 1, kps);
  const bool split = split_k > 1;
  float* out = split ? scratch : C;
  const int64_t ld = split ? N : ldc;
  const int64_t slab = split ? M * N : 0;
  int bn = gemm_tile_n(M, N);
  if (g_igcn_gemm_bn_cap > 0 && g_igcn_gemm_bn_cap < bn) bn = g_igcn_gemm_bn_cap;         // sweeps only (igcn_configure)
  dim3 grid((unsigned)igcn_cdiv(M, G_BM), (unsigned)igcn_cdiv(N, bn), (unsigned)split_k);
  const int vw = min_int(vec_width(A, sam, sak, M, K), vec_width(B, sbn, sbk, N, K));
  if (bf16) {
#define LAUNCH_G(BNV, VECV)                                                                                        \
  hipLaunchKernelGGL((k_gemm_bf16<BNV, VECV>), grid, dim3(256), gemm_bf16_lds_bytes(BNV, kps), st, M, N, K, A, sam,  \
                     sak, B, sbn, sbk, bias, out, ld, act, kps, slab, (int64_t)0, (int64_t)0, split_k)
    if (vw == 4) {
      if (bn == 16) { LAUNCH_G(16, 4); } else if (bn == 32) { LAUNCH_G(32, 4); } else { LAUNCH_G(64, 4); }
    } else if (vw == 2) {
      if (bn == 16) { LAUNCH_G(16, 2); } else if (bn == 32) { LAUNCH_G(32, 2); } else { LAUNCH_G(64, 2); }
    } else {
      if (bn == 16) { LAUNCH_G(16, 1); } else if (bn == 32) { LAUNCH_G(32, 1); } else { LAUNCH_G(64, 1); }
    }
#undef LAUNCH_G
  } else {
    const GemmArgs g = {M, N, K, A, sam, sak, B, sbn, sbk, bias, out, ld, act, kps, slab, 0, 0, split_k};
    launch_f32(bn, vw, grid, st, g);
  }
  IGCN_CHECK_LAUNCH(nm);
  if (leave_slabs) return IGCN_OK;                // igcn_gemm_effective_split(K, split_k) slabs in scratch (1: C is final)
  return gemm_sum_slabs(split_k, final_grad, M, N, scratch, bias, C, ldc, act, st);
}

// the number of K-slices a product asked to split `split_k` ways really runs in (whole 32-deep steps per slice): what a
// caller that sums the slabs itself (act | 0x200) has to read
extern "C" int igcn_gemm_effective_split(int64_t K, int split_k) {
  if (split_k < 1) split_k = 1;
  if (split_k > K / G_BK) split_k = (int)(K / G_BK > 0 ? K / G_BK : 1);
  int64_t kps = igcn_cdiv(igcn_cdiv(K, split_k), G_BK) * G_BK;
  if (kps == 0) kps = G_BK;
  return (int)igcn_cdiv(K > 0 ? K : 1, kps);
}

extern "C" int igcn_gemm_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t sam, int64_t sak,
                             const float* B, int64_t sbn, int64_t sbk, const float* bias, float* C, int64_t ldc,
                             int act, int split_k, float* scratch, void* stream) {
  return gemm_launch(false, M, N, K, A, sam, sak, B, sbn, sbk, bias, C, ldc, act, split_k, scratch, stream);
}

extern "C" int igcn_gemm_bf16(int64_t M, int64_t N, int64_t K, const float* A, int64_t sam, int64_t sak,
                              const float* B, int64_t sbn, int64_t sbk, const float* bias, float* C, int64_t ldc,
                              int act, int split_k, float* scratch, void* stream) {
  return gemm_launch(true, M, N, K, A, sam, sak, B, sbn, sbk, bias, C, ldc, act, split_k, scratch, stream);
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
  int vw = min_int(vec_width(A, sam, sak, M, K), vec_width(B, sbn, sbk, N, K));
  while (vw > 1 && (a_batch % vw != 0 || b_batch % vw != 0)) vw >>= 1;
  {
    const GemmArgs g = {M, N, K, A, sam, sak, B, sbn, sbk, nullptr, scratch, N, 0, kps, M * N, a_batch, b_batch, ksplit};
    launch_f32(bn, vw, grid, st, g);
  }
  IGCN_CHECK_LAUNCH("gemm_f32_batched_sum");
  if (ldc == N && M * N <= 4096 && slabs > 32)     // few outputs, many slabs: block-per-output tree reduce
    return igcn_launch_reduce_rows(scratch, slabs, M * N, (int)(M * N), C, 0, st);
  hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3((unsigned)igcn_cdiv(M * N, 256)), dim3(256), 0, st, M, N, slabs,
                     scratch, (const float*)nullptr, C, ldc, 0, (int64_t)0);
  IGCN_CHECK_LAUNCH("gemm_batched_reduce");
  return IGCN_OK;
}


// `batch` independent products of one shape in ONE launch (+ one slab sum when K is split):
//   C_z [M,N] = A_z B_z^T,  A_z = A + z a_batch, B_z = B + z b_batch, C_z = C + z c_batch   (element offsets)
// — the per-pass Gram matrices of the batched train step and their backward products: two half-empty grids and two
// slab sums become one launch each.  scratch: batch * split_k * M * N floats when split_k > 1.
extern "C" int igcn_gemm_f32_batched(int64_t M, int64_t N, int64_t K, int batch, const float* A, int64_t sam,
                                     int64_t sak, int64_t a_batch, const float* B, int64_t sbn, int64_t sbk,
                                     int64_t b_batch, float* C, int64_t c_batch, int64_t ldc, int split_k,
                                     float* scratch, void* stream) {
  IGCN_REQUIRE(M > 0 && N > 0 && K > 0 && batch >= 1 && split_k >= 1, "gemm_f32_batched: bad sizes");
  IGCN_REQUIRE(split_k == 1 || scratch != nullptr, "gemm_f32_batched: split_k > 1 needs scratch");
  hipStream_t st = (hipStream_t)stream;
  if (split_k > K / G_BK) split_k = (int)(K / G_BK > 0 ? K / G_BK : 1);
  int64_t kps = igcn_cdiv(igcn_cdiv(K, split_k), G_BK) * G_BK;
  split_k = (int)igcn_cdiv(K, kps);
  const bool split = split_k > 1;
  const int bn = gemm_tile_n(M, N);
  dim3 grid((unsigned)igcn_cdiv(M, G_BM), (unsigned)igcn_cdiv(N, bn), (unsigned)(batch * split_k));
  int vw = min_int(vec_width(A, sam, sak, M, K), vec_width(B, sbn, sbk, N, K));
  while (vw > 1 && (a_batch % vw != 0 || b_batch % vw != 0)) vw >>= 1;
  // split: slab (z, k) of the scratch, row stride N; unsplit: straight into C_z (the kernel's slab stride = c_batch)
  const GemmArgs g = {M, N, K, A, sam, sak, B, sbn, sbk, nullptr, split ? scratch : C, split ? N : ldc, 0, kps,
                      split ? M * N : c_batch, a_batch, b_batch, split_k};
  launch_f32(bn, vw, grid, st, g);
  IGCN_CHECK_LAUNCH("gemm_f32_batched");
  if (split) {
    // few outputs, many slabs (the 32 x 32 Gram matrices of configs[4]: 256 slabs each): a thread walking its output's
    // slabs is a chain of 32 dependent batches of loads (15.8 us for 2 MB); 16 x 16 tiles with 16 row groups instead
    if (ldc == N && M * N <= 4096 && split_k > 32)
      return igcn_launch_reduce_rows_batched(scratch, split_k, M * N, (int)(M * N), C, batch, (int64_t)split_k * M * N,
                                             c_batch, st);
    hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3((unsigned)igcn_cdiv(M * N, 256), (unsigned)batch), dim3(256), 0, st, M,
                       N, split_k, scratch, (const float*)nullptr, C, ldc, 0, c_batch);
    IGCN_CHECK_LAUNCH("gemm_f32_batched reduce");
  }
  return IGCN_OK;
}


// n (<= 4) products of different shapes in ONE launch: table [n][16] int64 =
//   {M, N, K, A, sam, sak, B, sbn, sbk, bias, C, ldc, act (| 0x100: parameter gradient), split_k, scratch, bf16}
// with the meaning of igcn_gemm_f32's arguments; bf16 != 0 (first problem's entry decides): operands rounded to bf16
// as in igcn_gemm_bf16.  Slab sums follow per problem as there.  Falls back to one launch per
// problem when the operands do not all allow at least 8-byte loads.
static size_t gemm_lds_bytes(int bn, int64_t k_per_split, bool a_rfast, bool b_rfast);
// ---- riders: products queued for a stream, carried by the NEXT grouped launch on it --------------------------------
// Two chains of a train step's forward end in products that are ready at the same time and depend on nothing of each
// other — the heads' first layers (lin1 | lin1_regr) and the per-pass Gram matrices of the batch losses — but are issued
// by different parts of the host code (the model's forward; the loss function).  The later one is QUEUED before the
// earlier one launches (igcn_gemm_rider, same table format) and joins its grid: one launch of four products instead of
// two launches of two.  igcn_gemm_rider_flush launches products nobody carried.
#include <map>
#include <mutex>
#include <vector>
static std::mutex g_gr_mutex;
static std::map<hipStream_t, std::vector<int64_t>> g_gemm_riders;      // 16 words per product

extern "C" int igcn_gemm_rider(void* stream, int n, const int64_t* table) {
  IGCN_REQUIRE(n >= 1 && n <= GG_MAX && table != nullptr, "gemm_rider: 1..%d products", GG_MAX);
  std::lock_guard<std::mutex> lk(g_gr_mutex);
  std::vector<int64_t>& q = g_gemm_riders[(hipStream_t)stream];
  IGCN_REQUIRE(q.size() / 16 + (size_t)n <= GG_MAX, "gemm_rider: more than %d products waiting on this stream", GG_MAX);
  q.insert(q.end(), table, table + 16 * n);
  return IGCN_OK;
}

static std::vector<int64_t> gemm_riders_take(hipStream_t st, int room, bool bf16) {
  std::lock_guard<std::mutex> lk(g_gr_mutex);
  auto it = g_gemm_riders.find(st);
  std::vector<int64_t> r;
  if (it == g_gemm_riders.end()) return r;
  // carried only whole, and only by a launch of the same operand type (the table's word 15)
  if ((int)(it->second.size() / 16) <= room && (it->second[15] != 0) == bf16) {
    r.swap(it->second);
    g_gemm_riders.erase(it);
  }
  return r;
}

void igcn_rider_dropout_cancel(hipStream_t st);         // plan.hip
// Forget every rider still waiting on the stream (mask job and products) WITHOUT launching it: for the start of a step
// that may follow one which failed between queueing a rider and the launch that would have carried it — the buffers it
// points at may be gone.
extern "C" int igcn_rider_cancel(void* stream) {
  igcn_rider_dropout_cancel((hipStream_t)stream);
  std::lock_guard<std::mutex> lk(g_gr_mutex);
  g_gemm_riders.erase((hipStream_t)stream);
  return IGCN_OK;
}

// Everything the library still holds for `stream` on the host side and has NOT launched: deferred reductions (entries),
// the dropout rider (0 / 1), product riders (products).  0 at the end of every step; see the header for each queue's
// ordering contract.
int igcn_reduce_pending_on(hipStream_t st);            // plan.hip
int igcn_rider_dropout_waiting(hipStream_t st);        // plan.hip
extern "C" int igcn_stream_pending(void* stream) {
  int n = igcn_reduce_pending_on((hipStream_t)stream) + igcn_rider_dropout_waiting((hipStream_t)stream);
  std::lock_guard<std::mutex> lk(g_gr_mutex);
  auto it = g_gemm_riders.find((hipStream_t)stream);
  if (it != g_gemm_riders.end()) n += (int)(it->second.size() / 16);
  return n;
}

extern "C" int igcn_gemm_f32_grouped(int n, const int64_t* table, void* stream);
extern "C" int igcn_gemm_rider_flush(void* stream) {
  std::vector<int64_t> r;
  {
    std::lock_guard<std::mutex> lk(g_gr_mutex);
    auto it = g_gemm_riders.find((hipStream_t)stream);
    if (it == g_gemm_riders.end()) return IGCN_OK;
    r.swap(it->second);
    g_gemm_riders.erase(it);
  }
  return igcn_gemm_f32_grouped((int)(r.size() / 16), r.data(), stream);
}

extern "C" int igcn_gemm_f32_grouped(int n, const int64_t* table, void* stream) {
  IGCN_REQUIRE(n >= 1 && n <= GG_MAX && table != nullptr, "gemm_f32_grouped: 1..%d problems", GG_MAX);
  hipStream_t st = (hipStream_t)stream;
  int64_t merged[16 * GG_MAX];
  {
    const std::vector<int64_t> r = gemm_riders_take(st, GG_MAX - n, table[15] != 0);
    if (!r.empty()) {                                  // the queued products join this launch
      for (int i = 0; i < 16 * n; ++i) merged[i] = table[i];
      for (size_t i = 0; i < r.size(); ++i) merged[16 * n + i] = r[i];
      n += (int)(r.size() / 16);
      table = merged;
    }
  }
  GemmGroup G;
  G.n = n;
  int bn = 64, vw = 4, pf = 1, wgs = 0, split_of[GG_MAX];
  bool fgrad[GG_MAX];
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t* t = table + 16 * i;
    GemmProb& p = G.p[i];
    p.M = t[0]; p.N = t[1]; p.K = t[2];
    p.A = (const float*)t[3]; p.sam = t[4]; p.sak = t[5];
    p.B = (const float*)t[6]; p.sbn = t[7]; p.sbk = t[8];
    p.bias = (const float*)t[9];
    float* C = (float*)t[10];
    const int64_t ldc = t[11];
    int act = (int)t[12], split_k = (int)t[13];
    float* scratch = (float*)t[14];
    fgrad[i] = (act & 0x100) != 0;
    act &= 0xff;
    IGCN_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && split_k >= 1 && (split_k == 1 || scratch != nullptr),
                 "gemm_f32_grouped: bad problem %d", i);
    if (split_k > p.K / G_BK) split_k = (int)(p.K / G_BK > 0 ? p.K / G_BK : 1);
    const int64_t kps = igcn_cdiv(igcn_cdiv(p.K, split_k), G_BK) * G_BK;
    split_k = (int)igcn_cdiv(p.K, kps);
    split_of[i] = split_k;
    const bool split = split_k > 1;
    p.C = split ? scratch : C;
    p.ldc = split ? p.N : ldc;
    p.act = act;
    p.kps = kps;
    p.slab = split ? p.M * p.N : 0;
    p.zsplit = split_k;
    p.arf = (p.sam == 1 && p.sak != 1);
    p.brf = (p.sbn == 1 && p.sbk != 1);
    const int b = gemm_tile_n(p.M, p.N);
    bn = b < bn ? b : bn;
    const int v = min_int(vec_width(p.A, p.sam, p.sak, p.M, p.K), vec_width(p.B, p.sbn, p.sbk, p.N, p.K));
    vw = v < vw ? v : vw;
    if (igcn_cdiv(kps < p.K ? kps : p.K, G_BK) > 2) pf = 2;
  }
  const bool bf16 = table[15] != 0;
  if (vw < 2 || n == 1) {                              // not groupable: the ordinary launches, one after the other
    for (int i = 0; i < n; ++i) {
      const int64_t* t = table + 16 * i;
      const int rc = gemm_launch(bf16, t[0], t[1], t[2], (const float*)t[3], t[4], t[5], (const float*)t[6], t[7], t[8],
                                 (const float*)t[9], (float*)t[10], t[11], (int)t[12], (int)t[13], (float*)t[14], stream);
      if (rc) return rc;
    }
    return IGCN_OK;
  }
  for (int i = 0; i < n; ++i) {
    GemmProb& p = G.p[i];
    p.gx = (int)igcn_cdiv(p.M, G_BM);
    p.gy = (int)igcn_cdiv(p.N, bn);
    p.gz = p.zsplit;
    p.wg0 = wgs;
    wgs += p.gx * p.gy * p.gz;
    const size_t l = bf16 ? gemm_bf16_lds_bytes(bn, p.kps) : gemm_lds_bytes(bn, p.kps, p.arf, p.brf);
    lds = l > lds ? l : lds;
  }
  for (int i = n; i < GG_MAX; ++i) G.p[i] = G.p[0];
#define LAUNCH_GG(BNV, PFV)                                                                                          \
  do {                                                                                                               \
    if (vw == 4) hipLaunchKernelGGL((k_gemm_f32_grouped<BNV, 4, PFV>), dim3((unsigned)wgs), dim3(256), lds, st, G);   \
    else hipLaunchKernelGGL((k_gemm_f32_grouped<BNV, 2, PFV>), dim3((unsigned)wgs), dim3(256), lds, st, G);           \
  } while (0)
#define LAUNCH_GB(BNV)                                                                                             \
  do {                                                                                                             \
    if (vw == 4) hipLaunchKernelGGL((k_gemm_bf16_grouped<BNV, 4>), dim3((unsigned)wgs), dim3(256), lds, st, G);      \
    else hipLaunchKernelGGL((k_gemm_bf16_grouped<BNV, 2>), dim3((unsigned)wgs), dim3(256), lds, st, G);              \
  } while (0)
  if (bf16) { if (bn == 16) LAUNCH_GB(16); else if (bn == 32) LAUNCH_GB(32); else LAUNCH_GB(64); }
  else if (pf == 1) { if (bn == 16) LAUNCH_GG(16, 1); else if (bn == 32) LAUNCH_GG(32, 1); else LAUNCH_GG(64, 1); }
  else              { if (bn == 16) LAUNCH_GG(16, 2); else if (bn == 32) LAUNCH_GG(32, 2); else LAUNCH_GG(64, 2); }
#undef LAUNCH_GB
#undef LAUNCH_GG
  IGCN_CHECK_LAUNCH("gemm_f32_grouped");
  // slab sums: the plain column-walk ones share ONE launch, whatever their output shapes
  bool done[GG_MAX] = {false, false, false, false};
  {
    SlabSums q = {};
    int cnt = 0;
    int64_t blocks = 0;
    for (int k = 0; k < n; ++k) {
      const int64_t* u = table + 16 * k;
      const bool tree = u[11] == u[1] && u[0] * u[1] <= 4096 && split_of[k] > 32 && u[9] == 0 && ((int)u[12] & 0xff) == 0;
      if (!(split_of[k] > 1 && !fgrad[k] && !tree)) continue;
      q.slabs[cnt] = (const float*)u[14]; q.bias[cnt] = (const float*)u[9]; q.C[cnt] = (float*)u[10];
      q.split[cnt] = split_of[k];
      q.M[cnt] = u[0]; q.N[cnt] = u[1]; q.ldc[cnt] = u[11]; q.act[cnt] = (int)u[12] & 0xff;
      const int64_t b = igcn_cdiv(u[0] * u[1], 256);
      blocks = b > blocks ? b : blocks;
      done[k] = true;
      ++cnt;
    }
    if (cnt > 0) {
      hipLaunchKernelGGL(k_gemm_splitk_reduce_multi, dim3((unsigned)blocks, (unsigned)cnt), dim3(256), 0, st, q);
      IGCN_CHECK_LAUNCH("gemm_splitk_reduce_multi");
    }
  }
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    const int64_t* t = table + 16 * i;
    const int rc = gemm_sum_slabs(split_of[i], fgrad[i], t[0], t[1], (float*)t[14], (const float*)t[9], (float*)t[10], t[11],
                                  (int)t[12] & 0xff, st);
    if (rc) return rc;
  }
  return IGCN_OK;
}
