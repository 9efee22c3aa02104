// Backward of the two heads' first layers (kernel/sgcn_img_snp.py:299 lin1, :302 lin1_regr: y = relu(x W^T + b) with
// x [R, C] = the head inputs, C ~ 3000 wide, W [64, C]) in ONE pass over the wide operands:
//   g  = dy * (y > 0)                [R, 64]   ReLU mask
//   db = column sums of g            [64]
//   dx = g W                         [R, C]
//   dW = g^T x                       [64, C]
// As four products of the general tiled GEMM (one grouped launch) these ran 32 us for 27 MB: the group's tile is the
// narrowest any member wants (16 columns) and its loads 8 bytes wide (C = 3182 leaves rows only 8-byte aligned), and a
// ReLU-mask / bias-gradient launch went in front.  Here a workgroup owns a 32-column block of x / W / dx / dW and 128
// rows: the W slab stays in LDS, g and x go through LDS in 64-row tiles (the next tile's loads are in flight while the
// current one is multiplied), dx tiles leave through LDS as whole 128-byte row pieces, dW accumulates in registers across the row
// tiles; the block-0 workgroups also sum g's columns.  x and W are read once, dx written once; g (128 KB) is re-read by
// every column block from L2.  Row splits leave partials of dW / db for a final (deferrable) reduction.
// Exact fp32 (v_mfma_f32_16x16x4_f32).  Hidden width 64, C even; anything else takes the grouped GEMM.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HB_H 64
#define HB_CB 32            // columns per workgroup
#define HB_ROWS 64          // rows per step
#define HB_RPW 128          // rows per workgroup
#define HB_LG (HB_H + 4)    // g tile row stride: operand reads of 16 rows x 4 k spread over all banks
#define HB_LX 48            // x / W slab row stride: reads of 4 k-rows x 16 columns spread over all banks

struct HbArgs {
  int R, C;
  const float *dy, *y, *W, *X;      // y may be NULL (no ReLU)
  float *dX, *dWp, *dbp;            // dWp [rsplit][64][C], dbp [rsplit][64]
  int cblocks, rsplit;
};

extern "C" int igcn_head_bwd_supported(int R, int H, int C) { return R > 0 && H == HB_H && C >= 2 && C % 2 == 0; }
static int hb_rsplit(int R) { return (R + HB_RPW - 1) / HB_RPW; }
// scratch floats for one problem: the row splits' partials of dW and db (nothing when one split covers the rows)
extern "C" size_t igcn_head_bwd_scratch_floats(int R, int C) {
  const int rs = hb_rsplit(R);
  return rs > 1 ? (size_t)rs * HB_H * ((size_t)C + 1) + 16 : 16;
}

__device__ __forceinline__ void head_bwd_body(float* lds, const HbArgs& a, int bid) {
  float (*Ws)[HB_LX] = reinterpret_cast<float (*)[HB_LX]>(lds);                             // [64 h][32 cols]
  float (*Xs)[HB_LX] = reinterpret_cast<float (*)[HB_LX]>(lds + HB_H * HB_LX);              // [64 rows][32 cols]
  float (*Gs)[HB_LG] = reinterpret_cast<float (*)[HB_LG]>(lds + (HB_H + HB_ROWS) * HB_LX);  // [64 rows][64 h]
  float (*Ds)[HB_CB + 4] = reinterpret_cast<float (*)[HB_CB + 4]>(lds + (HB_H + HB_ROWS) * HB_LX + HB_ROWS * HB_LG);   // dx tile
  const int R = a.R, C = a.C;
  const int cb = bid % a.cblocks, rs = bid / a.cblocks, c0 = cb * HB_CB;
  const int r_begin = rs * HB_RPW, r_end = min(R, r_begin + HB_RPW);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
  const float* __restrict__ dy = a.dy;
  const float* __restrict__ yr = a.y;
  const float* __restrict__ X = a.X;
  float* __restrict__ dX = a.dX;
  // W slab (zero columns past C)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = tid + 256 * k, r = i >> 4, c2 = i & 15, col = c0 + 2 * c2;
    const float2 v = col < C ? *reinterpret_cast<const float2*>(a.W + (int64_t)r * C + col) : make_float2(0.f, 0.f);
    *reinterpret_cast<float2*>(&Ws[r][2 * c2]) = v;
  }
  float4 gq[4];
  float2 xq[4];
  auto load = [&](int t0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + 256 * k, r = i >> 4, c = i & 15, row = t0 + r, col = c0 + 2 * c;
      gq[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      xq[k] = make_float2(0.f, 0.f);
      if (row < r_end) {
        gq[k] = *reinterpret_cast<const float4*>(dy + (int64_t)row * HB_H + 4 * c);
        if (yr) {
          const float4 u = *reinterpret_cast<const float4*>(yr + (int64_t)row * HB_H + 4 * c);
          gq[k].x = u.x > 0.f ? gq[k].x : 0.f; gq[k].y = u.y > 0.f ? gq[k].y : 0.f;
          gq[k].z = u.z > 0.f ? gq[k].z : 0.f; gq[k].w = u.w > 0.f ? gq[k].w : 0.f;
        }
        if (col < C) xq[k] = *reinterpret_cast<const float2*>(X + (int64_t)row * C + col);
      }
    }
  };
  load(r_begin);
  f32x4 dwa[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float dbacc = 0.f;
  for (int t0 = r_begin; t0 < r_end; t0 += HB_ROWS) {
    __syncthreads();                                    // the previous tile is fully consumed (and Ws is in)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + 256 * k, r = i >> 4, c = i & 15;
      *reinterpret_cast<float4*>(&Gs[r][4 * c]) = gq[k];
      *reinterpret_cast<float2*>(&Xs[r][2 * c]) = xq[k];
    }
    __syncthreads();
    if (t0 + HB_ROWS < r_end) load(t0 + HB_ROWS);
    if (cb == 0 && tid < HB_H) {                        // bias gradient: the tile's column sums, rows in order
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < HB_ROWS; ++r) s += Gs[r][tid];
      dbacc += s;
    }
    // dx tile, TRANSPOSED accumulator: dx^T[col][row] = sum_h W[h][col] g[row][h] — lane (g, n) owns four consecutive
    // columns 16 t + 4 g .. + 3 of row 16 w + n
    f32x4 dx[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < HB_H / 4; ++kk) {
      const float b = Gs[16 * w + n][4 * kk + g];
#pragma unroll
      for (int t = 0; t < 2; ++t) dx[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ws[4 * kk + g][16 * t + n], b, dx[t], 0, 0, 0);
    }
    // out through LDS: 16 lanes write one row's 128 contiguous bytes (rows are only 8-byte aligned when C % 4 != 0, so
    // straight from the accumulators a store instruction would put 8 bytes into every other 16-byte slot of 16 rows)
#pragma unroll
    for (int t = 0; t < 2; ++t)
      *reinterpret_cast<float4*>(&Ds[16 * w + n][16 * t + 4 * g]) = make_float4(dx[t][0], dx[t][1], dx[t][2], dx[t][3]);
    // dW += g_tile^T x_tile over the 64 rows: A[i = h 16 w + n][k = row 4 kk + g], B[k][col 16 t + n]
#pragma unroll
    for (int kk = 0; kk < HB_ROWS / 4; ++kk) {
      const float av = Gs[4 * kk + g][16 * w + n];
#pragma unroll
      for (int t = 0; t < 2; ++t) dwa[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Xs[4 * kk + g][16 * t + n], dwa[t], 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + 256 * k, r = i >> 4, c = i & 15, row = t0 + r, col = c0 + 2 * c;
      if (row < r_end && col < C) *reinterpret_cast<float2*>(dX + (int64_t)row * C + col) = *reinterpret_cast<const float2*>(&Ds[r][2 * c]);
    }
  }
  // this split's dW block: accumulator lane (g, n), register r = (h 16 w + 4 g + r, column c0 + 16 t + n)
  float* out = a.dWp + (int64_t)rs * HB_H * C;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = c0 + 16 * t + n;
    if (col < C) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(int64_t)(16 * w + 4 * g + r) * C + col] = dwa[t][r];
    }
  }
  if (cb == 0 && tid < HB_H) a.dbp[rs * HB_H + tid] = dbacc;
}

#define HB_LDS_FLOATS ((HB_H + HB_ROWS) * HB_LX + HB_ROWS * HB_LG + HB_ROWS * (HB_CB + 4))
__global__ void __launch_bounds__(256) k_head_bwd(HbArgs a, HbArgs b, int ablocks) {
  __shared__ __attribute__((aligned(16))) float lds[HB_LDS_FLOATS];
  if ((int)blockIdx.x < ablocks)
    head_bwd_body(lds, a, blockIdx.x);
  else
    head_bwd_body(lds, b, blockIdx.x - ablocks);
}

static int hb_check(int R, int H, int C, const float* dy, const float* y, const float* W, const float* X, float* dX,
                    float* dW, float* db, float* scratch) {
  if (!igcn_head_bwd_supported(R, H, C)) {
    igcn_set_error("head_bwd: needs hidden width 64 and an even input width (R=%d H=%d C=%d)", R, H, C);
    return IGCN_ERR_UNSUPPORTED;
  }
  IGCN_REQUIRE(dy && W && X && dX && dW && db && scratch, "head_bwd: null operand");
  IGCN_REQUIRE((((uintptr_t)dy | (uintptr_t)y) & 15) == 0 && (((uintptr_t)W | (uintptr_t)X | (uintptr_t)dX) & 7) == 0,
               "head_bwd: dy / y must be 16-byte aligned, W / x / dx 8-byte aligned");
  return IGCN_OK;
}

static HbArgs hb_args(int R, int C, const float* dy, const float* y, const float* W, const float* X, float* dX, float* dW,
                      float* db, float* scratch) {
  const int rs = hb_rsplit(R);
  HbArgs a = {R, C, dy, y, W, X, dX, rs > 1 ? scratch : dW, rs > 1 ? scratch + (size_t)rs * HB_H * C : db,
              (C + HB_CB - 1) / HB_CB, rs};
  return a;
}

// Two layers with the same row count and hidden width in one launch (C2 = 0: the first alone).  y_i NULL: no ReLU.
// dW_i, db_i are FINAL reductions in the sense of igcn_reduce_defer when the rows are split (R > 128); scratch_i:
// igcn_head_bwd_scratch_floats(R, C_i) floats.
extern "C" int igcn_head_bwd_pair(int R, int H, int C1, const float* dy1, const float* y1, const float* W1,
                                  const float* X1, float* dX1, float* dW1, float* db1, float* scratch1, int C2,
                                  const float* dy2, const float* y2, const float* W2, const float* X2, float* dX2,
                                  float* dW2, float* db2, float* scratch2, void* stream) {
  int rc = hb_check(R, H, C1, dy1, y1, W1, X1, dX1, dW1, db1, scratch1);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const HbArgs a = hb_args(R, C1, dy1, y1, W1, X1, dX1, dW1, db1, scratch1);
  HbArgs b = {};
  int bblocks = 0;
  if (C2 > 0) {
    rc = hb_check(R, H, C2, dy2, y2, W2, X2, dX2, dW2, db2, scratch2);
    if (rc) return rc;
    b = hb_args(R, C2, dy2, y2, W2, X2, dX2, dW2, db2, scratch2);
    bblocks = b.cblocks * b.rsplit;
  }
  const int ablocks = a.cblocks * a.rsplit;
  hipLaunchKernelGGL(k_head_bwd, dim3(ablocks + bblocks), dim3(256), 0, st, a, b, ablocks);
  IGCN_CHECK_LAUNCH("head_bwd_pair");
  if (a.rsplit > 1) {
    if ((rc = igcn_launch_reduce_rows_final(a.dWp, a.rsplit, (int64_t)HB_H * C1, HB_H * C1, dW1, st))) return rc;
    if ((rc = igcn_launch_reduce_rows_final(a.dbp, a.rsplit, HB_H, HB_H, db1, st))) return rc;
    if (C2 > 0) {
      if ((rc = igcn_launch_reduce_rows_final(b.dWp, b.rsplit, (int64_t)HB_H * C2, HB_H * C2, dW2, st))) return rc;
      if ((rc = igcn_launch_reduce_rows_final(b.dbp, b.rsplit, HB_H, HB_H, db2, st))) return rc;
    }
  }
  return IGCN_OK;
}
