// Graph-diffusion pre-transform of a batch of dense brain adjacencies, on the device:
//   util_gdc.py:7-15   get_ppr_matrix     P = alpha * (I - (1-alpha) * D^-1/2 A D^-1/2)^-1,  D = diag(row sums of A)
//   util_gdc.py:25-31  get_top_k_matrix   keep the k largest entries of every column, zero the rest, divide each
//                                         column by its sum (columns whose sum is <= 0 are left unscaled)
//   util_gdc.py:82-86  coo_matrix(A_res)  non-zeros in row-major order, edge_index = [row; col], edge_attr = f32
// plus the block-diagonal node offset of Batch.from_data_list (batch.py:98-104): graph g's indices are shifted by g*R.
//
// One workgroup per graph; the R x R system lives in LDS in fp64 (the reference computes in numpy float64 and the
// top-k SELECTION must not flip on rounding), inverted in place by Gauss-Jordan elimination with partial pivoting.
// Graph g owns the output slots [g*R*k, (g+1)*R*k): entries in (row, col) order, then (-1, -1, 0) padding when
// fewer than R*k non-zeros survive (a kept entry that is exactly 0 is not an edge for coo_matrix); counts[g] = #edges.
#include "common.h"

#define GDC_T 256

__device__ __forceinline__ int gdc_ld(int R) { return R | 1; }      // odd row stride (in doubles)

__global__ void __launch_bounds__(GDC_T)
k_gdc_topk(int R, int k, double alpha, const float* __restrict__ A, int64_t* __restrict__ ei_row,
           int64_t* __restrict__ ei_col, float* __restrict__ ew, int32_t* __restrict__ counts) {
  extern __shared__ double smem_d[];
  const int ld = gdc_ld(R), g = blockIdx.x, tid = threadIdx.x;
  double* M = smem_d;                                   // [R][ld]
  double* colf = M + (size_t)R * ld;                    // [R]  factors of the pivot column / dinv
  double* cand = colf + R;                              // [GDC_T] arg-max scratch (values)
  int* candi = (int*)(cand + GDC_T);                    // [GDC_T] arg-max scratch (rows)
  int* piv = candi + GDC_T;                             // [R]  row swapped with p at step p
  int* rowcnt = piv + R;                                // [R+1]
  unsigned char* keep = (unsigned char*)(rowcnt + R + 1);   // [R][R]
  const float* a = A + (int64_t)g * R * R;

  // ---- H = D^-1/2 A D^-1/2, M = I - (1 - alpha) H -------------------------------------------------
  for (int i = tid; i < R; i += GDC_T) {
    double s = 0.0;
    for (int j = 0; j < R; ++j) s += (double)a[i * R + j];
    colf[i] = 1.0 / sqrt(s);
  }
  __syncthreads();
  for (int t = tid; t < R * R; t += GDC_T) {
    const int i = t / R, j = t % R;
    const double h = colf[i] * (double)a[t] * colf[j];
    M[i * ld + j] = (i == j ? 1.0 : 0.0) - (1.0 - alpha) * h;
    keep[t] = 0;
  }
  __syncthreads();

  // ---- in-place Gauss-Jordan inversion with partial pivoting --------------------------------------
  int cw = 8;
  while (cw < R && cw < GDC_T) cw <<= 1;               // columns per pass (power of two >= R, R <= 128 < GDC_T)
  const int jc = tid & (cw - 1), i0 = tid / cw;
  for (int p = 0; p < R; ++p) {
    // partial-pivot search by ONE wave (shuffle arg-max, lowest row wins ties): one barrier instead of a
    // log2(256)-step LDS tree per pivot
    if (tid < 64) {
      double best = -1.0;
      int bi = p;
      for (int i = p + tid; i < R; i += 64) {
        const double v = fabs(M[i * ld + p]);
        if (v > best) { best = v; bi = i; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (tid == 0) candi[0] = bi;
    }
    __syncthreads();
    const int r = candi[0];
    if (tid == 0) piv[p] = r;
    if (r != p)
      for (int j = tid; j < R; j += GDC_T) {
        const double t0 = M[p * ld + j];
        M[p * ld + j] = M[r * ld + j];
        M[r * ld + j] = t0;
      }
    __syncthreads();
    const double inv = 1.0 / M[p * ld + p];
    for (int i = tid; i < R; i += GDC_T) colf[i] = M[i * ld + p];
    __syncthreads();
    for (int j = tid; j < R; j += GDC_T) M[p * ld + j] = (j == p ? 1.0 : M[p * ld + j]) * inv;
    __syncthreads();
    // rank-1 update of all other rows; thread = (row slice, column) with a power-of-two column count so that the
    // index arithmetic is shifts and masks (an integer division per element costs more than the update itself)
    if (jc < R) {
      const double pj = M[p * ld + jc];
      for (int i = i0; i < R; i += GDC_T / cw) {
        if (i == p) continue;
        const double base = (jc == p) ? 0.0 : M[i * ld + jc];
        M[i * ld + jc] = base - colf[i] * pj;
      }
    }
    __syncthreads();
  }
  for (int p = R - 1; p >= 0; --p) {                    // undo the row swaps as column swaps, last first
    const int r = piv[p];
    if (r != p)
      for (int i = tid; i < R; i += GDC_T) {
        const double t0 = M[i * ld + p];
        M[i * ld + p] = M[i * ld + r];
        M[i * ld + r] = t0;
      }
    __syncthreads();
  }

  // ---- top-k per column (thread = column), column normalisation ------------------------------------
  for (int j = tid; j < R; j += GDC_T) {
    double norm = 0.0;
    for (int s = 0; s < k && s < R; ++s) {
      double best = 0.0;
      int bi = -1;
      for (int i = 0; i < R; ++i) {
        if (keep[i * R + j]) continue;
        const double v = alpha * M[i * ld + j];
        if (bi < 0 || v >= best) { best = v; bi = i; }   // ties: the larger row (stable ascending argsort keeps it)
      }
      keep[bi * R + j] = 1;
      norm += best;
    }
    colf[j] = (norm <= 0.0) ? 1.0 : norm;
  }
  __syncthreads();

  // ---- COO emission in row-major order --------------------------------------------------------------
  for (int i = tid; i < R; i += GDC_T) {
    int c = 0;
    for (int j = 0; j < R; ++j)
      if (keep[i * R + j] && (alpha * M[i * ld + j]) / colf[j] != 0.0) ++c;
    rowcnt[i + 1] = c;
  }
  if (tid == 0) rowcnt[0] = 0;
  __syncthreads();
  if (tid == 0) {
    for (int i = 0; i < R; ++i) rowcnt[i + 1] += rowcnt[i];
    counts[g] = rowcnt[R];
  }
  __syncthreads();
  const int64_t slot0 = (int64_t)g * R * k, off = (int64_t)g * R;
  for (int i = tid; i < R; i += GDC_T) {
    int64_t s = slot0 + rowcnt[i];
    for (int j = 0; j < R; ++j) {
      if (!keep[i * R + j]) continue;
      const double v = (alpha * M[i * ld + j]) / colf[j];
      if (v == 0.0) continue;
      ei_row[s] = off + i;
      ei_col[s] = off + j;
      ew[s] = (float)v;
      ++s;
    }
  }
  for (int64_t s = slot0 + rowcnt[R] + tid; s < slot0 + (int64_t)R * k; s += GDC_T) {
    ei_row[s] = -1;
    ei_col[s] = -1;
    ew[s] = 0.f;
  }
}

static size_t gdc_lds_bytes(int R) {
  const size_t ld = (size_t)(R | 1);
  size_t b = ((size_t)R * ld + R + GDC_T) * sizeof(double);
  b += ((size_t)GDC_T + R + R + 1) * sizeof(int);
  b += (size_t)R * R;
  return (b + 15) & ~(size_t)15;
}

extern "C" int igcn_gdc_topk_max_rois(void) {
  int r = 8;
  while (gdc_lds_bytes(r + 1) <= 160 * 1024) ++r;
  return r;
}

extern "C" int igcn_gdc_topk(int B, int R, int k, double alpha, const float* A, int64_t* edge_index, float* edge_attr,
                             int32_t* counts, void* stream) {
  IGCN_REQUIRE(B >= 0 && R > 0 && k > 0 && k <= R, "gdc_topk: bad sizes B=%d R=%d k=%d", B, R, k);
  IGCN_REQUIRE(alpha > 0.0 && alpha <= 1.0, "gdc_topk: alpha=%g outside (0,1]", alpha);
  const size_t lds = gdc_lds_bytes(R);
  if (lds > 160 * 1024) {
    igcn_set_error("gdc_topk: R=%d needs %zu bytes of LDS (max R = %d)", R, lds, igcn_gdc_topk_max_rois());
    return IGCN_ERR_UNSUPPORTED;
  }
  if (B == 0) return IGCN_OK;
  IGCN_ALLOW_BIG_LDS(k_gdc_topk);
  const int64_t slots = (int64_t)B * R * k;
  hipLaunchKernelGGL(k_gdc_topk, dim3(B), dim3(GDC_T), lds, (hipStream_t)stream, R, k, alpha, A, edge_index,
                     edge_index + slots, edge_attr, counts);
  IGCN_CHECK_LAUNCH("gdc_topk");
  return IGCN_OK;
}
