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
//
// The elimination is a chain of R dependent rank-1 updates: its time is instruction issue and barriers, not arithmetic.
// Round 3 kept the matrix in LDS and spent five barriers and ~45 dependent LDS read-modify-writes per thread on every
// pivot (6.5 us per pivot, 590 us per batch of 256 graphs).  Now:
//   * the matrix lives in REGISTERS for the whole elimination — thread = (row slice i0, column jc) holds the NR
//     consecutive rows i0 NR .. i0 NR + NR - 1 of its column;
//   * rows are never swapped (implicit pivoting): the pivot of column p is the largest entry among the rows that have
//     not been a pivot yet (lowest row on ties), the working matrix W ends as P^-1 A^-1 P^-1 and is written back to LDS
//     through the two index maps, so the selection phases read the inverse itself;
//   * a pivot costs two barriers and ~3 instructions per row: (A) every WAVE finds the pivot for itself — lane l looks
//     at the multipliers of rows l and l + 64 (in LDS), a six-step wave maximum and a ballot give the largest
//     magnitude and its lowest row — and the slice that owns row r stages it in LDS; (B) every thread updates its
//     registers with one fma per row — the thread of column p, whose registers ARE the multipliers, gets the in-place
//     inverse column from the same fma with the factor 1 + 1/pivot — and the thread of column p + 1 leaves the next
//     multipliers in LDS.
#include "common.h"

#define GDC_T 256

__device__ __forceinline__ int gdc_ld(int R) { return R | 1; }      // odd row stride (in doubles)

// reg[kr] with a wave-uniform kr: a scalar switch (one taken branch) instead of NR compare-selects per access
#define GDC_ROWS(X)                                                                                                \
  X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20)  \
  X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39)    \
  X(40) X(41) X(42) X(43) X(44) X(45) X(46) X(47) X(48) X(49) X(50) X(51) X(52) X(53) X(54) X(55) X(56) X(57) X(58)    \
  X(59) X(60) X(61) X(62) X(63)
#define GDC_GET(K) case K: if constexpr (K < NR) v = reg[K]; break;
#define GDC_SET(K) case K: if constexpr (K < NR) reg[K] = newp; break;

#ifdef GDC_PROBE_ON
// phase stamps of workgroup 0, wave 0 (tools/gdc_probe.py, IGCN_HIPCC_EXTRA=-DGDC_PROBE_ON)
__device__ long long gdc_probe_buf[16];
#define GDC_PROBE(i)                                                                    \
  do {                                                                                  \
    if (threadIdx.x == 0 && blockIdx.x == 0) gdc_probe_buf[i] = wall_clock64();         \
  } while (0)
// inside the pivot loop: time since the previous stamp, accumulated per stage (wave 0 of workgroup 0)
#define GDC_LAP(i)                                                                      \
  do {                                                                                  \
    const long long now_ = wall_clock64();                                              \
    lap_[i] += now_ - last_;                                                            \
    last_ = now_;                                                                       \
  } while (0)
#define GDC_LAP_BEGIN long long lap_[6] = {0, 0, 0, 0, 0, 0}, last_ = wall_clock64()
#define GDC_LAP_END                                                                     \
  do {                                                                                  \
    if (threadIdx.x == 0 && blockIdx.x == 0)                                            \
      for (int q_ = 0; q_ < 6; ++q_) gdc_probe_buf[9 + q_] = lap_[q_];                  \
  } while (0)
extern "C" int igcn_debug_gdc_probe(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gdc_probe_buf), sizeof(long long) * 16);
}
#else
#define GDC_PROBE(i)
#define GDC_LAP(i)
#define GDC_LAP_BEGIN
#define GDC_LAP_END
#endif

template <int NR>                                       // rows of its column a thread keeps: NR * (GDC_T / cw) >= R
__global__ void __launch_bounds__(GDC_T)
k_gdc_topk(int R, int k, double alpha, const float* __restrict__ A, int64_t* __restrict__ ei_row,
           int64_t* __restrict__ ei_col, float* __restrict__ ew, int32_t* __restrict__ counts,
           const int64_t* __restrict__ subject, int64_t n_subjects) {
  extern __shared__ double smem_d[];
  const int ld = gdc_ld(R), g = blockIdx.x, tid = threadIdx.x;
  const int cw = R <= 64 ? 64 : 128, nsl = GDC_T / cw;  // a wave lies inside one row slice
  double* colf = smem_d;                                // [2][128]  multipliers (column p) of the current / next pivot
  double* rowP = colf + 2 * 128;                        // [128]  the pivot row
  double* cand = rowP + 128;                            // [128]  scale factors d^-1/2
  unsigned long long* rowmask = (unsigned long long*)(cand + 128);   // [128][2]  edges of row i as a column bit set
  unsigned long long* wmax = rowmask + 256;             // [GDC_T / 64]  per-wave slot of the pivot search
  double* M = (double*)(wmax + GDC_T / 64);             // [R][ld]
  int* piv = (int*)(M + (size_t)R * ld);                // [R]  row chosen as the pivot of column p
  int* rinv = piv + R;                                  // [R]  step at which row w was the pivot
  int* rowcnt = rinv + R;                               // [R+2]
  const int64_t sj = subject ? subject[g] : (int64_t)g;               // graph g of the batch = matrix subject[g]
  if (subject && (sj < 0 || sj >= n_subjects)) {
    // an index outside the dataset: no read — the graph is handed over EMPTY (all slots padding, count 0), which the
    // consumer's plan build reports in its status word (uniform over the workgroup: no barrier has been reached yet)
    const int64_t s0 = (int64_t)g * R * k;
    for (int64_t t = s0 + tid; t < s0 + (int64_t)R * k; t += GDC_T) {
      ei_row[t] = -1;
      ei_col[t] = -1;
      ew[t] = 0.f;
    }
    if (tid == 0) counts[g] = 0;
    return;
  }
  const float* a = A + sj * R * R;

  GDC_PROBE(0);
  // ---- H = D^-1/2 A D^-1/2, M = I - (1 - alpha) H -------------------------------------------------
  for (int t = tid; t < R * R; t += GDC_T) {            // coalesced: A enters LDS once
    const int i = t / R, j = t - i * R;
    M[i * ld + j] = (double)a[t];
  }
  for (int t = tid; t < 3 * 128; t += GDC_T) colf[t] = 0.0;          // multiplier buffers and the staged row, padded
  for (int t = tid; t < 256 + GDC_T / 64; t += GDC_T) rowmask[t] = 0ull;
  __syncthreads();
  GDC_PROBE(1);
  for (int i = tid; i < R; i += GDC_T) {
    double s = 0.0;
    for (int j = 0; j < R; ++j) s += M[i * ld + j];
    cand[i] = 1.0 / sqrt(s);
  }
  __syncthreads();
  GDC_PROBE(2);

  // ---- Gauss-Jordan inversion in registers, implicit partial pivoting -------------------------------
  const int jc = tid & (cw - 1), lane = tid & 63;
  const int i0 = __builtin_amdgcn_readfirstlane(tid / cw);          // wave-uniform: row predicates are scalar
  const int base = i0 * NR;
  const bool col_ok = jc < R;
  double reg[NR];
#pragma unroll
  for (int kk = 0; kk < NR; ++kk) {
    const int i = base + kk;
    double v = 0.0;
    if (col_ok && i < R) {
      const double h = cand[i] * M[i * ld + jc] * cand[jc];
      v = (i == jc ? 1.0 : 0.0) - (1.0 - alpha) * h;
    }
    reg[kk] = v;
  }
  // the thread of column c leaves its column — the multipliers of pivot c — in LDS (16 bytes per store)
  auto publish = [&](double* dst) {
#pragma unroll
    for (int kk = 0; kk < NR; kk += 2)
      *reinterpret_cast<double2*>(dst + base + kk) = make_double2(reg[kk], reg[kk + 1]);
  };
  if (jc == 0) publish(colf);
  // rows l and l + 64 of this lane: available as pivots while in range and not used yet
  bool free0 = lane < R, free1 = lane + 64 < R;
  __syncthreads();
  GDC_PROBE(3);
  GDC_LAP_BEGIN;
  for (int p = 0; p < R; ++p) {
    const double* cf = colf + (p & 1) * 128;            // column p: the multipliers of this pivot
    double* cn = colf + ((p + 1) & 1) * 128;            // column p + 1 after this pivot's update
    // (A) the pivot, found by every wave for itself: largest |.| among the free rows, lowest row on ties
    // (non-negative doubles order like their bit patterns: ONE LDS atomic maximum per lane on the wave's own slot
    // instead of a six-step shuffle chain; LDS operations of a wave complete in order, so no barrier is involved)
    const double a0 = free0 ? fabs(cf[lane]) : -1.0, a1 = free1 ? fabs(cf[lane + 64]) : -1.0;
    const double am = a0 >= a1 ? a0 : a1;
    unsigned long long* slot = wmax + (tid >> 6);
    if (am >= 0.0) atomicMax(slot, (unsigned long long)__double_as_longlong(am));
    const double wm = __longlong_as_double((long long)*(volatile unsigned long long*)slot);
    if (lane == 0) *(volatile unsigned long long*)slot = 0ull;      // (behind the read above: ready for the next pivot)
    const unsigned long long lo = __ballot(a0 == wm), hi = __ballot(a1 == wm);
    const int r = lo ? __ffsll((long long)lo) - 1 : 64 + __ffsll((long long)hi) - 1;       // (wave-uniform)
    free0 = free0 && lane != r;
    free1 = free1 && lane + 64 != r;
    const int own = r / NR, kr = r - own * NR;
    const bool mine = own == i0;                        // this slice holds the pivot row
    GDC_LAP(0);
    if (mine) {
      double v = 0.0;
      switch (kr) { GDC_ROWS(GDC_GET) default: break; }
      rowP[jc] = v;
    }
    if (tid == 0) { piv[p] = r; rinv[r] = p; }
    GDC_LAP(1);
    __syncthreads();
    GDC_LAP(2);
    // (B) one fma per row.  Column p's thread: its registers equal the multipliers, reg - c (1 + inv) = -c inv
    {
      const double inv = 1.0 / rowP[p];
      const double newp = ((jc == p) ? 1.0 : rowP[jc]) * inv;       // the scaled pivot row (stays in row r)
      const double fac = (jc == p) ? 1.0 + inv : newp;
#pragma unroll
      for (int k0 = 0; k0 < NR; k0 += 8) {
        double2 c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)                     // independent 16-byte LDS reads (same address in every lane)
          c[u] = *reinterpret_cast<const double2*>(cf + base + k0 + 2 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          reg[k0 + 2 * u] = fma(-c[u].x, fac, reg[k0 + 2 * u]);
          reg[k0 + 2 * u + 1] = fma(-c[u].y, fac, reg[k0 + 2 * u + 1]);
        }
      }
      if (mine) switch (kr) { GDC_ROWS(GDC_SET) default: break; }   // the pivot row itself: the scaled row
      GDC_LAP(3);
      if (jc == p + 1) publish(cn);
    }
    GDC_LAP(4);
    __syncthreads();
    GDC_LAP(5);
  }
  GDC_LAP_END;
  GDC_PROBE(4);
  // the inverse to LDS: A^-1[i][j] = W[piv[i]][rinv[j]], i.e. W[w][c] belongs at (rinv[w], piv[c])
  if (col_ok) {
    const int cj = piv[jc];
#pragma unroll
    for (int kk = 0; kk < NR; ++kk)
      if (base + kk < R) M[rinv[base + kk] * ld + cj] = reg[kk];
  }
  __syncthreads();

  GDC_PROBE(5);
  // ---- top-k per column (thread = column): the winners as a row bit set, their final weights in place ----------
  // order of preference: larger value, then LARGER row (a stable ascending argsort keeps the later of two equals)
  unsigned long long tk0 = 0ull, tk1 = 0ull;           // rows of this column that stay
  if (tid < R) {
    const int j = tid;
    double norm = 0.0;
    if (k <= 4) {
      // one sweep with the best four in registers: rows arrive in ascending order, so a newcomer beats an equal value
      double v0 = -HUGE_VAL, v1 = -HUGE_VAL, v2 = -HUGE_VAL, v3 = -HUGE_VAL;
      int b0 = -1, b1 = -1, b2 = -1, b3 = -1;
#pragma unroll 4
      for (int i = 0; i < R; ++i) {
        const double v = alpha * M[i * ld + j];
        if (v >= v3) {
          if (v >= v2) {
            v3 = v2; b3 = b2;
            if (v >= v1) {
              v2 = v1; b2 = b1;
              if (v >= v0) { v1 = v0; b1 = b0; v0 = v; b0 = i; } else { v1 = v; b1 = i; }
            } else { v2 = v; b2 = i; }
          } else { v3 = v; b3 = i; }
        }
      }
      const double vs[4] = {v0, v1, v2, v3};
      const int bs[4] = {b0, b1, b2, b3};
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
        if (s2 < k && bs[s2] >= 0) {
          norm += vs[s2];
          if (bs[s2] < 64) tk0 |= 1ull << bs[s2]; else tk1 |= 1ull << (bs[s2] - 64);
        }
    } else {
      for (int s2 = 0; s2 < k && s2 < R; ++s2) {
        double best = 0.0;
        int bi = -1;
        for (int i = 0; i < R; ++i) {
          const double v = alpha * M[i * ld + j];
          const bool taken = ((i < 64 ? tk0 >> i : tk1 >> (i - 64)) & 1ull) != 0;
          if (!taken && (bi < 0 || v >= best)) { best = v; bi = i; }
        }
        if (bi < 64) tk0 |= 1ull << bi; else tk1 |= 1ull << (bi - 64);
        norm += best;
      }
    }
    const double den = (norm <= 0.0) ? 1.0 : norm;
    for (int h = 0; h < 2; ++h) {
      unsigned long long m = h ? tk1 : tk0;
      while (m) {
        const int i = h * 64 + __ffsll((long long)m) - 1;
        m &= m - 1;
        const double w = (alpha * M[i * ld + j]) / den;
        M[i * ld + j] = w;
        if (w != 0.0) atomicOr(rowmask + 2 * i + (j >> 6), 1ull << (j & 63));     // (a kept 0 is not an edge for coo_matrix)
      }
    }
  }
  __syncthreads();
  GDC_PROBE(6);
  // ---- COO emission in row-major order: slot = row start + edges of the row in lower columns ---------------------
  if (tid < R) rowcnt[tid] = __popcll(rowmask[2 * tid]) + __popcll(rowmask[2 * tid + 1]);
  __syncthreads();
  GDC_PROBE(7);
  const int64_t slot0 = (int64_t)g * R * k, off = (int64_t)g * R;
  int total = 0;
  for (int t = 0; t < R; ++t) total += rowcnt[t];       // (every thread for itself: R independent LDS reads)
  if (tid < R) {
    const int j = tid;
    for (int h = 0; h < 2; ++h) {
      unsigned long long m = h ? tk1 : tk0;
      while (m) {
        const int i = h * 64 + __ffsll((long long)m) - 1;
        m &= m - 1;
        const double w = M[i * ld + j];
        if (w == 0.0) continue;
        int start = 0;
        for (int t = 0; t < i; ++t) start += rowcnt[t];
        const unsigned long long m0 = rowmask[2 * i], m1 = rowmask[2 * i + 1];
        const int before = j < 64 ? __popcll(m0 & ((1ull << j) - 1ull))
                                  : __popcll(m0) + __popcll(m1 & ((1ull << (j - 64)) - 1ull));
        const int64_t s2 = slot0 + start + before;
        ei_row[s2] = off + i;
        ei_col[s2] = off + j;
        ew[s2] = (float)w;
      }
    }
  }
  if (tid == 0) counts[g] = total;
  // padding behind the last edge of the graph
  for (int64_t s2 = slot0 + total + tid; s2 < slot0 + (int64_t)R * k; s2 += GDC_T) {
    ei_row[s2] = -1;
    ei_col[s2] = -1;
    ew[s2] = 0.f;
  }
  GDC_PROBE(8);
}

static int gdc_rows(int R) {                            // NR of the kernel instance that serves R
  const int nsl = GDC_T / (R <= 64 ? 64 : 128);
  const int rows = (R + nsl - 1) / nsl;
  for (int nr : {8, 16, 24, 32, 48, 64})
    if (rows <= nr) return nr;
  return 0;
}

static size_t gdc_lds_bytes(int R) {
  const size_t ld = (size_t)(R | 1);
  size_t b = ((size_t)R * ld + 4 * 128 + 256 + GDC_T / 64) * sizeof(double);
  b += ((size_t)R + R + R + 2) * sizeof(int);
  return (b + 15) & ~(size_t)15;
}

extern "C" int igcn_gdc_topk_max_rois(void) {
  int r = 8;
  while (r < 128 && gdc_lds_bytes(r + 1) <= 160 * 1024) ++r;       // (128: two row slices of 64 register rows)
  return r;
}

static int gdc_topk(int B, int R, int k, double alpha, const float* A, const int64_t* subject, int64_t n_subjects,
                    int64_t* edge_index, float* edge_attr, int32_t* counts, void* stream) {
  IGCN_REQUIRE(B >= 0 && R > 0 && k > 0 && k <= R, "gdc_topk: bad sizes B=%d R=%d k=%d", B, R, k);
  IGCN_REQUIRE(alpha > 0.0 && alpha <= 1.0, "gdc_topk: alpha=%g outside (0,1]", alpha);
  if (R > igcn_gdc_topk_max_rois()) {
    igcn_set_error("gdc_topk: R=%d exceeds the kernel's limit (max R = %d)", R, igcn_gdc_topk_max_rois());
    return IGCN_ERR_UNSUPPORTED;
  }
  const size_t lds = gdc_lds_bytes(R);
  if (B == 0) return IGCN_OK;
  const int64_t slots = (int64_t)B * R * k;
#define GDC_LAUNCH(NR)                                                                                              \
  case NR:                                                                                                          \
    IGCN_ALLOW_BIG_LDS(k_gdc_topk<NR>);                                                                             \
    hipLaunchKernelGGL(k_gdc_topk<NR>, dim3(B), dim3(GDC_T), lds, (hipStream_t)stream, R, k, alpha, A, edge_index,   \
                       edge_index + slots, edge_attr, counts, subject, n_subjects);                                             \
    break
  switch (gdc_rows(R)) {
    GDC_LAUNCH(8);
    GDC_LAUNCH(16);
    GDC_LAUNCH(24);
    GDC_LAUNCH(32);
    GDC_LAUNCH(48);
    GDC_LAUNCH(64);
  }
#undef GDC_LAUNCH
  IGCN_CHECK_LAUNCH("gdc_topk");
  return IGCN_OK;
}

extern "C" int igcn_gdc_topk(int B, int R, int k, double alpha, const float* A, int64_t* edge_index, float* edge_attr,
                             int32_t* counts, void* stream) {
  return gdc_topk(B, R, k, alpha, A, nullptr, B, edge_index, edge_attr, counts, stream);
}

extern "C" int igcn_gdc_topk_of(int B, int R, int k, double alpha, const float* A, int64_t n_subjects,
                                const int64_t* subject, int64_t* edge_index, float* edge_attr, int32_t* counts,
                                void* stream) {
  IGCN_REQUIRE(subject != nullptr && n_subjects > 0, "gdc_topk_of: the subject list is missing");
  return gdc_topk(B, R, k, alpha, A, subject, n_subjects, edge_index, edge_attr, counts, stream);
}
