// Cross-attention core on the CDNA4 matrix cores (exact-fp32 v_mfma_f32_16x16x4_f32), in place on the projection
// outputs like attn_core.hip:   q [B,Lq,H*hd]   kv [B,Lk,2,H*hd]   o [B,Lq,H*hd]
// (kernel/sgcn_img_snp.py:240, the nn.MultiheadAttention core; softmax(q k^T / sqrt(hd)) v per head).
// Any head_dim up to 32: the head's columns are zero-padded in LDS to HDP = the next multiple of 4 (template
// parameter; 10 -> 12, 15 -> 16, 24 -> 24 ...), the output / gradient tiles to multiples of 16 rows.
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

#define AM_MAX_WAVES 16

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// rows [0, rows_valid) of a row_stride-strided source (hd floats per row) -> LDS rows of HDP + 1 floats, columns
// hd..HDP-1 and rows up to rows_pad zero-filled.  vec: hd % 4 == 0 and every row start is 16-byte aligned.
template <int HDP>
__device__ __forceinline__ void am_stage(const float* __restrict__ src, int64_t row_stride, int hd, int vec,
                                         int rows_valid, int rows_pad, float* __restrict__ dst) {
  constexpr int LD = HDP + 1, NQ = HDP / 4;
  for (int t = threadIdx.x; t < rows_pad * NQ; t += blockDim.x) {
    const int j = t / NQ, c = (t % NQ) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < rows_valid) {
      const float* p = src + (int64_t)j * row_stride + c;
      if (vec) {
        const float4 t4 = *reinterpret_cast<const float4*>(p);
        v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (c + u < hd) v[u] = p[u];
      }
    }
    float* d = dst + j * LD + c;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
}

// Two matrices of the same geometry (K and V of a head; Q and dO) in ONE pass: every thread first issues all the loads
// of a batch of AM_SU slots of both sources, then stores — a workgroup's staging is then one or two memory round trips
// instead of one per 16-byte slot per matrix (eight serial trips for 400 keys on 384 threads, ~12 us per workgroup,
// which was most of what kept the matrix cores under 30 % busy).
#define AM_SU 5
// LD = LDS row stride (HDP + 1, or HDP for the unpadded 16-column layout of the forward kernel, where SWZ_A also XORs
// matrix A's 4-column group with bits 2-3 of the row: see k_attn_mfma_fwd).
template <int HDP, int LD = HDP + 1, bool SWZ_A = false>
__device__ __forceinline__ void am_stage_pair(const float* __restrict__ src_a, const float* __restrict__ src_b,
                                              int64_t row_stride, int hd, int vec, int rows_valid, int rows_pad,
                                              float* __restrict__ dst_a, float* __restrict__ dst_b) {
  constexpr int NQ = HDP / 4;
  const int total = rows_pad * NQ, step = (int)blockDim.x;
  for (int t0 = threadIdx.x; t0 < total; t0 += step * AM_SU) {
    float va[AM_SU][4], vb[AM_SU][4];
#pragma unroll
    for (int u = 0; u < AM_SU; ++u) {
      const int t = t0 + u * step, j = t / NQ, c = (t % NQ) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) va[u][e] = vb[u][e] = 0.f;
      if (t < total && j < rows_valid) {
        const float* pa = src_a + (int64_t)j * row_stride + c;
        const float* pb = src_b + (int64_t)j * row_stride + c;
        if (vec) {
          const float4 a4 = *reinterpret_cast<const float4*>(pa), b4 = *reinterpret_cast<const float4*>(pb);
          va[u][0] = a4.x; va[u][1] = a4.y; va[u][2] = a4.z; va[u][3] = a4.w;
          vb[u][0] = b4.x; vb[u][1] = b4.y; vb[u][2] = b4.z; vb[u][3] = b4.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c + e < hd) { va[u][e] = pa[e]; vb[u][e] = pb[e]; }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < AM_SU; ++u) {
      const int t = t0 + u * step, j = t / NQ, c = (t % NQ) * 4;
      if (t < total) {
        float* da = dst_a + j * LD + (SWZ_A ? (c ^ (((j >> 2) & 3) << 2)) : c);
        float* db = dst_b + j * LD + c;
        if constexpr (LD % 4 == 0) {                              // 16-byte aligned rows: one LDS store per slot
          *reinterpret_cast<float4*>(da) = make_float4(va[u][0], va[u][1], va[u][2], va[u][3]);
          *reinterpret_cast<float4*>(db) = make_float4(vb[u][0], vb[u][1], vb[u][2], vb[u][3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { da[e] = va[u][e]; db[e] = vb[u][e]; }
        }
      }
    }
  }
}

// four consecutive head columns [c0, c0+4) of one row of a [*, D] tensor, clipped to hd
__device__ __forceinline__ void am_store4(float* __restrict__ p, int c0, int hd, int vec, float a, float b, float c,
                                          float d) {
  if (vec && c0 + 3 < hd) {
    *reinterpret_cast<float4*>(p + c0) = make_float4(a, b, c, d);
  } else {
    if (c0 < hd) p[c0] = a;
    if (c0 + 1 < hd) p[c0 + 1] = b;
    if (c0 + 2 < hd) p[c0 + 2] = c;
    if (c0 + 3 < hd) p[c0 + 3] = d;
  }
}

// workgroup -> (sample, head) item of a B * H launch.  The hardware deals workgroup i to XCD i mod 8, and the H heads
// of a sample read the two halves of the SAME 128-byte lines of q / kv / dO (a [.., H * head_dim] row per token): with
// item = blockIdx the two heads of a sample sat on different XCDs, i.e. different L2s, and every line came from HBM
// twice — the K | V staging of the forward (768 workgroups at once) took 8-16 us for 39 MB.  Here XCD x takes the
// items [x per, (x + 1) per) in order: a sample's heads are neighbours in ONE L2.  Grid = 8 per (am_grid); -1 = no item.
__host__ __device__ inline int am_per_xcd(int items) { return (items + 7) >> 3; }
static inline unsigned am_grid(int items) { return 8u * (unsigned)am_per_xcd(items); }
__device__ __forceinline__ int am_item(int items) {
  const int per = am_per_xcd(items), slot = (int)(blockIdx.x >> 3);
  const int id = (int)(blockIdx.x & 7) * per + slot;
  return id < items ? id : -1;
}

#ifdef AM_PROBE_ON
__device__ long long am_probe_buf[16 * 8];
#define AM_PROBE(i) do { if ((threadIdx.x & 63) == 0 && (blockIdx.x % 128) == 0 && (threadIdx.x >> 6) < 2) { am_probe_buf[((blockIdx.x / 128) * 2 + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64(); } } while (0)
extern "C" int igcn_debug_attn_probe(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(am_probe_buf), sizeof(long long) * 128);
}
#else
#define AM_PROBE(i)
#endif

// EXACT: head_dim == HDP and 16-byte aligned rows — the clipping guards fold away at compile time
template <int HDP, bool EXACT>
__global__ void __launch_bounds__(64 * AM_MAX_WAVES)
k_attn_mfma_fwd(int H, int hd_rt, int vec_rt, int Lq, int Lk, const float* __restrict__ q,
                const float* __restrict__ kv, float* __restrict__ o, float* __restrict__ lse, int items) {
  // HDP == 16 (the bench model: two heads of 16): UNPADDED 16-float rows, 51 KB per workgroup at 400 keys instead of
  // 54 KB — three workgroups per CU instead of two, i.e. one can stage while two compute.  Conflict-free without the
  // pad column because (a) K's 4-column groups are XOR-swizzled with bits 2-3 of the row, and (b) the rows of a key
  // tile enter the score product in the order pi(m) = (m >> 2) + 4 (m & 3): register r of lane group g then holds key
  // g + 4r, so the four lane groups of a P.V step read V rows g (mod 4) = four different 16-bank windows.
  constexpr bool FLAT = (HDP == 16);
  constexpr int LD = FLAT ? HDP : HDP + 1, NC = HDP / 4, NO = (HDP + 15) / 16;
  const int hd = EXACT ? HDP : hd_rt, vec = EXACT ? 1 : vec_rt;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int item = am_item(items);
  if (item < 0) return;                                          // workgroup-uniform, before any barrier
  const int b = item / H, h = item % H, D = H * hd;
  const int Lkp = (Lk + 15) & ~15, nkt = Lkp >> 4, nqt = (Lq + 15) >> 4;
  float* Ks = smem;
  float* Vs = Ks + (size_t)Lkp * LD;
  const float* kbase = kv + (int64_t)b * Lk * 2 * D + h * hd;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6, n = lane & 15, g = lane >> 4;
  // scores in the LOG2 domain: q is scaled by log2(e) / sqrt(head_dim) once, every exponential of the tile loop is one
  // v_exp_f32 without the multiply __expf puts in front of it; lse is converted back where it is stored
  const float scale = rsqrtf((float)hd) * 1.44269504088896341f;
  const int krow = FLAT ? (n >> 2) + 4 * (n & 3) : n;            // K row (within a tile) behind score row n
  const int ksw = FLAT ? (krow >> 2) & 3 : 0;
  AM_PROBE(0);
  // the wave's first query tile is requested before the K/V staging, so that its round trip runs under the staging's
  float qn[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c)
    qn[c] = (w * 16 + n < Lq && 4 * c + g < hd) ? q[(int64_t)(b * Lq + w * 16 + n) * D + h * hd + 4 * c + g] * scale : 0.f;
  am_stage_pair<HDP, LD, FLAT>(kbase, kbase + D, 2 * D, hd, vec, Lk, Lkp, Ks, Vs);
  AM_PROBE(1);
  __syncthreads();
  AM_PROBE(2);
  for (int qt = w; qt < nqt; qt += nw) {
    const int qi = qt * 16 + n;
    float qb[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) qb[c] = qn[c];
    if (qt + nw < nqt) {                                        // wave-uniform: the next tile's queries, one tile ahead
#pragma unroll
      for (int c = 0; c < NC; ++c)
        qn[c] = (qi + nw * 16 < Lq && 4 * c + g < hd)
                    ? q[(int64_t)(b * Lq + qi + nw * 16) * D + h * hd + 4 * c + g] * scale : 0.f;
    }
    float m = -INFINITY, l = 0.f;
    f32x4 oacc[NO];
#pragma unroll
    for (int t = 0; t < NO; ++t) oacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt0 = 0; kt0 < nkt; kt0 += 4) {
      f32x4 s[4];
      float tmax = -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s[u] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (kt0 + u < nkt) {                                    // wave-uniform
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          const float* kr = Ks + ((kt0 + u) * 16 + krow) * LD + g;
#pragma unroll
          for (int c = 0; c < NC; ++c) acc = mfma4(kr[4 * (c ^ ksw)], qb[c], acc);
          if ((kt0 + u + 1) * 16 <= Lk) {                       // wave-uniform: a full key tile needs no masking (the
#pragma unroll                                                    // compare + select per score were a quarter of the loop's
            for (int r = 0; r < 4; ++r) {                        // vector instructions; only the last tile can be ragged)
              s[u][r] = acc[r];
              tmax = fmaxf(tmax, acc[r]);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = (kt0 + u) * 16 + (FLAT ? g + 4 * r : 4 * g + r);
              s[u][r] = (key < Lk) ? acc[r] : -INFINITY;
              tmax = fmaxf(tmax, s[u][r]);
            }
          }
        }
      }
      // LAZY running maximum: the reference m moves only when some lane of the wave sees a score more than 2^8 above it
      // (a wave-wide vote: one compare), so the two cross-lane exchanges, the rescaling exponential and its five
      // multiplies leave the loop's chain after the first groups; probabilities stay <= 2^8, sums far inside fp32
      if (__any(tmax > m + 8.f)) {                              // wave-uniform (true on the first group: m = -inf)
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);
        const float f = __builtin_amdgcn_exp2f(m - mn);         // 2^(-inf) = 0 on the first group
        m = mn;
        l *= f;
#pragma unroll
        for (int t = 0; t < NO; ++t) oacc[t] *= f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (kt0 + u < nkt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* vr = Vs + ((kt0 + u) * 16 + (FLAT ? g + 4 * r : 4 * g + r)) * LD + n;
            const float p = __builtin_amdgcn_exp2f(s[u][r] - m);
            l += p;
#pragma unroll
            for (int t = 0; t < NO; ++t) {
              const float a = (16 * t + n < HDP) ? vr[16 * t] : 0.f;          // A = V[key of (g, r)][hd 16t+n]
              oacc[t] = mfma4(a, p, oacc[t]);
            }
          }
        }
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    AM_PROBE(3);
    if (qi < Lq) {
      const float inv = 1.f / l;
      float* op = o + (int64_t)(b * Lq + qi) * D + h * hd;
#pragma unroll
      for (int t = 0; t < NO; ++t)
        am_store4(op, 16 * t + 4 * g, hd, vec, oacc[t][0] * inv, oacc[t][1] * inv, oacc[t][2] * inv, oacc[t][3] * inv);
      if (g == 0) lse[((int64_t)b * H + h) * Lq + qi] = m * 0.693147180559945309f + __logf(l);   // natural log
    }
  }
}

// backward: dq [B,Lq,D], dkv [B,Lk,2,D]
template <int HDP, bool EXACT>
__global__ void __launch_bounds__(64 * AM_MAX_WAVES)
k_attn_mfma_bwd(int H, int hd_rt, int vec_rt, int Lq, int Lk, const float* __restrict__ q,
                const float* __restrict__ kv, const float* __restrict__ o, const float* __restrict__ lse,
                const float* __restrict__ dout, float* __restrict__ dq, float* __restrict__ dkv, int items) {
  constexpr int LD = HDP + 1, NC = HDP / 4, NO = (HDP + 15) / 16;
  const int hd = EXACT ? HDP : hd_rt, vec = EXACT ? 1 : vec_rt;
  extern __shared__ float smem[];
  const int item = am_item(items);
  if (item < 0) return;                                          // workgroup-uniform, before any barrier
  const int b = item / H, h = item % H, D = H * hd;
  const int Lkp = (Lk + 15) & ~15, Lqp = (Lq + 15) & ~15, nkt = Lkp >> 4, nqt = Lqp >> 4;
  float* Ks = smem;
  float* Vs = Ks + (size_t)Lkp * LD;
  float* Qs = Vs + (size_t)Lkp * LD;
  float* dOs = Qs + (size_t)Lqp * LD;
  float* ls = dOs + (size_t)Lqp * LD;                           // lse [Lqp]   (+inf on padding rows: p = 0)
  float* dl = ls + Lqp;                                         // delta [Lqp]
  int* next_task = (int*)(dl + Lqp);                            // wave task counter
  const float* kbase = kv + (int64_t)b * Lk * 2 * D + h * hd;
  const float* qbase = q + (int64_t)b * Lq * D + h * hd;
  const float* dobase = dout + (int64_t)b * Lq * D + h * hd;
  // delta = rowsum(o * do) and lse: thread r's row, requested ahead of the staging passes (same reason as there)
  {
    const int r = threadIdx.x;
    float orow[HDP], drow[HDP], lv = INFINITY;
    const bool mine = r < Lqp, live = r < Lq;
    if (live) {
      const float* op = o + (int64_t)(b * Lq + r) * D + h * hd;
      const float* dp = dobase + (int64_t)r * D;
      if (vec) {
#pragma unroll
        for (int c = 0; c < HDP; c += 4) {
          const float4 a = *reinterpret_cast<const float4*>(op + c), d4 = *reinterpret_cast<const float4*>(dp + c);
          orow[c] = a.x; orow[c + 1] = a.y; orow[c + 2] = a.z; orow[c + 3] = a.w;
          drow[c] = d4.x; drow[c + 1] = d4.y; drow[c + 2] = d4.z; drow[c + 3] = d4.w;
        }
      } else {
#pragma unroll
        for (int c = 0; c < HDP; ++c) {
          orow[c] = c < hd ? op[c] : 0.f;
          drow[c] = c < hd ? dp[c] : 0.f;
        }
      }
      lv = lse[((int64_t)b * H + h) * Lq + r];
    }
    am_stage_pair<HDP>(kbase, kbase + D, 2 * D, hd, vec, Lk, Lkp, Ks, Vs);
    am_stage_pair<HDP>(qbase, dobase, D, hd, vec, Lq, Lqp, Qs, dOs);
    if (mine) {
      float d = 0.f;
      if (live) {
#pragma unroll
        for (int c = 0; c < HDP; ++c) d += orow[c] * drow[c];
      }
      dl[r] = d;
      ls[r] = lv;
    }
    for (int r2 = r + blockDim.x; r2 < Lqp; r2 += blockDim.x) {      // more query rows than threads
      float d = 0.f, l2 = INFINITY;
      if (r2 < Lq) {
        const float* op = o + (int64_t)(b * Lq + r2) * D + h * hd;
        const float* dp = dobase + (int64_t)r2 * D;
        for (int c = 0; c < hd; ++c) d += op[c] * dp[c];
        l2 = lse[((int64_t)b * H + h) * Lq + r2];
      }
      dl[r2] = d;
      ls[r2] = l2;
    }
  }
  if (threadIdx.x == 0) *next_task = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  const float scale = rsqrtf((float)hd);
  // wave tasks, long ones first: [0, nqt) own a query tile (loops over all key tiles), [nqt, nqt+nkt) own a key tile
  for (;;) {
    int task = 0;
    if (lane == 0) task = atomicAdd(next_task, 1);
    task = __shfl(task, 0, 64);
    if (task >= nqt + nkt) break;
    if (task < nqt) {
      // ---- dQ for queries [16*task, 16*task+16): S^T orientation, lane column n = query ----
      const int qrow = task * 16 + n;
      float qb[NC], dob[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        qb[c] = Qs[qrow * LD + 4 * c + g] * scale;
        dob[c] = dOs[qrow * LD + 4 * c + g];
      }
      const float lsn = ls[qrow], dln = dl[qrow];
      f32x4 acc[NO];                                            // dQ^T[hd 16t+4g+r][query n]
#pragma unroll
      for (int t = 0; t < NO; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int kt = 0; kt < nkt; ++kt) {
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
        const float* kr = Ks + (kt * 16 + n) * LD + g;
        const float* vr = Vs + (kt * 16 + n) * LD + g;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          st = mfma4(kr[4 * c], qb[c], st);                     // S^T = K Q^T (scaled)
          dpt = mfma4(vr[4 * c], dob[c], dpt);                  // dP^T = V dO^T
        }
        const float* kc = Ks + (kt * 16 + 4 * g) * LD + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ds = __expf(st[r] - lsn) * (dpt[r] - dln) * scale;   // padded keys: K row = 0 kills the term
#pragma unroll
          for (int t = 0; t < NO; ++t) {
            const float a = (16 * t + n < HDP) ? kc[r * LD + 16 * t] : 0.f;
            acc[t] = mfma4(a, ds, acc[t]);                      // dQ^T += K^T dS^T
          }
        }
      }
      if (qrow < Lq) {
        float* dst = dq + (int64_t)(b * Lq + qrow) * D + h * hd;
#pragma unroll
        for (int t = 0; t < NO; ++t) am_store4(dst, 16 * t + 4 * g, hd, vec, acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
      }
    } else {
      // ---- dK, dV for keys [16*kt, 16*kt+16): S orientation, lane column n = key ----
      const int kt = task - nqt, krow = kt * 16 + n;
      float kb[NC], vb[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        kb[c] = Ks[krow * LD + 4 * c + g] * scale;
        vb[c] = Vs[krow * LD + 4 * c + g];
      }
      f32x4 dka[NO], dva[NO];                                   // dK^T / dV^T [hd 16t+4g+r][key n]
#pragma unroll
      for (int t = 0; t < NO; ++t) dka[t] = dva[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int qt = 0; qt < nqt; ++qt) {
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
        const float* qr = Qs + (qt * 16 + n) * LD + g;
        const float* dr = dOs + (qt * 16 + n) * LD + g;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          st = mfma4(qr[4 * c], kb[c], st);                     // S = Q K^T (scaled): rows = queries 4g+r
          dpt = mfma4(dr[4 * c], vb[c], dpt);                   // dP = dO V^T
        }
        const float* qc = Qs + (qt * 16 + 4 * g) * LD + n;
        const float* dc = dOs + (qt * 16 + 4 * g) * LD + n;
        const float* lsr = ls + qt * 16 + 4 * g;
        const float* dlr = dl + qt * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(st[r] - lsr[r]);               // padding queries: lse = +inf -> p = 0
          const float ds = p * (dpt[r] - dlr[r]) * scale;
#pragma unroll
          for (int t = 0; t < NO; ++t) {
            const bool in = 16 * t + n < HDP;
            dva[t] = mfma4(in ? dc[r * LD + 16 * t] : 0.f, p, dva[t]);    // dV^T += dO^T P
            dka[t] = mfma4(in ? qc[r * LD + 16 * t] : 0.f, ds, dka[t]);   // dK^T += Q^T dS
          }
        }
      }
      if (krow < Lk) {
        float* base = dkv + ((int64_t)(b * Lk + krow) * 2) * D + h * hd;
#pragma unroll
        for (int t = 0; t < NO; ++t) {
          am_store4(base, 16 * t + 4 * g, hd, vec, dka[t][0], dka[t][1], dka[t][2], dka[t][3]);
          am_store4(base + D, 16 * t + 4 * g, hd, vec, dva[t][0], dva[t][1], dva[t][2], dva[t][3]);
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------------------
// Backward with SHARED score tiles (head_dim 16, <= 8 query tiles, at least as many key tiles as waves: the model's
// 90 queries x 400 keys).  k_attn_mfma_bwd computes every (key tile, query tile) pair TWICE — S^T, dP^T in the
// query-tile tasks (dQ reduces over keys), S, dP in the key-tile tasks (dK, dV reduce over queries): 28 matrix
// instructions and 8 exponentials per pair.  Here a pair is computed once, in the S orientation, and its score-gradient
// tile dS is TRANSPOSED through a wave-private 16 x 16 LDS tile (four 4-byte writes, one 16-byte read per lane) to feed
// the dQ product as well: 20 matrix instructions, 4 exponentials per pair.
//   * The pairs (key-tile major) are cut into one contiguous, equally long chunk per wave — a static split, so every
//     sum has a fixed order (the task queue of the kernel above hands out tiles by arrival).
//   * dQ: a wave keeps one accumulator per query tile (NQT x 4 registers) over its whole chunk; at the end the waves'
//     partials meet in LDS (over K | V, which are no longer needed) and are summed in wave order.
//   * dK | dV of a key tile whose pairs straddle two chunks: the first wave stores its part to the output rows, the
//     second adds that part (read back after the barrier) to its own and overwrites them.
// -------------------------------------------------------------------------------------------------------------
#define AM_TLD 20                                               // transpose tile row stride (16-byte aligned rows)
template <int NQT>
__global__ void __launch_bounds__(64 * 8)
k_attn_mfma_bwd_shared(int H, int Lq, int Lk, const float* __restrict__ q, const float* __restrict__ kv,
                       const float* __restrict__ o, const float* __restrict__ lse, const float* __restrict__ dout,
                       float* __restrict__ dq, float* __restrict__ dkv, int items) {
  constexpr int HDP = 16, LD = HDP + 1, NC = HDP / 4, hd = HDP;
  extern __shared__ float smem[];
  const int item = am_item(items);
  if (item < 0) return;                                          // workgroup-uniform, before any barrier
  const int b = item / H, h = item % H, D = H * hd;
  const int Lkp = (Lk + 15) & ~15, Lqp = NQT * 16, nkt = Lkp >> 4;
  float* Ks = smem;
  float* Vs = Ks + (size_t)Lkp * LD;
  float* Qs = Vs + (size_t)Lkp * LD;
  float* dOs = Qs + (size_t)Lqp * LD;
  float* ls = dOs + (size_t)Lqp * LD;                           // lse [Lqp]   (+inf on padding rows: p = 0)
  float* dl = ls + Lqp;                                         // delta [Lqp]
  float* Tw = dl + Lqp;                                         // [waves][16][AM_TLD] transpose tiles
  const float* kbase = kv + (int64_t)b * Lk * 2 * D + h * hd;
  const float* qbase = q + (int64_t)b * Lq * D + h * hd;
  const float* dobase = dout + (int64_t)b * Lq * D + h * hd;
  {
    const int r = threadIdx.x;                                  // delta = rowsum(o * do), lse: thread r's query row
    float orow[HDP], drow[HDP], lv = INFINITY;
    const bool mine = r < Lqp, live = r < Lq;
    if (live) {
      const float* op = o + (int64_t)(b * Lq + r) * D + h * hd;
      const float* dp = dobase + (int64_t)r * D;
#pragma unroll
      for (int c = 0; c < HDP; c += 4) {
        const float4 a = *reinterpret_cast<const float4*>(op + c), d4 = *reinterpret_cast<const float4*>(dp + c);
        orow[c] = a.x; orow[c + 1] = a.y; orow[c + 2] = a.z; orow[c + 3] = a.w;
        drow[c] = d4.x; drow[c + 1] = d4.y; drow[c + 2] = d4.z; drow[c + 3] = d4.w;
      }
      lv = lse[((int64_t)b * H + h) * Lq + r];
    }
    am_stage_pair<HDP>(kbase, kbase + D, 2 * D, hd, 1, Lk, Lkp, Ks, Vs);
    am_stage_pair<HDP>(qbase, dobase, D, hd, 1, Lq, Lqp, Qs, dOs);
    if (mine) {
      float d = 0.f;
      if (live) {
#pragma unroll
        for (int c = 0; c < HDP; ++c) d += orow[c] * drow[c];
      }
      dl[r] = d;
      ls[r] = lv;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6, n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // in a scalar register: the chunk bounds below are uniform
  const float scale = 0.25f;
  float* T = Tw + w * 16 * AM_TLD;
  const int P = nkt * NQT, lo = (w * P) / nw, hi = ((w + 1) * P) / nw;           // this wave's pairs [lo, hi)
  f32x4 dqa[NQT];                                               // dQ^T[hd 4g+r][query n] per query tile
#pragma unroll
  for (int t = 0; t < NQT; ++t) dqa[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 pend_k = {0.f, 0.f, 0.f, 0.f}, pend_v = {0.f, 0.f, 0.f, 0.f};            // first tile's part, if its head is elsewhere
  int pend_row = -1;
  const int kt_first = lo / NQT, kt_last = (hi - 1) / NQT;
  for (int kt = kt_first; kt <= kt_last; ++kt) {
    const int q_lo = kt == kt_first ? lo - kt * NQT : 0, q_hi = kt == kt_last ? hi - kt * NQT : NQT;
    const int krow = kt * 16 + n;
    float kb[NC], vb[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      kb[c] = Ks[krow * LD + 4 * c + g] * scale;
      vb[c] = Vs[krow * LD + 4 * c + g];
    }
    f32x4 dka = {0.f, 0.f, 0.f, 0.f}, dva = {0.f, 0.f, 0.f, 0.f};                // dK^T / dV^T [hd 4g+r][key n]
    const float* kc = Ks + (kt * 16 + 4 * g) * LD + n;           // K[key 4g + r][hd n]: the dQ product's A operand
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      if (qt >= q_lo && qt < q_hi) {                            // wave-uniform
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
        const float* qr = Qs + (qt * 16 + n) * LD + g;
        const float* dr = dOs + (qt * 16 + n) * LD + g;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          st = mfma4(qr[4 * c], kb[c], st);                     // S = Q K^T (scaled): rows = queries 4g+r, column = key n
          dpt = mfma4(dr[4 * c], vb[c], dpt);                   // dP = dO V^T
        }
        const float* qc = Qs + (qt * 16 + 4 * g) * LD + n;
        const float* dc = dOs + (qt * 16 + 4 * g) * LD + n;
        const float* lsr = ls + qt * 16 + 4 * g;
        const float* dlr = dl + qt * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(st[r] - lsr[r]);               // padding queries: lse = +inf -> p = 0
          const float ds = p * (dpt[r] - dlr[r]) * scale;
          T[(4 * g + r) * AM_TLD + n] = ds;
          dva = mfma4(dc[r * LD], p, dva);                      // dV^T += dO^T P
          dka = mfma4(qc[r * LD], ds, dka);                     // dK^T += Q^T dS
        }
        asm volatile("" ::: "memory");   // compiler-level order only: a wave's LDS operations execute in issue order, so
        const float4 dst = *reinterpret_cast<const float4*>(T + n * AM_TLD + 4 * g);   // the read sees the writes: dS^T, query n, keys 4g .. 4g+3
        asm volatile("" ::: "memory");   // (and the next pair's writes stay behind this read)
        dqa[qt] = mfma4(kc[0], dst.x, dqa[qt]);                 // dQ^T += K^T dS^T
        dqa[qt] = mfma4(kc[LD], dst.y, dqa[qt]);
        dqa[qt] = mfma4(kc[2 * LD], dst.z, dqa[qt]);
        dqa[qt] = mfma4(kc[3 * LD], dst.w, dqa[qt]);
      }
    }
    if (q_lo > 0) {                                             // the tile's first queries belong to the previous wave
      pend_k = dka;
      pend_v = dva;
      pend_row = krow;
    } else if (krow < Lk) {                                     // whole tile, or its first part (completed after the barrier)
      float* base = dkv + ((int64_t)(b * Lk + krow) * 2) * D + h * hd + 4 * g;
      *reinterpret_cast<float4*>(base) = make_float4(dka[0], dka[1], dka[2], dka[3]);
      *reinterpret_cast<float4*>(base + D) = make_float4(dva[0], dva[1], dva[2], dva[3]);
    }
  }
  // the first parts must be visible to the waves of THIS workgroup that complete them: the barrier's workgroup-scope
  // release / acquire is enough (one CU; a device-scope fence is an L2 write-back on this part: 107 -> 344 us)
  __syncthreads();                                              // ... and K | V are free
  if (pend_row >= 0 && pend_row < Lk) {
    float* base = dkv + ((int64_t)(b * Lk + pend_row) * 2) * D + h * hd + 4 * g;
    // (plain loads: these rows were never read before, so no stale line can sit in this CU's L1; the stores above went
    // through to L2 before the barrier)
    const float4 hk = *reinterpret_cast<const float4*>(base), hv = *reinterpret_cast<const float4*>(base + D);
    *reinterpret_cast<float4*>(base) = make_float4(hk.x + pend_k[0], hk.y + pend_k[1], hk.z + pend_k[2], hk.w + pend_k[3]);
    *reinterpret_cast<float4*>(base + D) = make_float4(hv.x + pend_v[0], hv.y + pend_v[1], hv.z + pend_v[2], hv.w + pend_v[3]);
  }
  // dQ: the waves' partials [wave][query][16 hd] over the K | V area, then summed in wave order
  float* R = smem;
#pragma unroll
  for (int t = 0; t < NQT; ++t)
    *reinterpret_cast<float4*>(R + ((size_t)(w * Lqp + t * 16 + n)) * HDP + 4 * g) =
        make_float4(dqa[t][0], dqa[t][1], dqa[t][2], dqa[t][3]);
  __syncthreads();
  for (int i = threadIdx.x; i < Lqp * 4; i += blockDim.x) {
    const int qrow = i >> 2, c4 = i & 3;
    if (qrow < Lq) {
      float4 a = *reinterpret_cast<const float4*>(R + (size_t)qrow * HDP + 4 * c4);
      for (int ww = 1; ww < nw; ++ww) {
        const float4 t4 = *reinterpret_cast<const float4*>(R + ((size_t)(ww * Lqp + qrow)) * HDP + 4 * c4);
        a.x += t4.x; a.y += t4.y; a.z += t4.z; a.w += t4.w;
      }
      *reinterpret_cast<float4*>(dq + (int64_t)(b * Lq + qrow) * D + h * hd + 4 * c4) = a;
    }
  }
}

static int am_hdp(int hd) { return (hd + 3) & ~3; }

static size_t am_lds_bytes(int hd, int Lq, int Lk, int backward) {
  const size_t Lkp = (size_t)((Lk + 15) & ~15), Lqp = (size_t)((Lq + 15) & ~15);
  const size_t ld = (size_t)am_hdp(hd) + ((!backward && am_hdp(hd) == 16) ? 0 : 1);   // forward, 16 columns: unpadded
  size_t fl = 2 * Lkp * ld;
  if (backward) fl += 2 * Lqp * ld + 2 * Lqp + 4;
  return fl * sizeof(float);
}

// 0 when the MFMA path does not cover the shape (head_dim 1..32; K, V (+ Q, dO) of one head must fit LDS)
size_t igcn_attn_mfma_lds_bytes(int D, int H, int Lq, int Lk, int backward) {
  if (H <= 0 || D <= 0 || D % H || Lq <= 0 || Lk <= 0) return 0;
  const int hd = D / H;
  if (hd > 32) return 0;
  const size_t bytes = am_lds_bytes(hd, Lq, Lk, backward);
  return bytes <= 150 * 1024 ? bytes : 0;
}

static int am_vec(int D, int hd, const void* a, const void* b, const void* c, const void* d) {
  return (hd % 4 == 0 && D % 4 == 0 && (uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0 && (uintptr_t)c % 16 == 0 &&
          (uintptr_t)d % 16 == 0) ? 1 : 0;
}

#define AM_DISPATCH(hdp, CALL)                                                                    \
  switch (hdp) {                                                                                   \
    case 4: CALL(4); break;   case 8: CALL(8); break;   case 12: CALL(12); break; case 16: CALL(16); break; \
    case 20: CALL(20); break; case 24: CALL(24); break; case 28: CALL(28); break; default: CALL(32); break; \
  }

int igcn_attn_mfma_fwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, float* o, float* lse,
                       hipStream_t st) {
  const int hd = D / H;
  const size_t lds = am_lds_bytes(hd, Lq, Lk, 0);
  const int vec = am_vec(D, hd, q, kv, o, o);
  int waves = (Lq + 15) / 16;
  if (waves > AM_MAX_WAVES) waves = AM_MAX_WAVES;
  if (waves < 4) waves = 4;                       // enough threads to stage K/V quickly
#define CALL(HDPV)                                                                                               \
  {                                                                                                              \
    IGCN_ALLOW_BIG_LDS((k_attn_mfma_fwd<HDPV, true>));                                        \
    IGCN_ALLOW_BIG_LDS((k_attn_mfma_fwd<HDPV, false>));                                        \
    if (vec && hd == HDPV)                                                                                       \
      hipLaunchKernelGGL((k_attn_mfma_fwd<HDPV, true>), dim3(am_grid(B * H)), dim3(64 * waves), lds, st, H, hd, vec, \
                         Lq, Lk, q, kv, o, lse, B * H);                                                                     \
    else                                                                                                         \
      hipLaunchKernelGGL((k_attn_mfma_fwd<HDPV, false>), dim3(am_grid(B * H)), dim3(64 * waves), lds, st, H, hd, vec, \
                         Lq, Lk, q, kv, o, lse, B * H);                                                                     \
  }
  AM_DISPATCH(am_hdp(hd), CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("attn_mfma_fwd");
  return IGCN_OK;
}

int igcn_attn_mfma_bwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, const float* o,
                       const float* lse, const float* dout, float* dq, float* dkv, hipStream_t st) {
  const int hd = D / H;
  const size_t lds = am_lds_bytes(hd, Lq, Lk, 1);
  const int vec = am_vec(D, hd, q, kv, dout, dq) && (uintptr_t)dkv % 16 == 0;
  const int tasks = (Lq + 15) / 16 + (Lk + 15) / 16;
  const int waves = tasks < 8 ? (tasks < 4 ? 4 : tasks) : 8;
  {
    // shared score tiles: head_dim 16 with 16-byte rows, <= 8 query tiles, >= 8 key tiles, partials fit over K | V
    const int nqt = (Lq + 15) / 16, nkt = (Lk + 15) / 16;
    const size_t lds2 = lds + (size_t)8 * 16 * AM_TLD * sizeof(float);
    if (vec && hd == 16 && nqt <= 8 && nkt >= 8 && (size_t)8 * nqt * 16 * 16 <= (size_t)2 * nkt * 16 * 17 &&
        lds2 <= 150 * 1024 && !igcn_opt(IGCN_OPT_ATTN_BWD_TWICE)) {
#define CALLS(NQTV)                                                                                              \
  case NQTV:                                                                                                     \
    IGCN_ALLOW_BIG_LDS((k_attn_mfma_bwd_shared<NQTV>));                                                          \
    hipLaunchKernelGGL((k_attn_mfma_bwd_shared<NQTV>), dim3(am_grid(B * H)), dim3(64 * 8), lds2, st, H, Lq, Lk, q, \
                       kv, o, lse, dout, dq, dkv, B * H);                                                        \
    break;
      switch (nqt) { CALLS(1) CALLS(2) CALLS(3) CALLS(4) CALLS(5) CALLS(6) CALLS(7) CALLS(8) }
#undef CALLS
      IGCN_CHECK_LAUNCH("attn_mfma_bwd_shared");
      return IGCN_OK;
    }
  }
#define CALL(HDPV)                                                                                               \
  {                                                                                                              \
    IGCN_ALLOW_BIG_LDS((k_attn_mfma_bwd<HDPV, true>));                                        \
    IGCN_ALLOW_BIG_LDS((k_attn_mfma_bwd<HDPV, false>));                                        \
    if (vec && hd == HDPV)                                                                                       \
      hipLaunchKernelGGL((k_attn_mfma_bwd<HDPV, true>), dim3(am_grid(B * H)), dim3(64 * waves), lds, st, H, hd, vec, \
                         Lq, Lk, q, kv, o, lse, dout, dq, dkv, B * H);                                           \
    else                                                                                                         \
      hipLaunchKernelGGL((k_attn_mfma_bwd<HDPV, false>), dim3(am_grid(B * H)), dim3(64 * waves), lds, st, H, hd, vec, \
                         Lq, Lk, q, kv, o, lse, dout, dq, dkv, B * H);                                           \
  }
  AM_DISPATCH(am_hdp(hd), CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("attn_mfma_bwd");
  return IGCN_OK;
}

// =================================================================================================
// Chunked variants for heads whose K, V (and Q, dO) do not fit LDS at once (e.g. 512 queries x 1300 keys): the
// same tile arithmetic, with the other side of the attention streamed through LDS in chunks.
//   forward / dQ : workgroup = (sample, head, block of 16*nw queries), wave = one 16-query tile, key chunks streamed
//   dK, dV       : workgroup = (sample, head, block of 16*nw keys),    wave = one 16-key tile,  query chunks streamed
// delta = rowsum(o * do) is written by the dQ kernel ([B,H,Lq] scratch) and read by the dK/dV kernel.
// =================================================================================================
template <int HDP>
__global__ void __launch_bounds__(64 * AM_MAX_WAVES)
k_attn_mfma_fwd_chunked(int H, int hd, int vec, int Lq, int Lk, int CH, const float* __restrict__ q,
                        const float* __restrict__ kv, float* __restrict__ o, float* __restrict__ lse) {
  constexpr int LD = HDP + 1, NC = HDP / 4, NO = (HDP + 15) / 16;
  extern __shared__ float smem[];
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * hd;
  float* Ks = smem;
  float* Vs = Ks + (size_t)CH * LD;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6, n = lane & 15, g = lane >> 4;
  const float scale = rsqrtf((float)hd);
  const int qi = (blockIdx.y * nw + w) * 16 + n;                 // this wave's query tile (may be past Lq)
  float qb[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c)
    qb[c] = (qi < Lq && 4 * c + g < hd) ? q[(int64_t)(b * Lq + qi) * D + h * hd + 4 * c + g] * scale : 0.f;
  float m = -INFINITY, l = 0.f;
  f32x4 oacc[NO];
#pragma unroll
  for (int t = 0; t < NO; ++t) oacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < Lk; k0 += CH) {
    const int kn = min(CH, Lk - k0), knp = (kn + 15) & ~15;
    __syncthreads();                                            // previous chunk fully consumed
    const float* kbase = kv + ((int64_t)b * Lk + k0) * 2 * D + h * hd;
    am_stage_pair<HDP>(kbase, kbase + D, 2 * D, hd, vec, kn, knp, Ks, Vs);
    __syncthreads();
    for (int kt = 0; kt < (knp >> 4); ++kt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* kr = Ks + (kt * 16 + n) * LD + g;
#pragma unroll
      for (int c = 0; c < NC; ++c) acc = mfma4(kr[4 * c], qb[c], acc);
      float s[4], tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[r] = (kt * 16 + 4 * g + r < kn) ? acc[r] : -INFINITY;
        tmax = fmaxf(tmax, s[r]);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float mn = fmaxf(m, tmax);
      const float f = __expf(m - mn);
      m = mn;
      l *= f;
#pragma unroll
      for (int t = 0; t < NO; ++t) oacc[t] *= f;
      const float* vr = Vs + (kt * 16 + 4 * g) * LD + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(s[r] - m);
        l += p;
#pragma unroll
        for (int t = 0; t < NO; ++t) oacc[t] = mfma4((16 * t + n < HDP) ? vr[r * LD + 16 * t] : 0.f, p, oacc[t]);
      }
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (qi < Lq) {
    const float inv = 1.f / l;
    float* op = o + (int64_t)(b * Lq + qi) * D + h * hd;
#pragma unroll
    for (int t = 0; t < NO; ++t)
      am_store4(op, 16 * t + 4 * g, hd, vec, oacc[t][0] * inv, oacc[t][1] * inv, oacc[t][2] * inv, oacc[t][3] * inv);
    if (g == 0) lse[((int64_t)b * H + h) * Lq + qi] = m + __logf(l);
  }
}

template <int HDP>
__global__ void __launch_bounds__(64 * AM_MAX_WAVES)
k_attn_mfma_bwd_dq_chunked(int H, int hd, int vec, int Lq, int Lk, int CH, const float* __restrict__ q,
                           const float* __restrict__ kv, const float* __restrict__ o, const float* __restrict__ lse,
                           const float* __restrict__ dout, float* __restrict__ dq, float* __restrict__ delta) {
  constexpr int LD = HDP + 1, NC = HDP / 4, NO = (HDP + 15) / 16;
  extern __shared__ float smem[];
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * hd;
  float* Ks = smem;
  float* Vs = Ks + (size_t)CH * LD;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6, n = lane & 15, g = lane >> 4;
  const float scale = rsqrtf((float)hd);
  const int qi = (blockIdx.y * nw + w) * 16 + n;
  const bool qlive = qi < Lq;
  const float* qrow = q + (int64_t)(b * Lq + (qlive ? qi : 0)) * D + h * hd;
  const float* dorow = dout + (int64_t)(b * Lq + (qlive ? qi : 0)) * D + h * hd;
  float qb[NC], dob[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const bool in = qlive && 4 * c + g < hd;
    qb[c] = in ? qrow[4 * c + g] * scale : 0.f;
    dob[c] = in ? dorow[4 * c + g] : 0.f;
  }
  // delta of this lane's query: the four lanes g = 0..3 of a column each sum a quarter of the head columns
  float dpart = 0.f;
  if (qlive) {
    const float* orow = o + (int64_t)(b * Lq + qi) * D + h * hd;
    for (int c = g; c < hd; c += 4) dpart += orow[c] * dorow[c];
  }
  dpart += __shfl_xor(dpart, 16, 64);
  dpart += __shfl_xor(dpart, 32, 64);
  const float dln = dpart;
  const float lsn = qlive ? lse[((int64_t)b * H + h) * Lq + qi] : INFINITY;
  if (qlive && g == 0) delta[((int64_t)b * H + h) * Lq + qi] = dln;
  f32x4 acc[NO];
#pragma unroll
  for (int t = 0; t < NO; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < Lk; k0 += CH) {
    const int kn = min(CH, Lk - k0), knp = (kn + 15) & ~15;
    __syncthreads();
    const float* kbase = kv + ((int64_t)b * Lk + k0) * 2 * D + h * hd;
    am_stage_pair<HDP>(kbase, kbase + D, 2 * D, hd, vec, kn, knp, Ks, Vs);
    __syncthreads();
    for (int kt = 0; kt < (knp >> 4); ++kt) {
      f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
      const float* kr = Ks + (kt * 16 + n) * LD + g;
      const float* vr = Vs + (kt * 16 + n) * LD + g;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        st = mfma4(kr[4 * c], qb[c], st);
        dpt = mfma4(vr[4 * c], dob[c], dpt);
      }
      const float* kc = Ks + (kt * 16 + 4 * g) * LD + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ds = __expf(st[r] - lsn) * (dpt[r] - dln) * scale;       // padded keys: zero K rows
#pragma unroll
        for (int t = 0; t < NO; ++t) acc[t] = mfma4((16 * t + n < HDP) ? kc[r * LD + 16 * t] : 0.f, ds, acc[t]);
      }
    }
  }
  if (qlive) {
    float* dst = dq + (int64_t)(b * Lq + qi) * D + h * hd;
#pragma unroll
    for (int t = 0; t < NO; ++t) am_store4(dst, 16 * t + 4 * g, hd, vec, acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  }
}

template <int HDP>
__global__ void __launch_bounds__(64 * AM_MAX_WAVES)
k_attn_mfma_bwd_dkv_chunked(int H, int hd, int vec, int Lq, int Lk, int CH, const float* __restrict__ q,
                            const float* __restrict__ kv, const float* __restrict__ lse,
                            const float* __restrict__ dout, const float* __restrict__ delta,
                            float* __restrict__ dkv) {
  constexpr int LD = HDP + 1, NC = HDP / 4, NO = (HDP + 15) / 16;
  extern __shared__ float smem[];
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * hd;
  float* Qs = smem;
  float* dOs = Qs + (size_t)CH * LD;
  float* ls = dOs + (size_t)CH * LD;                            // [CH]
  float* dl = ls + CH;                                          // [CH]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6, n = lane & 15, g = lane >> 4;
  const float scale = rsqrtf((float)hd);
  const int ki = (blockIdx.y * nw + w) * 16 + n;
  const bool klive = ki < Lk;
  const float* krow = kv + ((int64_t)(b * Lk + (klive ? ki : 0)) * 2) * D + h * hd;
  float kb[NC], vb[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const bool in = klive && 4 * c + g < hd;
    kb[c] = in ? krow[4 * c + g] * scale : 0.f;
    vb[c] = in ? krow[D + 4 * c + g] : 0.f;
  }
  f32x4 dka[NO], dva[NO];
#pragma unroll
  for (int t = 0; t < NO; ++t) dka[t] = dva[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int q0 = 0; q0 < Lq; q0 += CH) {
    const int qn = min(CH, Lq - q0), qnp = (qn + 15) & ~15;
    __syncthreads();
    am_stage_pair<HDP>(q + ((int64_t)b * Lq + q0) * D + h * hd, dout + ((int64_t)b * Lq + q0) * D + h * hd, D, hd, vec,
                       qn, qnp, Qs, dOs);
    for (int r = threadIdx.x; r < qnp; r += blockDim.x) {
      const bool in = r < qn;
      ls[r] = in ? lse[((int64_t)b * H + h) * Lq + q0 + r] : INFINITY;       // padding queries: p = 0
      dl[r] = in ? delta[((int64_t)b * H + h) * Lq + q0 + r] : 0.f;
    }
    __syncthreads();
    for (int qt = 0; qt < (qnp >> 4); ++qt) {
      f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
      const float* qr = Qs + (qt * 16 + n) * LD + g;
      const float* dr = dOs + (qt * 16 + n) * LD + g;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        st = mfma4(qr[4 * c], kb[c], st);
        dpt = mfma4(dr[4 * c], vb[c], dpt);
      }
      const float* qc = Qs + (qt * 16 + 4 * g) * LD + n;
      const float* dc = dOs + (qt * 16 + 4 * g) * LD + n;
      const float* lsr = ls + qt * 16 + 4 * g;
      const float* dlr = dl + qt * 16 + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(st[r] - lsr[r]);
        const float ds = p * (dpt[r] - dlr[r]) * scale;
#pragma unroll
        for (int t = 0; t < NO; ++t) {
          const bool in = 16 * t + n < HDP;
          dva[t] = mfma4(in ? dc[r * LD + 16 * t] : 0.f, p, dva[t]);
          dka[t] = mfma4(in ? qc[r * LD + 16 * t] : 0.f, ds, dka[t]);
        }
      }
    }
  }
  if (klive) {
    float* base = dkv + ((int64_t)(b * Lk + ki) * 2) * D + h * hd;
#pragma unroll
    for (int t = 0; t < NO; ++t) {
      am_store4(base, 16 * t + 4 * g, hd, vec, dka[t][0], dka[t][1], dka[t][2], dka[t][3]);
      am_store4(base + D, 16 * t + 4 * g, hd, vec, dva[t][0], dva[t][1], dva[t][2], dva[t][3]);
    }
  }
}

// rows of the other side streamed per chunk: as many as ~96 KB of LDS hold (two matrices of HDP+1 columns)
// `keys`: the chunk streams KEYS past resident query tiles (forward, dQ).  There a third of the LDS-filling chunk is
// better: 336 rows = 46 KB per workgroup, three workgroups per CU instead of one, so that one stages while the others
// multiply (configs[4], 512 x 1300: forward 99 -> 85 us, dQ 109 -> 97 us; 448 / 224 / 160 rows: 85 / 87 / 88 and
// 97 / 99 / 100).  The dK | dV kernel streams QUERIES (at most Lq rows: unchanged by the cap).
static int am_chunk_rows(int hd, bool keys = false) {
  const int ld = am_hdp(hd) + 1;
  int ch = (96 * 1024) / (2 * ld * 4 + 8);
  if (keys && ch > 336) ch = 336;
  if (g_igcn_attn_chunk_rows > 0 && g_igcn_attn_chunk_rows < ch) ch = g_igcn_attn_chunk_rows;   // sweeps (igcn_configure)
  ch &= ~15;
  return ch < 16 ? 16 : ch;
}

size_t igcn_attn_mfma_chunked_scratch_floats(int B, int H, int Lq) { return (size_t)B * H * Lq + 16; }

int igcn_attn_mfma_fwd_chunked(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, float* o,
                               float* lse, hipStream_t st) {
  const int hd = D / H, vec = am_vec(D, hd, q, kv, o, o);
  int ch = am_chunk_rows(hd, true);
  if (ch > ((Lk + 15) & ~15)) ch = (Lk + 15) & ~15;
  const size_t lds = (size_t)2 * ch * (am_hdp(hd) + 1) * sizeof(float);
  const int nqt = (Lq + 15) / 16, nw = nqt < 8 ? (nqt < 4 ? 4 : nqt) : 8;
  dim3 grid(B * H, (nqt + nw - 1) / nw);
#define CALL(HDPV)                                                                                               \
  {                                                                                                              \
    IGCN_ALLOW_BIG_LDS((k_attn_mfma_fwd_chunked<HDPV>));                                        \
    hipLaunchKernelGGL((k_attn_mfma_fwd_chunked<HDPV>), grid, dim3(64 * nw), lds, st, H, hd, vec, Lq, Lk, ch, q,  \
                       kv, o, lse);                                                                              \
  }
  AM_DISPATCH(am_hdp(hd), CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("attn_mfma_fwd_chunked");
  return IGCN_OK;
}

int igcn_attn_mfma_bwd_chunked(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, const float* o,
                               const float* lse, const float* dout, float* dq, float* dkv, float* delta,
                               hipStream_t st) {
  const int hd = D / H;
  const int vec = am_vec(D, hd, q, kv, dout, dq) && (uintptr_t)dkv % 16 == 0;
  int chk = am_chunk_rows(hd, true), chq = am_chunk_rows(hd);
  if (chk > ((Lk + 15) & ~15)) chk = (Lk + 15) & ~15;
  if (chq > ((Lq + 15) & ~15)) chq = (Lq + 15) & ~15;
  const size_t ld = (size_t)am_hdp(hd) + 1;
  const size_t lds_q = (size_t)2 * chk * ld * sizeof(float);
  const size_t lds_k = ((size_t)2 * chq * ld + 2 * chq) * sizeof(float);
  const int nqt = (Lq + 15) / 16, nkt = (Lk + 15) / 16;
  const int nwq = nqt < 8 ? (nqt < 4 ? 4 : nqt) : 8, nwk = nkt < 8 ? (nkt < 4 ? 4 : nkt) : 8;
  dim3 gq(B * H, (nqt + nwq - 1) / nwq), gk(B * H, (nkt + nwk - 1) / nwk);
#define CALL(HDPV)                                                                                               \
  {                                                                                                              \
    IGCN_ALLOW_BIG_LDS((k_attn_mfma_bwd_dq_chunked<HDPV>));                                        \
    IGCN_ALLOW_BIG_LDS((k_attn_mfma_bwd_dkv_chunked<HDPV>));                                        \
    hipLaunchKernelGGL((k_attn_mfma_bwd_dq_chunked<HDPV>), gq, dim3(64 * nwq), lds_q, st, H, hd, vec, Lq, Lk, chk, \
                       q, kv, o, lse, dout, dq, delta);                                                          \
    hipLaunchKernelGGL((k_attn_mfma_bwd_dkv_chunked<HDPV>), gk, dim3(64 * nwk), lds_k, st, H, hd, vec, Lq, Lk,    \
                       chq, q, kv, lse, dout, delta, dkv);                                                       \
  }
  AM_DISPATCH(am_hdp(hd), CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("attn_mfma_bwd_chunked");
  return IGCN_OK;
}
