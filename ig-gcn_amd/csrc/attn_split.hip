// Cross-attention core on SPLIT bf16 operands (kernel/sgcn_img_snp.py:240, the nn.MultiheadAttention core): fp32-grade
// results at the bf16 rate of the matrix cores.  Every fp32 operand x is carried as a bf16 head and a bf16 remainder,
//     x = hi + lo (+ 2^-17 |x|),   hi = bf16(x),  lo = bf16(x - hi),
// and every product of the attention runs on v_mfma_f32_16x16x32_bf16 (16 cycles per SIMD) instead of four
// v_mfma_f32_16x16x4_f32 (32 cycles each) per 16-deep reduction:
//   * reductions over the head dimension (16 = HALF the instruction's depth): the two halves of the depth carry the head
//     and the remainder of one operand, A = [a_hi | a_lo], against B = [b_hi | b_hi] and then B = [b_lo | b_lo] —
//     TWO instructions give all four partial products (a_hi + a_lo)(b_hi + b_lo);
//   * reductions over keys / queries (depth 32 = one step of the streamed side): hi.hi + lo.hi + hi.lo, THREE
//     instructions (the lo.lo term, 2^-16 of the product, is dropped).
// Softmax statistics, exponentials, the log-sum-exp and delta = rowsum(o do) are fp32 as everywhere; q, k | v, o and every
// gradient stay fp32 in HBM.  head_dim must be 16 (the model's 32-wide embedding over 2 heads).
// Error against an fp64 evaluation: tools/attn_error.py; tests/test_gpu_ops.py holds the 1e-4 / 1e-3 bounds of the exact
// fp32 kernels (csrc/attn_mfma.hip) on this path.
#include <stdlib.h>

#include "attn_bf16_common.h"

#define AS_MAX_WAVES 8
#define AS_LOG2E 1.4426950408889634f
#define AS_LN2 0.6931471805599453f
#define AS_LAZY 8.f

__device__ __forceinline__ float as_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// x = hi + lo
__device__ __forceinline__ __bf16 as_hi(float x) { return (__bf16)x; }
__device__ __forceinline__ __bf16 as_lo(float x, __bf16 hi) { return (__bf16)(x - (float)hi); }
__device__ __forceinline__ void as_split4(const float4 v, bf16x4& hi, bf16x4& lo) {
  hi = bf16x4{as_hi(v.x), as_hi(v.y), as_hi(v.z), as_hi(v.w)};
  lo = bf16x4{as_lo(v.x, hi[0]), as_lo(v.y, hi[1]), as_lo(v.z, hi[2]), as_lo(v.w, hi[3])};
}
// eight accumulator values of a 32-row step (tile 0 rows 4 g + r | tile 1 rows 16 + 4 g + r) -> the step's B operand, twice
__device__ __forceinline__ void as_pack_split(const float a[4], const float b[4], bf16x8& hi, bf16x8& lo) {
  hi = bf16x8{as_hi(a[0]), as_hi(a[1]), as_hi(a[2]), as_hi(a[3]), as_hi(b[0]), as_hi(b[1]), as_hi(b[2]), as_hi(b[3])};
  lo = bf16x8{as_lo(a[0], hi[0]), as_lo(a[1], hi[1]), as_lo(a[2], hi[2]), as_lo(a[3], hi[3]),
              as_lo(b[0], hi[4]), as_lo(b[1], hi[5]), as_lo(b[2], hi[6]), as_lo(b[3], hi[7])};
}

// Eight consecutive floats (columns 8 (g & 1) ..) of a resident row, scaled, as the operand pair of a head-dimension
// reduction: `hi` = heads in every lane group, `lo` = remainders in every lane group (the streamed side's fragment puts
// its heads in lane groups 0-1 and its remainders in 2-3: see as_rfrag)
__device__ __forceinline__ void as_row8(const float* __restrict__ row, bool live, float scale, bf16x8& hi, bf16x8& lo) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (live) {
    a = *reinterpret_cast<const float4*>(row);
    b = *reinterpret_cast<const float4*>(row + 4);
  }
  const float va[4] = {a.x * scale, a.y * scale, a.z * scale, a.w * scale};
  const float vb[4] = {b.x * scale, b.y * scale, b.z * scale, b.w * scale};
  as_pack_split(va, vb, hi, lo);
}

// row-major operand of tile `tile` of a [2][rows][16] (heads | remainders) array: lane groups 0-1 read the heads of row n
// (columns 8 (g & 1) ..), lane groups 2-3 the remainders of the same columns: one 16-byte LDS read per lane
__device__ __forceinline__ bf16x8 as_rfrag(const __bf16* __restrict__ r, int rows, int tile, int n, int g) {
  return *reinterpret_cast<const bf16x8*>(r + ((g >> 1) * rows + tile * 16 + n) * AB_HD + 8 * (g & 1));
}

// Rows [0, rows_valid) of K and V (row_stride-strided fp32, 16 floats per row) -> LDS: K row-major heads | remainders
// [2][rows_pad][16], V transposed heads and remainders [16][ldt] each.  Rows up to rows_pad (a multiple of 32): zeros.
__device__ __forceinline__ void as_stage_kv(const float* __restrict__ k, const float* __restrict__ v, int64_t row_stride,
                                            int rows_valid, int rows_pad, int ldt, __bf16* __restrict__ kr,
                                            __bf16* __restrict__ vth, __bf16* __restrict__ vtl) {
  const int items = rows_pad * 2;
  for (int t = threadIdx.x; t < items; t += blockDim.x) {
    const int c = (t & 3) * 4, r0 = (t >> 2) * 2;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a0 = z, a1 = z, b0 = z, b1 = z;
    if (r0 < rows_valid) {
      a0 = *reinterpret_cast<const float4*>(k + (int64_t)r0 * row_stride + c);
      b0 = *reinterpret_cast<const float4*>(v + (int64_t)r0 * row_stride + c);
    }
    if (r0 + 1 < rows_valid) {
      a1 = *reinterpret_cast<const float4*>(k + (int64_t)(r0 + 1) * row_stride + c);
      b1 = *reinterpret_cast<const float4*>(v + (int64_t)(r0 + 1) * row_stride + c);
    }
    bf16x4 h0, l0, h1, l1;
    as_split4(a0, h0, l0);
    as_split4(a1, h1, l1);
    *reinterpret_cast<bf16x4*>(kr + r0 * AB_HD + c) = h0;
    *reinterpret_cast<bf16x4*>(kr + (r0 + 1) * AB_HD + c) = h1;
    *reinterpret_cast<bf16x4*>(kr + (rows_pad + r0) * AB_HD + c) = l0;
    *reinterpret_cast<bf16x4*>(kr + (rows_pad + r0 + 1) * AB_HD + c) = l1;
    as_split4(b0, h0, l0);
    as_split4(b1, h1, l1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      *reinterpret_cast<bf16x2*>(vth + (c + e) * ldt + r0) = bf16x2{h0[e], h1[e]};
      *reinterpret_cast<bf16x2*>(vtl + (c + e) * ldt + r0) = bf16x2{l0[e], l1[e]};
    }
  }
}

// -------------------------------------------------------------------------------------------------------------
// forward: o = softmax(q k^T / 4) v, lse = log sum exp of the scaled scores.  Structure of k_attn_bf16_fwd (scores in the
// log2 domain, lazy running maximum, the next step's score products issued ahead of the softmax); per 32-key step
// 2 x 2 score instructions and 3 for p v.
// -------------------------------------------------------------------------------------------------------------
template <bool MASK>
__device__ __forceinline__ void as_fwd_step(const f32x4 s0, const f32x4 s1, int key0, int kn, const bf16x8 vh,
                                            const bf16x8 vl, float& m, float& l, f32x4& oacc) {
  float a[4], c[4], tl = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    a[r] = (!MASK || key0 + r < kn) ? s0[r] : -INFINITY;
    c[r] = (!MASK || key0 + 16 + r < kn) ? s1[r] : -INFINITY;
    tl = fmaxf(tl, fmaxf(a[r], c[r]));
  }
  if (__any(tl > m + AS_LAZY)) {                                  // wave-uniform; always taken by a head's first step
    tl = fmaxf(tl, __shfl_xor(tl, 16, 64));
    tl = fmaxf(tl, __shfl_xor(tl, 32, 64));
    const float mn = fmaxf(m, tl);
    const float f = as_exp2(m - mn);
    m = mn;
    l *= f;
    oacc *= f;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    a[r] = as_exp2(a[r] - m);
    c[r] = as_exp2(c[r] - m);
    l += a[r] + c[r];
  }
  bf16x8 ph, pl;
  as_pack_split(a, c, ph, pl);
  oacc = mfma32(vh, ph, oacc);
  oacc = mfma32(vl, ph, oacc);
  oacc = mfma32(vh, pl, oacc);
}

__global__ void __launch_bounds__(64 * AS_MAX_WAVES)
k_attn_split_fwd(int H, int Lq, int Lk, int CH, const float* __restrict__ q, const float* __restrict__ kv,
                 float* __restrict__ o, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char as_smem[];
  const int ldt = ab_ldt(CH);
  __bf16* Kr = reinterpret_cast<__bf16*>(as_smem);                // [2][CH][16]  heads | remainders
  __bf16* Vth = Kr + (size_t)2 * CH * AB_HD;                      // [16][ldt]
  __bf16* Vtl = Vth + (size_t)AB_HD * ldt;                        // [16][ldt]
  const int item = ab_item(), b = item / H, h = item % H, D = H * AB_HD;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6, n = lane & 15, g = lane >> 4;
  const int qi = (blockIdx.y * nw + w) * 16 + n;                  // this wave's query tile (may be past Lq)
  const bool qlive = qi < Lq;
  bf16x8 qh, ql;
  as_row8(q + (int64_t)(b * Lq + (qlive ? qi : 0)) * D + h * AB_HD + 8 * (g & 1), qlive, 0.25f * AS_LOG2E, qh, ql);
  float m = -INFINITY, l = 0.f;
  f32x4 oacc = {0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  auto score = [&](int tile) {
    const bf16x8 kf = as_rfrag(Kr, CH, tile, n, g);               // [k_hi | k_lo] against [q_hi | q_hi], then [q_lo | q_lo]
    return mfma32(kf, ql, mfma32(kf, qh, zero));
  };
  for (int k0 = 0; k0 < Lk; k0 += CH) {
    const int kn = min(CH, Lk - k0), knp = (kn + 31) & ~31;
    __syncthreads();                                              // previous chunk fully consumed
    const float* kbase = kv + ((int64_t)b * Lk + k0) * 2 * D + h * AB_HD;
    as_stage_kv(kbase, kbase + D, 2 * D, kn, CH, ldt, Kr, Vth, Vtl);
    __syncthreads();
    const int nst = knp >> 5, nfull = kn >> 5;                    // steps; steps without padded keys
    f32x4 s0 = score(0), s1 = score(1);
    for (int st = 0; st < nfull; ++st) {
      const int nx = min(st + 1, nst - 1);
      const f32x4 n0 = score(2 * nx), n1 = score(2 * nx + 1);
      as_fwd_step<false>(s0, s1, 0, 0, ab_tfrag(Vth, ldt, n, g, st), ab_tfrag(Vtl, ldt, n, g, st), m, l, oacc);
      s0 = n0;
      s1 = n1;
    }
    if (nfull < nst)
      as_fwd_step<true>(s0, s1, nfull * 32 + 4 * g, kn, ab_tfrag(Vth, ldt, n, g, nfull), ab_tfrag(Vtl, ldt, n, g, nfull), m, l,
                        oacc);
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (qlive) {
    const float inv = 1.f / l;
    *reinterpret_cast<float4*>(o + (int64_t)(b * Lq + qi) * D + h * AB_HD + 4 * g) =
        make_float4(oacc[0] * inv, oacc[1] * inv, oacc[2] * inv, oacc[3] * inv);
    if (g == 0) lse[((int64_t)b * H + h) * Lq + qi] = (m + __log2f(l)) * AS_LN2;
  }
}

// -------------------------------------------------------------------------------------------------------------
// backward, one workgroup per (sample, head) with SHARED score tiles (the decomposition of k_attn_mfma_bwd_shared,
// csrc/attn_mfma.hip: every (key tile, query tile) pair is computed once, in the S orientation; the pairs are cut into one
// contiguous chunk per wave; dQ accumulates per query tile in registers and the waves' partials meet in LDS at the end;
// a key tile split between two waves is completed by the second).  Per pair TEN matrix instructions of 16 cycles instead
// of twenty of 32:
//   S  = Q K^T, dP = dO V^T      A = [x_hi | x_lo] of the query rows (one 16-byte LDS read), B = the key tile's K / V from
//                                 REGISTERS (loaded from HBM once per key tile: K | V are not staged at all)
//   dV^T += dO^T P, dK^T += Q^T dS   reduction over the tile's 16 queries: the lane's own four accumulator values, split,
//                                 ARE the B operand ([p_hi(4) | p_lo(4)]); A = the transposed operand's four values, twice
//   dQ^T += K^T dS^T              dS transposed through a wave-private LDS tile (as the fp32 kernel), split after the
//                                 read; A = the key tile's K transposed through a second wave-private tile, once per tile
// LDS: Q, dO row-major (hi | lo) and transposed (hi, lo), lse, delta, one small tile per wave — and, over it at the end,
// the waves' dQ partials (a tree: at most four at a time): 37 KB at 96 queries, four workgroups per CU.
// -------------------------------------------------------------------------------------------------------------
#define AS_TLD 20                                               // transpose tile row stride (16-byte aligned rows)
#define AS_KLD 17                                               // K tile row stride (conflict-free column reads)
__host__ __device__ inline int as_ldt(int rows) { return rows + 16; }     // rows % 32 == 0: rows start 8 banks apart

__device__ __forceinline__ bf16x8 as_dup(const bf16x4 a) { return bf16x8{a[0], a[1], a[2], a[3], a[0], a[1], a[2], a[3]}; }
// four fp32 values -> [hi(4) | lo(4)]: the B operand of a 16-deep reduction whose A operand holds its values twice
__device__ __forceinline__ bf16x8 as_hilo(const float v[4]) {
  const __bf16 h0 = as_hi(v[0]), h1 = as_hi(v[1]), h2 = as_hi(v[2]), h3 = as_hi(v[3]);
  return bf16x8{h0, h1, h2, h3, as_lo(v[0], h0), as_lo(v[1], h1), as_lo(v[2], h2), as_lo(v[3], h3)};
}

// rows of a [*, D]-strided fp32 source (16 floats per row) -> row-major heads | remainders [2][rows_pad][16] and
// transposed heads, remainders [16][ldt]; rows up to rows_pad are zeros
__device__ __forceinline__ void as_stage_rt(const float* __restrict__ src, int64_t row_stride, int rows_valid, int rows_pad,
                                            int ldt, __bf16* __restrict__ rr, __bf16* __restrict__ th,
                                            __bf16* __restrict__ tl) {
  for (int t = threadIdx.x; t < rows_pad * 2; t += blockDim.x) {
    const int c = (t & 3) * 4, r0 = (t >> 2) * 2;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (r0 < rows_valid) a0 = *reinterpret_cast<const float4*>(src + (int64_t)r0 * row_stride + c);
    if (r0 + 1 < rows_valid) a1 = *reinterpret_cast<const float4*>(src + (int64_t)(r0 + 1) * row_stride + c);
    bf16x4 h0, l0, h1, l1;
    as_split4(a0, h0, l0);
    as_split4(a1, h1, l1);
    *reinterpret_cast<bf16x4*>(rr + r0 * AB_HD + c) = h0;
    *reinterpret_cast<bf16x4*>(rr + (r0 + 1) * AB_HD + c) = h1;
    *reinterpret_cast<bf16x4*>(rr + (rows_pad + r0) * AB_HD + c) = l0;
    *reinterpret_cast<bf16x4*>(rr + (rows_pad + r0 + 1) * AB_HD + c) = l1;
    // transposed, every group of four rows TWICE in a row ([x0 x1 x2 x3 x0 x1 x2 x3]): a 16-byte read is then the A operand
    // of a 16-deep reduction against B = [b_hi(4) | b_lo(4)] as it stands (eight register moves less per pair)
    const int pos = (r0 >> 2) * 8 + (r0 & 3);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bf16x2 hh = bf16x2{h0[e], h1[e]}, ll = bf16x2{l0[e], l1[e]};
      *reinterpret_cast<bf16x2*>(th + (c + e) * ldt + pos) = hh;
      *reinterpret_cast<bf16x2*>(th + (c + e) * ldt + pos + 4) = hh;
      *reinterpret_cast<bf16x2*>(tl + (c + e) * ldt + pos) = ll;
      *reinterpret_cast<bf16x2*>(tl + (c + e) * ldt + pos + 4) = ll;
    }
  }
}

template <int NQT>
__global__ void __launch_bounds__(64 * 8)
k_attn_split_bwd(int H, int Lq, int Lk, const float* __restrict__ q, const float* __restrict__ kv,
                 const float* __restrict__ o, const float* __restrict__ lse, const float* __restrict__ dout,
                 float* __restrict__ dq, float* __restrict__ dkv) {
  constexpr int Lqp = NQT * 16, LDT = 2 * Lqp + 16;             // transposed rows hold every value twice (as_stage_rt)
  extern __shared__ __attribute__((aligned(16))) unsigned char as_smem[];
  const int item = ab_item(), b = item / H, h = item % H, D = H * AB_HD;
  const int nkt = (Lk + 15) >> 4;
  __bf16* Qr = reinterpret_cast<__bf16*>(as_smem);              // [2][Lqp][16]
  __bf16* Or = Qr + 2 * Lqp * AB_HD;                            // [2][Lqp][16]   dO
  __bf16* Qth = Or + 2 * Lqp * AB_HD;                           // [16][LDT]
  __bf16* Qtl = Qth + AB_HD * LDT;
  __bf16* Oth = Qtl + AB_HD * LDT;
  __bf16* Otl = Oth + AB_HD * LDT;
  float* ls = reinterpret_cast<float*>(Otl + AB_HD * LDT);      // [Lqp]  lse in the log2 domain (+inf on padding rows)
  float* dl = ls + Lqp;                                         // [Lqp]  delta
  float* Tw = dl + Lqp;                                         // [8][16][AS_TLD]  per wave: the key tile's K (stride AS_KLD),
                                                                //                  then the dS transpose tile of each pair
  const float* qbase = q + (int64_t)b * Lq * D + h * AB_HD;
  const float* dobase = dout + (int64_t)b * Lq * D + h * AB_HD;
  {
    const int r = threadIdx.x;                                  // delta = rowsum(o * do), lse: thread r's query row
    float d = 0.f, lv = INFINITY;
    if (r < Lq) {
      const float* op = o + (int64_t)(b * Lq + r) * D + h * AB_HD;
      const float* dp = dobase + (int64_t)r * D;
#pragma unroll
      for (int c = 0; c < AB_HD; c += 4) {
        const float4 a = *reinterpret_cast<const float4*>(op + c), d4 = *reinterpret_cast<const float4*>(dp + c);
        d += a.x * d4.x + a.y * d4.y + a.z * d4.z + a.w * d4.w;
      }
      lv = lse[((int64_t)b * H + h) * Lq + r] * AS_LOG2E;
    }
    as_stage_rt(qbase, D, Lq, Lqp, LDT, Qr, Qth, Qtl);
    as_stage_rt(dobase, D, Lq, Lqp, LDT, Or, Oth, Otl);
    if (r < Lqp) {
      dl[r] = d;
      ls[r] = lv;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6, n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* T = Tw + w * 16 * AS_TLD;
  float* KT = T;
  const int P = nkt * NQT, lo = (w * P) / nw, hi = ((w + 1) * P) / nw;           // this wave's pairs [lo, hi)
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 dqa[NQT];                                               // dQ^T[hd 4g+r][query n] per query tile
#pragma unroll
  for (int t = 0; t < NQT; ++t) dqa[t] = zero;
  f32x4 pend_k = zero, pend_v = zero;                           // first tile's part, if its head is elsewhere
  int pend_row = -1;
  const int kt_first = lo / NQT, kt_last = (hi - 1) / NQT;
  // a key tile's K and V rows come straight from HBM: lane (n, g) columns 8 (g & 1) .. + 7 of key kt * 16 + n; the NEXT
  // tile's are requested while the current tile's pairs run
  float4 nk0 = make_float4(0.f, 0.f, 0.f, 0.f), nk1 = nk0, nv0 = nk0, nv1 = nk0;
  auto fetch = [&](int kt) {
    const int row = kt * 16 + n;
    nk0 = nk1 = nv0 = nv1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < Lk) {
      const float* kp = kv + ((int64_t)(b * Lk + row) * 2) * D + h * AB_HD + 8 * (g & 1);
      nk0 = *reinterpret_cast<const float4*>(kp);
      nk1 = *reinterpret_cast<const float4*>(kp + 4);
      nv0 = *reinterpret_cast<const float4*>(kp + D);
      nv1 = *reinterpret_cast<const float4*>(kp + D + 4);
    }
  };
  if (lo < hi) fetch(kt_first);
  for (int kt = kt_first; kt <= kt_last && lo < hi; ++kt) {
    const int q_lo = kt == kt_first ? lo - kt * NQT : 0, q_hi = kt == kt_last ? hi - kt * NQT : NQT;
    const int krow = kt * 16 + n;
    const float4 k0 = nk0, k1 = nk1, v0 = nv0, v1 = nv1;
    if (kt < kt_last) fetch(kt + 1);
    bf16x8 kh, kl, vh, vl;
    {
      const float c = 0.25f * AS_LOG2E;                         // scores in the log2 domain
      const float ka[4] = {k0.x * c, k0.y * c, k0.z * c, k0.w * c}, kb[4] = {k1.x * c, k1.y * c, k1.z * c, k1.w * c};
      as_pack_split(ka, kb, kh, kl);
      const float va[4] = {v0.x, v0.y, v0.z, v0.w}, vb[4] = {v1.x, v1.y, v1.z, v1.w};
      as_pack_split(va, vb, vh, vl);
    }
    // K^T of the tile for the dQ product: lane (n = hd, g) needs K[keys 4g .. 4g+3][hd n]
    if (g < 2) {
      float* kw = KT + n * AS_KLD + 8 * g;
      kw[0] = k0.x; kw[1] = k0.y; kw[2] = k0.z; kw[3] = k0.w;
      kw[4] = k1.x; kw[5] = k1.y; kw[6] = k1.z; kw[7] = k1.w;
    }
    asm volatile("" ::: "memory");       // (a wave's LDS operations execute in issue order: no barrier for a private tile)
    float kc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) kc[r] = KT[(4 * g + r) * AS_KLD + n];
    asm volatile("" ::: "memory");
    bf16x8 kth, ktl;
    {
      const bf16x8 hl = as_hilo(kc);
      kth = bf16x8{hl[0], hl[1], hl[2], hl[3], hl[0], hl[1], hl[2], hl[3]};
      ktl = bf16x8{hl[4], hl[5], hl[6], hl[7], hl[4], hl[5], hl[6], hl[7]};
    }
    f32x4 dka = zero, dva = zero;                               // dK^T / dV^T [hd 4g+r][key n]
    auto pair = [&](const int qt) {
        const bf16x8 qf = as_rfrag(Qr, Lqp, qt, n, g), of = as_rfrag(Or, Lqp, qt, n, g);
        const f32x4 st = mfma32(qf, kl, mfma32(qf, kh, zero));  // S (log2 domain): rows = queries 4g+r, column = key n
        const f32x4 dpt = mfma32(of, vl, mfma32(of, vh, zero)); // dP = dO V^T
        const float4 l4 = *reinterpret_cast<const float4*>(ls + qt * 16 + 4 * g);
        const float4 e4 = *reinterpret_cast<const float4*>(dl + qt * 16 + 4 * g);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
        float p[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = as_exp2(st[r] - lv[r]);                        // padding queries: lse = +inf -> p = 0
          ds[r] = p[r] * (dpt[r] - ev[r]) * 0.25f;
          T[(4 * g + r) * AS_TLD + n] = ds[r];
        }
        const bf16x8 pb = as_hilo(p), db = as_hilo(ds);
        const int to = n * LDT + (qt * 16 + 4 * g) * 2;          // [x(4) | x(4)] of queries 4g .. 4g+3 of the tile, row n
        dva = mfma32(*reinterpret_cast<const bf16x8*>(Oth + to), pb, dva);     // dV^T += dO^T P
        dva = mfma32(*reinterpret_cast<const bf16x8*>(Otl + to), pb, dva);
        dka = mfma32(*reinterpret_cast<const bf16x8*>(Qth + to), db, dka);     // dK^T += Q^T dS
        dka = mfma32(*reinterpret_cast<const bf16x8*>(Qtl + to), db, dka);
        asm volatile("" ::: "memory");   // compiler-level order only: a wave's LDS operations execute in issue order
        const float4 dst = *reinterpret_cast<const float4*>(T + n * AS_TLD + 4 * g);   // dS^T: query n, keys 4g .. 4g+3
        asm volatile("" ::: "memory");   // (and the next pair's writes stay behind this read)
        const float dt[4] = {dst.x, dst.y, dst.z, dst.w};
        const bf16x8 tb = as_hilo(dt);
        dqa[qt] = mfma32(kth, tb, dqa[qt]);                     // dQ^T += K^T dS^T
        dqa[qt] = mfma32(ktl, tb, dqa[qt]);
    };
    // [All NQT pairs of a whole key tile as one straight-line block — so that the scheduler may overlap neighbouring pairs —
    // needs 168 registers (one workgroup per CU: 94 us against 58); so does a second transpose tile taken in turn.  At 126
    // registers two workgroups share a CU and the kernel is bound by its vector-instruction issue: ~95 per pair.]
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
      if (qt >= q_lo && qt < q_hi) pair(qt);                    // wave-uniform
    if (q_lo > 0) {                                             // the tile's first queries belong to the previous wave
      pend_k = dka;
      pend_v = dva;
      pend_row = krow;
    } else if (krow < Lk) {                                     // whole tile, or its first part (completed after the barrier)
      float* base = dkv + ((int64_t)(b * Lk + krow) * 2) * D + h * AB_HD + 4 * g;
      *reinterpret_cast<float4*>(base) = make_float4(dka[0], dka[1], dka[2], dka[3]);
      *reinterpret_cast<float4*>(base + D) = make_float4(dva[0], dva[1], dva[2], dva[3]);
    }
  }
  __syncthreads();                                              // first parts visible to this workgroup; LDS operands dead
  if (pend_row >= 0 && pend_row < Lk) {
    float* base = dkv + ((int64_t)(b * Lk + pend_row) * 2) * D + h * AB_HD + 4 * g;
    const float4 hk = *reinterpret_cast<const float4*>(base), hv = *reinterpret_cast<const float4*>(base + D);
    *reinterpret_cast<float4*>(base) = make_float4(hk.x + pend_k[0], hk.y + pend_k[1], hk.z + pend_k[2], hk.w + pend_k[3]);
    *reinterpret_cast<float4*>(base + D) = make_float4(hv.x + pend_v[0], hv.y + pend_v[1], hv.z + pend_v[2], hv.w + pend_v[3]);
  }
  // dQ: the waves' partials meet pairwise over the (dead) operand area — waves 4-7 hand theirs to 0-3, 2-3 to 0-1, 1 to 0 —
  // a fixed tree, ((w0 + w4) + (w2 + w6)) + ((w1 + w5) + (w3 + w7)); wave 0 stores the sums
  float* R = reinterpret_cast<float*>(as_smem);
  for (int half = 4; half >= 1; half >>= 1) {
    if (w >= half && w < 2 * half) {
#pragma unroll
      for (int t = 0; t < NQT; ++t)
        *reinterpret_cast<float4*>(R + ((size_t)((w - half) * Lqp + t * 16 + n)) * AB_HD + 4 * g) =
            make_float4(dqa[t][0], dqa[t][1], dqa[t][2], dqa[t][3]);
    }
    __syncthreads();
    if (w < half) {
#pragma unroll
      for (int t = 0; t < NQT; ++t) {
        const float4 a = *reinterpret_cast<const float4*>(R + ((size_t)(w * Lqp + t * 16 + n)) * AB_HD + 4 * g);
        dqa[t][0] += a.x; dqa[t][1] += a.y; dqa[t][2] += a.z; dqa[t][3] += a.w;
      }
    }
    __syncthreads();
  }
  if (w == 0) {
#pragma unroll
    for (int t = 0; t < NQT; ++t)
      if (t * 16 + n < Lq)
        *reinterpret_cast<float4*>(dq + (int64_t)(b * Lq + t * 16 + n) * D + h * AB_HD + 4 * g) =
            make_float4(dqa[t][0], dqa[t][1], dqa[t][2], dqa[t][3]);
  }
}

// -------------------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------------------
// keys per LDS chunk of the forward: 30 KB of LDS (five workgroups per CU) at 224; sweep at 512 x 2 heads, 90 x 400:
// 448 / 224 / 160 / 128 keys -> 31.0 / 27.2 / 27.3 / 28.3 us
#define AS_KEY_CHUNK 224
static int as_waves(int tiles) { return tiles < AS_MAX_WAVES ? (tiles < 4 ? 4 : tiles) : AS_MAX_WAVES; }
static int as_chunk(int rows, int cap) {
  const int padded = (rows + 31) & ~31;
  return padded < cap ? padded : cap;
}
static bool as_shape_ok(int D, int H, int Lq, int Lk) { return H > 0 && D == H * AB_HD && Lq > 0 && Lk > 0; }

extern "C" int igcn_attn_core_split_supported(int D, int H, int Lq, int Lk) { return as_shape_ok(D, H, Lq, Lk); }

static int as_check(const char* what, int B, int D, int H, int Lq, int Lk, const void* a, const void* b, const void* c,
                    const void* d) {
  if (!as_shape_ok(D, H, Lq, Lk) || B <= 0) {
    igcn_set_error("%s: head_dim must be 16 (B=%d D=%d H=%d Lq=%d Lk=%d)", what, B, D, H, Lq, Lk);
    return IGCN_ERR_UNSUPPORTED;
  }
  IGCN_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0,
               "attn_core_split: operands must be 16-byte aligned");
  IGCN_REQUIRE((int64_t)B * H <= 0x7fffffff && (int64_t)B * (Lq > Lk ? Lq : Lk) <= 0x7fffffff,
               "attn_core_split: batch too large");
  return IGCN_OK;
}

extern "C" int igcn_attn_core_split_fwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv, float* o,
                                        float* lse, void* stream) {
  int rc = as_check("attn_core_split_fwd", B, D, H, Lq, Lk, q, kv, o, o);
  if (rc) return rc;
  static const int cap = getenv("IGCN_AS_CHUNK") ? atoi(getenv("IGCN_AS_CHUNK")) : AS_KEY_CHUNK;   // (experiments)
  const int ch = as_chunk(Lk, cap), ldt = ab_ldt(ch);
  const size_t lds = ((size_t)2 * ch * AB_HD + (size_t)2 * AB_HD * ldt) * 2;
  const int nqt = (Lq + 15) / 16, nw = as_waves(nqt);
  if (lds > 64 * 1024) IGCN_ALLOW_BIG_LDS(k_attn_split_fwd);
  hipLaunchKernelGGL(k_attn_split_fwd, dim3(B * H, (nqt + nw - 1) / nw), dim3(64 * nw), lds, (hipStream_t)stream, H, Lq,
                     Lk, ch, q, kv, o, lse);
  IGCN_CHECK_LAUNCH("attn_core_split_fwd");
  return IGCN_OK;
}

static size_t as_bwd_lds(int nqt) {
  const size_t lqp = (size_t)nqt * 16, ldt = 2 * lqp + 16;
  const size_t work = (2 * 2 * lqp * AB_HD + 4 * AB_HD * ldt) * 2 + (2 * lqp + 8 * 16 * AS_TLD) * sizeof(float);
  const size_t part = (size_t)4 * lqp * AB_HD * sizeof(float);          // the dQ tree's widest level: four waves' partials
  return work > part ? work : part;
}

// the one-workgroup-per-(sample, head) backward covers the shape (<= 8 query tiles; any number of keys)
extern "C" int igcn_attn_core_split_bwd_supported(int D, int H, int Lq, int Lk) {
  return as_shape_ok(D, H, Lq, Lk) && (Lq + 15) / 16 <= 8 && (Lk + 15) / 16 >= 8;
}

extern "C" int igcn_attn_core_split_bwd(int B, int D, int H, int Lq, int Lk, const float* q, const float* kv,
                                        const float* o, const float* lse, const float* dout, float* dq, float* dkv,
                                        float* scratch, void* stream) {
  (void)scratch;
  int rc = as_check("attn_core_split_bwd", B, D, H, Lq, Lk, q, kv, o, dout);
  if (rc) return rc;
  IGCN_REQUIRE((((uintptr_t)dq | (uintptr_t)dkv) & 15) == 0, "attn_core_split_bwd: aligned gradients needed");
  if (!igcn_attn_core_split_bwd_supported(D, H, Lq, Lk)) {
    igcn_set_error("attn_core_split_bwd: needs <= 128 queries and >= 128 keys (Lq=%d Lk=%d)", Lq, Lk);
    return IGCN_ERR_UNSUPPORTED;
  }
  const int nqt = (Lq + 15) / 16;
  const size_t lds = as_bwd_lds(nqt);
#define AS_BWD(NQTV)                                                                                             \
  case NQTV:                                                                                                     \
    if (lds > 64 * 1024) IGCN_ALLOW_BIG_LDS((k_attn_split_bwd<NQTV>));                                           \
    hipLaunchKernelGGL((k_attn_split_bwd<NQTV>), dim3(B * H), dim3(64 * 8), lds, (hipStream_t)stream, H, Lq, Lk, q, \
                       kv, o, lse, dout, dq, dkv);                                                               \
    break;
  switch (nqt) { AS_BWD(1) AS_BWD(2) AS_BWD(3) AS_BWD(4) AS_BWD(5) AS_BWD(6) AS_BWD(7) AS_BWD(8) }
#undef AS_BWD
  IGCN_CHECK_LAUNCH("attn_core_split_bwd");
  return IGCN_OK;
}
