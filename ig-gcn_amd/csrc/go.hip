// GO-hierarchical attention network kernels (include/igcn.h).  Activations are channel-major
// [B, f, N]: lanes map to consecutive nodes, so every direct access is a coalesced dword stream and the
// neighbour gathers stay inside one sample's f*N*4-byte slab (L2 resident).
// One launch covers all samples: this replaces the reference's per-sample python loop
// (kernel/go_model.py:236-244) and its dense N x N autograd temporaries.
#include <stdlib.h>

#include "common.h"
#include <algorithm>
#include <vector>

#define GO_T 256

// =================================================================================================
// sparse maps with learnable non-zeros (gene encode / decode)
// =================================================================================================
#define SPMM_HEAVY 12
// thread = (sample, row); rows longer than SPMM_HEAVY (the GO root is linked to every SNP) are walked by the whole
// wave so that one thread's serial walk does not set the kernel's duration
__global__ void k_spmm_fwd(int C, int I, int J, int64_t nnz, const int32_t* __restrict__ row_ptr,
                           const int32_t* __restrict__ col, const float* __restrict__ val,
                           const float* __restrict__ x, float* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const bool live = i < I;
  const float* xb = x + (int64_t)b * J;
  // lanes past the end get an EMPTY range: shadowing the last row would make them walk the hub serially
  const int32_t p0 = live ? row_ptr[i] : 0, p1 = live ? row_ptr[i + 1] : 0;
  const bool heavy = p1 - p0 > SPMM_HEAVY;
  for (int c = 0; c < C; ++c) {
    float acc = 0.f;
    if (!heavy)
      for (int32_t p = p0; p < p1; ++p) acc += val[(int64_t)c * nnz + p] * xb[col[p]];
    unsigned long long hm = __ballot(heavy);
    while (hm) {
      const int src = __ffsll((long long)hm) - 1;
      hm &= hm - 1;
      const int32_t h0 = __shfl(p0, src, 64), h1 = __shfl(p1, src, 64);
      float part = 0.f;
      for (int32_t p = h0 + lane; p < h1; p += 64) part += val[(int64_t)c * nnz + p] * xb[col[p]];
      part = wave_sum_all(part);
      if (lane == src) acc = part;
    }
    if (live) y[((int64_t)b * C + c) * I + i] = acc;
  }
}

// long rows (gene decode: ~150 GO nodes per SNP): one wave per (sample, row), lanes stride the row
__global__ void __launch_bounds__(GO_T)
k_spmm_fwd_wave(int B, int C, int I, int J, int64_t nnz, const int32_t* __restrict__ row_ptr,
                const int32_t* __restrict__ col, const float* __restrict__ val, const float* __restrict__ x,
                float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * (GO_T / 64) + (threadIdx.x >> 6);
  if (wid >= (int64_t)B * I) return;
  const int b = (int)(wid / I), i = (int)(wid - (int64_t)b * I);
  const float* xb = x + (int64_t)b * J;
  const int32_t p0 = row_ptr[i], p1 = row_ptr[i + 1];
  for (int c = 0; c < C; ++c) {
    float acc = 0.f;
    for (int32_t p = p0 + lane; p < p1; p += 64) acc += val[(int64_t)c * nnz + p] * xb[col[p]];
    acc = wave_sum(acc);
    if (lane == 0) y[((int64_t)b * C + c) * I + i] = acc;
  }
}

// -------------------------------------------------------------------------------------------------
// The structure of a map is the SAME for every sample, and small (8 k non-zeros): the kernels below read it once per
// workgroup and reuse it across a tile of samples whose operand rows sit in LDS.  The first versions (above / below:
// kept as the fallback for structures that do not fit) re-read pointers, indices and values for every (sample, row)
// — 130 us per step for the two maps of the model, which is why the model used to scatter the values into a dense
// image and run six dense GEMMs (114 us) instead.  Three shapes cover the six products of the two maps:
//   rows_tiled : SHORT lists (a GO node's SNPs; a node's readers), operand rows of <= 1024 floats: thread = output
//                row, SPMM_SB samples per workgroup, their operand rows in LDS
//   long_lds   : LONG lists (a SNP's ~150 GO nodes): workgroup = sample, its operand vector (C x L floats) in LDS,
//                wave per output, lanes stride the list
//   dval_lds   : value gradients: workgroup = SPMM_DB samples, their long vectors and short rows in LDS, thread per
//                non-zero; one partial row per workgroup, summed by the (deferrable) final reduction
// Summation order per output is that of the first versions (list order / lanes stride + fixed tree).
// -------------------------------------------------------------------------------------------------
#define SPMM_SB 8
// sum_c == 0: out[b, c, i] = sum_{p in list i} val[c, vk?] * x[b, idx[p]]                      (x rows [J])
// sum_c == 1: out[b, i]    = sum_{p in list i} sum_c val[c, vk?] * x[b, c, idx[p]]              (x rows [C][J])
// (vk?: val index vk[p] when a transposed list is walked, p otherwise); row length (sum_c ? C : 1) * J <= 1024 floats
__global__ void __launch_bounds__(GO_T)
k_spmm_rows_tiled(int B, int C, int I, int J, int64_t nnz, int sum_c, const int32_t* __restrict__ ptr_,
                  const int32_t* __restrict__ idx, const int32_t* __restrict__ vk, const float* __restrict__ val,
                  const float* __restrict__ x, float* __restrict__ y) {
  extern __shared__ float sp_lds[];                    // [SPMM_SB][RL]
  const int RL = sum_c ? C * J : J;
  const int b0 = blockIdx.y * SPMM_SB, nb = min(SPMM_SB, B - b0);
  for (int t = threadIdx.x; t < nb * RL; t += GO_T) sp_lds[t] = x[(int64_t)b0 * RL + t];
  for (int t = nb * RL + threadIdx.x; t < SPMM_SB * RL; t += GO_T) sp_lds[t] = 0.f;
  __syncthreads();
  const int i = blockIdx.x * GO_T + threadIdx.x, lane = threadIdx.x & 63;
  const bool live = i < I;
  const int32_t p0 = live ? ptr_[i] : 0, p1 = live ? ptr_[i + 1] : 0;      // empty range past the end
  const bool heavy = p1 - p0 > SPMM_HEAVY;
  const int nc_out = sum_c ? 1 : C, nc_in = sum_c ? C : 1;
  for (int co = 0; co < nc_out; ++co) {
    float acc[SPMM_SB];
#pragma unroll
    for (int s = 0; s < SPMM_SB; ++s) acc[s] = 0.f;
    if (!heavy)
      // four list entries per trip, their index / value loads requested together (clamped slots carry a zero value): one
      // entry per trip was a dependent round trip each — a node's three SNPs were three round trips in front of its store
      for (int32_t pb = p0; pb < p1; pb += 4) {
        int32_t rr[4];
        float vv[4];
        if (nc_in == 1) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int32_t p = pb + u < p1 ? pb + u : p1 - 1;
            const int32_t k = vk ? vk[p] : p;
            rr[u] = idx[p];
            vv[u] = val[(int64_t)co * nnz + k];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (pb + u >= p1) break;
            const float* xr = sp_lds + rr[u];
#pragma unroll
            for (int s = 0; s < SPMM_SB; ++s) acc[s] += vv[u] * xr[s * RL];
          }
        } else if (nc_in == 2) {                       // (the model's two-channel maps, summed over the channels)
          float v2[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int32_t p = pb + u < p1 ? pb + u : p1 - 1;
            const int32_t k = vk ? vk[p] : p;
            rr[u] = idx[p];
            vv[u] = val[(int64_t)co * nnz + k];
            v2[u] = val[(int64_t)(co + 1) * nnz + k];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (pb + u >= p1) break;
            const float* xr = sp_lds + rr[u];
#pragma unroll
            for (int s = 0; s < SPMM_SB; ++s) acc[s] += vv[u] * xr[s * RL];
#pragma unroll
            for (int s = 0; s < SPMM_SB; ++s) acc[s] += v2[u] * xr[J + s * RL];
          }
        } else {
          for (int32_t p = pb; p < p1 && p < pb + 4; ++p) {
            const int32_t k = vk ? vk[p] : p, r = idx[p];
            for (int ci = 0; ci < nc_in; ++ci) {
              const float v = val[(int64_t)(co + ci) * nnz + k];
              const float* xr = sp_lds + ci * J + r;
#pragma unroll
              for (int s = 0; s < SPMM_SB; ++s) acc[s] += v * xr[s * RL];
            }
          }
        }
      }
    unsigned long long hm = __ballot(heavy);
    while (hm) {                                       // hub lists (the GO root): the whole wave strides the list
      const int src = __ffsll((long long)hm) - 1;
      hm &= hm - 1;
      const int32_t h0 = __shfl(p0, src, 64), h1 = __shfl(p1, src, 64);
      float part[SPMM_SB];
#pragma unroll
      for (int s = 0; s < SPMM_SB; ++s) part[s] = 0.f;
      for (int32_t p = h0 + lane; p < h1; p += 64) {
        const int32_t k = vk ? vk[p] : p, r = idx[p];
        for (int ci = 0; ci < nc_in; ++ci) {
          const float v = val[(int64_t)(co + ci) * nnz + k];
          const float* xr = sp_lds + ci * J + r;
#pragma unroll
          for (int s = 0; s < SPMM_SB; ++s) part[s] += v * xr[s * RL];
        }
      }
#pragma unroll
      for (int s = 0; s < SPMM_SB; ++s) {
        part[s] = wave_sum_all(part[s]);
        if (lane == src) acc[s] = part[s];
      }
    }
    if (live)
#pragma unroll
      for (int s = 0; s < SPMM_SB; ++s)
        if (s < nb) y[((int64_t)(b0 + s) * nc_out + co) * I + i] = acc[s];
  }
}

// sum_c == 0: out[b, c, j] = sum_{q in list j} val[c, vk?] * vec[b, idx[q]]                  (vec [B][L])
// sum_c == 1: out[b, j]    = sum_{q in list j} sum_c val[c, vk?] * vec[b, c, idx[q]]          (vec [B][C][L])
// A wave owns every 8th list and walks SPMM_RM of them AT ONCE: pointers of all, then the first SPMM_LU stride-64
// entries of all (indices, then values, then the LDS gathers), then the wave sums.  [A list costs ~2-3.5 us of
// dependent latency — pointer, index, value, gather, six-step tree — whatever is in flight inside it (phase probe:
// tools/spmm_probe.py); one after the other, a wave's seven lists were the whole 12 / 24 us of the kernel.
// Entry-parallel products parked in LDS + a second phase of list sums: 88 KB of LDS, one workgroup per CU, 31 us.]
#ifdef SPMM_PROBE_ON
__device__ long long spmm_probe_buf[2 * 8 * 8];
#define SPMM_PROBE(i) do { if (threadIdx.x == 0 && blockIdx.x < 8) spmm_probe_buf[(sum_c * 8 + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
extern "C" int igcn_debug_spmm_probe(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(spmm_probe_buf), sizeof(long long) * 128);
}
#else
#define SPMM_PROBE(i)
#endif
__device__ __forceinline__ float spmm_wave_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
  const int ri = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(ri, 0)) + __int_as_float(__builtin_amdgcn_readlane(ri, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(ri, 32)) + __int_as_float(__builtin_amdgcn_readlane(ri, 48)));
}
#define SPMM_LT 512
#define SPMM_LU 3
#define SPMM_RM 7
__global__ void __launch_bounds__(SPMM_LT, 4)
k_spmm_long_lds(int C, int L, int n_out, int64_t nnz, int sum_c, const int32_t* __restrict__ ptr_,
                const int32_t* __restrict__ idx, const int32_t* __restrict__ vk, const float* __restrict__ val,
                const float* __restrict__ vec, float* __restrict__ out) {
  extern __shared__ float sp_lds[];                    // [sum_c ? C : 1][L]
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nv = sum_c ? C * L : L;
  const float* vb = vec + (int64_t)b * nv;
  SPMM_PROBE(0);
  if (nv % 4 == 0 && ((uintptr_t)vb & 15) == 0) {
    for (int t = threadIdx.x * 4; t < nv; t += SPMM_LT * 4)
      *reinterpret_cast<float4*>(sp_lds + t) = *reinterpret_cast<const float4*>(vb + t);
  } else {
    for (int t = threadIdx.x; t < nv; t += SPMM_LT) sp_lds[t] = vb[t];
  }
  __syncthreads();
  SPMM_PROBE(1);
  const int nc_out = sum_c ? 1 : C, nc_in = sum_c ? C : 1;
  constexpr int NWV = SPMM_LT / 64;
  for (int jb = w; jb < n_out; jb += NWV * SPMM_RM) {
    // SPMM_RM lists per wave AT ONCE: their pointers, then the first SPMM_LU stride-64 entries of every list, then the
    // values, then the LDS gathers — the dependent round trips of a list overlap with those of its siblings
    int32_t q0[SPMM_RM], q1[SPMM_RM];
#pragma unroll
    for (int m = 0; m < SPMM_RM; ++m) {
      const int j = jb + m * NWV;
      q0[m] = j < n_out ? ptr_[j] : 0;
      q1[m] = j < n_out ? ptr_[j + 1] : 0;
    }
    for (int co = 0; co < nc_out; ++co) {
      float acc[SPMM_RM];
      int32_t r[SPMM_RM][SPMM_LU], k[SPMM_RM][SPMM_LU];
#pragma unroll
      for (int m = 0; m < SPMM_RM; ++m) {
        acc[m] = 0.f;
#pragma unroll
        for (int u = 0; u < SPMM_LU; ++u) {
          const int32_t qq = q0[m] + lane + 64 * u;
          const bool ok = qq < q1[m];
          r[m][u] = ok ? idx[qq] : -1;
          k[m][u] = ok ? (vk ? vk[qq] : qq) : 0;
        }
      }
#pragma unroll
      for (int m = 0; m < SPMM_RM; ++m) {
#pragma unroll
        for (int u = 0; u < SPMM_LU; ++u)
          for (int ci = 0; ci < nc_in; ++ci) {
            const float v = r[m][u] >= 0 ? val[(int64_t)(co + ci) * nnz + k[m][u]] : 0.f;
            acc[m] += v * sp_lds[ci * L + (r[m][u] >= 0 ? r[m][u] : 0)];
          }
        for (int32_t q = q0[m] + lane + 64 * SPMM_LU; q < q1[m]; q += 64)       // lists beyond 64 SPMM_LU entries
          for (int ci = 0; ci < nc_in; ++ci)
            acc[m] += val[(int64_t)(co + ci) * nnz + (vk ? vk[q] : q)] * sp_lds[ci * L + idx[q]];
      }
#pragma unroll
      for (int m = 0; m < SPMM_RM; ++m) {                // seven wave sums on the VALU (DPP rows + v_readlane), not as seven
        const int j = jb + m * NWV;                      // six-step chains of LDS-pipe permutes
        const float t = spmm_wave_sum(acc[m]);
        if (lane == 0 && j < n_out) out[((int64_t)b * nc_out + co) * n_out + j] = t;
      }
    }
    if (jb == w) SPMM_PROBE(2);
  }
  SPMM_PROBE(3);
}
static size_t spmm_long_lds_bytes(int C, int L, int64_t nnz, int sum_c) {
  (void)nnz;
  return (size_t)(sum_c ? C * L : L) * sizeof(float);
}

// partial[wg][c][k] = sum_{b in tile} big[b, c, bidx[k]] * small[b, sidx[k]]     (big [B][C][L], small [B][S])
#define SPMM_DT 1024
__device__ __forceinline__ void
spmm_dval_lds_body(const int bx, int B, int C, int L, int S, int64_t nnz, int tile, const int32_t* __restrict__ bidx,
                   const int32_t* __restrict__ sidx, const float* __restrict__ big, const float* __restrict__ small_,
                   float* __restrict__ partial) {
  extern __shared__ float sp_lds[];                    // [tile][C*L] then [tile][S]
  const int b0 = bx * tile, nb = min(tile, B - b0), CL = C * L;
  float* bs = sp_lds;
  float* ss = sp_lds + (size_t)tile * CL;
  const float* bsrc = big + (int64_t)b0 * CL;
  if (CL % 4 == 0 && ((uintptr_t)bsrc & 15) == 0) {
    for (int t = threadIdx.x * 4; t < nb * CL; t += SPMM_DT * 4)
      *reinterpret_cast<float4*>(bs + t) = *reinterpret_cast<const float4*>(bsrc + t);
  } else {
    for (int t = threadIdx.x; t < nb * CL; t += SPMM_DT) bs[t] = bsrc[t];
  }
  for (int t = threadIdx.x; t < nb * S; t += SPMM_DT) ss[t] = small_[(int64_t)b0 * S + t];
  __syncthreads();
  float* prow = partial + (int64_t)bx * C * nnz;
  for (int64_t k = threadIdx.x; k < nnz; k += SPMM_DT) {
    const int32_t r = bidx[k], j = sidx[k];
    for (int c = 0; c < C; ++c) {
      float acc = 0.f;
      for (int s = 0; s < nb; ++s) acc += bs[s * CL + c * L + r] * ss[s * S + j];
      prow[(int64_t)c * nnz + k] = acc;
    }
  }
}

__global__ void __launch_bounds__(SPMM_DT)
k_spmm_dval_lds(int B, int C, int L, int S, int64_t nnz, int tile, const int32_t* __restrict__ bidx,
                const int32_t* __restrict__ sidx, const float* __restrict__ big, const float* __restrict__ small_,
                float* __restrict__ partial) {
  spmm_dval_lds_body((int)blockIdx.x, B, C, L, S, nnz, tile, bidx, sidx, big, small_, partial);
}

// The value-gradient passes of TWO maps (the SNP -> GO encoding and the GO -> SNP decoding of one backward) as one flat
// grid: each alone is B / 4 = 128 workgroups of 1024 threads with ~100 KB of LDS — half the CUs — and both are
// parameter gradients that nothing reads before the optimiser, so the first can wait for the second.
struct SpmmDvalProb {
  int B, C, L, S, tile, wg0;
  int64_t nnz;
  const int32_t *bidx, *sidx;
  const float *big, *small_;
  float* partial;
};
struct SpmmDvalGroup { int n; SpmmDvalProb p[2]; };
__global__ void __launch_bounds__(SPMM_DT) k_spmm_dval_lds_multi(const SpmmDvalGroup G) {
  const int pi = (G.n > 1 && (int)blockIdx.x >= G.p[1].wg0) ? 1 : 0;
  const SpmmDvalProb& p = G.p[pi];
  spmm_dval_lds_body((int)blockIdx.x - p.wg0, p.B, p.C, p.L, p.S, p.nnz, p.tile, p.bidx, p.sidx, p.big, p.small_, p.partial);
}

// samples per workgroup of k_spmm_dval_lds: what ~120 KB of LDS hold, at most 4 (0: the vectors do not fit)
static int spmm_dval_tile(int C, int L, int S) {
  const size_t per = ((size_t)C * L + S) * sizeof(float);
  const int t = (int)((size_t)120 * 1024 / per);
  return t > 4 ? 4 : t;
}
static bool spmm_no_lds(void) { return igcn_opt(IGCN_OPT_SPMM_NO_LDS); }     // the first versions (A/B runs)

// vs = floats between the value rows of consecutive channels (nnz when val is one [C, nnz] tensor; the kernels below
// take it in the place of `nnz`, which they use for nothing else)
static int spmm_fwd_impl(int B, int C, int I, int J, int64_t nnz, int64_t vs, const int32_t* row_ptr, const int32_t* col,
                         const float* val, const float* x, float* y, void* stream) {
  IGCN_REQUIRE(B > 0 && C > 0 && I > 0 && J > 0 && vs >= nnz, "spmm_fwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const bool longrows = nnz > (int64_t)8 * I;
  if (!spmm_no_lds() && longrows && spmm_long_lds_bytes(C, J, nnz, 0) <= 150 * 1024) {
    const size_t lds = spmm_long_lds_bytes(C, J, nnz, 0);
    if (lds > 64 * 1024) IGCN_ALLOW_BIG_LDS(k_spmm_long_lds);
    hipLaunchKernelGGL(k_spmm_long_lds, dim3(B), dim3(SPMM_LT), lds, st, C, J, I, vs, 0, row_ptr, col,
                       (const int32_t*)nullptr, val, x, y);
  } else if (!spmm_no_lds() && !longrows && J <= 1024) {
    hipLaunchKernelGGL(k_spmm_rows_tiled, dim3((unsigned)igcn_cdiv(I, GO_T), (unsigned)igcn_cdiv(B, SPMM_SB)),
                       dim3(GO_T), (size_t)SPMM_SB * J * sizeof(float), st, B, C, I, J, vs, 0, row_ptr, col,
                       (const int32_t*)nullptr, val, x, y);
  } else if (longrows) {
    hipLaunchKernelGGL(k_spmm_fwd_wave, dim3((unsigned)igcn_cdiv((int64_t)B * I, GO_T / 64)), dim3(GO_T), 0, st, B, C,
                       I, J, vs, row_ptr, col, val, x, y);
  } else {
    hipLaunchKernelGGL(k_spmm_fwd, dim3((unsigned)igcn_cdiv(I, GO_T), B), dim3(GO_T), 0, st, C, I, J, vs, row_ptr, col,
                       val, x, y);
  }
  IGCN_CHECK_LAUNCH("spmm_fwd");
  return IGCN_OK;
}

extern "C" int igcn_spmm_fwd(int B, int C, int I, int J, int64_t nnz, const int32_t* row_ptr, const int32_t* col,
                             const float* val, const float* x, float* y, void* stream) {
  return spmm_fwd_impl(B, C, I, J, nnz, nnz, row_ptr, col, val, x, y, stream);
}
// the C value rows val + c * val_stride need not be one contiguous tensor (per-channel parameters laid out at a
// constant stride in the optimiser's flat buffer: no torch.stack in front of every step)
extern "C" int igcn_spmm_fwd_strided(int B, int C, int I, int J, int64_t nnz, const int32_t* row_ptr,
                                     const int32_t* col, const float* val, int64_t val_stride, const float* x, float* y,
                                     void* stream) {
  return spmm_fwd_impl(B, C, I, J, nnz, val_stride, row_ptr, col, val, x, y, stream);
}

__global__ void k_spmm_bwd_dx(int C, int I, int J, int64_t nnz, const int32_t* __restrict__ t_ptr,
                              const int32_t* __restrict__ t_row, const int32_t* __restrict__ t_k,
                              const float* __restrict__ val, const float* __restrict__ dy, float* __restrict__ dx) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const bool live = j < J;
  const int32_t q0 = live ? t_ptr[j] : 0, q1 = live ? t_ptr[j + 1] : 0;     // empty range past the end
  const bool heavy = q1 - q0 > SPMM_HEAVY;
  float acc = 0.f;
  if (!heavy) {
    for (int32_t q = q0; q < q1; ++q) {
      const int32_t r = t_row[q], k = t_k[q];
      for (int c = 0; c < C; ++c) acc += val[(int64_t)c * nnz + k] * dy[((int64_t)b * C + c) * I + r];
    }
  }
  unsigned long long hm = __ballot(heavy);
  while (hm) {
    const int src = __ffsll((long long)hm) - 1;
    hm &= hm - 1;
    const int32_t h0 = __shfl(q0, src, 64), h1 = __shfl(q1, src, 64);
    float part = 0.f;
    for (int32_t q = h0 + lane; q < h1; q += 64) {
      const int32_t r = t_row[q], k = t_k[q];
      for (int c = 0; c < C; ++c) part += val[(int64_t)c * nnz + k] * dy[((int64_t)b * C + c) * I + r];
    }
    part = wave_sum_all(part);
    if (lane == src) acc = part;
  }
  if (live) dx[(int64_t)b * J + j] = acc;
}

__global__ void __launch_bounds__(GO_T)
k_spmm_bwd_dx_wave(int B, int C, int I, int J, int64_t nnz, const int32_t* __restrict__ t_ptr,
                   const int32_t* __restrict__ t_row, const int32_t* __restrict__ t_k,
                   const float* __restrict__ val, const float* __restrict__ dy, float* __restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * (GO_T / 64) + (threadIdx.x >> 6);
  if (wid >= (int64_t)B * J) return;
  const int b = (int)(wid / J), j = (int)(wid - (int64_t)b * J);
  float acc = 0.f;
  for (int32_t q = t_ptr[j] + lane; q < t_ptr[j + 1]; q += 64) {
    const int32_t r = t_row[q], k = t_k[q];
    for (int c = 0; c < C; ++c) acc += val[(int64_t)c * nnz + k] * dy[((int64_t)b * C + c) * I + r];
  }
  acc = wave_sum(acc);
  if (lane == 0) dx[(int64_t)b * J + j] = acc;
}

// dval[c,k] = sum_b dy[b,c,row_k] x[b,col_k]: samples split into gridDim.z chunks -> partial[z][c][k]
#define SPMM_BCH 16
__global__ void k_spmm_bwd_dval(int B, int C, int I, int J, int64_t nnz, const int32_t* __restrict__ col,
                                const int32_t* __restrict__ row_of, const float* __restrict__ x,
                                const float* __restrict__ dy, float* __restrict__ partial) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (k >= nnz) return;
  const int per = (B + gridDim.z - 1) / gridDim.z;
  const int b0 = blockIdx.z * per, b1 = min(B, b0 + per);
  const int32_t r = row_of[k], j = col[k];
  float acc = 0.f;
  for (int b = b0; b < b1; ++b) acc += dy[((int64_t)b * C + c) * I + r] * x[(int64_t)b * J + j];
  partial[((int64_t)blockIdx.z * C + c) * nnz + k] = acc;
}

// scratch of igcn_spmm_bwd (value gradients): one partial row [C][nnz] per workgroup of the tiled kernel, or the
// SPMM_BCH sample chunks of the fallback
extern "C" size_t igcn_spmm_bwd_scratch_floats(int B, int C, int I, int J, int64_t nnz) {
  const int L = I > J ? I : J, S = I > J ? J : I;
  const int tile = spmm_dval_tile(C, L, S);
  const int64_t rows = (tile >= 1 && !spmm_no_lds()) ? igcn_cdiv(B, tile) : SPMM_BCH;
  return (size_t)((rows > SPMM_BCH ? rows : SPMM_BCH) * C * (nnz > 0 ? nnz : 1));
}

static int spmm_bwd_impl(int B, int C, int I, int J, int64_t nnz, int64_t vs, const int32_t* row_ptr, const int32_t* col,
                             const int32_t* row_of, const int32_t* t_ptr, const int32_t* t_row, const int32_t* t_k,
                             const float* val, const float* x, const float* dy, float* dx, float* dval,
                             float* scratch, void* stream) {
  (void)row_ptr;
  IGCN_REQUIRE(B > 0 && C > 0 && I > 0 && J > 0, "spmm_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (dx) {
    // dx[b, j] = sum over column j's entries q, over channels c: val[c, t_k[q]] * dy[b, c, t_row[q]]
    const bool longcols = nnz > (int64_t)8 * J;
    const size_t vec_bytes = spmm_long_lds_bytes(C, I, nnz, 1);
    if (!spmm_no_lds() && longcols && vec_bytes <= 150 * 1024) {
      if (vec_bytes > 64 * 1024) IGCN_ALLOW_BIG_LDS(k_spmm_long_lds);
      hipLaunchKernelGGL(k_spmm_long_lds, dim3(B), dim3(SPMM_LT), vec_bytes, st, C, I, J, vs, 1, t_ptr, t_row, t_k,
                         val, dy, dx);
    } else if (!spmm_no_lds() && !longcols && (int64_t)C * I <= 1024) {
      hipLaunchKernelGGL(k_spmm_rows_tiled, dim3((unsigned)igcn_cdiv(J, GO_T), (unsigned)igcn_cdiv(B, SPMM_SB)),
                         dim3(GO_T), (size_t)SPMM_SB * C * I * sizeof(float), st, B, C, J, I, vs, 1, t_ptr, t_row, t_k,
                         val, dy, dx);
    } else if (longcols) {
      hipLaunchKernelGGL(k_spmm_bwd_dx_wave, dim3((unsigned)igcn_cdiv((int64_t)B * J, GO_T / 64)), dim3(GO_T), 0, st,
                         B, C, I, J, vs, t_ptr, t_row, t_k, val, dy, dx);
    } else {
      hipLaunchKernelGGL(k_spmm_bwd_dx, dim3((unsigned)igcn_cdiv(J, GO_T), B), dim3(GO_T), 0, st, C, I, J, vs, t_ptr,
                         t_row, t_k, val, dy, dx);
    }
  }
  IGCN_CHECK_LAUNCH("spmm_bwd_dx");
  if (dval && nnz > 0) {
    IGCN_REQUIRE(scratch != nullptr, "spmm_bwd: dval needs scratch (igcn_spmm_bwd_scratch_floats)");
    // dval[c, k] = sum_b dy[b, c, row_of[k]] * x[b, col[k]]: the LONG side of the map (per-sample vector with a channel
    // axis only if it is dy) goes to LDS per sample, the short side rides along
    const bool rows_long = I >= J;
    const int L = rows_long ? I : J, S = rows_long ? J : I;
    const int tile = (rows_long || C == 1) ? spmm_dval_tile(C, L, S) : 0;     // x has no channel axis
    if (!spmm_no_lds() && tile >= 1) {
      const int64_t wgs = igcn_cdiv(B, tile);
      const size_t lds = (size_t)tile * ((size_t)C * L + S) * sizeof(float);
      if (lds > 64 * 1024) IGCN_ALLOW_BIG_LDS(k_spmm_dval_lds);
      hipLaunchKernelGGL(k_spmm_dval_lds, dim3((unsigned)wgs), dim3(SPMM_DT), lds, st, B, C, L, S, nnz, tile,
                         rows_long ? row_of : col, rows_long ? col : row_of, rows_long ? dy : x, rows_long ? x : dy,
                         scratch);
      IGCN_CHECK_LAUNCH("spmm_bwd_dval(lds)");
      return igcn_launch_reduce_rows_final(scratch, wgs, (int64_t)C * nnz, (int)((int64_t)C * nnz), dval, st);
    }
    const int bch = B < SPMM_BCH ? B : SPMM_BCH;
    hipLaunchKernelGGL(k_spmm_bwd_dval, dim3((unsigned)igcn_cdiv(nnz, GO_T), C, bch), dim3(GO_T), 0, st, B, C, I, J,
                       nnz, col, row_of, x, dy, scratch);
    IGCN_CHECK_LAUNCH("spmm_bwd_dval");
    return igcn_launch_reduce_rows_final(scratch, bch, (int64_t)C * nnz, (int)((int64_t)C * nnz), dval, st);
  }
  return IGCN_OK;
}

extern "C" int igcn_spmm_bwd(int B, int C, int I, int J, int64_t nnz, const int32_t* row_ptr, const int32_t* col,
                             const int32_t* row_of, const int32_t* t_ptr, const int32_t* t_row, const int32_t* t_k,
                             const float* val, const float* x, const float* dy, float* dx, float* dval,
                             float* scratch, void* stream) {
  return spmm_bwd_impl(B, C, I, J, nnz, nnz, row_ptr, col, row_of, t_ptr, t_row, t_k, val, x, dy, dx, dval, scratch, stream);
}
// value rows at a constant stride (see igcn_spmm_fwd_strided); dval stays one contiguous [C, nnz] output
extern "C" int igcn_spmm_bwd_strided(int B, int C, int I, int J, int64_t nnz, const int32_t* row_ptr,
                                     const int32_t* col, const int32_t* row_of, const int32_t* t_ptr, const int32_t* t_row,
                                     const int32_t* t_k, const float* val, int64_t val_stride, const float* x,
                                     const float* dy, float* dx, float* dval, float* scratch, void* stream) {
  IGCN_REQUIRE(val_stride >= nnz, "spmm_bwd_strided: val_stride < nnz");
  return spmm_bwd_impl(B, C, I, J, nnz, val_stride, row_ptr, col, row_of, t_ptr, t_row, t_k, val, x, dy, dx, dval, scratch,
                       stream);
}

// table [n <= 2][12] int64 = {B, C, I, J, nnz, col, row_of, x, dy, dval, scratch, 0} per map: the dval half of
// igcn_spmm_bwd (same arguments, same scratch) for up to two maps in one launch.  A map whose vectors do not fit LDS
// gets its own launch of the global-memory kernel.
extern "C" int igcn_spmm_bwd_dval_multi(int n, const int64_t* table, void* stream) {
  IGCN_REQUIRE(n >= 1 && n <= 2 && table != nullptr, "spmm_bwd_dval_multi: one or two maps");
  hipStream_t st = (hipStream_t)stream;
  SpmmDvalGroup G = {};
  int wgs = 0;
  size_t lds = 0;
  int64_t rows_of[2] = {0, 0};
  for (int i = 0; i < n; ++i) {
    const int64_t* t = table + 12 * i;
    const int B = (int)t[0], C = (int)t[1], I = (int)t[2], J = (int)t[3];
    const int64_t nnz = t[4];
    const int32_t *col = (const int32_t*)t[5], *row_of = (const int32_t*)t[6];
    const float *x = (const float*)t[7], *dy = (const float*)t[8];
    float* scratch = (float*)t[10];
    IGCN_REQUIRE(B > 0 && C > 0 && I > 0 && J > 0 && scratch != nullptr, "spmm_bwd_dval_multi: bad map %d", i);
    if (nnz <= 0) continue;
    const bool rows_long = I >= J;
    const int L = rows_long ? I : J, S = rows_long ? J : I;
    const int tile = (rows_long || C == 1) ? spmm_dval_tile(C, L, S) : 0;
    if (spmm_no_lds() || tile < 1) {
      const int bch = B < SPMM_BCH ? B : SPMM_BCH;
      hipLaunchKernelGGL(k_spmm_bwd_dval, dim3((unsigned)igcn_cdiv(nnz, GO_T), C, bch), dim3(GO_T), 0, st, B, C, I, J,
                         nnz, col, row_of, x, dy, scratch);
      rows_of[i] = bch;
      continue;
    }
    SpmmDvalProb& p = G.p[G.n++];
    p.B = B; p.C = C; p.L = L; p.S = S; p.tile = tile; p.nnz = nnz;
    p.bidx = rows_long ? row_of : col;
    p.sidx = rows_long ? col : row_of;
    p.big = rows_long ? dy : x;
    p.small_ = rows_long ? x : dy;
    p.partial = scratch;
    p.wg0 = wgs;
    const int w = (int)igcn_cdiv(B, tile);
    wgs += w;
    rows_of[i] = w;
    const size_t l = (size_t)tile * ((size_t)C * L + S) * sizeof(float);
    lds = l > lds ? l : lds;
  }
  if (G.n > 0) {
    if (G.n == 1) G.p[1] = G.p[0];
    if (lds > 64 * 1024) IGCN_ALLOW_BIG_LDS(k_spmm_dval_lds_multi);
    hipLaunchKernelGGL(k_spmm_dval_lds_multi, dim3((unsigned)wgs), dim3(SPMM_DT), lds, st, G);
  }
  IGCN_CHECK_LAUNCH("spmm_bwd_dval_multi");
  for (int i = 0; i < n; ++i) {
    const int64_t* t = table + 12 * i;
    if (t[4] <= 0) continue;
    const int64_t cn = t[1] * t[4];
    const int rc = igcn_launch_reduce_rows_final((float*)t[10], rows_of[i], cn, (int)cn, (float*)t[9], st);
    if (rc) return rc;
  }
  return IGCN_OK;
}

// =================================================================================================
// attention-GCN encoder layer
// =================================================================================================
// hardware exp2-based exp / tanh (v_exp_f32): ~1e-6 relative / ~1e-7 absolute error, a fraction of the instructions
// of the libm versions; the attention kernels evaluate one tanh + one exp per edge per sample
// (reciprocals on v_rcp_f32, 1 ulp: a plain `/` is the IEEE sequence — ten VALU instructions per edge per sample)
__device__ __forceinline__ float go_exp(float z) { return __expf(z); }
__device__ __forceinline__ float go_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float go_tanh(float z) { return 1.f - 2.f * go_rcp(1.f + __expf(2.f * z)); }

template <int FIN, int FOUT>
struct AttnW {
  float wi[FOUT][FIN], ws[FOUT][FIN], a1[FOUT], a2[FOUT], as[FOUT];
  __device__ __forceinline__ void load(const float* w_inc, const float* w_s, const float* a_in, const float* a_s) {
#pragma unroll
    for (int c = 0; c < FOUT; ++c) {
#pragma unroll
      for (int d = 0; d < FIN; ++d) {
        wi[c][d] = w_inc[c * FIN + d];
        ws[c][d] = w_s[c * FIN + d];
      }
      a1[c] = a_in[c];
      a2[c] = a_in[FOUT + c];
      as[c] = a_s[c];
    }
  }
};

template <int FIN, int FOUT>
__device__ __forceinline__ void transform(const float (&w)[FOUT][FIN], const float (&x)[FIN], float (&o)[FOUT]) {
#pragma unroll
  for (int c = 0; c < FOUT; ++c) {
    float t = 0.f;
#pragma unroll
    for (int d = 0; d < FIN; ++d) t += w[c][d] * x[d];
    o[c] = t;
  }
}

// y[d] = sum_c w[c][d] * v[c]: a FOUT-vector pulled back to the input features.  Scores and row sums only ever
// need x_in = W_inc x of a NEIGHBOUR inside a dot product, a . (W x) = (W^T a) . x: FIN multiply-adds per edge instead
// of a FIN x FOUT transform plus a FOUT dot
template <int FIN, int FOUT>
__device__ __forceinline__ void pull_back(const float (&w)[FOUT][FIN], const float (&v)[FOUT], float (&y)[FIN]) {
#pragma unroll
  for (int d = 0; d < FIN; ++d) {
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < FOUT; ++c) t += w[c][d] * v[c];
    y[d] = t;
  }
}

template <int FIN>
__device__ __forceinline__ void load_node(const float* __restrict__ xb, int N, int n, float (&v)[FIN]) {
#pragma unroll
  for (int d = 0; d < FIN; ++d) v[d] = xb[d * N + n];       // 32-bit offset from the sample's base pointer
}

template <int F>
__device__ __forceinline__ float dot(const float (&a)[F], const float (&b)[F]) {
  float t = 0.f;
#pragma unroll
  for (int c = 0; c < F; ++c) t += a[c] * b[c];
  return t;
}

// workgroup -> (node block, sample) of a (node blocks, B) grid.  The hardware deals workgroup i (x fastest) to XCD i mod 8,
// so with (blockIdx.x, blockIdx.y) taken as they come the node blocks of ONE sample sit on all eight XCDs and every
// neighbour gather pulls the sample's rows into eight L2s: k_go_attn_bwd_main<2,5> at 64 samples x 10 000 nodes read
// 294 MB for 30 MB of operands (PMC, profiles/r04_step_traffic_stress.csv) and ran at the HBM rate.  Here XCD c takes
// the items [c T/8, (c+1) T/8) in order: a sample's blocks run back to back on one XCD, its rows cross the fabric once.
__device__ __forceinline__ void go_block(int& bx, int& b) {
  const int nx = gridDim.x, total = nx * gridDim.y, id = blockIdx.y * nx + blockIdx.x;
  int j = id;
  if ((total & 7) == 0) j = (id & 7) * (total >> 3) + (id >> 3);
  b = j / nx;
  bx = j - b * nx;
}

template <int FIN, int FOUT>
__global__ void __launch_bounds__(GO_T)
k_go_attn_fwd(int N, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
              const float* __restrict__ x, const float* __restrict__ w_inc, const float* __restrict__ w_s,
              const float* __restrict__ a_in, const float* __restrict__ a_s, float* __restrict__ y) {
  AttnW<FIN, FOUT> W;
  W.load(w_inc, w_s, a_in, a_s);
  int bx, b;
  go_block(bx, b);
  const int n = bx * GO_T + threadIdx.x;
  if (n >= N) return;
  const float* xb = x + (int64_t)b * FIN * N;
  float xr[FIN], xin[FOUT], xs[FOUT];
  load_node<FIN>(xb, N, n, xr);
  transform<FIN, FOUT>(W.wi, xr, xin);
  transform<FIN, FOUT>(W.ws, xr, xs);
  const float p = dot<FOUT>(W.a1, xin);
  float Z = 0.f, agg[FOUT];
#pragma unroll
  for (int c = 0; c < FOUT; ++c) agg[c] = 0.f;
  const int32_t p0 = row_ptr[n], p1 = row_ptr[n + 1];
  // two edges per step: both neighbour indices, then both neighbour rows, are in flight together (a GO term has
  // 1-2 parents: most walks are one dependent index->row chain instead of two); summation order is unchanged
  for (int32_t e = p0; e < p1; e += 2) {
    const bool two = e + 1 < p1;
    const int m0 = col[e], m1 = col[two ? e + 1 : e];
    float xm0[FIN], xm1[FIN], xi0[FOUT], xi1[FOUT];
    load_node<FIN>(xb, N, m0, xm0);
    load_node<FIN>(xb, N, m1, xm1);
    transform<FIN, FOUT>(W.wi, xm0, xi0);
    transform<FIN, FOUT>(W.wi, xm1, xi1);
    const float s0 = go_exp(go_tanh(p + dot<FOUT>(W.a2, xi0)));
    const float s1 = two ? go_exp(go_tanh(p + dot<FOUT>(W.a2, xi1))) : 0.f;
    Z += s0;
#pragma unroll
    for (int c = 0; c < FOUT; ++c) agg[c] += s0 * xi0[c];
    if (two) {
      Z += s1;
#pragma unroll
      for (int c = 0; c < FOUT; ++c) agg[c] += s1 * xi1[c];
    }
  }
  const float zinv = p1 > p0 ? go_rcp(Z) : 0.f;
  const float g = go_rcp(1.f + go_exp(-dot<FOUT>(W.as, xs)));
  float* yb = y + (int64_t)b * FOUT * N;
#pragma unroll
  for (int c = 0; c < FOUT; ++c) yb[c * N + n] = agg[c] * zinv + xs[c] * g;
}

#define GO_DISPATCH(fin, fout, CALL)                         \
  if (fin == 2 && fout == 5) { CALL(2, 5); }                 \
  else if (fin == 5 && fout == 5) { CALL(5, 5); }            \
  else if (fin == 5 && fout == 2) { CALL(5, 2); }            \
  else if (fin == 2 && fout == 2) { CALL(2, 2); }            \
  else {                                                     \
    igcn_set_error("unsupported GO feature dims fin=%d fout=%d (built: 2->5, 5->5, 5->2, 2->2)", fin, fout); \
    return IGCN_ERR_UNSUPPORTED;                             \
  }

extern "C" int igcn_go_attn_fwd(int B, int N, int fin, int fout, const int32_t* row_ptr, const int32_t* col,
                                const float* x, const float* w_inc, const float* w_s, const float* a_in,
                                const float* a_s, float* y, void* stream) {
  IGCN_REQUIRE(B > 0 && N > 0, "go_attn_fwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)igcn_cdiv(N, GO_T), B);
#define CALL(FI, FO) \
  hipLaunchKernelGGL((k_go_attn_fwd<FI, FO>), grid, dim3(GO_T), 0, st, N, row_ptr, col, x, w_inc, w_s, a_in, a_s, y)
  GO_DISPATCH(fin, fout, CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("go_attn_fwd");
  return IGCN_OK;
}

// ---- backward, kernel A: per-row softmax statistics + the row-side score gradient --------------
// stats = (p (a1.x_in), q (a2.x_in), 1/Z, tr = dy . (agg/Z)) per node as one float4, then dp [B,N].
// With s_e = exp(tanh(p + q_m)), w_e = s_e (1 - tanh^2), D_e = dy_n . x_in,m = (W_inc^T dy_n) . x_m, one walk over a
// node's own edges yields everything the row side needs:
//   Z = sum s_e,  tr = (sum s_e D_e) / Z,  dp = sum_e (D_e - tr) (s_e / Z) (1 - tanh^2) = (sum w_e D_e - tr sum w_e) / Z
// — the second walk of the same edges the first version made (recomputing the transforms, tanh and exp) is gone, and
// a neighbour costs 2 FIN multiply-adds instead of a FIN x FOUT transform and two FOUT dots.
template <int FIN, int FOUT>
struct RowPass {
  float v2[FIN], g[FIN], p, Z, A, Bw, Cw;
  __device__ __forceinline__ void begin(const float (&wi)[FOUT][FIN], const float (&a2)[FOUT], const float (&dyn)[FOUT],
                                        float p_) {
    pull_back<FIN, FOUT>(wi, a2, v2);
    pull_back<FIN, FOUT>(wi, dyn, g);
    p = p_; Z = 0.f; A = 0.f; Bw = 0.f; Cw = 0.f;
  }
  __device__ __forceinline__ void edge(const float (&xm)[FIN]) {
    const float th = go_tanh(p + dot<FIN>(v2, xm));
    const float s = go_exp(th), w = s * (1.f - th * th), D = dot<FIN>(g, xm);
    Z += s; A += s * D; Bw += w * D; Cw += w;
  }
  // (1/Z, tr, dp)
  __device__ __forceinline__ void end(bool any, float& zinv, float& tr, float& dp) const {
    zinv = any ? go_rcp(Z) : 0.f;
    tr = A * zinv;
    dp = (Bw - tr * Cw) * zinv;
  }
};

template <int FIN, int FOUT>
__global__ void __launch_bounds__(GO_T)
k_go_attn_bwd_stats(int B, int N, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                    const float* __restrict__ x, const float* __restrict__ w_inc, const float* __restrict__ a_in,
                    const float* __restrict__ dy, float* __restrict__ stats) {
  float wi[FOUT][FIN], a1[FOUT], a2[FOUT];
#pragma unroll
  for (int c = 0; c < FOUT; ++c) {
#pragma unroll
    for (int d = 0; d < FIN; ++d) wi[c][d] = w_inc[c * FIN + d];
    a1[c] = a_in[c];
    a2[c] = a_in[FOUT + c];
  }
  int bx, b;
  go_block(bx, b);
  const int n = bx * GO_T + threadIdx.x;
  if (n >= N) return;
  const float* xb = x + (int64_t)b * FIN * N;
  float xr[FIN], xin[FOUT], dyn[FOUT];
  load_node<FIN>(xb, N, n, xr);
  load_node<FOUT>(dy + (int64_t)b * FOUT * N, N, n, dyn);
  transform<FIN, FOUT>(wi, xr, xin);
  RowPass<FIN, FOUT> rp;
  rp.begin(wi, a2, dyn, dot<FOUT>(a1, xin));
  const int32_t p0 = row_ptr[n], p1 = row_ptr[n + 1];
  for (int32_t e = p0; e < p1; e += 2) {            // two edges per step, as in the forward kernel
    const bool two = e + 1 < p1;
    const int m0 = col[e], m1 = col[two ? e + 1 : e];
    float xm0[FIN], xm1[FIN];
    load_node<FIN>(xb, N, m0, xm0);
    load_node<FIN>(xb, N, m1, xm1);
    rp.edge(xm0);
    if (two) rp.edge(xm1);
  }
  float zinv, tr, dp;
  rp.end(p1 > p0, zinv, tr, dp);
  // one 16-byte record per node: the walks of kernel B fetch a neighbour's statistics with one load
  reinterpret_cast<float4*>(stats)[(int64_t)b * N + n] = make_float4(rp.p, dot<FOUT>(a2, xin), zinv, tr);
  stats[4 * (int64_t)B * N + (int64_t)b * N + n] = dp;
}

// dparams = (dW_inc [FOUT,FIN], dW_s [FOUT,FIN], da_in [2 FOUT], da_s [FOUT]) from G [2 FOUT + 3, FIN]:
//   da1 = W_inc (sum dp x), da2 = W_inc (sum dq x), da_s = W_s (sum dgate x)      (x_in = W_inc x, x_s = W_s x)
// ONE workgroup: sums the workgroups' partials gpart [ROWS * FIN][parts] itself — a wave per entry, lanes striding the
// partials, fixed tree — and writes the parameter gradients (the separate "sum the partials" launch of the first
// version is gone).  [Tried: the LAST workgroup of the main kernel doing this behind a device-scope ticket.  On this
// 8-XCD part 512 arrivals at one ticket serialise at about a microsecond each — 35 -> 600 us — whether the cost is
// the release fence (an L2 write-back per arrival) or the compare-and-swap retries; per-tile tickets in the split-K
// GEMMs cost more per product (+8 us) than the reduction launch they replaced.  Not kept.]
// One workgroup of FIN waves per OUTPUT (grid = 2 FOUT FIN + 3 FOUT): a weight-gradient output is one entry of G, an
// attention-vector output the dot product of a weight row with FIN entries — wave d sums entry d's partials (lanes
// stride them, fixed tree).  [One 1024-thread workgroup walking all 65 entries, 4-5 per wave: 6.4 us of latency.]
#define GO_FIN_T 1024
template <int FIN, int FOUT>
__global__ void __launch_bounds__(64 * FIN)
k_go_attn_bwd_finish(const float* __restrict__ gpart, int64_t parts, const float* __restrict__ w_inc,
                     const float* __restrict__ w_s, float* __restrict__ dparams) {
  __shared__ float G[FIN];
  go_attn_finish_output(gpart, parts, FIN, FOUT, w_inc, w_s, dparams, blockIdx.x, G);
}

// ---- backward, kernel B: input gradient + block partials of the parameter gradients ------------
// One thread per (sample, node).  A node's row list (its parents) is short, but its COLUMN list (the rows that
// read it = its children) can be long for hub nodes (the root feeds every level-1 term): columns longer than
// GO_HEAVY are walked by the whole wave (lanes stride the list, wave reduction), so the kernel's duration is
// not set by one thread's serial walk.
#define GO_SB 1
#define GO_HEAVY 12
template <int FIN, int FOUT>
__global__ void __launch_bounds__(GO_T)
k_go_attn_bwd_main(int B, int N, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                   const int32_t* __restrict__ t_ptr, const int32_t* __restrict__ t_row,
                   const float* __restrict__ x, const float* __restrict__ w_inc, const float* __restrict__ w_s,
                   const float* __restrict__ a_in, const float* __restrict__ a_s, const float* __restrict__ dy,
                   const float* __restrict__ stats, float* __restrict__ dx, float* __restrict__ gpart) {
  // parameter-gradient products G[2 FOUT + 3, FIN] = sum_nodes u[:,node] (x) x[:,node] are reduced over the block's
  // nodes on the matrix cores: u and x go through LDS once (node-major -> MFMA operand layout)
  constexpr int ROWS = 2 * FOUT + 3, TP = GO_T + 4;
  __shared__ float us[ROWS][TP];
  __shared__ float xt[FIN][TP];
  AttnW<FIN, FOUT> W;
  W.load(w_inc, w_s, a_in, a_s);
  int bx, b;
  go_block(bx, b);
  const int n = bx * GO_T + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool live = n < N;
  const int nn = live ? n : N - 1;                 // dead lanes shadow a valid node, results discarded
  const float* xb = x + (int64_t)b * FIN * N;
  const float* dyb = dy + (int64_t)b * FOUT * N;
  const float4* sp = reinterpret_cast<const float4*>(stats) + (int64_t)b * N;     // (p, q, zinv, tr) per node
  // ... but with EMPTY edge ranges: N-1 is the root, and a dead lane walking its column list serially (it is not
  // "heavy", being dead) used to set the duration of the whole kernel
  const int32_t c0 = live ? t_ptr[n] : 0, c1 = live ? t_ptr[n + 1] : 0;
  float xr[FIN], xin[FOUT], xs[FOUT], dyn[FOUT];
  load_node<FIN>(xb, N, nn, xr);
  load_node<FOUT>(dyb, N, nn, dyn);
  transform<FIN, FOUT>(W.wi, xr, xin);
  transform<FIN, FOUT>(W.ws, xr, xs);
  const float4 st_n = sp[nn];
  const float q_n = st_n.y;
  const float dp = live ? stats[4 * (int64_t)B * N + (int64_t)b * N + n] : 0.f;     // n as ROW: from kernel A
  // n as COLUMN: what the rows reading n send back
  float dq = 0.f, dxin[FOUT];
#pragma unroll
  for (int c = 0; c < FOUT; ++c) dxin[c] = 0.f;
  const bool heavy = c1 - c0 > GO_HEAVY;
  if (!heavy) {
    for (int32_t e = c0; e < c1; e += 2) {
      const bool two = e + 1 < c1;
      const int ra = t_row[e], rb = t_row[two ? e + 1 : e];
      float dya[FOUT], dyb2[FOUT];
      load_node<FOUT>(dyb, N, ra, dya);
      load_node<FOUT>(dyb, N, rb, dyb2);
      const float4 sa = sp[ra], sb = sp[rb];
      const float tha = go_tanh(sa.x + q_n), thb = go_tanh(sb.x + q_n);
      const float ala = go_exp(tha) * sa.z, alb = two ? go_exp(thb) * sb.z : 0.f;
      dq += (dot<FOUT>(dya, xin) - sa.w) * ala * (1.f - tha * tha);
#pragma unroll
      for (int c = 0; c < FOUT; ++c) dxin[c] += ala * dya[c];
      if (two) {
        dq += (dot<FOUT>(dyb2, xin) - sb.w) * alb * (1.f - thb * thb);
#pragma unroll
        for (int c = 0; c < FOUT; ++c) dxin[c] += alb * dyb2[c];
      }
    }
  }
  unsigned long long hmask = __ballot(heavy);
  while (hmask) {                                   // wave-uniform loop over this wave's hub nodes
    const int src = __ffsll((long long)hmask) - 1;
    hmask &= hmask - 1;
    const int32_t hc0 = __shfl(c0, src, 64), hc1 = __shfl(c1, src, 64);
    const float hq = __shfl(q_n, src, 64);
    float hxin[FOUT];
#pragma unroll
    for (int c = 0; c < FOUT; ++c) hxin[c] = __shfl(xin[c], src, 64);
    float pdq = 0.f, pdx[FOUT];
#pragma unroll
    for (int c = 0; c < FOUT; ++c) pdx[c] = 0.f;
    for (int32_t e = hc0 + lane; e < hc1; e += 64) {
      const int r = t_row[e];
      float dyr[FOUT];
      load_node<FOUT>(dyb, N, r, dyr);
      const float4 sr = sp[r];
      const float th = go_tanh(sr.x + hq);
      const float alpha = go_exp(th) * sr.z;
      pdq += (dot<FOUT>(dyr, hxin) - sr.w) * alpha * (1.f - th * th);
#pragma unroll
      for (int c = 0; c < FOUT; ++c) pdx[c] += alpha * dyr[c];
    }
    pdq = wave_sum_all(pdq);
#pragma unroll
    for (int c = 0; c < FOUT; ++c) pdx[c] = wave_sum_all(pdx[c]);
    if (lane == src) {
      dq = pdq;
#pragma unroll
      for (int c = 0; c < FOUT; ++c) dxin[c] = pdx[c];
    }
  }
  if (live) {
#pragma unroll
    for (int c = 0; c < FOUT; ++c) dxin[c] += dp * W.a1[c] + dq * W.a2[c];
    // gated self term
    const float g = go_rcp(1.f + go_exp(-dot<FOUT>(W.as, xs)));
    const float dgate = dot<FOUT>(dyn, xs) * g * (1.f - g);
    float dxs[FOUT];
#pragma unroll
    for (int c = 0; c < FOUT; ++c) dxs[c] = dyn[c] * g + dgate * W.as[c];
    // input gradient
    float* dxb = dx + (int64_t)b * FIN * N;
#pragma unroll
    for (int d = 0; d < FIN; ++d) {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < FOUT; ++c) t += W.wi[c][d] * dxin[c] + W.ws[c][d] * dxs[c];
      dxb[d * N + n] = t;
    }
    // rows of the parameter-gradient product: u = (dxin[FOUT], dxs[FOUT], dp, dq, dgate)
#pragma unroll
    for (int c = 0; c < FOUT; ++c) {
      us[c][threadIdx.x] = dxin[c];
      us[FOUT + c][threadIdx.x] = dxs[c];
    }
    us[2 * FOUT][threadIdx.x] = dp;
    us[2 * FOUT + 1][threadIdx.x] = dq;
    us[2 * FOUT + 2][threadIdx.x] = dgate;
#pragma unroll
    for (int d = 0; d < FIN; ++d) xt[d][threadIdx.x] = xr[d];
  } else {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) us[r][threadIdx.x] = 0.f;
#pragma unroll
    for (int d = 0; d < FIN; ++d) xt[d][threadIdx.x] = 0.f;
  }
  __syncthreads();
  {
    // wave w reduces nodes [64 w, 64 w + 64): 16 MFMAs (K = 4 nodes each); acc[r] = G[row 4g+r][feature m]
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int w = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* ua = &us[m < ROWS ? m : 0][64 * w + g];
    const float* xb2 = &xt[m < FIN ? m : 0][64 * w + g];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float a = (m < ROWS) ? ua[4 * c] : 0.f;
      const float bq = (m < FIN) ? xb2[4 * c] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, acc, 0, 0, 0);
    }
    // the four waves' accumulators are summed through LDS (us is free again after the barrier): one partial per
    // block, layout [ROWS * FIN][blocks] so that the second-stage sum of an entry reads contiguous memory
    __syncthreads();
    float* wsum = &us[0][0];
#pragma unroll
    for (int r = 0; r < 4; ++r) wsum[(w * 4 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (w == 0 && m < FIN) {
      const int64_t parts = (int64_t)gridDim.x * gridDim.y;
      float* gp = gpart + (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * g + r < ROWS) {
          const float t = (wsum[r * 64 + lane] + wsum[(4 + r) * 64 + lane]) +
                          (wsum[(8 + r) * 64 + lane] + wsum[(12 + r) * 64 + lane]);
          gp[(int64_t)((4 * g + r) * FIN + m) * parts] = t;
        }
    }
  }
}

int igcn_gemm_f32_batched_sum_impl(int64_t M, int64_t N, int64_t K, int batch, const float* A, int64_t sam,
                                   int64_t sak, int64_t a_batch, const float* B, int64_t sbn, int64_t sbk,
                                   int64_t b_batch, float* C, int64_t ldc, float* scratch, hipStream_t st);

extern "C" size_t igcn_go_attn_bwd_scratch_floats(int B, int N, int fin, int fout) {
  // channel-major kernels: statistics [B,N] float4 + one partial per wave; the LDS-resident kernel needs only one
  // partial per sample, which is less
  const int64_t rows = 2 * fout + 3, parts = igcn_cdiv(N, GO_T) * (int64_t)B * (GO_T / 64);
  return (size_t)(5 * (int64_t)B * N + parts * rows * fin + rows * fin + 64);
}

// =================================================================================================
// LDS-resident formulation of the backward: ONE 1024-thread workgroup per sample.  A sample's operands — x [FIN][N],
// dy [FOUT][N] and the per-node softmax statistics — are 132 KB at most in the shapes of the model (N = 3000,
// 2 -> 5), i.e. they fit the 160 KB of a CU.  The workgroup copies x and dy in once (coalesced), computes the
// statistics of all nodes into LDS (what k_go_attn_bwd_stats writes to HBM), and then runs the walks of
// k_go_attn_bwd_main with every gather — neighbour features, neighbour gradients, neighbour statistics — served
// from LDS; only the CSR indices (identical for all samples, L2-resident) and the coalesced dx rows touch global
// memory.  The parameter-gradient rows u (x) x of a thread's nodes wait in registers until the walks are done, then
// go through the (now free) LDS to the matrix cores exactly as in the channel-major kernel.
// =================================================================================================
#ifdef GO_ABL_PROBE
__device__ long long go_probe_buf[8 * 16];
#define GO_PROBE(i) do { if (threadIdx.x == 0 && blockIdx.x < 8) go_probe_buf[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
extern "C" int igcn_debug_go_probe(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(go_probe_buf), sizeof(long long) * 8 * 16);
}
#else
#define GO_PROBE(i)
#endif
// sum of `acc[0..F)` over the 64 lanes of a wave: while a lane still holds more than one value, lanes M apart exchange
// the half the partner keeps (HALF shuffles instead of 2 HALF), then plain xor steps; the total of value f ends up in
// acc[0] of lane 2 f (F = 32).  All indices are compile-time constants; fixed order.
template <int F, int HALF, int M>
__device__ __forceinline__ void go_butterfly(float (&acc)[F], int lane) {
  if constexpr (M >= 1) {
    if constexpr (HALF >= 1) {
      const bool up = (lane & M) != 0;
#pragma unroll
      for (int j = 0; j < HALF; ++j) {
        const float send = up ? acc[j] : acc[j + HALF];
        const float recv = __shfl_xor(send, M, 64);
        acc[j] = (up ? acc[j + HALF] : acc[j]) + recv;
      }
      go_butterfly<F, HALF / 2, M / 2>(acc, lane);
    } else {
      acc[0] += __shfl_xor(acc[0], M, 64);
      go_butterfly<F, 0, M / 2>(acc, lane);
    }
  }
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- LayerNorm-over-nodes backward folded into its consumer's copy-in -------------------------------------------
// The gradient a GO layer's backward receives is the LayerNorm's input gradient (go_model.py:246-251: y = layer(x),
// x' = dropout(relu(LN(y)))).  The LDS-resident backward kernels hold ONE sample — all FOUT rows of its N nodes — so
// they can form that gradient themselves from (y, dz, gamma, beta, keep, mean, rstd) while they copy it into LDS: the
// LayerNorm's own dX launch (one workgroup per (sample, channel) row: 12.8 us x 4 in the default step), the 15 MB it
// writes and the 15 MB read back disappear, and since a workgroup sees all channels of a sample it also leaves the
// affine gradients summed over them — part [B][2][N - pool] (d gamma | d beta rows of the kept nodes), reduced over samples by the deferred
// final reduction instead of k_nodes_ln_bwd_affine_multi's second pass over y and dz.  Arithmetic per element as
// k_nodes_ln_bwd_dy_v / ln_bwd_affine_v_body.  Preconditions (igcn_go_ln_fused_ok): N, pool multiples of 4, N / 4 <= T,
// 16-byte aligned tensors.
struct LnFuse {
  const float *y, *dz, *gamma, *beta, *keep, *mean, *rstd;
  float* part;
  int pool;
  const float *dz2, *dz3;          // further consumers' gradients of z (same layout as dz), added on load in this order; or NULL
  float* dgb;                      // [2][N] d gamma | d beta: the kernel writes the zeros of the pooled nodes' columns
};

// sum over the 16 lanes of a DPP row, in every lane of the row (quad permutes, then the half-row and row mirrors)
__device__ __forceinline__ float go_row16_sum_dpp(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
  return v;
}

// NV block-wide sums, the totals in s[] of every thread.  Wave totals on the VALU (DPP inside the 16-lane rows, the four
// row totals through v_readlane: a __shfl chain is six dependent LDS-pipe permutes per value), one LDS row per wave,
// summed in wave order by the first NV threads, the totals broadcast.  red: >= (T / 64 + 1) * NV floats; two barriers.
// fin(total, i) is applied by the ONE thread that holds total i before the broadcast: a division or a reciprocal square
// root every thread would otherwise repeat (10-15 VALU instructions each; with 16 waves per workgroup and two
// workgroups per CU, 100 instructions per thread are 1.3 us of the launch).
struct GoIdentity { __device__ float operator()(float t, int) const { return t; } };
template <int NV, int T, typename Fin = GoIdentity>
__device__ __forceinline__ void go_block_sums(float (&s)[NV], float* __restrict__ red, Fin fin = Fin()) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int ri = __float_as_int(go_row16_sum_dpp(s[i]));
    const float t = (__int_as_float(__builtin_amdgcn_readlane(ri, 0)) + __int_as_float(__builtin_amdgcn_readlane(ri, 16))) +
                    (__int_as_float(__builtin_amdgcn_readlane(ri, 32)) + __int_as_float(__builtin_amdgcn_readlane(ri, 48)));
    if (lane == i) red[w * NV + i] = t;
  }
  __syncthreads();
  float* tot = red + (T / 64) * NV;
  if (tid < NV) {
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < T / 64; ++ww) t += red[ww * NV + tid];
    tot[tid] = fin(t, tid);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) s[i] = tot[i];
}

// dys [FOUT][N] (LDS) = d loss / d y of sample b; red: >= (T / 64 + 1) * 2 * FOUT floats of LDS nobody else uses yet.
// Every thread of the workgroup must call (barrier inside); the caller's next barrier publishes dys.
template <int FOUT, int T>
__device__ __forceinline__ void ln_bwd_into_lds(const LnFuse& L, int b, int N, float* __restrict__ dys,
                                                float* __restrict__ red) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = 4 * tid, M = N - L.pool;
  const bool live = n < N, act = live && n >= L.pool;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 yv[FOUT], up[FOUT];
  float mu[FOUT], rs[FOUT];
  float4 g = zero4, be = zero4, kp = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
  for (int c = 0; c < FOUT; ++c) {                      // every load of the phase in flight at once
    const int64_t row = (int64_t)b * FOUT + c;
    mu[c] = L.mean[row];
    rs[c] = L.rstd[row];
    yv[c] = live ? ld4(L.y + row * N + n) : zero4;
    up[c] = act ? ld4(L.dz + row * M + (n - L.pool)) : zero4;
  }
  if (L.dz2) {                                          // workgroup-uniform: z had several consumers (igcn_sum_n's order)
#pragma unroll
    for (int c = 0; c < FOUT; ++c) {
      const int64_t off = ((int64_t)b * FOUT + c) * M + (n - L.pool);
      if (act) {
        const float4 t = ld4(L.dz2 + off);
        up[c].x += t.x; up[c].y += t.y; up[c].z += t.z; up[c].w += t.w;
        if (L.dz3) {
          const float4 t3 = ld4(L.dz3 + off);
          up[c].x += t3.x; up[c].y += t3.y; up[c].z += t3.z; up[c].w += t3.w;
        }
      }
    }
  }
  if (live) g = ld4(L.gamma + n);
  if (act) {
    be = ld4(L.beta + n);
    if (L.keep) kp = ld4(L.keep + (int64_t)b * N + n);
  }
  float s[2 * FOUT];
  float4 dg = zero4, db = zero4;
#pragma unroll
  for (int c = 0; c < FOUT; ++c) {
    float4 xh = zero4, dx = zero4;
    if (live) xh = make_float4((yv[c].x - mu[c]) * rs[c], (yv[c].y - mu[c]) * rs[c], (yv[c].z - mu[c]) * rs[c],
                               (yv[c].w - mu[c]) * rs[c]);
    if (act) {
      float4 u = up[c];
      u.x *= kp.x; u.y *= kp.y; u.z *= kp.z; u.w *= kp.w;
      u.x = xh.x * g.x + be.x > 0.f ? u.x : 0.f;
      u.y = xh.y * g.y + be.y > 0.f ? u.y : 0.f;
      u.z = xh.z * g.z + be.z > 0.f ? u.z : 0.f;
      u.w = xh.w * g.w + be.w > 0.f ? u.w : 0.f;
      dx = make_float4(u.x * g.x, u.y * g.y, u.z * g.z, u.w * g.w);
      dg.x += u.x * xh.x; dg.y += u.y * xh.y; dg.z += u.z * xh.z; dg.w += u.w * xh.w;
      db.x += u.x; db.y += u.y; db.z += u.z; db.w += u.w;
    }
    s[c] = (dx.x + dx.y) + (dx.z + dx.w);
    s[FOUT + c] = (dx.x * xh.x + dx.y * xh.y) + (dx.z * xh.z + dx.w * xh.w);
    yv[c] = xh;                                         // the registers now hold xhat and the masked, scaled upstream
    up[c] = dx;
  }
  if (act) {                                            // per-sample affine partials [B][2][N - pool] of the KEPT nodes
    float* pr = L.part + (int64_t)b * 2 * M + (n - L.pool);
    *reinterpret_cast<float4*>(pr) = dg;
    *reinterpret_cast<float4*>(pr + M) = db;
  } else if (live && b == 0) {                          // pooled nodes feed nothing: exact zeros, written once
    *reinterpret_cast<float4*>(L.dgb + n) = zero4;
    *reinterpret_cast<float4*>(L.dgb + N + n) = zero4;
  }
  // [first version: wave_sum per value and every thread adding up all T / 64 rows itself — 60 permutes and 160 LDS reads
  // per thread, 6.8 us of a 10.4 us copy-in; with go_block_sums 3.8 of 7.3]
  const float fN = (float)N;
  go_block_sums<2 * FOUT, T>(s, red, [fN](float t, int) { return t / fN; });
  if (live) {
#pragma unroll
    for (int c = 0; c < FOUT; ++c) {
      const float s1 = s[c], s2 = s[FOUT + c];
      float4 o;
      o.x = rs[c] * (up[c].x - s1 - yv[c].x * s2);
      o.y = rs[c] * (up[c].y - s1 - yv[c].y * s2);
      o.z = rs[c] * (up[c].z - s1 - yv[c].z * s2);
      o.w = rs[c] * (up[c].w - s1 - yv[c].w * s2);
      *reinterpret_cast<float4*>(dys + c * N + n) = o;
    }
  }
}

#define GO_ABL_T 1024                                   // default workgroup; 512 when two workgroups then share a CU
#define GO_ABL_MAXIT 4
template <int FIN, int FOUT, int MAXIT, int T>
__global__ void __launch_bounds__(T, 4)
k_go_attn_bwd_lds(int N, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                  const int32_t* __restrict__ t_ptr, const int32_t* __restrict__ t_row, const float* __restrict__ x,
                  const float* __restrict__ w_inc, const float* __restrict__ w_s, const float* __restrict__ a_in,
                  const float* __restrict__ a_s, const float* __restrict__ dy, const int32_t* __restrict__ order,
                  float* __restrict__ dx, float* __restrict__ gpart, const LnFuse L) {
  extern __shared__ float go_abl[];
  constexpr int ROWS = 2 * FOUT + 3, TP = T + 4;
  const int NP = (N + 3) & ~3;
  float* xs = go_abl;                                   // [FIN][NP]
  float* dys = xs + FIN * NP;                           // [FOUT][NP]
  float4* st = reinterpret_cast<float4*>(dys + FOUT * NP);      // [N]: (p, dp, 1/Z, tr/Z)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  GO_PROBE(0);
  const float* xb = x + (int64_t)b * FIN * N;
  const float* dyb = dy + (int64_t)b * FOUT * N;
  // Two thread -> node maps.  The row side (statistics walk: 1-2 parents per GO term, no imbalance) takes nodes in
  // order, slot it * 1024 + tid.  The COLUMN side (children lists: 0 for the leaves, ~4 +- 3 for inner terms, 99 for
  // the root) takes them through `order` (igcn_go_attn_walk_order): nodes sorted by column degree, dealt to the
  // (wave, pass) slots in groups of 64 so that the lanes of a wave walk lists of (nearly) equal length and the 16
  // waves carry (nearly) equal totals — in node order the waves holding the upper levels walked 5-10 steps per
  // pass while the leaf waves idled at the barrier (9.7 us of a 23.5 us workgroup; 3 us of it was work).
  // CSR pointers: issued before the slab copy so that their latency hides behind it
  int32_t pr0[MAXIT], pr1[MAXIT], pc0[MAXIT], pc1[MAXIT], ncol[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int slot = it * T + tid;
    ncol[it] = order ? order[slot] : (slot < N ? slot : -1);
  }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int n = it * T + tid;
    const bool live = n < N;                            // lanes past the end: EMPTY edge ranges
    pr0[it] = live ? row_ptr[n] : 0; pr1[it] = live ? row_ptr[n + 1] : 0;
    const int nc = ncol[it];
    pc0[it] = nc >= 0 ? t_ptr[nc] : 0; pc1[it] = nc >= 0 ? t_ptr[nc + 1] : 0;
  }
  // 16-byte slab copies: every load issued before the first LDS store (a load / store loop is one memory round trip
  // per trip: 2-4 of them in front of a phase that is nothing but latency)
  constexpr int XC = (FIN * MAXIT + 3) / 4, DC = (FOUT * MAXIT + 3) / 4;
  if (L.y) {                                            // block-uniform: dy formed here (ln_bwd_into_lds; NP == N)
    float4 xv[XC];
#pragma unroll
    for (int k = 0; k < XC; ++k) {
      const int i = (tid + k * T) * 4;
      if (i < FIN * N) xv[k] = ld4(xb + i);
    }
    ln_bwd_into_lds<FOUT, T>(L, b, N, dys, reinterpret_cast<float*>(st));
#pragma unroll
    for (int k = 0; k < XC; ++k) {
      const int i = (tid + k * T) * 4;
      if (i < FIN * N) *reinterpret_cast<float4*>(xs + i) = xv[k];
    }
  } else if (NP == N && (((uintptr_t)xb | (uintptr_t)dyb) & 15) == 0) {
    float4 xv[XC], dv[DC];
#pragma unroll
    for (int k = 0; k < XC; ++k) {
      const int i = (tid + k * T) * 4;
      if (i < FIN * N) xv[k] = ld4(xb + i);
    }
#pragma unroll
    for (int k = 0; k < DC; ++k) {
      const int i = (tid + k * T) * 4;
      if (i < FOUT * N) dv[k] = ld4(dyb + i);
    }
#pragma unroll
    for (int k = 0; k < XC; ++k) {
      const int i = (tid + k * T) * 4;
      if (i < FIN * N) *reinterpret_cast<float4*>(xs + i) = xv[k];
    }
#pragma unroll
    for (int k = 0; k < DC; ++k) {
      const int i = (tid + k * T) * 4;
      if (i < FOUT * N) *reinterpret_cast<float4*>(dys + i) = dv[k];
    }
  } else {
    for (int d = 0; d < FIN; ++d)
      for (int n = tid; n < N; n += T) xs[d * NP + n] = xb[d * N + n];
    for (int c = 0; c < FOUT; ++c)
      for (int n = tid; n < N; n += T) dys[c * NP + n] = dyb[c * N + n];
  }
  // ... and the first two neighbours of each list (the GO DAG rarely has more: the walks then never wait on L2)
  int pm0[MAXIT], pm1[MAXIT], pra[MAXIT], prb[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int dr = pr1[it] - pr0[it], dc = pc1[it] - pc0[it];
    pm0[it] = dr > 0 ? col[pr0[it]] : 0;
    pm1[it] = dr > 1 ? col[pr0[it] + 1] : 0;
    pra[it] = dc > 0 ? t_row[pc0[it]] : 0;
    prb[it] = dc > 1 ? t_row[pc0[it] + 1] : 0;
  }
  AttnW<FIN, FOUT> W;
  W.load(w_inc, w_s, a_in, a_s);
  __syncthreads();
  GO_PROBE(1);

  // ---- statistics of every node of the sample + the row-side score gradient (RowPass) ---------------
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int n = it * T + tid;
    if (n >= N) continue;
    float xr[FIN], xin[FOUT], dyn[FOUT];
    load_node<FIN>(xs, NP, n, xr);
    load_node<FOUT>(dys, NP, n, dyn);
    transform<FIN, FOUT>(W.wi, xr, xin);
    RowPass<FIN, FOUT> rp;
    rp.begin(W.wi, W.a2, dyn, dot<FOUT>(W.a1, xin));
    const int32_t p0 = pr0[it], p1 = pr1[it];
    for (int32_t e = p0; e < p1; e += 2) {              // two edges per step, the first pair already in registers
      const bool two = e + 1 < p1;
      const int m0 = e == p0 ? pm0[it] : col[e], m1 = two ? (e == p0 ? pm1[it] : col[e + 1]) : m0;
      float xm0[FIN], xm1[FIN];
      load_node<FIN>(xs, NP, m0, xm0);
      load_node<FIN>(xs, NP, m1, xm1);
      rp.edge(xm0);
      if (two) rp.edge(xm1);
    }
    float zinv, tr, dp;
    rp.end(p1 > p0, zinv, tr, dp);
    st[n] = make_float4(rp.p, dp, zinv, tr);
  }
  __syncthreads();
  GO_PROBE(2);

  // ---- column walks: input gradient, parameter-gradient rows kept in registers ---------------------
  float uu[MAXIT][ROWS], xx[MAXIT][FIN];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) uu[it][r] = 0.f;
#pragma unroll
    for (int d = 0; d < FIN; ++d) xx[it][d] = 0.f;
    if (it * T < N) {                            // block-uniform
      const int n = ncol[it];
      const bool live = n >= 0;
      const int nn = live ? n : N - 1;                  // dead lanes shadow a valid node with EMPTY edge ranges
      const int32_t c0 = pc0[it], c1 = pc1[it];
      float xr[FIN], xin[FOUT], xsl[FOUT], dyn[FOUT];
      load_node<FIN>(xs, NP, nn, xr);
      load_node<FOUT>(dys, NP, nn, dyn);
      transform<FIN, FOUT>(W.wi, xr, xin);
      transform<FIN, FOUT>(W.ws, xr, xsl);
      const float q_n = dot<FOUT>(W.a2, xin);
      const float dp = st[nn].y;                        // row side, from the statistics walk
      float dq = 0.f, dxin[FOUT];                       // n as COLUMN: what the rows reading n send back
#pragma unroll
      for (int c = 0; c < FOUT; ++c) dxin[c] = 0.f;
      const bool heavy = c1 - c0 > GO_HEAVY;
      if (!heavy) {
        for (int32_t e = c0; e < c1; e += 2) {
          const bool two = e + 1 < c1;
          const int ra = e == c0 ? pra[it] : t_row[e], rb = two ? (e == c0 ? prb[it] : t_row[e + 1]) : ra;
          float dya[FOUT], dyb2[FOUT];
          load_node<FOUT>(dys, NP, ra, dya);
          load_node<FOUT>(dys, NP, rb, dyb2);
          const float4 sa = st[ra], sb = st[rb];
          const float tha = go_tanh(sa.x + q_n), thb = go_tanh(sb.x + q_n);
          const float ala = go_exp(tha) * sa.z, alb = two ? go_exp(thb) * sb.z : 0.f;
          dq += (dot<FOUT>(dya, xin) - sa.w) * ala * (1.f - tha * tha);
#pragma unroll
          for (int c = 0; c < FOUT; ++c) dxin[c] += ala * dya[c];
          if (two) {
            dq += (dot<FOUT>(dyb2, xin) - sb.w) * alb * (1.f - thb * thb);
#pragma unroll
            for (int c = 0; c < FOUT; ++c) dxin[c] += alb * dyb2[c];
          }
        }
      }
      unsigned long long hmask = __ballot(heavy);
      while (hmask) {                                   // hub columns: the whole wave strides the list
        const int src = __ffsll((long long)hmask) - 1;
        hmask &= hmask - 1;
        const int32_t hc0 = __shfl(c0, src, 64), hc1 = __shfl(c1, src, 64);
        const float hq = __shfl(q_n, src, 64);
        float hxin[FOUT];
#pragma unroll
        for (int c = 0; c < FOUT; ++c) hxin[c] = __shfl(xin[c], src, 64);
        float pdq = 0.f, pdx[FOUT];
#pragma unroll
        for (int c = 0; c < FOUT; ++c) pdx[c] = 0.f;
        for (int32_t e = hc0 + lane; e < hc1; e += 64) {
          const int r = t_row[e];
          float dyr[FOUT];
          load_node<FOUT>(dys, NP, r, dyr);
          const float4 sr = st[r];
          const float th = go_tanh(sr.x + hq);
          const float alpha = go_exp(th) * sr.z;
          pdq += (dot<FOUT>(dyr, hxin) - sr.w) * alpha * (1.f - th * th);
#pragma unroll
          for (int c = 0; c < FOUT; ++c) pdx[c] += alpha * dyr[c];
        }
        pdq = wave_sum_all(pdq);
#pragma unroll
        for (int c = 0; c < FOUT; ++c) pdx[c] = wave_sum_all(pdx[c]);
        if (lane == src) {
          dq = pdq;
#pragma unroll
          for (int c = 0; c < FOUT; ++c) dxin[c] = pdx[c];
        }
      }
      if (live) {
#pragma unroll
        for (int c = 0; c < FOUT; ++c) dxin[c] += dp * W.a1[c] + dq * W.a2[c];
        const float g = go_rcp(1.f + go_exp(-dot<FOUT>(W.as, xsl)));       // gated self term
        const float dgate = dot<FOUT>(dyn, xsl) * g * (1.f - g);
        float dxs[FOUT];
#pragma unroll
        for (int c = 0; c < FOUT; ++c) dxs[c] = dyn[c] * g + dgate * W.as[c];
#pragma unroll
        for (int d = 0; d < FIN; ++d) {
          float t = 0.f;
#pragma unroll
          for (int c = 0; c < FOUT; ++c) t += W.wi[c][d] * dxin[c] + W.ws[c][d] * dxs[c];
          xs[d * NP + n] = t;                             // dx over x in place (nobody else reads x[., n] any more):
        }                                                 // leaves LDS in one coalesced pass below
#pragma unroll
        for (int c = 0; c < FOUT; ++c) {
          uu[it][c] = dxin[c];
          uu[it][FOUT + c] = dxs[c];
        }
        uu[it][2 * FOUT] = dp;
        uu[it][2 * FOUT + 1] = dq;
        uu[it][2 * FOUT + 2] = dgate;
#pragma unroll
        for (int d = 0; d < FIN; ++d) xx[it][d] = xr[d];
      }
    }
  }
  GO_PROBE(3);
  __syncthreads();                                      // dy and the statistics are dead: MFMA staging area
  GO_PROBE(4);

  // ---- G[ROWS, FIN] += u (x) x over the sample's nodes on the matrix cores ----------------------
  // wave w stages and reduces the nodes of its own 64 threads: the staging columns [64 w, 64 w + 64) are private to
  // the wave, so the iterations need no workgroup barrier (LDS serves a wave's accesses in order)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  {                                                     // dx [FIN][N] out of LDS, 16 bytes per lane
    float* dxb = dx + (int64_t)b * FIN * N;
    if (NP == N && ((uintptr_t)dxb & 15) == 0) {
      for (int i = tid * 4; i < FIN * N; i += T * 4)
        *reinterpret_cast<float4*>(dxb + i) = *reinterpret_cast<const float4*>(xs + i);
    } else {
      for (int d = 0; d < FIN; ++d)
        for (int n = tid; n < N; n += T) dxb[d * N + n] = xs[d * NP + n];
    }
  }
  const int w = tid >> 6;
  if constexpr (ROWS * FIN <= 32) {
    // few products (f_in = 2: 13 x 2): per-thread sums over its nodes, a halving butterfly over the wave (32 shuffles
    // for 32 values), the 16 wave totals through LDS.  [On the matrix cores the same products cost 768 16x16x4
    // instructions per workgroup for 26 useful outputs each — 4.6 us of a 20 us workgroup (phase probe).]
    float gv[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) gv[e] = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int d = 0; d < FIN; ++d) gv[r * FIN + d] += uu[it][r] * xx[it][d];
    go_butterfly<32, 16, 32>(gv, lane);
    float* wpart = dys;                                 // [T / 64][32]
    if ((lane & 1) == 0) wpart[w * 32 + (lane >> 1)] = gv[0];
    __syncthreads();
    if (tid < ROWS * FIN) {
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < T / 64; ++ww) t += wpart[ww * 32 + tid];
      gpart[(int64_t)tid * gridDim.x + b] = t;
    }
  } else {
  float* us = dys;                                      // [ROWS][TP], behind the dx slab
  float* xt = us + ROWS * TP;                           // [FIN][TP]
  const int m = lane & 15, g4 = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    if (it * T < N) {                            // block-uniform
#pragma unroll
      for (int r = 0; r < ROWS; ++r) us[r * TP + tid] = uu[it][r];
#pragma unroll
      for (int d = 0; d < FIN; ++d) xt[d * TP + tid] = xx[it][d];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const float* ua = us + (m < ROWS ? m : 0) * TP + 64 * w + g4;
      const float* xa = xt + (m < FIN ? m : 0) * TP + 64 * w + g4;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float a = (m < ROWS) ? ua[4 * c] : 0.f;
        const float bq = (m < FIN) ? xa[4 * c] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, acc, 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  GO_PROBE(5);
  float* wsum = xt + FIN * TP;                          // [16 waves * 4][64], behind the staging area
#pragma unroll
  for (int r = 0; r < 4; ++r) wsum[(w * 4 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (w == 0 && m < FIN) {
    const int64_t parts = gridDim.x;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * g4 + r < ROWS) {
        float t = 0.f;
#pragma unroll
        for (int ww = 0; ww < T / 64; ++ww) t += wsum[(ww * 4 + r) * 64 + lane];
        gpart[(int64_t)((4 * g4 + r) * FIN + m) * parts + b] = t;
      }
  }
  }
  GO_PROBE(6);
}

static size_t go_abl_lds_bytes(int N, int fin, int fout, int T) {
  const size_t np = ((size_t)N + 3) & ~(size_t)3;
  const size_t walk = (size_t)(fout + 4) * np * sizeof(float);          // dy + statistics, then (same bytes) ...
  const size_t stage = ((size_t)(2 * fout + 3 + fin) * (T + 4) + (T / 64) * 4 * 64) * sizeof(float);
  return (size_t)fin * np * sizeof(float) + (walk > stage ? walk : stage);      // x / dx slab in front
}
// Workgroup size: 512 threads when that lets TWO workgroups share a CU (<= 80 KB of LDS each, <= 128 registers at
// 4 waves per SIMD) — one stages its sample while the other computes, and all samples of a 256-graph step are
// resident at once; 1024 threads otherwise (N = 3000: 132 KB, one workgroup per CU whatever its size).
static int go_abl_threads(int N, int fin, int fout) {
  if (N <= 512 * GO_ABL_MAXIT && go_abl_lds_bytes(N, fin, fout, 512) <= 80 * 1024) return 512;
  return GO_ABL_T;
}
extern "C" int igcn_go_attn_bwd_threads(int N, int fin, int fout) { return go_abl_threads(N, fin, fout); }

// Thread -> node map of the LDS-resident backward's column walks (HOST code; structure-only, computed once per
// hierarchy).  T = igcn_go_attn_bwd_threads(N, fin, fout).  Slots: pass it, thread tid -> order[it * T + tid] (-1 = idle); wave w owns slots [64 w, 64 w + 64) of
// every pass.  Nodes are sorted by column degree (descending, ties by id) and cut into groups of 64 = one wave-pass
// each (hub nodes last), so a wave's lanes walk lists of nearly equal length; a group's cost ~ fixed per-node work + list steps of its
// longest ordinary list + the wave-cooperative hub lists; groups go to the wave with the smallest load so far (longest
// first).  Any permutation gives the same numbers per node; the order only moves work between waves.
extern "C" int igcn_go_attn_walk_slots(int N, int fin, int fout) {
  const int T = go_abl_threads(N, fin, fout);
  return (int)(igcn_cdiv(N > 0 ? N : 1, T) * T);
}

extern "C" int igcn_go_attn_walk_order(int N, int fin, int fout, const int32_t* t_ptr_host, int32_t* order_host) {
  IGCN_REQUIRE(N > 0 && t_ptr_host && order_host, "go_attn_walk_order: bad arguments");
  const int T = go_abl_threads(N, fin, fout);
  const int passes = (int)igcn_cdiv(N, T), waves = T / 64;
  std::vector<int> nodes(N);
  for (int i = 0; i < N; ++i) nodes[i] = i;
  auto deg = [&](int n) { return t_ptr_host[n + 1] - t_ptr_host[n]; };
  // hubs (walked by the whole wave, whatever their lane) sort BEHIND the leaves: they join wave-passes that have no
  // list work of their own instead of the pass with the longest ordinary lists
  auto key = [&](int n) { const int d = deg(n); return d > GO_HEAVY ? -1 : d; };
  std::stable_sort(nodes.begin(), nodes.end(), [&](int a, int b) { return key(a) > key(b); });
  const int ngroups = (int)igcn_cdiv(N, 64);
  std::vector<double> cost(ngroups);
  for (int g = 0; g < ngroups; ++g) {
    int longest = 0, hubs = 0;
    for (int k = g * 64; k < N && k < g * 64 + 64; ++k) {
      const int d = deg(nodes[k]);
      if (d > GO_HEAVY) hubs += 1 + d / 64; else if (d > longest) longest = d;
    }
    cost[g] = 1.5 + (longest + 1) / 2 + 3.0 * hubs;     // in units of one two-edge list step
  }
  std::vector<int> gorder(ngroups);
  for (int g = 0; g < ngroups; ++g) gorder[g] = g;
  std::stable_sort(gorder.begin(), gorder.end(), [&](int a, int b) { return cost[a] > cost[b]; });
  std::vector<double> load(waves, 0.0);
  std::vector<int> used(waves, 0);
  for (int i = 0; i < passes * T; ++i) order_host[i] = -1;
  for (int g : gorder) {
    int best = -1;
    for (int w = 0; w < waves; ++w)
      if (used[w] < passes && (best < 0 || load[w] < load[best])) best = w;
    IGCN_REQUIRE(best >= 0, "go_attn_walk_order: no free slot");
    const int base = used[best] * T + best * 64;
    for (int k = g * 64, j = 0; k < N && j < 64; ++k, ++j) order_host[base + j] = nodes[k];
    used[best] += 1;
    load[best] += cost[g];
  }
  return IGCN_OK;
}

// IGCN_GO_ATTN_CM=1 (A/B runs): the global-memory kernels (attention backward, decoder forward / backward) even when a
// sample fits LDS — these are also the fallbacks for hierarchies too large for LDS
static bool go_attn_force_cm(void) { return igcn_opt(IGCN_OPT_GO_ATTN_CM); }

int igcn_queue_go_finish(const float* gpart, int64_t parts, int fin, int fout, const float* w_inc, const float* w_s,
                         float* dparams, hipStream_t st);                                       // plan.hip

static bool go_ln_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static bool go_attn_bwd_in_lds(int N, int fin, int fout) {
  const int TT = go_abl_threads(N, fin, fout);
  return !go_attn_force_cm() && go_abl_lds_bytes(N, fin, fout, TT) <= 160 * 1024 && N <= TT * GO_ABL_MAXIT;
}
// 1 when igcn_go_attn_ln_bwd runs for these sizes (the LDS-resident kernel takes the layer and LnFuse's shape
// preconditions hold); otherwise call igcn_nodes_ln_bwd* and igcn_go_attn_bwd
extern "C" int igcn_go_attn_ln_fused_ok(int N, int fin, int fout, int pool) {
  if (N <= 0 || pool < 0 || pool >= N || N % 4 || pool % 4) return 0;
  return go_attn_bwd_in_lds(N, fin, fout) && N / 4 <= go_abl_threads(N, fin, fout) ? 1 : 0;
}
extern "C" size_t igcn_go_ln_part_floats(int B, int N) { return (size_t)B * 2 * (size_t)N + 64; }   // (N - pool columns used)

static int go_attn_bwd_impl(int B, int N, int fin, int fout, const int32_t* row_ptr, const int32_t* col,
                            const int32_t* t_ptr, const int32_t* t_row, const int32_t* walk_order, const float* x,
                            const float* w_inc, const float* w_s, const float* a_in, const float* a_s, const float* dy,
                            float* dx, float* dparams, float* scratch, const LnFuse& L, hipStream_t st) {
  const int64_t rows = 2 * fout + 3;
  const int TT = go_abl_threads(N, fin, fout);
  const size_t abl_lds = go_abl_lds_bytes(N, fin, fout, TT);
  if (go_attn_bwd_in_lds(N, fin, fout)) {
    float* gpart = scratch;                                         // [rows * fin][B] block partials
    const int iters = (int)igcn_cdiv(N, TT);
#define CALLLI(FI, FO, MI, TV)                                                                                    \
  {                                                                                                               \
    IGCN_ALLOW_BIG_LDS((k_go_attn_bwd_lds<FI, FO, MI, TV>));                                                      \
    hipLaunchKernelGGL((k_go_attn_bwd_lds<FI, FO, MI, TV>), dim3(B), dim3(TV), abl_lds, st, N, row_ptr, col,      \
                       t_ptr, t_row, x, w_inc, w_s, a_in, a_s, dy, walk_order, dx, gpart, L);                     \
  }
#define CALLLT(FI, FO, TV)                                                                                        \
  if (iters <= 1) CALLLI(FI, FO, 1, TV) else if (iters == 2) CALLLI(FI, FO, 2, TV)                                \
  else if (iters == 3) CALLLI(FI, FO, 3, TV) else CALLLI(FI, FO, 4, TV)
#define CALLL(FI, FO)                                                                                             \
  if (TT == 512) { CALLLT(FI, FO, 512) } else { CALLLT(FI, FO, 1024) }
    GO_DISPATCH(fin, fout, CALLL)
#undef CALLL
#undef CALLLT
#undef CALLLI
    IGCN_CHECK_LAUNCH("go_attn_bwd(lds)");
    if (igcn_queue_go_finish(gpart, (int64_t)B, fin, fout, w_inc, w_s, dparams, st)) return IGCN_OK;
#define CALLF(FI, FO) \
  hipLaunchKernelGGL((k_go_attn_bwd_finish<FI, FO>), dim3(2 * FO * FI + 3 * FO), dim3(64 * FI), 0, st, gpart, (int64_t)B, w_inc, w_s, dparams)
    GO_DISPATCH(fin, fout, CALLF)
#undef CALLF
    IGCN_CHECK_LAUNCH("go_attn_bwd_finish");
    return IGCN_OK;
  }
  IGCN_REQUIRE(L.y == nullptr, "go_attn_ln_bwd: this layer does not run LDS-resident (igcn_go_attn_ln_fused_ok)");
  float* stats = scratch;                                         // float4 [B,N] + dp [B,N]
  float* gpart = stats + 5 * (int64_t)B * N;                       // [blocks * 4 waves][rows * fin] block partials
  dim3 grid((unsigned)igcn_cdiv(N, GO_T), B);
  const int64_t parts = (int64_t)grid.x * grid.y;
#define CALL(FI, FO)                                                                                             \
  hipLaunchKernelGGL((k_go_attn_bwd_stats<FI, FO>), grid, dim3(GO_T), 0, st, B, N, row_ptr, col, x, w_inc, a_in,  \
                     dy, stats);                                                                                  \
  hipLaunchKernelGGL((k_go_attn_bwd_main<FI, FO>), grid, dim3(GO_T), 0, st, B, N, row_ptr, col, t_ptr, t_row, x,  \
                     w_inc, w_s, a_in, a_s, dy, stats, dx, gpart)
  GO_DISPATCH(fin, fout, CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("go_attn_bwd");
  if (igcn_queue_go_finish(gpart, parts, fin, fout, w_inc, w_s, dparams, st)) return IGCN_OK;
#define CALLF(FI, FO) \
  hipLaunchKernelGGL((k_go_attn_bwd_finish<FI, FO>), dim3(2 * FO * FI + 3 * FO), dim3(64 * FI), 0, st, gpart, parts, w_inc, w_s, dparams)
  GO_DISPATCH(fin, fout, CALLF)
#undef CALLF
  IGCN_CHECK_LAUNCH("go_attn_bwd_finish");
  return IGCN_OK;
}

extern "C" int igcn_go_attn_bwd(int B, int N, int fin, int fout, const int32_t* row_ptr, const int32_t* col,
                                const int32_t* t_ptr, const int32_t* t_row, const int32_t* walk_order,
                                const float* x, const float* w_inc,
                                const float* w_s, const float* a_in, const float* a_s, const float* dy, float* dx,
                                float* dparams, float* scratch, void* stream) {
  IGCN_REQUIRE(B > 0 && N > 0, "go_attn_bwd: bad sizes");
  const LnFuse none = {};
  return go_attn_bwd_impl(B, N, fin, fout, row_ptr, col, t_ptr, t_row, walk_order, x, w_inc, w_s, a_in, a_s, dy, dx,
                          dparams, scratch, none, (hipStream_t)stream);
}

// GO attention layer backward WITH the backward of the LayerNorm block that follows it (go_model.py:219-251): dz is the
// gradient of z = dropout(relu(LN(y)))[.., pool:], y the layer's own output; dx and dparams as igcn_go_attn_bwd,
// dgb [2, N] = d gamma | d beta, part: igcn_go_ln_part_floats(B, N) floats (alive until the deferred reductions ran).
extern "C" int igcn_go_attn_ln_bwd(int B, int N, int fin, int fout, const int32_t* row_ptr, const int32_t* col,
                                   const int32_t* t_ptr, const int32_t* t_row, const int32_t* walk_order,
                                   const float* x, const float* w_inc, const float* w_s, const float* a_in,
                                   const float* a_s, int pool, const float* y, const float* gamma, const float* beta,
                                   const float* keep, const float* mean, const float* rstd, const float* dz,
                                   const float* dz2, const float* dz3, float* dx, float* dparams, float* dgb,
                                   float* scratch, float* part, void* stream) {
  IGCN_REQUIRE(B > 0 && N > 0, "go_attn_ln_bwd: bad sizes");
  IGCN_REQUIRE(igcn_go_attn_ln_fused_ok(N, fin, fout, pool), "go_attn_ln_bwd: sizes not supported (igcn_go_attn_ln_fused_ok)");
  IGCN_REQUIRE(y && gamma && beta && mean && rstd && dz && dgb && part, "go_attn_ln_bwd: null operand");
  IGCN_REQUIRE(go_ln_al16(x) && go_ln_al16(y) && go_ln_al16(dz) && go_ln_al16(gamma) && go_ln_al16(beta) &&
               go_ln_al16(keep) && go_ln_al16(part) && go_ln_al16(dx) && go_ln_al16(dz2) && go_ln_al16(dz3),
               "go_attn_ln_bwd: operands must be 16-byte aligned");
  IGCN_REQUIRE(dz2 != nullptr || dz3 == nullptr, "go_attn_ln_bwd: dz3 without dz2");
  const LnFuse L = {y, dz, gamma, beta, keep, mean, rstd, part, pool, dz2, dz3, dgb};
  int rc = go_attn_bwd_impl(B, N, fin, fout, row_ptr, col, t_ptr, t_row, walk_order, x, w_inc, w_s, a_in, a_s,
                            nullptr, dx, dparams, scratch, L, (hipStream_t)stream);
  if (rc) return rc;
  const int M = N - pool;                               // part rows: [d gamma (M) | d beta (M)] of the kept nodes
  if (pool == 0) return igcn_launch_reduce_rows_final(part, B, 2 * (int64_t)M, 2 * M, dgb, (hipStream_t)stream);
  if ((rc = igcn_launch_reduce_rows_final(part, B, 2 * (int64_t)M, M, dgb + pool, (hipStream_t)stream))) return rc;
  return igcn_launch_reduce_rows_final(part + M, B, 2 * (int64_t)M, M, dgb + N + pool, (hipStream_t)stream);
}

// =================================================================================================
// LayerNorm over nodes + ReLU + node dropout + pooling
// =================================================================================================
__global__ void __launch_bounds__(GO_T)
k_nodes_ln_fwd(int f, int N, int pool, float eps, const float* __restrict__ y, const float* __restrict__ gamma,
               const float* __restrict__ beta, const float* __restrict__ keep, float* __restrict__ z,
               float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  __shared__ float red[16];
  const int row = blockIdx.x;  // b*f + c
  const int b = row / f;
  const float* yr = y + (int64_t)row * N;
  float s = 0.f;
  for (int n = threadIdx.x; n < N; n += GO_T) s += yr[n];
  const float mean = block_sum_all(s, red) / (float)N;
  float v = 0.f;
  for (int n = threadIdx.x; n < N; n += GO_T) {
    const float d = yr[n] - mean;
    v += d * d;
  }
  const float var = block_sum_all(v, red) / (float)N;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
  const int M = N - pool;
  float* zr = z + (int64_t)row * M;
  for (int n = pool + threadIdx.x; n < N; n += GO_T) {
    float t = (yr[n] - mean) * rstd * gamma[n] + beta[n];
    t = fmaxf(t, 0.f);
    if (keep) t *= keep[(int64_t)b * N + n];
    zr[n - pool] = t;
  }
}

// Register-resident form: the row (N <= 4096 floats, N and pool multiples of 4) is read ONCE with 16-byte loads and
// kept in registers for the mean, the variance and the normalisation (the generic kernel makes three passes).
#define LN_VPT 4
#define LN_TMAX 1024

// Which (sample, channel) row a workgroup takes: XCD c the rows [c R/8, (c+1) R/8) in order, so that the f channel rows
// of a sample — which all read the sample's dropout factors keep[b, :] — meet in one L2 (blockIdx.x taken as it comes
// puts them on f different XCDs).
__device__ __forceinline__ int ln_row() {
  const int n = (int)gridDim.x, x = (int)blockIdx.x;
  return (n & 7) ? x : (x & 7) * (n >> 3) + (x >> 3);
}

// (256 threads for rows up to 4096 nodes, 1024 threads up to 16384: the 10 000-node hierarchy of configs[4])
__global__ void __launch_bounds__(LN_TMAX)
k_nodes_ln_fwd_v(int f, int N, int pool, float eps, const float* __restrict__ y, const float* __restrict__ gamma,
                 const float* __restrict__ beta, const float* __restrict__ keep, float* __restrict__ z,
                 float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  __shared__ float red[16];
  const int row = ln_row(), b = row / f, nv = N / 4;
  const float* yr = y + (int64_t)row * N;
  float4 v[LN_VPT];
  // the affine parameters and the dropout factors of the thread's slots are requested with the row, in front of the two
  // block sums (asked for behind them, their round trip was the tail of every workgroup of a 2560-workgroup latency chain)
  float4 g4[LN_VPT], be4[LN_VPT], k4[LN_VPT];
#pragma unroll
  for (int i = 0; i < LN_VPT; ++i) {
    const int q = threadIdx.x + i * (int)blockDim.x, n = 4 * q;
    v[i] = q < nv ? ld4(yr + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    g4[i] = be4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    k4[i] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (q < nv && n >= pool) {
      g4[i] = ld4(gamma + n);
      be4[i] = ld4(beta + n);
      if (keep) k4[i] = ld4(keep + (int64_t)b * N + n);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_VPT; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = block_sum_all(s, red) / (float)N;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < LN_VPT; ++i) {
    if (threadIdx.x + i * (int)blockDim.x < nv) {
      const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      var += (a * a + bb * bb) + (c * c + d * d);
    }
  }
  var = block_sum_all(var, red) / (float)N;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
  float* zr = z + (int64_t)row * (N - pool);
#pragma unroll
  for (int i = 0; i < LN_VPT; ++i) {
    const int q = threadIdx.x + i * (int)blockDim.x, n = 4 * q;
    if (q < nv && n >= pool) {
      const float4 g = g4[i], be = be4[i];
      float4 o;
      o.x = fmaxf((v[i].x - mean) * rstd * g.x + be.x, 0.f);
      o.y = fmaxf((v[i].y - mean) * rstd * g.y + be.y, 0.f);
      o.z = fmaxf((v[i].z - mean) * rstd * g.z + be.z, 0.f);
      o.w = fmaxf((v[i].w - mean) * rstd * g.w + be.w, 0.f);
      if (keep) {
        const float4 k = k4[i];
        o.x *= k.x; o.y *= k.y; o.z *= k.z; o.w *= k.w;
      }
      *reinterpret_cast<float4*>(zr + (n - pool)) = o;
    }
  }
}

__global__ void __launch_bounds__(LN_TMAX)
k_nodes_ln_bwd_dy_v(int f, int N, int pool, const float* __restrict__ y, const float* __restrict__ gamma,
                    const float* __restrict__ beta, const float* __restrict__ keep, const float* __restrict__ mean,
                    const float* __restrict__ rstd, const float* __restrict__ dz, float* __restrict__ dy) {
  __shared__ float red[16];
  const int row = ln_row(), b = row / f, nv = N / 4;
  const float mu = mean[row], rs = rstd[row];
  const float* yr = y + (int64_t)row * N;
  const float* dzr = dz + (int64_t)row * (N - pool);
  float4 xh[LN_VPT], dx[LN_VPT];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < LN_VPT; ++i) {
    const int q = threadIdx.x + i * (int)blockDim.x, n = 4 * q;
    xh[i] = dx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < nv) {
      const float4 yv = ld4(yr + n), g = ld4(gamma + n);
      xh[i] = make_float4((yv.x - mu) * rs, (yv.y - mu) * rs, (yv.z - mu) * rs, (yv.w - mu) * rs);
      if (n >= pool) {
        const float4 be = ld4(beta + n);
        float4 up = ld4(dzr + (n - pool));
        if (keep) {
          const float4 k = ld4(keep + (int64_t)b * N + n);
          up.x *= k.x; up.y *= k.y; up.z *= k.z; up.w *= k.w;
        }
        dx[i].x = xh[i].x * g.x + be.x > 0.f ? up.x * g.x : 0.f;
        dx[i].y = xh[i].y * g.y + be.y > 0.f ? up.y * g.y : 0.f;
        dx[i].z = xh[i].z * g.z + be.z > 0.f ? up.z * g.z : 0.f;
        dx[i].w = xh[i].w * g.w + be.w > 0.f ? up.w * g.w : 0.f;
      }
      s1 += (dx[i].x + dx[i].y) + (dx[i].z + dx[i].w);
      s2 += (dx[i].x * xh[i].x + dx[i].y * xh[i].y) + (dx[i].z * xh[i].z + dx[i].w * xh[i].w);
    }
  }
  s1 = block_sum_all(s1, red) / (float)N;
  s2 = block_sum_all(s2, red) / (float)N;
  float* dyr = dy + (int64_t)row * N;
#pragma unroll
  for (int i = 0; i < LN_VPT; ++i) {
    const int q = threadIdx.x + i * (int)blockDim.x;
    if (q < nv) {
      float4 o;
      o.x = rs * (dx[i].x - s1 - xh[i].x * s2);
      o.y = rs * (dx[i].y - s1 - xh[i].y * s2);
      o.z = rs * (dx[i].z - s1 - xh[i].z * s2);
      o.w = rs * (dx[i].w - s1 - xh[i].w * s2);
      *reinterpret_cast<float4*>(dyr + 4 * q) = o;
    }
  }
}

static int ln_threads(int N) { return N <= GO_T * 4 * LN_VPT ? GO_T : LN_TMAX; }
static bool ln_vec_ok(int N, int pool, const void* a, const void* b, const void* c, const void* d, const void* e,
                      const void* k) {
  auto al = [](const void* p) { return p == nullptr || ((uintptr_t)p % 16) == 0; };
  return N % 4 == 0 && pool % 4 == 0 && N <= LN_TMAX * 4 * LN_VPT && al(a) && al(b) && al(c) && al(d) && al(e) && al(k);
}

extern "C" int igcn_nodes_ln_fwd(int B, int f, int N, int pool, float eps, const float* y, const float* gamma,
                                 const float* beta, const float* keep, float* z, float* mean, float* rstd,
                                 void* stream) {
  IGCN_REQUIRE(B > 0 && f > 0 && N > 0 && pool >= 0 && pool < N, "nodes_ln_fwd: bad sizes");
  if (ln_vec_ok(N, pool, y, gamma, beta, z, nullptr, keep)) {
    hipLaunchKernelGGL(k_nodes_ln_fwd_v, dim3(B * f), dim3(ln_threads(N)), 0, (hipStream_t)stream, f, N, pool, eps, y, gamma,
                       beta, keep, z, mean, rstd);
    IGCN_CHECK_LAUNCH("nodes_ln_fwd_v");
    return IGCN_OK;
  }
  hipLaunchKernelGGL(k_nodes_ln_fwd, dim3(B * f), dim3(GO_T), 0, (hipStream_t)stream, f, N, pool, eps, y, gamma,
                     beta, keep, z, mean, rstd);
  IGCN_CHECK_LAUNCH("nodes_ln_fwd");
  return IGCN_OK;
}

__global__ void __launch_bounds__(GO_T)
k_nodes_ln_bwd_dy(int f, int N, int pool, const float* __restrict__ y, const float* __restrict__ gamma,
                  const float* __restrict__ beta, const float* __restrict__ keep, const float* __restrict__ mean,
                  const float* __restrict__ rstd, const float* __restrict__ dz, float* __restrict__ dy) {
  __shared__ float red[16];
  const int row = blockIdx.x;
  const int b = row / f;
  const float mu = mean[row], rs = rstd[row];
  const float* yr = y + (int64_t)row * N;
  const int M = N - pool;
  const float* dzr = dz + (int64_t)row * M;
  float s1 = 0.f, s2 = 0.f;
  for (int n = threadIdx.x; n < N; n += GO_T) {
    const float xh = (yr[n] - mu) * rs;
    float up = 0.f;
    if (n >= pool && xh * gamma[n] + beta[n] > 0.f) {
      up = dzr[n - pool];
      if (keep) up *= keep[(int64_t)b * N + n];
    }
    const float dxh = up * gamma[n];
    s1 += dxh;
    s2 += dxh * xh;
  }
  s1 = block_sum_all(s1, red) / (float)N;
  s2 = block_sum_all(s2, red) / (float)N;
  float* dyr = dy + (int64_t)row * N;
  for (int n = threadIdx.x; n < N; n += GO_T) {
    const float xh = (yr[n] - mu) * rs;
    float up = 0.f;
    if (n >= pool && xh * gamma[n] + beta[n] > 0.f) {
      up = dzr[n - pool];
      if (keep) up *= keep[(int64_t)b * N + n];
    }
    dyr[n] = rs * (up * gamma[n] - s1 - xh * s2);
  }
}

// dgamma[n] = sum_rows up*xhat ; dbeta[n] = sum_rows up.  Block = 64 node lanes x 4 row groups over a
// chunk of LN_RC rows; partial[chunk][2][N] is summed over chunks (in order) by k_reduce_rows.
#define LN_RC 64
__global__ void __launch_bounds__(256)
k_nodes_ln_bwd_affine(int rows, int f, int N, int pool, const float* __restrict__ y,
                      const float* __restrict__ gamma, const float* __restrict__ beta,
                      const float* __restrict__ keep, const float* __restrict__ mean,
                      const float* __restrict__ rstd, const float* __restrict__ dz,
                      float* __restrict__ partial) {
  __shared__ float sg[4][64], sb[4][64];
  const int nl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + nl;
  const int r0 = blockIdx.y * LN_RC, r1 = min(rows, r0 + LN_RC);
  float dg = 0.f, db = 0.f;
  if (n < N && n >= pool) {
    const float ga = gamma[n], be = beta[n];
    const int M = N - pool;
#pragma unroll 4
    for (int row = r0 + rg; row < r1; row += 4) {      // unconditional, independent loads: several rows in flight
      const float yv = y[(int64_t)row * N + n];
      float up = dz[(int64_t)row * M + (n - pool)];
      if (keep) up *= keep[(int64_t)(row / f) * N + n];
      const float xh = (yv - mean[row]) * rstd[row];
      up = (xh * ga + be > 0.f) ? up : 0.f;
      dg += up * xh;
      db += up;
    }
  }
  sg[rg][nl] = dg;
  sb[rg][nl] = db;
  __syncthreads();
  if (rg == 0 && n < N) {
    float* prow = partial + (int64_t)blockIdx.y * 2 * N;
    prow[n] = (sg[0][nl] + sg[1][nl]) + (sg[2][nl] + sg[3][nl]);
    prow[N + n] = (sb[0][nl] + sb[1][nl]) + (sb[2][nl] + sb[3][nl]);
  }
}

// 16-bytes-per-lane form (N, pool multiples of 4, aligned rows): thread = (quad of nodes, row lane); one float4 of y,
// dz and keep per row instead of four scalar loads each.  Same chunk layout of the partials.
__device__ __forceinline__ void
ln_bwd_affine_v_body(const int bx, const int by, int rows, int f, int N, int pool, const float* __restrict__ y,
                     const float* __restrict__ gamma, const float* __restrict__ beta,
                     const float* __restrict__ keep, const float* __restrict__ mean,
                     const float* __restrict__ rstd, const float* __restrict__ dz,
                     float* __restrict__ partial) {
  __shared__ float4 sg[16][16], sb[16][16];
  const int nq = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int n = (bx * 16 + nq) * 4;
  const int r0 = by * LN_RC, r1 = min(rows, r0 + LN_RC);
  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
  if (n < N && n >= pool) {
    const float4 ga = ld4(gamma + n), be = ld4(beta + n);
    const int M = N - pool;
#pragma unroll 4
    for (int row = r0 + rg; row < r1; row += 16) {
      const float4 yv = ld4(y + (int64_t)row * N + n);
      float4 up = ld4(dz + (int64_t)row * M + (n - pool));
      if (keep) {
        const float4 k = ld4(keep + (int64_t)(row / f) * N + n);
        up.x *= k.x; up.y *= k.y; up.z *= k.z; up.w *= k.w;
      }
      const float mu = mean[row], rs = rstd[row];
      const float x0 = (yv.x - mu) * rs, x1 = (yv.y - mu) * rs, x2 = (yv.z - mu) * rs, x3 = (yv.w - mu) * rs;
      up.x = (x0 * ga.x + be.x > 0.f) ? up.x : 0.f;
      up.y = (x1 * ga.y + be.y > 0.f) ? up.y : 0.f;
      up.z = (x2 * ga.z + be.z > 0.f) ? up.z : 0.f;
      up.w = (x3 * ga.w + be.w > 0.f) ? up.w : 0.f;
      dg.x += up.x * x0; dg.y += up.y * x1; dg.z += up.z * x2; dg.w += up.w * x3;
      db.x += up.x; db.y += up.y; db.z += up.z; db.w += up.w;
    }
  }
  sg[rg][nq] = dg;
  sb[rg][nq] = db;
  __syncthreads();
  if (rg == 0 && n < N) {
    float4 tg = make_float4(0.f, 0.f, 0.f, 0.f), tb = tg;
#pragma unroll
    for (int r = 0; r < 16; ++r) {                     // row lanes summed in order
      const float4 a = sg[r][nq], c = sb[r][nq];
      tg.x += a.x; tg.y += a.y; tg.z += a.z; tg.w += a.w;
      tb.x += c.x; tb.y += c.y; tb.z += c.z; tb.w += c.w;
    }
    float* prow = partial + (int64_t)by * 2 * N;
    *reinterpret_cast<float4*>(prow + n) = tg;
    *reinterpret_cast<float4*>(prow + N + n) = tb;
  }
}

__global__ void __launch_bounds__(256)
k_nodes_ln_bwd_affine_v(int rows, int f, int N, int pool, const float* __restrict__ y,
                        const float* __restrict__ gamma, const float* __restrict__ beta,
                        const float* __restrict__ keep, const float* __restrict__ mean,
                        const float* __restrict__ rstd, const float* __restrict__ dz,
                        float* __restrict__ partial) {
  ln_bwd_affine_v_body((int)blockIdx.x, (int)blockIdx.y, rows, f, N, pool, y, gamma, beta, keep, mean, rstd, dz, partial);
}

// The affine-gradient passes of SEVERAL LayerNorm layers in one flat grid: d gamma / d beta are parameter gradients —
// nothing in the backward reads them — so the passes of a whole backward can wait for its end and share one launch
// (four grids of 150-750 workgroups each, which otherwise run one after the other between the layers' dX kernels).
#define LN_AFF_MAX 4
struct LnAffProb {
  int rows, f, N, pool, gx, wg0;
  const float *y, *gamma, *beta, *keep, *mean, *rstd, *dz;
  float* partial;
};
struct LnAffGroup { int n; LnAffProb p[LN_AFF_MAX]; };
__global__ void __launch_bounds__(256) k_nodes_ln_bwd_affine_multi(const LnAffGroup G) {
  int pi = 0;
#pragma unroll
  for (int i = 1; i < LN_AFF_MAX; ++i)
    if (i < G.n && (int)blockIdx.x >= G.p[i].wg0) pi = i;
  const LnAffProb& p = G.p[pi];
  const int l = (int)blockIdx.x - p.wg0;
  ln_bwd_affine_v_body(l % p.gx, l / p.gx, p.rows, p.f, p.N, p.pool, p.y, p.gamma, p.beta, p.keep, p.mean, p.rstd, p.dz,
                       p.partial);
}

extern "C" size_t igcn_nodes_ln_bwd_scratch_floats(int B, int f, int N) {
  return (size_t)(igcn_cdiv((int64_t)B * f, LN_RC) * 2 * N + 64);
}

extern "C" int igcn_nodes_ln_bwd(int B, int f, int N, int pool, const float* y, const float* gamma,
                                 const float* beta, const float* keep, const float* mean, const float* rstd,
                                 const float* dz, float* dy, float* dgb /*[2,N]: dgamma then dbeta*/,
                                 float* scratch, void* stream) {
  IGCN_REQUIRE(B > 0 && f > 0 && N > 0 && pool >= 0 && pool < N, "nodes_ln_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (ln_vec_ok(N, pool, y, gamma, beta, dz, dy, keep)) {
    hipLaunchKernelGGL(k_nodes_ln_bwd_dy_v, dim3(B * f), dim3(ln_threads(N)), 0, st, f, N, pool, y, gamma, beta, keep, mean,
                       rstd, dz, dy);
  } else {
    hipLaunchKernelGGL(k_nodes_ln_bwd_dy, dim3(B * f), dim3(GO_T), 0, st, f, N, pool, y, gamma, beta, keep, mean, rstd,
                       dz, dy);
  }
  const int64_t chunks = igcn_cdiv((int64_t)B * f, LN_RC);
  if (ln_vec_ok(N, pool, y, gamma, beta, dz, scratch, keep)) {
    hipLaunchKernelGGL(k_nodes_ln_bwd_affine_v, dim3((unsigned)igcn_cdiv(N, 64), (unsigned)chunks), dim3(256), 0, st,
                       B * f, f, N, pool, y, gamma, beta, keep, mean, rstd, dz, scratch);
  } else {
    hipLaunchKernelGGL(k_nodes_ln_bwd_affine, dim3((unsigned)igcn_cdiv(N, 64), (unsigned)chunks), dim3(256), 0, st,
                       B * f, f, N, pool, y, gamma, beta, keep, mean, rstd, dz, scratch);
  }
  IGCN_CHECK_LAUNCH("nodes_ln_bwd");
  return igcn_launch_reduce_rows_final(scratch, chunks, 2 * (int64_t)N, 2 * N, dgb, st);
}

// The backward in two calls: igcn_nodes_ln_bwd_dy now, and the affine gradients of up to LN_AFF_MAX layers later in one
// launch (igcn_nodes_ln_bwd_affine_multi) — see k_nodes_ln_bwd_affine_multi.
extern "C" int igcn_nodes_ln_bwd_dy(int B, int f, int N, int pool, const float* y, const float* gamma,
                                    const float* beta, const float* keep, const float* mean, const float* rstd,
                                    const float* dz, float* dy, void* stream) {
  IGCN_REQUIRE(B > 0 && f > 0 && N > 0 && pool >= 0 && pool < N, "nodes_ln_bwd_dy: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (ln_vec_ok(N, pool, y, gamma, beta, dz, dy, keep)) {
    hipLaunchKernelGGL(k_nodes_ln_bwd_dy_v, dim3(B * f), dim3(ln_threads(N)), 0, st, f, N, pool, y, gamma, beta, keep, mean,
                       rstd, dz, dy);
  } else {
    hipLaunchKernelGGL(k_nodes_ln_bwd_dy, dim3(B * f), dim3(GO_T), 0, st, f, N, pool, y, gamma, beta, keep, mean, rstd,
                       dz, dy);
  }
  IGCN_CHECK_LAUNCH("nodes_ln_bwd_dy");
  return IGCN_OK;
}

// table [n][13] int64 = {B, f, N, pool, y, gamma, beta, keep, mean, rstd, dz, scratch, dgb} per layer (pointers as
// integers; scratch of igcn_nodes_ln_bwd_scratch_floats, dgb [2,N]).  Layers whose tensors do not allow 16-byte
// accesses get their own launch of the scalar kernel.
extern "C" int igcn_nodes_ln_bwd_affine_multi(int n, const int64_t* table, void* stream) {
  IGCN_REQUIRE(n >= 1 && n <= LN_AFF_MAX && table != nullptr, "nodes_ln_bwd_affine_multi: 1..%d layers", LN_AFF_MAX);
  hipStream_t st = (hipStream_t)stream;
  LnAffGroup G = {};
  int wgs = 0;
  int64_t chunks_of[LN_AFF_MAX];
  bool grouped[LN_AFF_MAX];
  for (int i = 0; i < n; ++i) {
    const int64_t* t = table + 13 * i;
    const int B = (int)t[0], f = (int)t[1], N = (int)t[2], pool = (int)t[3];
    const float *y = (const float*)t[4], *gamma = (const float*)t[5], *beta = (const float*)t[6], *keep = (const float*)t[7];
    const float *mean = (const float*)t[8], *rstd = (const float*)t[9], *dz = (const float*)t[10];
    float* scratch = (float*)t[11];
    IGCN_REQUIRE(B > 0 && f > 0 && N > 0 && pool >= 0 && pool < N && scratch != nullptr,
                 "nodes_ln_bwd_affine_multi: bad layer %d", i);
    const int64_t chunks = igcn_cdiv((int64_t)B * f, LN_RC);
    chunks_of[i] = chunks;
    grouped[i] = ln_vec_ok(N, pool, y, gamma, beta, dz, scratch, keep);
    if (!grouped[i]) {
      hipLaunchKernelGGL(k_nodes_ln_bwd_affine, dim3((unsigned)igcn_cdiv(N, 64), (unsigned)chunks), dim3(256), 0, st,
                         B * f, f, N, pool, y, gamma, beta, keep, mean, rstd, dz, scratch);
      continue;
    }
    LnAffProb& p = G.p[G.n++];
    p.rows = B * f; p.f = f; p.N = N; p.pool = pool;
    p.y = y; p.gamma = gamma; p.beta = beta; p.keep = keep; p.mean = mean; p.rstd = rstd; p.dz = dz; p.partial = scratch;
    p.gx = (int)igcn_cdiv(N, 64);
    p.wg0 = wgs;
    wgs += p.gx * (int)chunks;
  }
  if (G.n > 0) {
    for (int i = G.n; i < LN_AFF_MAX; ++i) G.p[i] = G.p[0];
    hipLaunchKernelGGL(k_nodes_ln_bwd_affine_multi, dim3((unsigned)wgs), dim3(256), 0, st, G);
  }
  IGCN_CHECK_LAUNCH("nodes_ln_bwd_affine_multi");
  for (int i = 0; i < n; ++i) {
    const int64_t* t = table + 13 * i;
    const int N = (int)t[2];
    const int rc = igcn_launch_reduce_rows_final((float*)t[11], chunks_of[i], 2 * (int64_t)N, 2 * N, (float*)t[12], st);
    if (rc) return rc;
  }
  return IGCN_OK;
}

// =================================================================================================
// decoder layer: mean aggregation + zero-padded self term
// =================================================================================================
template <int FIN, int FOUT>
__global__ void __launch_bounds__(GO_T)
k_go_decode_fwd(int Nin, int Nout, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                const float* __restrict__ x, const float* __restrict__ w_out, const float* __restrict__ w_sout,
                float* __restrict__ y) {
  float wo[FOUT][FIN], wso[FOUT][FIN];
#pragma unroll
  for (int c = 0; c < FOUT; ++c)
#pragma unroll
    for (int d = 0; d < FIN; ++d) {
      wo[c][d] = w_out[c * FIN + d];
      wso[c][d] = w_sout[c * FIN + d];
    }
  int bx, b;
  go_block(bx, b);
  const int r = bx * GO_T + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool live = r < Nout;
  const int rr = live ? r : Nout - 1;
  const int off = Nout - Nin;
  const float* xb = x + (int64_t)b * FIN * Nin;
  float acc[FIN];
#pragma unroll
  for (int d = 0; d < FIN; ++d) acc[d] = 0.f;
  const int32_t p0 = live ? row_ptr[rr] : 0, p1 = live ? row_ptr[rr + 1] : 0;     // empty range past the end
  const bool heavy = p1 - p0 > GO_HEAVY;             // hub rows (a parent with many children): whole wave
  // the self term does not depend on the walk: its loads go out first
  float xs[FIN];
#pragma unroll
  for (int d = 0; d < FIN; ++d) xs[d] = 0.f;
  if (live && r >= off) load_node<FIN>(xb, Nin, r - off, xs);
  if (!heavy) {
    for (int32_t e = p0; e < p1; e += 2) {           // two edges per step: indices, then rows, in flight together
      const bool two = e + 1 < p1;
      const int m0 = col[e], m1 = col[two ? e + 1 : e];
      float x0[FIN], x1[FIN];
      load_node<FIN>(xb, Nin, m0, x0);
      load_node<FIN>(xb, Nin, m1, x1);
#pragma unroll
      for (int d = 0; d < FIN; ++d) {
        acc[d] += x0[d];
        if (two) acc[d] += x1[d];
      }
    }
  }
  unsigned long long hmask = __ballot(heavy);
  while (hmask) {
    const int src = __ffsll((long long)hmask) - 1;
    hmask &= hmask - 1;
    const int32_t h0 = __shfl(p0, src, 64), h1 = __shfl(p1, src, 64);
    float part[FIN];
#pragma unroll
    for (int d = 0; d < FIN; ++d) part[d] = 0.f;
    for (int32_t e = h0 + lane; e < h1; e += 64) {
      const int m = col[e];
#pragma unroll
      for (int d = 0; d < FIN; ++d) part[d] += xb[d * Nin + m];
    }
#pragma unroll
    for (int d = 0; d < FIN; ++d) part[d] = wave_sum_all(part[d]);
    if (lane == src) {
#pragma unroll
      for (int d = 0; d < FIN; ++d) acc[d] = part[d];
    }
  }
  if (!live) return;
  const float inv = p1 > p0 ? 1.f / (float)(p1 - p0) : 0.f;
  float out[FOUT];
  transform<FIN, FOUT>(wo, acc, out);
#pragma unroll
  for (int c = 0; c < FOUT; ++c) out[c] *= inv;
  if (r >= off) {
    float o2[FOUT];
    transform<FIN, FOUT>(wso, xs, o2);
#pragma unroll
    for (int c = 0; c < FOUT; ++c) out[c] += o2[c];
  }
  float* yb = y + (int64_t)b * FOUT * Nout;
#pragma unroll
  for (int c = 0; c < FOUT; ++c) yb[c * Nout + r] = out[c];
}

// LDS-staged form.  The GO graph is very sparse (most rows of a decoder layer have no edge at all, a few have up to
// a dozen, one hub has ~100) and identical for every sample, and a sample's [FIN][Nin] slab is 24 KB at most in the
// shapes of the model.  In the global-memory kernel above a wave pays one dependent L2 round trip per edge step of
// its LONGEST row (index, then FIN scalar gathers).  Here a 1024-thread workgroup copies the slab, its rows'
// pointers and their column indices into LDS once (coalesced), and the walk never leaves LDS.
#define GO_DEC_T 1024
#define GO_DEC_ITERS 3
#define GO_DEC_COLCAP 4096
template <int FIN, int FOUT>
__global__ void __launch_bounds__(GO_DEC_T)
k_go_decode_fwd_lds(int Nin, int Nout, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                    const float* __restrict__ x, const float* __restrict__ w_out, const float* __restrict__ w_sout,
                    float* __restrict__ y) {
  extern __shared__ float go_slab[];                   // [FIN][Nin] | rows+1 pointers | GO_DEC_COLCAP indices
  constexpr int ROWS = GO_DEC_T * GO_DEC_ITERS;
  const int total = FIN * Nin;
  int32_t* rp = reinterpret_cast<int32_t*>(go_slab + ((total + 3) & ~3));
  int32_t* cl = rp + ROWS + 4;
  int bx, b;
  go_block(bx, b);
  const int lane = threadIdx.x & 63;
  const int r_base = bx * ROWS, nrows = min(Nout - r_base, ROWS);
  const int32_t e_base = row_ptr[r_base], e_cnt = row_ptr[r_base + nrows] - e_base;
  const bool col_lds = e_cnt <= GO_DEC_COLCAP;         // block-uniform; larger ranges are read from global memory
  const float* xb = x + (int64_t)b * FIN * Nin;
  if ((total & 3) == 0) {                              // slab base = b * total floats: 16-byte aligned with x
    for (int i = threadIdx.x * 4; i < total; i += GO_DEC_T * 4)
      *reinterpret_cast<float4*>(go_slab + i) = *reinterpret_cast<const float4*>(xb + i);
  } else {
    for (int i = threadIdx.x; i < total; i += GO_DEC_T) go_slab[i] = xb[i];
  }
  for (int i = threadIdx.x; i <= nrows; i += GO_DEC_T) rp[i] = row_ptr[r_base + i] - e_base;
  if (col_lds)
    for (int i = threadIdx.x; i < e_cnt; i += GO_DEC_T) cl[i] = col[e_base + i];
  float wo[FOUT][FIN], wso[FOUT][FIN];
#pragma unroll
  for (int c = 0; c < FOUT; ++c)
#pragma unroll
    for (int d = 0; d < FIN; ++d) {
      wo[c][d] = w_out[c * FIN + d];
      wso[c][d] = w_sout[c * FIN + d];
    }
  __syncthreads();
  const int32_t* cg = col + e_base;
  const int off = Nout - Nin;
  float* yb = y + (int64_t)b * FOUT * Nout;
#pragma unroll 1
  for (int it = 0; it < GO_DEC_ITERS; ++it) {
    if (it * GO_DEC_T >= nrows) break;                 // block-uniform
    const int rl = it * GO_DEC_T + threadIdx.x, r = r_base + rl;
    const bool live = rl < nrows;
    float acc[FIN];
#pragma unroll
    for (int d = 0; d < FIN; ++d) acc[d] = 0.f;
    const int32_t p0 = live ? rp[rl] : 0, p1 = live ? rp[rl + 1] : 0;     // lanes past the end walk nothing
    const bool heavy = p1 - p0 > GO_HEAVY;
    if (!heavy) {
      for (int32_t e = p0; e < p1; e += 4) {           // four indices, then their gathers: two LDS round trips
        int m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = (e + i < p1) ? (col_lds ? cl[e + i] : cg[e + i]) : -1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (m[i] >= 0) {
#pragma unroll
            for (int d = 0; d < FIN; ++d) acc[d] += go_slab[d * Nin + m[i]];
          }
      }
    }
    unsigned long long hmask = __ballot(heavy);
    while (hmask) {                                    // hub rows: the whole wave walks the row
      const int src = __ffsll((long long)hmask) - 1;
      hmask &= hmask - 1;
      const int32_t h0 = __shfl(p0, src, 64), h1 = __shfl(p1, src, 64);
      float part[FIN];
#pragma unroll
      for (int d = 0; d < FIN; ++d) part[d] = 0.f;
      for (int32_t e = h0 + lane; e < h1; e += 64) {
        const int mm = col_lds ? cl[e] : cg[e];
#pragma unroll
        for (int d = 0; d < FIN; ++d) part[d] += go_slab[d * Nin + mm];
      }
#pragma unroll
      for (int d = 0; d < FIN; ++d) part[d] = wave_sum_all(part[d]);
      if (lane == src) {
#pragma unroll
        for (int d = 0; d < FIN; ++d) acc[d] = part[d];
      }
    }
    if (live) {
      const float inv = p1 > p0 ? 1.f / (float)(p1 - p0) : 0.f;
      float out[FOUT];
      transform<FIN, FOUT>(wo, acc, out);
#pragma unroll
      for (int c = 0; c < FOUT; ++c) out[c] *= inv;
      if (r >= off) {
        float xs[FIN], o2[FOUT];
#pragma unroll
        for (int d = 0; d < FIN; ++d) xs[d] = go_slab[d * Nin + (r - off)];
        transform<FIN, FOUT>(wso, xs, o2);
#pragma unroll
        for (int c = 0; c < FOUT; ++c) out[c] += o2[c];
      }
#pragma unroll
      for (int c = 0; c < FOUT; ++c) yb[c * Nout + r] = out[c];
    }
  }
}

static size_t go_dec_lds_bytes(int fin, int Nin) {
  return ((((size_t)fin * Nin + 3) & ~(size_t)3) + GO_DEC_T * GO_DEC_ITERS + 4 + GO_DEC_COLCAP) * sizeof(float);
}

extern "C" int igcn_go_decode_fwd(int B, int Nin, int Nout, int fin, int fout, const int32_t* row_ptr,
                                  const int32_t* col, const float* x, const float* w_out, const float* w_sout,
                                  float* y, void* stream) {
  IGCN_REQUIRE(B > 0 && Nin > 0 && Nout >= Nin, "go_decode_fwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = go_dec_lds_bytes(fin, Nin);
  if (!go_attn_force_cm() && lds <= 64 * 1024 && ((uintptr_t)x % 16) == 0) {
    dim3 lgrid((unsigned)igcn_cdiv(Nout, GO_DEC_T * GO_DEC_ITERS), B);
#define CALL(FI, FO)                                                                                                \
  hipLaunchKernelGGL((k_go_decode_fwd_lds<FI, FO>), lgrid, dim3(GO_DEC_T), lds, st, Nin, Nout, row_ptr, col, x, w_out, \
                     w_sout, y)
    GO_DISPATCH(fin, fout, CALL)
#undef CALL
    IGCN_CHECK_LAUNCH("go_decode_fwd_lds");
    return IGCN_OK;
  }
  dim3 grid((unsigned)igcn_cdiv(Nout, GO_T), B);
#define CALL(FI, FO) \
  hipLaunchKernelGGL((k_go_decode_fwd<FI, FO>), grid, dim3(GO_T), 0, st, Nin, Nout, row_ptr, col, x, w_out, w_sout, y)
  GO_DISPATCH(fin, fout, CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("go_decode_fwd");
  return IGCN_OK;
}

template <int FIN, int FOUT>
__global__ void __launch_bounds__(GO_T)
k_go_decode_bwd(int B, int Nin, int Nout, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ t_ptr,
                const int32_t* __restrict__ t_row, const float* __restrict__ x, const float* __restrict__ w_out,
                const float* __restrict__ w_sout, const float* __restrict__ dy, float* __restrict__ dx,
                float* __restrict__ partial) {
  constexpr int NW = 2 * FOUT * FIN;
  __shared__ float red[(GO_T / 64) * NW];
  float wo[FOUT][FIN], wso[FOUT][FIN];
#pragma unroll
  for (int c = 0; c < FOUT; ++c)
#pragma unroll
    for (int d = 0; d < FIN; ++d) {
      wo[c][d] = w_out[c * FIN + d];
      wso[c][d] = w_sout[c * FIN + d];
    }
  int bx, by;
  go_block(bx, by);                                   // (a sample's node blocks on one XCD: its dy rows are gathered)
  const int m = bx * GO_T + threadIdx.x;
  const int off = Nout - Nin;
  float gw[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) gw[j] = 0.f;
  if (m < Nin) {
    const int32_t c0 = t_ptr[m], c1 = t_ptr[m + 1];
    const int b_end = min(B, (by + 1) * GO_SB);
    for (int b = by * GO_SB; b < b_end; ++b) {
      const float* dyb = dy + (int64_t)b * FOUT * Nout;
      float G[FOUT], Gs[FOUT], xr[FIN];
#pragma unroll
      for (int c = 0; c < FOUT; ++c) G[c] = 0.f;
      load_node<FOUT>(dyb, Nout, m + off, Gs);          // independent of the walk: issued first
      load_node<FIN>(x + (int64_t)b * FIN * Nin, Nin, m, xr);
      for (int32_t e = c0; e < c1; e += 2) {              // two edges per step
        const bool two = e + 1 < c1;
        const int ra = t_row[e], rb = t_row[two ? e + 1 : e];
        const int32_t da = row_ptr[ra + 1] - row_ptr[ra], db = row_ptr[rb + 1] - row_ptr[rb];
        float ya[FOUT], yb2[FOUT];
        load_node<FOUT>(dyb, Nout, ra, ya);
        load_node<FOUT>(dyb, Nout, rb, yb2);
        const float ia = 1.f / (float)da, ib = 1.f / (float)db;
#pragma unroll
        for (int c = 0; c < FOUT; ++c) {
          G[c] += ya[c] * ia;
          if (two) G[c] += yb2[c] * ib;
        }
      }
      float* dxb = dx + (int64_t)b * FIN * Nin;
#pragma unroll
      for (int d = 0; d < FIN; ++d) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < FOUT; ++c) t += wo[c][d] * G[c] + wso[c][d] * Gs[c];
        dxb[d * Nin + m] = t;
      }
#pragma unroll
      for (int c = 0; c < FOUT; ++c)
#pragma unroll
        for (int d = 0; d < FIN; ++d) {
          gw[c * FIN + d] += G[c] * xr[d];
          gw[FOUT * FIN + c * FIN + d] += Gs[c] * xr[d];
        }
    }
  }
  block_reduce_vec<NW>(gw, red, partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * NW);
}

// LDS-resident form, one 1024-thread workgroup per sample: the sample's dy slab [FOUT][Nout] and the rows' inverse
// degrees live in LDS, so the walk over a node's readers gathers from there; the parameter-gradient products
// (G | G_self) (x) x go through LDS to the matrix cores once per 1024 nodes and leave the workgroup as ONE partial per
// sample (the thread-per-node kernel above reduces 2*FOUT*FIN values per 256 nodes and writes 5x as many partials).
#define GO_DBL_T 1024
template <int FIN, int FOUT, int T>
__global__ void __launch_bounds__(T)
k_go_decode_bwd_lds(int Nin, int Nout, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ t_ptr,
                    const int32_t* __restrict__ t_row, const float* __restrict__ x, const float* __restrict__ w_out,
                    const float* __restrict__ w_sout, const float* __restrict__ dy, float* __restrict__ dx,
                    float* __restrict__ partial, const LnFuse L) {
  extern __shared__ float go_dbl[];
  constexpr int ROWS = 2 * FOUT, TP = T + 4, NW = 2 * FOUT * FIN;
  const int NPo = (Nout + 3) & ~3;
  float* dys = go_dbl;                                  // [FOUT][NPo]
  float* inv = dys + FOUT * NPo;                        // [NPo]  1 / (row degree)
  float* us = inv + NPo;                                // [ROWS][TP]
  float* xt = us + ROWS * TP;                           // [FIN][TP]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const float* dyb = dy + (int64_t)b * FOUT * Nout;
  if (L.y) {                                            // block-uniform: dy formed here (ln_bwd_into_lds; NPo == Nout)
    ln_bwd_into_lds<FOUT, T>(L, b, Nout, dys, us);
  } else if (NPo == Nout && ((uintptr_t)dyb & 15) == 0) {
    for (int i = tid * 4; i < FOUT * Nout; i += T * 4)
      *reinterpret_cast<float4*>(dys + i) = *reinterpret_cast<const float4*>(dyb + i);
  } else {
    for (int c = 0; c < FOUT; ++c)
      for (int r = tid; r < Nout; r += T) dys[c * NPo + r] = dyb[c * Nout + r];
  }
  for (int r = tid; r < Nout; r += T) {
    const int32_t d = row_ptr[r + 1] - row_ptr[r];
    inv[r] = d > 0 ? 1.f / (float)d : 0.f;
  }
  float wo[FOUT][FIN], wso[FOUT][FIN];
#pragma unroll
  for (int c = 0; c < FOUT; ++c)
#pragma unroll
    for (int d = 0; d < FIN; ++d) {
      wo[c][d] = w_out[c * FIN + d];
      wso[c][d] = w_sout[c * FIN + d];
    }
  __syncthreads();
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int w = tid >> 6, mm = lane & 15, g4 = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float gv[32];
#pragma unroll
  for (int e = 0; e < 32; ++e) gv[e] = 0.f;
  const int off = Nout - Nin;
  const float* xb = x + (int64_t)b * FIN * Nin;
  float* dxb = dx + (int64_t)b * FIN * Nin;
#pragma unroll 1
  for (int base = 0; base < Nin; base += T) {    // block-uniform
    const int m = base + tid;
    float G[FOUT], Gs[FOUT], xr[FIN];
#pragma unroll
    for (int c = 0; c < FOUT; ++c) G[c] = Gs[c] = 0.f;
#pragma unroll
    for (int d = 0; d < FIN; ++d) xr[d] = 0.f;
    if (m < Nin) {
      const int32_t c0 = t_ptr[m], c1 = t_ptr[m + 1];
      load_node<FIN>(xb, Nin, m, xr);
      load_node<FOUT>(dys, NPo, m + off, Gs);
      for (int32_t e = c0; e < c1; e += 2) {            // two reader indices in flight, then LDS gathers
        const bool two = e + 1 < c1;
        const int ra = t_row[e], rb = t_row[two ? e + 1 : e];
        float ya[FOUT], yb2[FOUT];
        load_node<FOUT>(dys, NPo, ra, ya);
        load_node<FOUT>(dys, NPo, rb, yb2);
        const float ia = inv[ra], ib = inv[rb];
#pragma unroll
        for (int c = 0; c < FOUT; ++c) {
          G[c] += ya[c] * ia;
          if (two) G[c] += yb2[c] * ib;
        }
      }
#pragma unroll
      for (int d = 0; d < FIN; ++d) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < FOUT; ++c) t += wo[c][d] * G[c] + wso[c][d] * Gs[c];
        dxb[d * Nin + m] = t;
      }
    }
    if constexpr (NW <= 32) {                           // few products: per-thread sums, one butterfly at the end
#pragma unroll
      for (int c = 0; c < FOUT; ++c)
#pragma unroll
        for (int d = 0; d < FIN; ++d) {
          gv[c * FIN + d] += G[c] * xr[d];
          gv[(FOUT + c) * FIN + d] += Gs[c] * xr[d];
        }
      continue;
    }
#pragma unroll
    for (int c = 0; c < FOUT; ++c) {
      us[c * TP + tid] = G[c];
      us[(FOUT + c) * TP + tid] = Gs[c];
    }
#pragma unroll
    for (int d = 0; d < FIN; ++d) xt[d * TP + tid] = xr[d];
    // the staging columns [64 w, 64 w + 64) belong to wave w alone (written and read by it, LDS serves a wave's
    // accesses in order): no workgroup barrier inside the loop
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float* ua = us + (mm < ROWS ? mm : 0) * TP + 64 * w + g4;
    const float* xa = xt + (mm < FIN ? mm : 0) * TP + 64 * w + g4;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float a = (mm < ROWS) ? ua[4 * c] : 0.f;
      const float bq = (mm < FIN) ? xa[4 * c] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, acc, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (NW <= 32) {
    go_butterfly<32, 16, 32>(gv, lane);                 // total of value f in lane 2 f
    float* wpart = us;                                  // [16 waves][32]
    if ((lane & 1) == 0) wpart[w * 32 + (lane >> 1)] = gv[0];
    __syncthreads();
    if (tid < NW) {
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < T / 64; ++ww) t += wpart[ww * 32 + tid];
      partial[(int64_t)b * NW + tid] = t;
    }
    return;
  }
  __syncthreads();                                      // every wave is done with the staging area
  float* wsum = us;                                     // [16 waves * 4][64] <= ROWS * TP floats (ROWS >= 4)
#pragma unroll
  for (int r = 0; r < 4; ++r) wsum[(w * 4 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (w == 0 && mm < FIN) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * g4 + r < ROWS) {
        float t = 0.f;
#pragma unroll
        for (int ww = 0; ww < T / 64; ++ww) t += wsum[(ww * 4 + r) * 64 + lane];
        partial[(int64_t)b * NW + (4 * g4 + r) * FIN + mm] = t;
      }
  }
}

static size_t go_dbl_lds_bytes(int Nout, int fin, int fout, int T) {
  const size_t npo = ((size_t)Nout + 3) & ~(size_t)3;
  // few parameter-gradient products (VALU + butterfly path): one row of 32 totals per wave instead of the MFMA staging
  const size_t stage = 2 * fout * fin <= 32 ? (size_t)(T / 64) * 32 : (size_t)(2 * fout + fin) * (T + 4);
  return ((size_t)(fout + 1) * npo + stage) * sizeof(float);
}
// 512 threads when the layer's INPUT nodes fit one pass of them and two workgroups then share a CU (the 400 -> 1200
// layer: 60 KB instead of 90 KB, and no 624 idle threads), 1024 otherwise
static int go_dbl_threads(int Nin, int Nout, int fin, int fout) {
  return (Nin <= 512 && go_dbl_lds_bytes(Nout, fin, fout, 512) <= 80 * 1024) ? 512 : GO_DBL_T;
}

extern "C" size_t igcn_go_decode_bwd_scratch_floats(int B, int Nin, int fin, int fout) {
  return (size_t)(igcn_cdiv(Nin, GO_T) * igcn_cdiv(B, GO_SB) * 2 * fout * fin + 64);
}

static bool go_decode_bwd_in_lds(int Nin, int Nout, int fin, int fout) {
  const int TT = go_dbl_threads(Nin, Nout, fin, fout);
  return !go_attn_force_cm() && go_dbl_lds_bytes(Nout, fin, fout, TT) <= 160 * 1024 && 2 * fout >= 4;
}
// 1 when igcn_go_decode_ln_bwd runs for these sizes (see igcn_go_attn_ln_fused_ok; the decoder's LayerNorm has no pooling)
extern "C" int igcn_go_decode_ln_fused_ok(int Nin, int Nout, int fin, int fout) {
  if (Nin <= 0 || Nout < Nin || Nout % 4) return 0;
  return go_decode_bwd_in_lds(Nin, Nout, fin, fout) && Nout / 4 <= go_dbl_threads(Nin, Nout, fin, fout) ? 1 : 0;
}

static int go_decode_bwd_impl(int B, int Nin, int Nout, int fin, int fout, const int32_t* row_ptr, const int32_t* t_ptr,
                              const int32_t* t_row, const float* x, const float* w_out, const float* w_sout,
                              const float* dy, float* dx, float* dparams, float* scratch, const LnFuse& L,
                              hipStream_t st) {
  const int nw = 2 * fout * fin;
  const int TT = go_dbl_threads(Nin, Nout, fin, fout);
  const size_t lds = go_dbl_lds_bytes(Nout, fin, fout, TT);
  // one workgroup per sample fills the chip from ~128 samples on; below that (configs[4]: 64) the thread-per-node
  // kernel — 2 500 workgroups, a sample's blocks on one XCD — is the faster one (12.1 against 14.4 us at 64 x 10 000)
  if (go_decode_bwd_in_lds(Nin, Nout, fin, fout) && (B >= 128 || L.y != nullptr)) {
#define CALLT(FI, FO, TV)                                                                                        \
  {                                                                                                               \
    IGCN_ALLOW_BIG_LDS((k_go_decode_bwd_lds<FI, FO, TV>));                                                        \
    hipLaunchKernelGGL((k_go_decode_bwd_lds<FI, FO, TV>), dim3(B), dim3(TV), lds, st, Nin, Nout, row_ptr, t_ptr,  \
                       t_row, x, w_out, w_sout, dy, dx, scratch, L);                                              \
  }
#define CALL(FI, FO) \
  if (TT == 512) CALLT(FI, FO, 512) else CALLT(FI, FO, 1024)
    GO_DISPATCH(fin, fout, CALL)
#undef CALL
#undef CALLT
    IGCN_CHECK_LAUNCH("go_decode_bwd(lds)");
    return igcn_launch_reduce_rows_final(scratch, B, nw, nw, dparams, st);
  }
  IGCN_REQUIRE(L.y == nullptr, "go_decode_ln_bwd: this layer does not run LDS-resident (igcn_go_decode_ln_fused_ok)");
  dim3 grid((unsigned)igcn_cdiv(Nin, GO_T), (unsigned)igcn_cdiv(B, GO_SB));
#define CALL(FI, FO)                                                                                             \
  hipLaunchKernelGGL((k_go_decode_bwd<FI, FO>), grid, dim3(GO_T), 0, st, B, Nin, Nout, row_ptr, t_ptr, t_row, x, \
                     w_out, w_sout, dy, dx, scratch)
  GO_DISPATCH(fin, fout, CALL)
#undef CALL
  IGCN_CHECK_LAUNCH("go_decode_bwd");
  return igcn_launch_reduce_rows_final(scratch, (int64_t)grid.x * grid.y, nw, nw, dparams, st);
}

extern "C" int igcn_go_decode_bwd(int B, int Nin, int Nout, int fin, int fout, const int32_t* row_ptr,
                                  const int32_t* t_ptr, const int32_t* t_row, const float* x, const float* w_out,
                                  const float* w_sout, const float* dy, float* dx, float* dparams, float* scratch,
                                  void* stream) {
  IGCN_REQUIRE(B > 0 && Nin > 0 && Nout >= Nin, "go_decode_bwd: bad sizes");
  const LnFuse none = {};
  return go_decode_bwd_impl(B, Nin, Nout, fin, fout, row_ptr, t_ptr, t_row, x, w_out, w_sout, dy, dx, dparams, scratch,
                            none, (hipStream_t)stream);
}

// GO decoder layer backward WITH the backward of the LayerNorm block behind it (go_model.py:262-275; pool = 0): dz is
// the gradient of z = dropout(relu(LN(y))), y [B, fout, Nout] the layer's own output; dgb [2, Nout], part:
// igcn_go_ln_part_floats(B, Nout) floats (alive until the deferred reductions ran).
extern "C" int igcn_go_decode_ln_bwd(int B, int Nin, int Nout, int fin, int fout, const int32_t* row_ptr,
                                     const int32_t* t_ptr, const int32_t* t_row, const float* x, const float* w_out,
                                     const float* w_sout, const float* y, const float* gamma, const float* beta,
                                     const float* keep, const float* mean, const float* rstd, const float* dz,
                                     float* dx, float* dparams, float* dgb, float* scratch, float* part, void* stream) {
  IGCN_REQUIRE(B > 0 && Nin > 0 && Nout >= Nin, "go_decode_ln_bwd: bad sizes");
  IGCN_REQUIRE(igcn_go_decode_ln_fused_ok(Nin, Nout, fin, fout), "go_decode_ln_bwd: sizes not supported (igcn_go_decode_ln_fused_ok)");
  IGCN_REQUIRE(y && gamma && beta && mean && rstd && dz && dgb && part, "go_decode_ln_bwd: null operand");
  IGCN_REQUIRE(go_ln_al16(y) && go_ln_al16(dz) && go_ln_al16(gamma) && go_ln_al16(beta) && go_ln_al16(keep) &&
               go_ln_al16(part), "go_decode_ln_bwd: operands must be 16-byte aligned");
  const LnFuse L = {y, dz, gamma, beta, keep, mean, rstd, part, 0, nullptr, nullptr, dgb};
  const int rc = go_decode_bwd_impl(B, Nin, Nout, fin, fout, row_ptr, t_ptr, t_row, x, w_out, w_sout, nullptr, dx,
                                    dparams, scratch, L, (hipStream_t)stream);
  if (rc) return rc;
  return igcn_launch_reduce_rows_final(part, B, 2 * (int64_t)Nout, 2 * Nout, dgb, (hipStream_t)stream);
}
