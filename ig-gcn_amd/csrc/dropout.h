// Dropout masks (see misc.hip: igcn_dropout_masks) — the definitions shared with the launches the mask generation can
// RIDE in (plan.hip: the per-graph plan build; igcn_rider_dropout).
#pragma once
#include "common.h"

#define DM_MAXSEG 32
struct DropSegs {
  int64_t end[DM_MAXSEG];
  float p[DM_MAXSEG];
  int n;
};
// int64 device counters the launch bumps by `inc` (BatchNorm's num_batches_tracked of the model's five BatchNorms: the
// masks are drawn once per training forward, which is exactly when those counters advance — a torch._foreach_add_
// launch less per step)
#define DM_MAXCNT 8
struct DropCounters {
  long long* c[DM_MAXCNT];
  int n;
  long long inc;
};

__device__ __forceinline__ uint32_t dm_hash(uint32_t x) {       // lowbias32
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

// `state` (device uint64[IGCN_DROPOUT_STATE_WORDS]): [0] stream counter, [1] groups done, [2 + 32 g] workgroups done of
// group g (DM_GROUPS groups, their words 256 bytes apart).  "Last one out advances the counter" in two levels: a
// single word takes ~90 atomics per microsecond, so 6000 one-shot workgroups spent 68 us there and the first remedy —
// at most 512 grid-striding workgroups — left two waves per SIMD to do the hashing (19 us for 18 MB of factors).
#define DM_GROUPS 16
#define DM_WORDS (2 + 32 * DM_GROUPS)
// 256-thread block `blk` of `nblk` (a launch of its own, or the tail of another launch's grid).  `tid` = the thread's
// index inside that block: threadIdx.x, or — in a carrier whose workgroups are 512 threads wide — threadIdx.x & 255 with
// blk = 2 * workgroup + (threadIdx.x >> 8), so that no half of a carrier workgroup idles (a block index >= nblk takes
// part in the barrier and does nothing else).
__device__ __forceinline__ void dropout_masks_body(unsigned blk, unsigned nblk, int64_t total, const DropSegs& segs,
                                                   unsigned long long* __restrict__ state, float* __restrict__ out,
                                                   const DropCounters& cnt, unsigned tid = threadIdx.x) {
  const bool live = blk < nblk;
  if (blk == 0 && (int)tid < cnt.n) *cnt.c[tid] += cnt.inc;
  const unsigned long long c = state[0];
  const uint32_t k0 = dm_hash((uint32_t)c ^ 0x9E3779B9u), k1 = dm_hash((uint32_t)(c >> 32) + 0x85EBCA6Bu + k0);
  for (int64_t i0 = live ? ((int64_t)blk * 256 + tid) * 4 : total; i0 < total; i0 += (int64_t)nblk * 1024) {
    float v[4];
    float p = 0.f;                                     // a quad never straddles sites (sites start on multiples of 4)
    for (int sgm = 0; sgm < segs.n; ++sgm)
      if (i0 < segs.end[sgm]) { p = segs.p[sgm]; break; }
    const float scale = 1.0f / (1.0f - p);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = i0 + j;
      const uint32_t h = dm_hash(((uint32_t)i * 0x9E3779B1u) ^ k0) + (uint32_t)(i >> 32) * 0x85EBCA77u;
      const float u = (float)(dm_hash(h ^ k1) >> 8) * (1.0f / 16777216.0f);             // [0, 1)
      v[j] = u < p ? 0.f : scale;
    }
    if (i0 + 3 < total) {
      *reinterpret_cast<float4*>(out + i0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int j = 0; j < 4 && i0 + j < total; ++j) out[i0 + j] = v[j];
    }
  }
  __syncthreads();
  if (tid == 0 && live) {
    // every workgroup has read the counter before it arrives here; group g = blockIdx % DM_GROUPS has
    // ceil((nblk - g) / DM_GROUPS) members
    const unsigned g = blk % DM_GROUPS, ng = nblk < DM_GROUPS ? nblk : DM_GROUPS;
    const unsigned members = (nblk - g + DM_GROUPS - 1) / DM_GROUPS;
    unsigned long long* gw = state + 2 + 32 * g;
    if (atomicAdd(gw, 1ull) == (unsigned long long)members - 1) {
      *gw = 0;
      if (atomicAdd(&state[1], 1ull) == (unsigned long long)ng - 1) {
        state[0] = c + 1;
        state[1] = 0;
      }
    }
  }
}


// one mask-generation job as the host hands it over (igcn_dropout_masks / igcn_rider_dropout)
struct DropJob {
  int64_t total;
  DropSegs sg;
  unsigned long long* state;
  float* out;
  DropCounters cnt;
  unsigned blocks;
};
int igcn_dropout_launch(const DropJob& job, hipStream_t st);
int igcn_dropout_job(DropJob& job, const char* who, int64_t total, int n_segments, const int64_t* seg_end, const float* seg_p,
                     void* state, float* out, int n_counters, int64_t* const* counters, int64_t counter_inc);
