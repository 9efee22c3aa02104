// Shared by the bf16-operand attention kernels (attn_bf16.hip: operands rounded once; attn_split.hip: operands split into
// a bf16 head and a bf16 remainder): operand types, the 16x16x32 matrix instruction and its LDS fragment reads.
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define AB_HD 16
#define AB_MAX_WAVES 8
#define AB_KEY_CHUNK 448          // keys per chunk of the forward / dQ kernels (31 / 46 KB of LDS: several workgroups per CU)
#define AB_QUERY_CHUNK 256        // queries per chunk of the dK | dV kernel (35 KB)

// (sample, head) item of a workgroup.  Grid x = B * H items, y = row blocks; the hardware deals workgroup i (x fastest) to
// XCD i mod 8, and the H heads of a sample read the two halves of the SAME 128-byte lines of q / k | v / do (a [..,
// H * head_dim] row per token): with item = blockIdx.x the heads of a sample sat on different XCDs and every line came from
// HBM once per head (k_attn_bf16_fwd read 51.8 MB for 25.5 MB of operands, PMC).  Here XCD c takes the items
// [c n/8, (c+1) n/8): a sample's heads — and, the grid's x extent being a multiple of 8, all row blocks of an item —
// share one L2.  (csrc/attn_mfma.hip does the same through am_per_xcd.)
__device__ __forceinline__ int ab_item() {
  const int n = (int)gridDim.x, x = (int)blockIdx.x;
  return (n & 7) ? x : (x & 7) * (n >> 3) + (x >> 3);
}

__device__ __forceinline__ f32x4 mfma32(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// row length of a transposed tile: >= rows, and = 16 (mod 128) elements, i.e. 32 bytes (mod 256): the sixteen
// head-dim rows an operand read touches then start 8 banks apart
__host__ __device__ inline int ab_ldt(int rows) { return ((rows + 111) / 128) * 128 + 16; }

__device__ __forceinline__ bf16x4 ab_cvt4(const float4 v) {
  return bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
}

__device__ __forceinline__ bf16x8 ab_pack(const float a[4], const float b[4]) {
  return bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
}

// eight consecutive floats of a row -> the resident side's operand half (lanes g < 2); lanes g >= 2 hold zeros
__device__ __forceinline__ bf16x8 ab_row8(const float* __restrict__ row, bool live, float scale) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)0.f;
  if (live) {
    const float4 a = *reinterpret_cast<const float4*>(row), b = *reinterpret_cast<const float4*>(row + 4);
    r = bf16x8{(__bf16)(a.x * scale), (__bf16)(a.y * scale), (__bf16)(a.z * scale), (__bf16)(a.w * scale),
               (__bf16)(b.x * scale), (__bf16)(b.y * scale), (__bf16)(b.z * scale), (__bf16)(b.w * scale)};
  }
  return r;
}

// the transposed operand of a 32-row step: slots 0..3 = rows 4 g .. 4 g + 3, slots 4..7 = rows 16 + 4 g .. of the step
__device__ __forceinline__ bf16x8 ab_tfrag(const __bf16* __restrict__ t, int ldt, int n, int g, int step) {
  const __bf16* p = t + n * ldt + step * 32 + 4 * g;
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(p), b = *reinterpret_cast<const bf16x4*>(p + 16);
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// row-major operand of tile `tile` (16 rows): lane (g, n) reads columns 8 (g & 1) .. + 7 of row n; lanes g >= 2 read
// the same bytes again — their products meet the resident side's zeros
__device__ __forceinline__ bf16x8 ab_rfrag(const __bf16* __restrict__ r, int tile, int n, int g) {
  return *reinterpret_cast<const bf16x8*>(r + (tile * 16 + n) * AB_HD + 8 * (g & 1));
}

