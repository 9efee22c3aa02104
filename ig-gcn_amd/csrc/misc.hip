// Optimiser step over the flat parameter buffer (include/igcn.h: igcn_adam_step).
#include "common.h"

__global__ void k_adam_tick(int32_t* step) { *step += 1; }

// torch.optim.Adam (amsgrad=False, weight_decay=0, maximize=False), single-tensor formulation:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void k_adam(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, const int32_t* __restrict__ step, const float* __restrict__ lr_dev,
                       float b1, float b2, float eps, float gscale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float t = (float)(*step);
  const float lr = *lr_dev;                         // device scalar: a schedule reaches the captured launch
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  const float gi = g[i] * gscale;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[i] -= (lr / bc1) * (mi / denom);
}

static int adam_step_impl(bool tick, int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                          int32_t* step, const float* lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (tick) hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, st, step);
  if (n > 0)
    hipLaunchKernelGGL(k_adam, dim3((unsigned)igcn_cdiv(n, 256)), dim3(256), 0, st, n, param, grad, exp_avg,
                       exp_avg_sq, step, lr, beta1, beta2, eps, grad_scale);
  IGCN_CHECK_LAUNCH("adam_step");
  return IGCN_OK;
}
extern "C" int igcn_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                              int32_t* step, const float* lr, float beta1, float beta2, float eps, float grad_scale,
                              void* stream) {
  return adam_step_impl(true, n, param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, grad_scale, stream);
}
// *step has been advanced already (igcn_reduce_flush_tick): no counter launch in front
extern "C" int igcn_adam_step_ticked(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                     int32_t* step, const float* lr, float beta1, float beta2, float eps,
                                     float grad_scale, void* stream) {
  return adam_step_impl(false, n, param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, grad_scale, stream);
}

// ---- multi-tensor variants: one launch walks a device table of tensors --------------------------------
// table[t] = {param, grad, exp_avg, exp_avg_sq} (device pointers as int64), numel[t].  grad == 0: skipped
// (torch.optim.Adam skips parameters whose .grad is None).  The table lets autograd hand over freshly
// written gradient tensors (no AccumulateGrad add into a pre-zeroed flat buffer, no zero_grad memset).
__global__ void __launch_bounds__(256)
k_adam_multi(const int64_t* __restrict__ table, const int64_t* __restrict__ numel,
             const int32_t* __restrict__ step, const float* __restrict__ lr_dev, float b1, float b2, float eps,
             float gscale) {
  const int t = blockIdx.y;
  const int64_t n = numel[t];
  if ((int64_t)blockIdx.x * 256 >= n) return;       // most tensors are tiny: their surplus workgroups leave at once
  const float* g = reinterpret_cast<const float*>(table[4 * t + 1]);
  if (g == nullptr) return;
  float* p = reinterpret_cast<float*>(table[4 * t]);
  float* m = reinterpret_cast<float*>(table[4 * t + 2]);
  float* v = reinterpret_cast<float*>(table[4 * t + 3]);
  const float ts = (float)(*step);
  const float lr = *lr_dev;
  const float bc1 = 1.f - powf(b1, ts), bc2s = sqrtf(1.f - powf(b2, ts));
  auto upd = [&](float gi, float& mi, float& vi, float& pi) {
    gi *= gscale;
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    pi -= (lr / bc1) * (mi / (sqrtf(vi) / bc2s + eps));
  };
  // 16 bytes per lane on all four streams when the tensors allow it (the big matrices: two passes instead of eight)
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  const int64_t nv = vec ? n / 4 : 0;
#pragma unroll 2
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    const float4 g4 = reinterpret_cast<const float4*>(g)[i];
    float4 m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i];
    float4 p4 = reinterpret_cast<float4*>(p)[i];
    upd(g4.x, m4.x, v4.x, p4.x);
    upd(g4.y, m4.y, v4.y, p4.y);
    upd(g4.z, m4.z, v4.z, p4.z);
    upd(g4.w, m4.w, v4.w, p4.w);
    reinterpret_cast<float4*>(m)[i] = m4;
    reinterpret_cast<float4*>(v)[i] = v4;
    reinterpret_cast<float4*>(p)[i] = p4;
  }
#pragma unroll 4
  for (int64_t i = nv * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float mi = m[i], vi = v[i], pi = p[i];
    upd(g[i], mi, vi, pi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
}

// The same update over a precomputed block list: block b covers elements [blk_off[b], blk_off[b] + ADAM_CHUNK) of tensor
// blk_tensor[b].  The (96, n_tensors) grid above launches 7 680 workgroups for the ~80 tensors of the model, of which
// ~7 400 find nothing to do (most tensors are a few hundred floats): their dispatch was most of the kernel's 10 us.
#define ADAM_CHUNK 1024
__global__ void __launch_bounds__(256)
k_adam_blocks(const int64_t* __restrict__ table, const int64_t* __restrict__ numel, const int32_t* __restrict__ blk_tensor,
              const int32_t* __restrict__ blk_off, const int32_t* __restrict__ step, const float* __restrict__ lr_dev,
              float b1, float b2, float eps, float gscale) {
  const int t = blk_tensor[blockIdx.x];
  const float* g = reinterpret_cast<const float*>(table[4 * t + 1]);
  if (g == nullptr) return;                           // (no gradient: torch's Adam skips the parameter)
  float* p = reinterpret_cast<float*>(table[4 * t]);
  float* m = reinterpret_cast<float*>(table[4 * t + 2]);
  float* v = reinterpret_cast<float*>(table[4 * t + 3]);
  const int64_t n = numel[t], lo = blk_off[blockIdx.x], hi = lo + ADAM_CHUNK < n ? lo + ADAM_CHUNK : n;
  const float ts = (float)(*step);
  const float lr = *lr_dev;
  const float bc1 = 1.f - powf(b1, ts), bc2s = sqrtf(1.f - powf(b2, ts));
  auto upd = [&](float gi, float& mi, float& vi, float& pi) {
    gi *= gscale;
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    pi -= (lr / bc1) * (mi / (sqrtf(vi) / bc2s + eps));
  };
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);   // (lo is a multiple of 4)
  int64_t i = lo + 4 * (int64_t)threadIdx.x;
  if (vec) {
    for (; i + 4 <= hi; i += 1024) {
      const float4 g4 = *reinterpret_cast<const float4*>(g + i);
      float4 m4 = *reinterpret_cast<float4*>(m + i), v4 = *reinterpret_cast<float4*>(v + i);
      float4 p4 = *reinterpret_cast<float4*>(p + i);
      upd(g4.x, m4.x, v4.x, p4.x);
      upd(g4.y, m4.y, v4.y, p4.y);
      upd(g4.z, m4.z, v4.z, p4.z);
      upd(g4.w, m4.w, v4.w, p4.w);
      *reinterpret_cast<float4*>(m + i) = m4;
      *reinterpret_cast<float4*>(v + i) = v4;
      *reinterpret_cast<float4*>(p + i) = p4;
    }
    if (i < hi && i + 4 > hi) {                       // the ragged last quad of the tensor
      for (int64_t j = i; j < hi; ++j) {
        float mi = m[j], vi = v[j], pi = p[j];
        upd(g[j], mi, vi, pi);
        m[j] = mi; v[j] = vi; p[j] = pi;
      }
    }
  } else {
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) {
      float mi = m[j], vi = v[j], pi = p[j];
      upd(g[j], mi, vi, pi);
      m[j] = mi; v[j] = vi; p[j] = pi;
    }
  }
}

extern "C" int igcn_adam_chunk(void) { return ADAM_CHUNK; }

extern "C" int igcn_adam_step_blocks(int n_blocks, const int64_t* table, const int64_t* numel, const int32_t* blk_tensor,
                                     const int32_t* blk_off, int32_t* step, const float* lr, float beta1, float beta2,
                                     float eps, float grad_scale, int ticked, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!ticked) hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, st, step);
  if (n_blocks > 0)
    hipLaunchKernelGGL(k_adam_blocks, dim3((unsigned)n_blocks), dim3(256), 0, st, table, numel, blk_tensor, blk_off, step, lr,
                       beta1, beta2, eps, grad_scale);
  IGCN_CHECK_LAUNCH("adam_step_blocks");
  return IGCN_OK;
}

static int adam_step_multi_impl(bool tick, int n_tensors, const int64_t* table, const int64_t* numel, int32_t* step,
                                const float* lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (tick) hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, st, step);
  if (n_tensors > 0)
    hipLaunchKernelGGL(k_adam_multi, dim3(96, n_tensors), dim3(256), 0, st, table, numel, step, lr, beta1, beta2,
                       eps, grad_scale);
  IGCN_CHECK_LAUNCH("adam_step_multi");
  return IGCN_OK;
}
extern "C" int igcn_adam_step_multi(int n_tensors, const int64_t* table, const int64_t* numel, int32_t* step,
                                    const float* lr, float beta1, float beta2, float eps, float grad_scale,
                                    void* stream) {
  return adam_step_multi_impl(true, n_tensors, table, numel, step, lr, beta1, beta2, eps, grad_scale, stream);
}
extern "C" int igcn_adam_step_multi_ticked(int n_tensors, const int64_t* table, const int64_t* numel, int32_t* step,
                                           const float* lr, float beta1, float beta2, float eps, float grad_scale,
                                           void* stream) {
  return adam_step_multi_impl(false, n_tensors, table, numel, step, lr, beta1, beta2, eps, grad_scale, stream);
}

// dst_flat[off[t] .. off[t]+numel[t]) = grad tensor t (zeros when it has no gradient): packs the gradients
// into the flat bucket that the data-parallel all-reduce exchanges.
__global__ void __launch_bounds__(256)
k_pack_grads(const int64_t* __restrict__ table, const int64_t* __restrict__ numel,
             const int64_t* __restrict__ offset, float* __restrict__ flat) {
  const int t = blockIdx.y;
  const float* g = reinterpret_cast<const float*>(table[4 * t + 1]);
  const int64_t n = numel[t];
  float* dst = flat + offset[t];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dst[i] = g ? g[i] : 0.f;
}

extern "C" int igcn_pack_grads(int n_tensors, const int64_t* table, const int64_t* numel, const int64_t* offset,
                               float* flat, void* stream) {
  if (n_tensors > 0)
    hipLaunchKernelGGL(k_pack_grads, dim3(32, n_tensors), dim3(256), 0, (hipStream_t)stream, table, numel, offset,
                       flat);
  IGCN_CHECK_LAUNCH("pack_grads");
  return IGCN_OK;
}

// =================================================================================================
// igcn_copy_multi: n independent device-to-device copies in one launch (the hand-over of a batch to a captured step).
// blockIdx.y = copy, blockIdx.x strides over it; the descriptors travel as kernel arguments (no table upload).
// =================================================================================================
struct CopyMulti {
  void* dst[IGCN_COPY_MULTI_MAX];
  const void* src[IGCN_COPY_MULTI_MAX];
  int64_t nbytes[IGCN_COPY_MULTI_MAX];
};
__global__ void __launch_bounds__(256) k_copy_multi(CopyMulti cm) {
  const int c = blockIdx.y;
  const int64_t nb = cm.nbytes[c];
  char* d = reinterpret_cast<char*>(cm.dst[c]);
  const char* s = reinterpret_cast<const char*>(cm.src[c]);
  const int64_t first = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  if ((((uintptr_t)d | (uintptr_t)s) & 15) == 0) {
    const int64_t nv = nb >> 4;
    for (int64_t i = first; i < nv; i += stride)
      reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
    for (int64_t i = (nv << 4) + first; i < nb; i += stride) d[i] = s[i];
  } else {
    for (int64_t i = first; i < nb; i += stride) d[i] = s[i];
  }
}

extern "C" int igcn_copy_multi(int n, void* const* dst, const void* const* src, const int64_t* nbytes, void* stream) {
  IGCN_REQUIRE(n >= 0 && n <= IGCN_COPY_MULTI_MAX, "copy_multi: n=%d outside [0, %d]", n, IGCN_COPY_MULTI_MAX);
  if (n == 0) return IGCN_OK;
  CopyMulti cm;
  int64_t big = 0;
  for (int c = 0; c < n; ++c) {
    IGCN_REQUIRE(nbytes[c] >= 0 && (nbytes[c] == 0 || (dst[c] && src[c])), "copy_multi: bad descriptor %d", c);
    cm.dst[c] = dst[c];
    cm.src[c] = src[c];
    cm.nbytes[c] = nbytes[c];
    big = nbytes[c] > big ? nbytes[c] : big;
  }
  for (int c = n; c < IGCN_COPY_MULTI_MAX; ++c) { cm.dst[c] = nullptr; cm.src[c] = nullptr; cm.nbytes[c] = 0; }
  // enough workgroups for the largest copy at 16 bytes per lane and ~4 vectors per thread, at most 64 per copy
  int64_t gx = igcn_cdiv(igcn_cdiv(big, 16), 256 * 4);
  gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
  hipLaunchKernelGGL(k_copy_multi, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, cm);
  IGCN_CHECK_LAUNCH("copy_multi");
  return IGCN_OK;
}

// =================================================================================================
// Dense image of a sparse map (ops.SparseMap, small batches: the SNP <-> GO maps of configs[4] run as dense products on
// the matrix cores): image[c][pos[k]] = val_c[k] before the products, dval[c][k] = dimage[c][pos[k]] after them.  The
// channels' value vectors are read where they are (`val_stride` floats apart: the ParameterList entries inside
// train.FlatAdam's flat buffer) — no stack, no index_copy / index_select launches of the tensor library.
// =================================================================================================
__global__ void __launch_bounds__(256)
k_image_put(int64_t nnz, const int64_t* __restrict__ pos, const float* __restrict__ val, int64_t val_stride,
            float* __restrict__ image, int64_t image_stride) {
  const int c = blockIdx.y;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * 256)
    image[c * image_stride + pos[k]] = val[c * val_stride + k];
}
__global__ void __launch_bounds__(256)
k_image_take(int64_t nnz, const int64_t* __restrict__ pos, const float* __restrict__ image, int64_t image_stride,
             float* __restrict__ out) {
  const int c = blockIdx.y;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * 256)
    out[c * nnz + k] = image[c * image_stride + pos[k]];
}

extern "C" int igcn_image_put(int channels, int64_t nnz, const int64_t* pos, const float* val, int64_t val_stride,
                              float* image, int64_t image_stride, void* stream) {
  IGCN_REQUIRE(channels > 0 && channels <= 65535 && nnz >= 0 && val_stride >= nnz, "image_put: bad sizes");
  if (nnz == 0) return IGCN_OK;
  int64_t gx = igcn_cdiv(nnz, 256 * 4);
  gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
  hipLaunchKernelGGL(k_image_put, dim3((unsigned)gx, (unsigned)channels), dim3(256), 0, (hipStream_t)stream, nnz, pos, val,
                     val_stride, image, image_stride);
  IGCN_CHECK_LAUNCH("image_put");
  return IGCN_OK;
}

extern "C" int igcn_image_take(int channels, int64_t nnz, const int64_t* pos, const float* image, int64_t image_stride,
                               float* out, void* stream) {
  IGCN_REQUIRE(channels > 0 && channels <= 65535 && nnz >= 0, "image_take: bad sizes");
  if (nnz == 0) return IGCN_OK;
  int64_t gx = igcn_cdiv(nnz, 256 * 4);
  gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
  hipLaunchKernelGGL(k_image_take, dim3((unsigned)gx, (unsigned)channels), dim3(256), 0, (hipStream_t)stream, nnz, pos,
                     image, image_stride, out);
  IGCN_CHECK_LAUNCH("image_take");
  return IGCN_OK;
}

// =================================================================================================
// igcn_gather_batch: the block-diagonal collation of Batch.from_data_list (batch.py:24-123) for a dataset of UNIFORM
// graphs held as stacked device tensors — every key of the batch in ONE launch (blockIdx.y = key):
//   kind 0  rows:   dst[b, :] = src[idx[b], :]                       (x, edge_attr, snps_feat, y, ... : `row_bytes` each)
//   kind 1  index:  dst[r, b E + e] = src[idx[b], r, e] + b * nodes   (edge_index [2, E] int64 per graph -> [2, B E]:
//                   concatenated along the last dim, offset by the cumulative node count, batch.py:98-104)
// A feeder on a side stream then costs the train step one launch instead of a dozen gathers.
// =================================================================================================
struct GatherBatch {
  void* dst[IGCN_COPY_MULTI_MAX];
  const void* src[IGCN_COPY_MULTI_MAX];
  int64_t row_bytes[IGCN_COPY_MULTI_MAX];             // bytes per graph (kind 1: 2 * E * 8)
  int kind[IGCN_COPY_MULTI_MAX];
};
__global__ void __launch_bounds__(256)
k_gather_batch(int B, int64_t nodes, int64_t n_src, const int64_t* __restrict__ idx, GatherBatch gb) {
  const int k = blockIdx.y;
  const int64_t rb = gb.row_bytes[k];
  const int64_t first = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  if (gb.kind[k] == 0) {
    const int64_t w = rb >> 2, total = (int64_t)B * w;                        // 4-byte words
    const uint32_t* s = reinterpret_cast<const uint32_t*>(gb.src[k]);
    uint32_t* d = reinterpret_cast<uint32_t*>(gb.dst[k]);
    for (int64_t i = first; i < total; i += stride) {
      const int64_t b = i / w, o = i - b * w, sj = idx[b];
      d[i] = (sj >= 0 && sj < n_src) ? s[sj * w + o] : 0u;                   // (an index outside the dataset: zeros, no read)
    }
  } else {
    const int64_t e2 = rb >> 3, E = e2 >> 1, total = (int64_t)B * e2;         // int64 entries: [2, E] per graph
    const int64_t* s = reinterpret_cast<const int64_t*>(gb.src[k]);
    int64_t* d = reinterpret_cast<int64_t*>(gb.dst[k]);
    for (int64_t i = first; i < total; i += stride) {
      const int64_t r = i / ((int64_t)B * E), rem = i - r * (int64_t)B * E, b = rem / E, e = rem - b * E, sj = idx[b];
      d[i] = (sj >= 0 && sj < n_src) ? s[sj * e2 + r * E + e] + b * nodes : -1;      // (-1: the plan build reports it)
    }
  }
}

extern "C" int igcn_gather_batch(int n, int B, int64_t nodes_per_graph, int64_t n_subjects, const int64_t* idx,
                                 void* const* dst, const void* const* src, const int64_t* row_bytes, const int* kind,
                                 void* stream) {
  IGCN_REQUIRE(n >= 0 && n <= IGCN_COPY_MULTI_MAX && B > 0 && idx != nullptr && n_subjects > 0,
               "gather_batch: n=%d outside [0, %d] or bad B / dataset size", n, IGCN_COPY_MULTI_MAX);
  if (n == 0) return IGCN_OK;
  GatherBatch gb;
  int64_t big = 0;
  for (int c = 0; c < IGCN_COPY_MULTI_MAX; ++c) {
    gb.dst[c] = c < n ? dst[c] : nullptr;
    gb.src[c] = c < n ? src[c] : nullptr;
    gb.row_bytes[c] = c < n ? row_bytes[c] : 0;
    gb.kind[c] = c < n ? kind[c] : 0;
    if (c < n) {
      IGCN_REQUIRE(dst[c] && src[c] && row_bytes[c] > 0 && row_bytes[c] % (kind[c] ? 16 : 4) == 0,
                   "gather_batch: key %d: rows must be whole 4-byte words (index keys: [2, E] int64)", c);
      big = row_bytes[c] > big ? row_bytes[c] : big;
    }
  }
  int64_t gx = igcn_cdiv(igcn_cdiv(big * B, 4), 256 * 4);
  gx = gx < 1 ? 1 : (gx > 128 ? 128 : gx);
  hipLaunchKernelGGL(k_gather_batch, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, B, nodes_per_graph,
                     n_subjects, idx, gb);
  IGCN_CHECK_LAUNCH("gather_batch");
  return IGCN_OK;
}

// =================================================================================================
// Backward glue of y = act(x W^T + b) (ops.Linear): g = dy * [y > 0] (when y is given) and db[c] = sum_r g[r,c] in
// ONE pass over dy — instead of compare + multiply + (memset + two-stage sum) as four library launches.
// Rows are split over workgroups; each writes one [cols] partial, summed in order by the common row reduction.
// =================================================================================================
#define BG_T 256
// (blockIdx.y selects one of up to two problems of the same shape: igcn_bias_grad_pair)
struct BiasGradPtrs { const float* dy[2]; const float* y[2]; float* g[2]; float* partial[2]; int zero_cols[2]; };
template <int VW>
__global__ void __launch_bounds__(BG_T)
k_bias_grad(int64_t rows, int cols, int64_t rows_per_block, BiasGradPtrs pp) {
  const float* __restrict__ dy = pp.dy[blockIdx.y];
  const float* __restrict__ y = pp.y[blockIdx.y];
  float* __restrict__ g = pp.g[blockIdx.y];
  float* __restrict__ partial = pp.partial[blockIdx.y];
  const int zero_cols = pp.zero_cols[blockIdx.y];
  // thread = (row lane, column group of VW): cpr column groups per row, BG_T / cpr row lanes
  __shared__ float red[BG_T * VW];
  const int cpr = cols / VW, rl = threadIdx.x / cpr, cg = threadIdx.x % cpr, lanes = BG_T / cpr;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float acc[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) acc[j] = 0.f;
  if (rl < lanes) {
#pragma unroll 4
    for (int64_t r = r0 + rl; r < r1; r += lanes) {
      const int64_t o = r * cols + cg * VW;
      float v[VW];
      if constexpr (VW == 4) {
        const float4 t = *reinterpret_cast<const float4*>(dy + o);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        if (y) {
          const float4 u = *reinterpret_cast<const float4*>(y + o);
          v[0] = u.x > 0.f ? v[0] : 0.f; v[1] = u.y > 0.f ? v[1] : 0.f;
          v[2] = u.z > 0.f ? v[2] : 0.f; v[3] = u.w > 0.f ? v[3] : 0.f;
          *reinterpret_cast<float4*>(g + o) = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
        v[0] = dy[o];
        if (y) {
          v[0] = y[o] > 0.f ? v[0] : 0.f;
          g[o] = v[0];
        }
      }
#pragma unroll
      for (int j = 0; j < VW; ++j) acc[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VW; ++j) red[threadIdx.x * VW + j] = (rl < lanes) ? acc[j] : 0.f;
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += BG_T) {      // column c = group c / VW, slot c % VW; row lanes in order
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[(l * cpr + c / VW) * VW + c % VW];
    partial[(int64_t)blockIdx.x * (cols + zero_cols) + c] = t;
  }
  // `zero_cols` structurally zero outputs behind the sums (igcn_col_sums): they ride through the same reduction
  for (int c = threadIdx.x; c < zero_cols; c += BG_T) partial[(int64_t)blockIdx.x * (cols + zero_cols) + cols + c] = 0.f;
}

extern "C" size_t igcn_bias_grad_scratch_floats(int64_t rows, int cols) {
  const int64_t blocks = rows < 512 * 64 ? igcn_cdiv(rows, 64) : 512;
  return (size_t)(blocks * cols + 64);
}

// one or two column-sum problems of the same [rows, cols] shape in one launch
static int bias_grad_launch(int64_t rows, int cols, int n, const float* const* dy, const float* const* y, float* const* g,
                            float* const* out, float* const* scratch, const int* zero_cols, hipStream_t st,
                            const char* nm) {
  const int64_t blocks = rows < 512 * 64 ? igcn_cdiv(rows, 64) : 512;       // >= 64 rows per workgroup, <= 512 of them
  const int64_t rpb = igcn_cdiv(rows, blocks);
  const int64_t nb = igcn_cdiv(rows, rpb);
  bool vec = cols % 4 == 0 && BG_T % (cols / 4) == 0;
  BiasGradPtrs pp = {};
  for (int i = 0; i < 2; ++i) {
    const int k = i < n ? i : 0;
    pp.dy[i] = dy[k]; pp.y[i] = y[k]; pp.g[i] = g[k]; pp.partial[i] = scratch[k]; pp.zero_cols[i] = zero_cols[k];
    vec = vec && (((uintptr_t)dy[k] | (uintptr_t)y[k] | (uintptr_t)g[k]) & 15) == 0;
  }
  if (vec)
    hipLaunchKernelGGL((k_bias_grad<4>), dim3((unsigned)nb, (unsigned)n), dim3(BG_T), 0, st, rows, cols, rpb, pp);
  else
    hipLaunchKernelGGL((k_bias_grad<1>), dim3((unsigned)nb, (unsigned)n), dim3(BG_T), 0, st, rows, cols, rpb, pp);
  IGCN_CHECK_LAUNCH(nm);
  for (int i = 0; i < n; ++i) {
    const int w = cols + zero_cols[i];
    const int rc = igcn_launch_reduce_rows_final(scratch[i], nb, w, w, out[i], st);
    if (rc) return rc;
  }
  return IGCN_OK;
}

extern "C" int igcn_bias_grad(int64_t rows, int cols, const float* dy, const float* y, float* g, float* db,
                              float* scratch, void* stream) {
  IGCN_REQUIRE(rows > 0 && cols > 0 && cols <= BG_T && (y == nullptr || g != nullptr), "bias_grad: bad arguments");
  const int z = 0;
  return bias_grad_launch(rows, cols, 1, &dy, &y, &g, &db, &scratch, &z, (hipStream_t)stream, "bias_grad");
}

// Two igcn_bias_grad / igcn_col_sums problems of the SAME [rows, cols] shape in one launch (the two heads' first
// layers; d b_q and d b_v of the attention projection): per problem dy, optional ReLU reference y with masked copy g,
// output db [cols + zero_cols] (zero_cols structurally zero entries behind the sums), scratch as for one problem.
extern "C" int igcn_bias_grad_pair(int64_t rows, int cols, const float* dy0, const float* y0, float* g0, float* db0,
                                   int zero_cols0, float* scratch0, const float* dy1, const float* y1, float* g1,
                                   float* db1, int zero_cols1, float* scratch1, void* stream) {
  IGCN_REQUIRE(rows > 0 && cols > 0 && cols <= BG_T && (y0 == nullptr || g0 != nullptr) &&
                   (y1 == nullptr || g1 != nullptr) && zero_cols0 >= 0 && zero_cols1 >= 0 && zero_cols0 <= 4096 &&
                   zero_cols1 <= 4096, "bias_grad_pair: bad arguments");
  const float* dy[2] = {dy0, dy1};
  const float* y[2] = {y0, y1};
  float* g[2] = {g0, g1};
  float* out[2] = {db0, db1};
  float* scr[2] = {scratch0, scratch1};
  const int z[2] = {zero_cols0, zero_cols1};
  return bias_grad_launch(rows, cols, 2, dy, y, g, out, scr, z, (hipStream_t)stream, "bias_grad_pair");
}

// out[0:cols] = column sums of x [rows, cols]; out[cols : cols + zero_cols] = 0 — a gradient block that is known to
// vanish (the key bias of a softmax over keys, ops.ProjectedAttention) written by the launch that sums its neighbour.
// scratch: igcn_bias_grad_scratch_floats(rows, cols + zero_cols).  Final-gradient semantics (deferrable).
extern "C" int igcn_col_sums(int64_t rows, int cols, int zero_cols, const float* x, float* out, float* scratch,
                             void* stream) {
  IGCN_REQUIRE(rows > 0 && cols > 0 && cols <= BG_T && zero_cols >= 0 && zero_cols <= 4096, "col_sums: bad arguments");
  const float* y = nullptr;
  float* g = nullptr;
  return bias_grad_launch(rows, cols, 1, &x, &y, &g, &out, &scratch, &zero_cols, (hipStream_t)stream, "col_sums");
}

// =================================================================================================
// out[r, p*F + c] = part_p[r, c]: the jumping-knowledge concatenation of the GCN layer outputs
// (kernel/sgcn_img_snp.py:223-224 `torch.cat(xs, dim=1)`), 16 bytes per lane on both sides.  The library's
// generic concatenation moves this shape 4 bytes at a time (11.7 us for 5.9 MB at the bench shape).
// =================================================================================================
struct CatParts { const float* p[4]; };
__global__ void __launch_bounds__(256)
k_concat_cols(int64_t rows, int F, int nparts, CatParts parts, float* __restrict__ out) {
  const int fq = F / 4, per_row = nparts * fq;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * per_row) return;
  const int64_t r = i / per_row;
  const int j = (int)(i - r * per_row), p = j / fq, q = j - p * fq;
  const float* src = p == 0 ? parts.p[0] : (p == 1 ? parts.p[1] : (p == 2 ? parts.p[2] : parts.p[3]));
  reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(src)[r * fq + q];
}

// out = sum of n <= 4 tensors of `numel` floats (16 bytes per lane, scalar tail).  The gradient of a tensor with several
// consumers: autograd would add the incoming gradients pairwise, one launch per add (ops.GradFan).
struct SumParts {
  const float* p[4];
};
__global__ void __launch_bounds__(256) k_sum_n(int64_t numel, int n, SumParts sp, float* __restrict__ out) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 + 3 < numel && n >= 1) {
    float4 a = *reinterpret_cast<const float4*>(sp.p[0] + i4);
    for (int k = 1; k < n; ++k) {
      const float4 b = *reinterpret_cast<const float4*>(sp.p[k] + i4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    *reinterpret_cast<float4*>(out + i4) = a;
  } else {
    for (int64_t i = i4; i < numel && i < i4 + 4; ++i) {
      float a = sp.p[0][i];
      for (int k = 1; k < n; ++k) a += sp.p[k][i];
      out[i] = a;
    }
  }
}

extern "C" int igcn_sum_n(int64_t numel, int n, const float* const* parts /*HOST array*/, float* out, void* stream) {
  IGCN_REQUIRE(numel >= 0 && n >= 1 && n <= 4 && parts && out, "sum_n: 1..4 parts");
  SumParts sp = {};
  for (int k = 0; k < n; ++k) {
    IGCN_REQUIRE(parts[k] != nullptr && ((uintptr_t)parts[k] & 15) == 0, "sum_n: parts must be 16-byte aligned");
    sp.p[k] = parts[k];
  }
  IGCN_REQUIRE(((uintptr_t)out & 15) == 0, "sum_n: out must be 16-byte aligned");
  if (numel == 0) return IGCN_OK;
  hipLaunchKernelGGL(k_sum_n, dim3((unsigned)igcn_cdiv(numel, 1024)), dim3(256), 0, (hipStream_t)stream, numel, n, sp,
                     out);
  IGCN_CHECK_LAUNCH("sum_n");
  return IGCN_OK;
}

// igcn_sum_n for a gradient nothing reads before the optimiser (a LEAF with several consumers): joins the deferred
// final reductions (igcn_reduce_defer / igcn_reduce_flush) when they are on — the buffers must then stay alive until
// the flush — and is igcn_sum_n otherwise.  Same summation order either way.
int igcn_queue_sum_final(const float* const* parts, int k, int64_t numel, float* out, hipStream_t st);   // plan.hip
extern "C" int igcn_sum_n_final(int64_t numel, int n, const float* const* parts /*HOST array*/, float* out,
                                void* stream) {
  IGCN_REQUIRE(numel >= 0 && n >= 1 && n <= 4 && parts && out, "sum_n_final: 1..4 parts");
  for (int k = 0; k < n; ++k) IGCN_REQUIRE(parts[k] != nullptr, "sum_n_final: null part");
  if (numel > 0 && igcn_queue_sum_final(parts, n, numel, out, (hipStream_t)stream)) return IGCN_OK;
  return igcn_sum_n(numel, n, parts, out, stream);
}

extern "C" int igcn_concat_cols(int64_t rows, int F, int nparts, const float* const* parts /*HOST array*/, float* out,
                                void* stream) {
  IGCN_REQUIRE(rows >= 0 && F > 0 && F % 4 == 0 && nparts >= 1 && nparts <= 4, "concat_cols: F %% 4 == 0, <= 4 parts");
  CatParts cp = {};
  for (int k = 0; k < nparts; ++k) {
    IGCN_REQUIRE(((uintptr_t)parts[k] & 15) == 0, "concat_cols: parts must be 16-byte aligned");
    cp.p[k] = parts[k];
  }
  IGCN_REQUIRE(((uintptr_t)out & 15) == 0, "concat_cols: out must be 16-byte aligned");
  const int64_t total = rows * nparts * (F / 4);
  if (total == 0) return IGCN_OK;
  hipLaunchKernelGGL(k_concat_cols, dim3((unsigned)igcn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, rows, F,
                     nparts, cp, out);
  IGCN_CHECK_LAUNCH("concat_cols");
  return IGCN_OK;
}

// =================================================================================================
// Inputs of the two MLP heads (kernel/sgcn_img_snp.py:284-297) in one pass:
//   out_z   [R, W]         = (img + cross) / 2
//   out_lin [R, W + L]     = out_z | latent
//   feat    [R, W + L + P] = out_lin | (x * prob) of the row's sample          (P = 0: no regression features)
// R = passes * bsz rows (pass-major), x [bsz, P], prob [P].  All widths are even: 8 bytes per lane (W + L + P = 3182
// floats per row leaves odd rows only 8-byte aligned).  The composite is seven library launches forward and a dozen
// backward (add, scale, two concatenations, broadcast multiply, repeat, their gradients and reductions).
// =================================================================================================
__global__ void __launch_bounds__(256)
k_head_inputs_fwd(int64_t R, int bsz, int W, int L, int P, const float* __restrict__ img,
                  const float* __restrict__ cross, const float* __restrict__ latent, const float* __restrict__ x,
                  const float* __restrict__ prob, float* __restrict__ out_z, float* __restrict__ out_lin,
                  float* __restrict__ feat) {
  const int wf = (W + L + P) / 2;                               // float2 per row of the widest output
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * wf) return;
  const int64_t r = i / wf;
  const int c = (int)(i - r * wf) * 2;
  float2 v;
  if (c < W) {
    const float2 a = *reinterpret_cast<const float2*>(img + r * W + c);
    const float2 b = *reinterpret_cast<const float2*>(cross + r * W + c);
    v = make_float2((a.x + b.x) * 0.5f, (a.y + b.y) * 0.5f);
    *reinterpret_cast<float2*>(out_z + r * W + c) = v;
  } else if (c < W + L) {
    v = *reinterpret_cast<const float2*>(latent + r * L + (c - W));
  } else {
    const int j = c - W - L;
    const int64_t b = r % bsz;
    const float2 xv = *reinterpret_cast<const float2*>(x + b * P + j);
    const float2 pv = *reinterpret_cast<const float2*>(prob + j);
    v = make_float2(xv.x * pv.x, xv.y * pv.y);
  }
  if (c < W + L) *reinterpret_cast<float2*>(out_lin + r * (W + L) + c) = v;
  if (feat) *reinterpret_cast<float2*>(feat + r * (W + L + P) + c) = v;
}

// The same outputs with the layer in front computed on the way: cross[r, node * D + f] = relu(o[r, node, :] . Wp[f, :] + bp[f])
// (relu(out_proj(attention output)), kernel/sgcn_img_snp.py:241-242; a 32 x 32 product per graph node — as a GEMM launch of
// its own 6.4 us and 5.9 MB written and read back).  Workgroup (chunk of 512 columns, HOF_ROWS rows): the chunk's
// 512 / D nodes of o and Wp^T are staged in LDS; thread t owns columns chunk * 512 + 2t, + 1 as in k_head_inputs_fwd.
// `cross` is still written: the backward's ReLU mask reads it (igcn_head_inputs_bwd_relu).  D a power of two, 2 <= D <= 64.
// Both operands come from LDS, four k at a time: the thread's two columns of Wp^T for those k (4 x 8 bytes) serve all
// HOF_ROWS rows, and a row's o values are one 16-byte read — 8 LDS instructions per 32 FMAs.  [First form, one k and one row
// at a time (a 4-byte and an 8-byte read per two FMAs): 13.4 us, LDS-issue-bound, against 6.4 + 8.5 for the two launches it
// replaces.  Wp in registers (2 x 32 floats per thread): the compiler reloads them per row, 15.5-17 us.]
#define HOF_ROWS 4
__global__ void __launch_bounds__(256)
k_outproj_head_inputs_fwd(int64_t R, int bsz, int W, int L, int P, int D, const float* __restrict__ o,
                          const float* __restrict__ Wp, const float* __restrict__ bp, const float* __restrict__ img,
                          const float* __restrict__ latent, const float* __restrict__ x, const float* __restrict__ prob,
                          float* __restrict__ cross, float* __restrict__ out_z, float* __restrict__ out_lin,
                          float* __restrict__ feat, int chunks) {
  __shared__ __attribute__((aligned(16))) float wt[64 * 64];                 // Wp^T [k][f]
  __shared__ __attribute__((aligned(16))) float os[HOF_ROWS][512 + 256];     // o of the chunk's nodes per row: [node][k], node stride D + 4
  const int chunk = blockIdx.x % chunks, rp = blockIdx.x / chunks;
  const int tid = threadIdx.x;
  const int c0 = chunk * 512, c = c0 + 2 * tid;
  const int f = c & (D - 1), nl = (2 * tid) / D;         // feature pair (f, f + 1) of local node nl
  const int ost = D >= 8 ? D + 4 : D + 1;                 // <= 768 floats per row for every D; 16-byte aligned rows for D >= 8
  // every global operand of the thread's HOF_ROWS rows is requested before the first one is used
  float2 ov2[HOF_ROWS], in2[HOF_ROWS];
  float2 bias = make_float2(0.f, 0.f), pv = make_float2(0.f, 0.f);
  if (c < W) bias = *reinterpret_cast<const float2*>(bp + f);
  else if (c >= W + L && c < W + L + P) pv = *reinterpret_cast<const float2*>(prob + (c - W - L));
#pragma unroll
  for (int k = 0; k < HOF_ROWS; ++k) {
    const int64_t r0 = (int64_t)rp * HOF_ROWS + k, r = r0 < R ? r0 : R - 1;
    ov2[k] = in2[k] = make_float2(0.f, 0.f);
    if (c < W) {
      ov2[k] = *reinterpret_cast<const float2*>(o + r * W + c);
      in2[k] = *reinterpret_cast<const float2*>(img + r * W + c);
    } else if (c < W + L) {
      in2[k] = *reinterpret_cast<const float2*>(latent + r * L + (c - W));
    } else if (c < W + L + P) {
      in2[k] = *reinterpret_cast<const float2*>(x + (r % bsz) * P + (c - W - L));
    }
  }
  if (c0 < W) {
    for (int i = tid; i < D * D; i += 256) wt[(i % D) * D + i / D] = Wp[i];            // Wp [f][k] -> wt[k][f]
#pragma unroll
    for (int k = 0; k < HOF_ROWS; ++k)
      if (c < W) {
        const int at = nl * ost + f;
        os[k][at] = ov2[k].x;
        os[k][at + 1] = ov2[k].y;
      }
    __syncthreads();
  }
  if (c >= W + L + P) return;
  float2 a[HOF_ROWS];
#pragma unroll
  for (int k = 0; k < HOF_ROWS; ++k) a[k] = bias;
  if (c < W) {
    if (D >= 8) {
      for (int kk = 0; kk < D; kk += 4) {
        float2 w2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w2[i] = *reinterpret_cast<const float2*>(&wt[(kk + i) * D + f]);
#pragma unroll
        for (int k = 0; k < HOF_ROWS; ++k) {
          const float4 ov = *reinterpret_cast<const float4*>(&os[k][nl * ost + kk]);
          a[k].x = fmaf(ov.x, w2[0].x, a[k].x);     a[k].y = fmaf(ov.x, w2[0].y, a[k].y);
          a[k].x = fmaf(ov.y, w2[1].x, a[k].x);     a[k].y = fmaf(ov.y, w2[1].y, a[k].y);
          a[k].x = fmaf(ov.z, w2[2].x, a[k].x);     a[k].y = fmaf(ov.z, w2[2].y, a[k].y);
          a[k].x = fmaf(ov.w, w2[3].x, a[k].x);     a[k].y = fmaf(ov.w, w2[3].y, a[k].y);
        }
      }
    } else {
      for (int kk = 0; kk < D; ++kk) {
        const float2 w2 = *reinterpret_cast<const float2*>(&wt[kk * D + f]);
#pragma unroll
        for (int k = 0; k < HOF_ROWS; ++k) {
          const float ov = os[k][nl * ost + kk];
          a[k].x = fmaf(ov, w2.x, a[k].x);
          a[k].y = fmaf(ov, w2.y, a[k].y);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < HOF_ROWS; ++k) {
    const int64_t r = (int64_t)rp * HOF_ROWS + k;
    if (r >= R) break;
    float2 v;
    if (c < W) {
      const float2 y = make_float2(fmaxf(a[k].x, 0.f), fmaxf(a[k].y, 0.f));
      *reinterpret_cast<float2*>(cross + r * W + c) = y;
      v = make_float2((in2[k].x + y.x) * 0.5f, (in2[k].y + y.y) * 0.5f);
      *reinterpret_cast<float2*>(out_z + r * W + c) = v;
    } else if (c < W + L) {
      v = in2[k];
    } else {
      v = make_float2(in2[k].x * pv.x, in2[k].y * pv.y);
    }
    if (c < W + L) *reinterpret_cast<float2*>(out_lin + r * (W + L) + c) = v;
    if (feat) *reinterpret_cast<float2*>(feat + r * (W + L + P) + c) = v;
  }
}

// d_mid [R, W] = (d_out_z + d_out_lin[:, :W] + d_feat[:, :W]) / 2  (the gradient of img AND of cross);
// d_latent [R, L] = d_out_lin[:, W:] + d_feat[:, W:W+L].  Any of the three incoming gradients may be NULL.
__global__ void __launch_bounds__(256)
k_head_inputs_bwd_main(int64_t R, int W, int L, int P, const float* __restrict__ d_out_z,
                       const float* __restrict__ d_out_lin, const float* __restrict__ d_feat,
                       float* __restrict__ d_mid, float* __restrict__ d_latent) {
  const int wf = (W + L) / 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * wf) return;
  const int64_t r = i / wf;
  const int c = (int)(i - r * wf) * 2;
  float2 t = make_float2(0.f, 0.f);
  if (d_out_lin) {
    const float2 a = *reinterpret_cast<const float2*>(d_out_lin + r * (W + L) + c);
    t.x += a.x; t.y += a.y;
  }
  if (d_feat) {
    const float2 a = *reinterpret_cast<const float2*>(d_feat + r * (W + L + P) + c);
    t.x += a.x; t.y += a.y;
  }
  if (c < W) {
    if (d_out_z) {
      const float2 a = *reinterpret_cast<const float2*>(d_out_z + r * W + c);
      t.x += a.x; t.y += a.y;
    }
    *reinterpret_cast<float2*>(d_mid + r * W + c) = make_float2(t.x * 0.5f, t.y * 0.5f);
  } else {
    *reinterpret_cast<float2*>(d_latent + r * L + (c - W)) = t;
  }
}

// regression features: g[b, j] = sum_passes d_feat[pass*bsz + b, W+L+j];  dx[b, j] = g * prob[j];
// dprob[j] = sum_b g[b, j] * x[b, j].  One workgroup per column j, threads stride the samples.
__device__ __forceinline__ void head_inputs_bwd_prob_body(int j, int64_t R, int bsz, int W, int L, int P,
                                                          const float* __restrict__ d_feat,
                                                          const float* __restrict__ x, const float* __restrict__ prob,
                                                          float* __restrict__ dx, float* __restrict__ dprob,
                                                          float* red) {
  const int passes = (int)(R / bsz);
  const float pj = prob[j];
  float acc = 0.f;
  for (int b = threadIdx.x; b < bsz; b += 256) {
    float g = 0.f;
    if (d_feat)
      for (int k = 0; k < passes; ++k) g += d_feat[((int64_t)k * bsz + b) * (W + L + P) + W + L + j];
    dx[(int64_t)b * P + j] = g * pj;
    acc += g * x[(int64_t)b * P + j];
  }
  acc = block_sum_all(acc, red);
  if (threadIdx.x == 0) dprob[j] = acc;
}

// The same sums by workgroups of HIP_COLS = 16 COLUMNS x all samples (thread = (sample group tid / 8, column pair tid % 8)):
// a row's 16 columns are half a 128-byte line shared by 8 lanes, and 256 samples are ONE trip of eight per thread.  [One workgroup per column touches a different line per lane — 270
// columns were ~35 MB of line traffic for 0.8 MB of data and cost the launch they rode in 3.7 us.]  red: 512 floats.
#define HIP_COLS 16
__device__ __forceinline__ void head_inputs_bwd_probc_body(int blk, int64_t R, int bsz, int W, int L, int P,
                                                            const float* __restrict__ d_feat,
                                                            const float* __restrict__ x, const float* __restrict__ prob,
                                                            float* __restrict__ dx, float* __restrict__ dprob,
                                                            float* red) {
  const int passes = (int)(R / bsz), sg = threadIdx.x >> 3, jl = threadIdx.x & 7;
  const int j = HIP_COLS * blk + 2 * jl;
  const bool live = j < P;
  float2 acc = make_float2(0.f, 0.f);
  if (live) {
    const float2 pj = *reinterpret_cast<const float2*>(prob + j);
    // eight samples per trip, every load of the trip requested before the first sum (few workgroups carry this part: a
    // dependent round trip per sample made them the tail of the launch)
    for (int b0 = sg; b0 < bsz; b0 += 256) {
      float2 t0[8], t1[8], xv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int b = b0 + 32 * i < bsz ? b0 + 32 * i : bsz - 1;
        t0[i] = t1[i] = make_float2(0.f, 0.f);
        if (d_feat) {
          t0[i] = *reinterpret_cast<const float2*>(d_feat + (int64_t)b * (W + L + P) + W + L + j);
          if (passes > 1) t1[i] = *reinterpret_cast<const float2*>(d_feat + ((int64_t)bsz + b) * (W + L + P) + W + L + j);
        }
        xv[i] = *reinterpret_cast<const float2*>(x + (int64_t)b * P + j);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int b = b0 + 32 * i;
        if (b >= bsz) break;
        float2 g = make_float2(t0[i].x + t1[i].x, t0[i].y + t1[i].y);
        if (d_feat)
          for (int k = 2; k < passes; ++k) {
            const float2 t = *reinterpret_cast<const float2*>(d_feat + ((int64_t)k * bsz + b) * (W + L + P) + W + L + j);
            g.x += t.x; g.y += t.y;
          }
        *reinterpret_cast<float2*>(dx + (int64_t)b * P + j) = make_float2(g.x * pj.x, g.y * pj.y);
        acc.x = fmaf(g.x, xv[i].x, acc.x);
        acc.y = fmaf(g.y, xv[i].y, acc.y);
      }
    }
  }
  red[2 * threadIdx.x] = acc.x;
  red[2 * threadIdx.x + 1] = acc.y;
  __syncthreads();
  if (threadIdx.x < HIP_COLS && HIP_COLS * blk + (int)threadIdx.x < P) {
    float a = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) a += red[2 * (8 * g + (threadIdx.x >> 1)) + (threadIdx.x & 1)];
    dprob[HIP_COLS * blk + threadIdx.x] = a;
  }
}

__global__ void __launch_bounds__(256)
k_head_inputs_bwd_prob(int64_t R, int bsz, int W, int L, int P, const float* __restrict__ d_feat,
                       const float* __restrict__ x, const float* __restrict__ prob, float* __restrict__ dx,
                       float* __restrict__ dprob) {
  __shared__ float red[16];
  head_inputs_bwd_prob_body(blockIdx.x, R, bsz, W, L, P, d_feat, x, prob, dx, dprob, red);
}

// The same sums when `cross` [R, W] is the POST-ReLU output of the layer in front (relu(out_proj(attention)),
// kernel/sgcn_img_snp.py:241-242): the launch also writes d_cross [R, W] = d_mid where cross > 0, else 0 (that layer's
// ReLU backward — k_bias_grad's mask pass was a launch of its own, 7.6 us) and the workgroup's share of the layer's bias
// gradient, db_part [blocks][D] (column c of a row belongs to output feature c % D; D a power of two, 2 <= D <= 64), for
// the deferred reduction.  Workgroup (chunk, row pair): thread t owns columns chunk * 512 + 2t, 2t + 1 of HIB_ROWS rows,
// so its feature pair is (2t) % D whatever the chunk and the bias sums are lane shuffles (strides D/2 .. 32) and one
// 4-wave LDS step — a fixed order.  The first n_prob workgroups of the grid are the regression-feature columns
// (head_inputs_bwd_probc_body, HIP_COLS columns each — a launch of its own, 5.9 us, otherwise).
#define HIB_ROWS 2
__global__ void __launch_bounds__(256)
k_head_inputs_bwd_relu(int64_t R, int W, int L, int P, const float* __restrict__ d_out_z,
                       const float* __restrict__ d_out_lin, const float* __restrict__ d_feat,
                       float* __restrict__ d_mid, float* __restrict__ d_latent, const float* __restrict__ cross,
                       float* __restrict__ d_cross, float* __restrict__ db_part, int D, int chunks, int n_prob, int bsz,
                       const float* __restrict__ x, const float* __restrict__ prob, float* __restrict__ dx,
                       float* __restrict__ dprob) {
  __shared__ float wsum[8][64];
  // the n_prob workgroups of the regression-feature columns (HIP_COLS columns each) come first in the grid
  if ((int)blockIdx.x < n_prob) {
    head_inputs_bwd_probc_body(blockIdx.x, R, bsz, W, L, P, d_feat, x, prob, dx, dprob, &wsum[0][0]);
    return;
  }
  const int blk = blockIdx.x - n_prob;
  const int chunk = blk % chunks, rp = blk / chunks;
  const int c = chunk * 512 + 2 * threadIdx.x;
  float2 m = make_float2(0.f, 0.f);                     // this thread's masked pair summed over its rows
  if (c < W + L) {
    // the loads of every row first, then the sums and stores (a row at a time, a row's loads waited behind the stores of the
    // row before)
    float2 l1[HIB_ROWS], l2[HIB_ROWS], l3[HIB_ROWS], y[HIB_ROWS];
#pragma unroll
    for (int k = 0; k < HIB_ROWS; ++k) {
      const int64_t r0 = (int64_t)rp * HIB_ROWS + k, r = r0 < R ? r0 : R - 1;
      l1[k] = l2[k] = l3[k] = y[k] = make_float2(0.f, 0.f);
      if (d_out_lin) l1[k] = *reinterpret_cast<const float2*>(d_out_lin + r * (W + L) + c);
      if (d_feat) l2[k] = *reinterpret_cast<const float2*>(d_feat + r * (W + L + P) + c);
      if (c < W) {
        if (d_out_z) l3[k] = *reinterpret_cast<const float2*>(d_out_z + r * W + c);
        y[k] = *reinterpret_cast<const float2*>(cross + r * W + c);
      }
    }
#pragma unroll
    for (int k = 0; k < HIB_ROWS; ++k) {
      const int64_t r = (int64_t)rp * HIB_ROWS + k;
      if (r >= R) break;
      float2 t = make_float2(0.f, 0.f);                  // (the order of the additions is the plain kernel's)
      t.x += l1[k].x; t.y += l1[k].y;
      t.x += l2[k].x; t.y += l2[k].y;
      if (c < W) {
        t.x += l3[k].x; t.y += l3[k].y;
        const float2 h = make_float2(t.x * 0.5f, t.y * 0.5f);
        const float2 g = make_float2(y[k].x > 0.f ? h.x : 0.f, y[k].y > 0.f ? h.y : 0.f);
        *reinterpret_cast<float2*>(d_mid + r * W + c) = h;
        *reinterpret_cast<float2*>(d_cross + r * W + c) = g;
        m.x += g.x; m.y += g.y;
      } else {
        *reinterpret_cast<float2*>(d_latent + r * L + (c - W)) = t;
      }
    }
  }
  for (int s = D / 2; s < 64; s *= 2) {                 // lanes that share (2t) % D
    m.x += __shfl_xor(m.x, s);
    m.y += __shfl_xor(m.y, s);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < D / 2) { wsum[wave][2 * lane] = m.x; wsum[wave][2 * lane + 1] = m.y; }
  __syncthreads();
  if ((int)threadIdx.x < D)
    db_part[(int64_t)blk * D + threadIdx.x] =
        (wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + (wsum[2][threadIdx.x] + wsum[3][threadIdx.x]);
}

extern "C" int igcn_head_inputs_fwd(int64_t R, int bsz, int W, int L, int P, const float* img, const float* cross,
                                    const float* latent, const float* x, const float* prob, float* out_z,
                                    float* out_lin, float* feat, void* stream) {
  IGCN_REQUIRE(R > 0 && bsz > 0 && R % bsz == 0 && W > 0 && L > 0 && P >= 0 && W % 2 == 0 && L % 2 == 0 && P % 2 == 0,
               "head_inputs_fwd: even widths, R divisible by bsz");
  IGCN_REQUIRE((P == 0) == (feat == nullptr), "head_inputs_fwd: feat goes with P > 0");
  const int64_t total = R * ((W + L + P) / 2);
  hipLaunchKernelGGL(k_head_inputs_fwd, dim3((unsigned)igcn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, R,
                     bsz, W, L, P, img, cross, latent, x, prob, out_z, out_lin, feat);
  IGCN_CHECK_LAUNCH("head_inputs_fwd");
  return IGCN_OK;
}

// relu(out_proj(o)) and the heads' inputs in one launch (k_outproj_head_inputs_fwd): o [R, W] = [R, W / D nodes, D], Wp [D, D],
// bp [D]; cross [R, W] is written too (the backward's ReLU mask).  D a power of two in [2, 64] dividing 512 and W.
extern "C" int igcn_outproj_head_inputs_fwd(int64_t R, int bsz, int W, int L, int P, int D, const float* o, const float* Wp,
                                            const float* bp, const float* img, const float* latent, const float* x,
                                            const float* prob, float* cross, float* out_z, float* out_lin, float* feat,
                                            void* stream) {
  IGCN_REQUIRE(R > 0 && bsz > 0 && R % bsz == 0 && W > 0 && L > 0 && P >= 0 && W % 2 == 0 && L % 2 == 0 && P % 2 == 0,
               "outproj_head_inputs_fwd: even widths, R divisible by bsz");
  IGCN_REQUIRE((P == 0) == (feat == nullptr), "outproj_head_inputs_fwd: feat goes with P > 0");
  IGCN_REQUIRE(D >= 2 && D <= 64 && (D & (D - 1)) == 0 && W % D == 0 && o && Wp && bp && cross,
               "outproj_head_inputs_fwd: D a power of two in [2, 64] dividing W");
  const int chunks = (int)igcn_cdiv(W + L + P, 512);
  const int64_t blocks = (int64_t)chunks * igcn_cdiv(R, HOF_ROWS);
hipLaunchKernelGGL(k_outproj_head_inputs_fwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, R, bsz, W, L, P,
                     D, o, Wp, bp, img, latent, x, prob, cross, out_z, out_lin, feat, chunks);
  IGCN_CHECK_LAUNCH("outproj_head_inputs_fwd");
  return IGCN_OK;
}

extern "C" int igcn_head_inputs_bwd(int64_t R, int bsz, int W, int L, int P, const float* d_out_z,
                                    const float* d_out_lin, const float* d_feat, const float* x, const float* prob,
                                    float* d_mid, float* d_latent, float* dx, float* dprob, void* stream) {
  IGCN_REQUIRE(R > 0 && bsz > 0 && R % bsz == 0 && W % 2 == 0 && L % 2 == 0 && P % 2 == 0, "head_inputs_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = R * ((W + L) / 2);
  hipLaunchKernelGGL(k_head_inputs_bwd_main, dim3((unsigned)igcn_cdiv(total, 256)), dim3(256), 0, st, R, W, L, P,
                     d_out_z, d_out_lin, d_feat, d_mid, d_latent);
  if (P > 0 && dx && dprob)
    hipLaunchKernelGGL(k_head_inputs_bwd_prob, dim3(P), dim3(256), 0, st, R, bsz, W, L, P, d_feat, x, prob, dx, dprob);
  IGCN_CHECK_LAUNCH("head_inputs_bwd");
  return IGCN_OK;
}

// The same backward that also takes the ReLU backward and the bias gradient of the layer that produced `cross` (its
// post-ReLU output [R, W], W = rows x D features): d_cross [R, W] = the gradient of that layer's PRE-activation, db [D] = its
// column sums by feature — through db_part [igcn_head_inputs_bwd_blocks(R, W, L)][D] and igcn_reduce_rows_final (a deferred
// reduction while the stream defers).  D a power of two, 2 <= D <= 64, W % D == 0.
extern "C" int igcn_head_inputs_bwd_blocks(int64_t R, int W, int L) {
  return (int)(igcn_cdiv(W + L, 512) * igcn_cdiv(R, HIB_ROWS));
}
extern "C" int igcn_head_inputs_bwd_relu(int64_t R, int bsz, int W, int L, int P, const float* d_out_z,
                                         const float* d_out_lin, const float* d_feat, const float* x, const float* prob,
                                         float* d_mid, float* d_latent, float* dx, float* dprob, const float* cross,
                                         float* d_cross, int D, float* db_part, float* db, void* stream) {
  IGCN_REQUIRE(R > 0 && bsz > 0 && R % bsz == 0 && W % 2 == 0 && L % 2 == 0 && P % 2 == 0, "head_inputs_bwd_relu: bad sizes");
  IGCN_REQUIRE(cross && d_cross && db_part && db && D >= 2 && D <= 64 && (D & (D - 1)) == 0 && W % D == 0,
               "head_inputs_bwd_relu: D a power of two in [2, 64] dividing W, non-null outputs");
  hipStream_t st = (hipStream_t)stream;
  const int64_t blocks = igcn_head_inputs_bwd_blocks(R, W, L);
  const int chunks = (int)igcn_cdiv(W + L, 512);
  const bool with_prob = P > 0 && dx && dprob;
  const int n_prob = with_prob ? (int)igcn_cdiv(P, HIP_COLS) : 0;
  hipLaunchKernelGGL(k_head_inputs_bwd_relu, dim3((unsigned)(blocks + n_prob)), dim3(256), 0, st, R, W, L, P,
                     d_out_z, d_out_lin, d_feat, d_mid, d_latent, cross, d_cross, db_part, D, chunks, n_prob, bsz, x,
                     prob, dx, dprob);
  IGCN_CHECK_LAUNCH("head_inputs_bwd_relu");
  return igcn_launch_reduce_rows_final(db_part, blocks, D, D, db, st);
}

// =================================================================================================
// SNP importance mask (kernel/sgcn_img_snp.py:147-151): out[b, j] = snps[b, j] * sigmoid(p[j]), sp[j] = sigmoid(p[j]);
// backward dp[j] = (sum_b dout[b, j] snps[b, j] + dsp[j]) * sp (1 - sp).  One workgroup per SNP column.
// =================================================================================================
__global__ void __launch_bounds__(256)
k_snps_mask_fwd(int B, int S, const float* __restrict__ snps, const float* __restrict__ p, float* __restrict__ out,
                float* __restrict__ sp, float* __restrict__ plain) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * S) return;
  const int j = (int)(i % S);
  const float sg = 1.f / (1.f + __expf(-p[j]));
  const float v = snps[i];
  out[i] = v * sg;
  if (plain) plain[i] = v;                              // first half of the stacked (plain | masked) batch
  if (i < S) sp[j] = sg;
}

__global__ void __launch_bounds__(256)
k_snps_mask_bwd(int B, int S, const float* __restrict__ snps, const float* __restrict__ p,
                const float* __restrict__ dout, const float* __restrict__ dsp, float* __restrict__ dp) {
  __shared__ float red[16];
  const int j = blockIdx.x;
  float acc = 0.f;
  if (dout)
    for (int b = threadIdx.x; b < B; b += 256) acc += dout[(int64_t)b * S + j] * snps[(int64_t)b * S + j];
  acc = block_sum_all(acc, red);
  if (threadIdx.x == 0) {
    const float sg = 1.f / (1.f + __expf(-p[j]));
    dp[j] = (acc + (dsp ? dsp[j] : 0.f)) * sg * (1.f - sg);
  }
}

extern "C" int igcn_snps_mask_fwd(int B, int S, const float* snps, const float* p, float* out, float* sp,
                                  float* plain, void* stream) {
  IGCN_REQUIRE(B > 0 && S > 0, "snps_mask_fwd: bad sizes");
  hipLaunchKernelGGL(k_snps_mask_fwd, dim3((unsigned)igcn_cdiv((int64_t)B * S, 256)), dim3(256), 0, (hipStream_t)stream,
                     B, S, snps, p, out, sp, plain);
  IGCN_CHECK_LAUNCH("snps_mask_fwd");
  return IGCN_OK;
}

extern "C" int igcn_snps_mask_bwd(int B, int S, const float* snps, const float* p, const float* dout, const float* dsp,
                                  float* dp, void* stream) {
  IGCN_REQUIRE(B > 0 && S > 0, "snps_mask_bwd: bad sizes");
  hipLaunchKernelGGL(k_snps_mask_bwd, dim3(S), dim3(256), 0, (hipStream_t)stream, B, S, snps, p, dout, dsp, dp);
  IGCN_CHECK_LAUNCH("snps_mask_bwd");
  return IGCN_OK;
}

// =================================================================================================
// Narrow output layers: y[r, c] = sum_k x[r, k] W[c, k] + b[c] with C <= 4 outputs (lin2 / lin2_regr of
// kernel/sgcn_img_snp.py:289-301: 64 -> 3).  On the GEMM path such a layer costs a launch forward and five
// backward (bias-gradient pass + its reduction, dX, split-K dW + its reduction) for a few hundred KB of data; here
// it is one VALU kernel forward and one (+ a short reduction of block partials) backward.
// Thread = (row, quad of K): KQ = K/4 lanes per row, 256/KQ rows per workgroup pass.
// =================================================================================================
#define SL_MAXC 4
// (blockIdx.y selects one of up to two layers over the same [R, K] input shape: igcn_small_linear_pair_*)
struct SmallLinPtrs {
  const float* x[2]; const float* keep[2]; const float* W[2]; const float* b[2]; float* y[2];
  const float* dy[2]; float* dx[2]; float* partial[2]; int C[2];
};
__global__ void __launch_bounds__(256)
k_small_linear_fwd(int64_t R, int K, SmallLinPtrs pp) {
  const float* __restrict__ x = pp.x[blockIdx.y];
  const float* __restrict__ keep = pp.keep[blockIdx.y];
  const float* __restrict__ W = pp.W[blockIdx.y];
  const float* __restrict__ b = pp.b[blockIdx.y];
  float* __restrict__ y = pp.y[blockIdx.y];
  const int C = pp.C[blockIdx.y];
  const int kq = K / 4, q = threadIdx.x % kq, rl = threadIdx.x / kq, rpb = 256 / kq;
  const int64_t r = (int64_t)blockIdx.x * rpb + rl;
  float acc[SL_MAXC];
#pragma unroll
  for (int c = 0; c < SL_MAXC; ++c) acc[c] = 0.f;
  if (rl < rpb && r < R) {
    float4 xv = *reinterpret_cast<const float4*>(x + r * K + 4 * q);
    if (keep) {                                          // dropout of the input, fused: x * keep
      const float4 kv = *reinterpret_cast<const float4*>(keep + r * K + 4 * q);
      xv.x *= kv.x; xv.y *= kv.y; xv.z *= kv.z; xv.w *= kv.w;
    }
#pragma unroll
    for (int c = 0; c < SL_MAXC; ++c)
      if (c < C) {
        const float4 w = *reinterpret_cast<const float4*>(W + c * K + 4 * q);
        acc[c] = (xv.x * w.x + xv.y * w.y) + (xv.z * w.z + xv.w * w.w);
      }
  }
  // sum over the kq lanes of a row (kq is a power of two <= 64, rows do not straddle waves)
#pragma unroll
  for (int c = 0; c < SL_MAXC; ++c)
    for (int o = 1; o < kq; o <<= 1) acc[c] += __shfl_xor(acc[c], o, 64);
  if (q == 0 && rl < rpb && r < R) {
#pragma unroll
    for (int c = 0; c < SL_MAXC; ++c)
      if (c < C) y[r * C + c] = acc[c] + (b ? b[c] : 0.f);
  }
}

// dx[r, k] = sum_c dy[r, c] W[c, k];  partial[blk][c*K + k] = sum_{r in blk} dy[r, c] x[r, k];
// partial[blk][C*K + c] = sum_{r in blk} dy[r, c].  Rows of a workgroup: rows_per_block, walked 256/KQ at a time.
__global__ void __launch_bounds__(256)
k_small_linear_bwd(int64_t R, int K, int rows_per_block, SmallLinPtrs pp) {
  const float* __restrict__ x = pp.x[blockIdx.y];
  const float* __restrict__ keep = pp.keep[blockIdx.y];
  const float* __restrict__ W = pp.W[blockIdx.y];
  const float* __restrict__ dy = pp.dy[blockIdx.y];
  float* __restrict__ dx = pp.dx[blockIdx.y];
  float* __restrict__ partial = pp.partial[blockIdx.y];
  const int C = pp.C[blockIdx.y];
  __shared__ float red[256 * 4 * SL_MAXC + 256 * SL_MAXC];
  const int kq = K / 4, q = threadIdx.x % kq, rl = threadIdx.x / kq, rpb = 256 / kq;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  float4 w[SL_MAXC], gw[SL_MAXC];
  float gb[SL_MAXC];
#pragma unroll
  for (int c = 0; c < SL_MAXC; ++c) {
    w[c] = (c < C) ? *reinterpret_cast<const float4*>(W + c * K + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    gw[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    gb[c] = 0.f;
  }
  if (rl < rpb) {
#pragma unroll 4
    for (int64_t r = r0 + rl; r < r1; r += rpb) {
      float4 xv = *reinterpret_cast<const float4*>(x + r * K + 4 * q);
      float4 kv = make_float4(1.f, 1.f, 1.f, 1.f);
      if (keep) {
        kv = *reinterpret_cast<const float4*>(keep + r * K + 4 * q);
        xv.x *= kv.x; xv.y *= kv.y; xv.z *= kv.z; xv.w *= kv.w;
      }
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < SL_MAXC; ++c)
        if (c < C) {
          const float g = dy[r * C + c];
          d.x += g * w[c].x; d.y += g * w[c].y; d.z += g * w[c].z; d.w += g * w[c].w;
          gw[c].x += g * xv.x; gw[c].y += g * xv.y; gw[c].z += g * xv.z; gw[c].w += g * xv.w;
          gb[c] += g;
        }
      if (dx) *reinterpret_cast<float4*>(dx + r * K + 4 * q) = make_float4(d.x * kv.x, d.y * kv.y, d.z * kv.z, d.w * kv.w);
    }
  }
  // row lanes summed in order through LDS: every channel's partials staged at once (one barrier pair instead of four
  // per channel), one thread per (channel, column) / per channel
  float* prow = partial + (int64_t)blockIdx.x * (C * K + C);
  float* redb = red + 256 * 4 * SL_MAXC;               // [SL_MAXC][256] bias partials
#pragma unroll
  for (int c = 0; c < SL_MAXC; ++c)
    if (c < C) {
      float* rc = red + c * 1024;
      rc[threadIdx.x * 4 + 0] = gw[c].x; rc[threadIdx.x * 4 + 1] = gw[c].y;
      rc[threadIdx.x * 4 + 2] = gw[c].z; rc[threadIdx.x * 4 + 3] = gw[c].w;
      redb[c * 256 + threadIdx.x] = (q == 0 && rl < rpb) ? gb[c] : 0.f;
    }
  __syncthreads();
  for (int idx = threadIdx.x; idx < C * K; idx += 256) {
    const int c = idx / K, k = idx - c * K, qq = k / 4, j = k % 4;
    float t = 0.f;
    for (int l = 0; l < rpb; ++l) t += red[c * 1024 + (l * kq + qq) * 4 + j];
    prow[c * K + k] = t;
  }
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int l = 0; l < rpb; ++l) t += redb[threadIdx.x * 256 + l * kq];
    prow[C * K + threadIdx.x] = t;
  }
}

static bool small_linear_ok(int K, int C) {
  const int kq = K / 4;
  return K % 4 == 0 && kq >= 1 && kq <= 64 && (kq & (kq - 1)) == 0 && C >= 1 && C <= SL_MAXC;
}

// rows per workgroup: 64, or 16 while that leaves fewer than a few hundred workgroups (512 rows: 8 workgroups were the
// whole launch, 10 us of latency for 130 KB)
static int small_linear_rpb(int64_t R) { return R < 16384 ? 16 : 64; }
extern "C" size_t igcn_small_linear_bwd_scratch_floats(int64_t R, int K, int C) {
  return (size_t)(igcn_cdiv(R, small_linear_rpb(R)) * (C * K + C) + 64);
}

static int small_linear_fwd_launch(int64_t R, int K, int n, const SmallLinPtrs& pp, hipStream_t st) {
  for (int i = 0; i < n; ++i) {
    IGCN_REQUIRE(R > 0 && small_linear_ok(K, pp.C[i]), "small_linear: K/4 a power of two <= 64, 1 <= C <= 4 (K=%d C=%d)",
                 K, pp.C[i]);
    IGCN_REQUIRE((((uintptr_t)pp.x[i] | (uintptr_t)pp.W[i] | (uintptr_t)pp.keep[i]) & 15) == 0,
                 "small_linear: x, keep and W must be 16-byte aligned");
  }
  const int rpb = 256 / (K / 4);
  hipLaunchKernelGGL(k_small_linear_fwd, dim3((unsigned)igcn_cdiv(R, rpb), (unsigned)n), dim3(256), 0, st, R, K, pp);
  IGCN_CHECK_LAUNCH("small_linear_fwd");
  return IGCN_OK;
}

extern "C" int igcn_small_linear_fwd(int64_t R, int K, int C, const float* x, const float* keep, const float* W,
                                     const float* b, float* y, void* stream) {
  SmallLinPtrs pp = {};
  pp.x[0] = pp.x[1] = x; pp.keep[0] = pp.keep[1] = keep; pp.W[0] = pp.W[1] = W; pp.b[0] = pp.b[1] = b;
  pp.y[0] = pp.y[1] = y; pp.C[0] = pp.C[1] = C;
  return small_linear_fwd_launch(R, K, 1, pp, (hipStream_t)stream);
}

// Two narrow layers over inputs of the same [R, K] shape in one launch (lin2 and lin2_regr of the two heads).
extern "C" int igcn_small_linear_pair_fwd(int64_t R, int K, int C0, const float* x0, const float* keep0,
                                          const float* W0, const float* b0, float* y0, int C1, const float* x1,
                                          const float* keep1, const float* W1, const float* b1, float* y1,
                                          void* stream) {
  SmallLinPtrs pp = {};
  pp.x[0] = x0; pp.keep[0] = keep0; pp.W[0] = W0; pp.b[0] = b0; pp.y[0] = y0; pp.C[0] = C0;
  pp.x[1] = x1; pp.keep[1] = keep1; pp.W[1] = W1; pp.b[1] = b1; pp.y[1] = y1; pp.C[1] = C1;
  return small_linear_fwd_launch(R, K, 2, pp, (hipStream_t)stream);
}

static int small_linear_bwd_launch(int64_t R, int K, int n, const SmallLinPtrs& pp, float* const* dwb, hipStream_t st) {
  for (int i = 0; i < n; ++i) {
    IGCN_REQUIRE(R > 0 && small_linear_ok(K, pp.C[i]), "small_linear: K/4 a power of two <= 64, 1 <= C <= 4 (K=%d C=%d)",
                 K, pp.C[i]);
    IGCN_REQUIRE((((uintptr_t)pp.x[i] | (uintptr_t)pp.W[i] | (uintptr_t)pp.dx[i] | (uintptr_t)pp.keep[i]) & 15) == 0,
                 "small_linear: 16-byte aligned tensors");
  }
  const int rows_per_block = small_linear_rpb(R);
  const int64_t nb = igcn_cdiv(R, rows_per_block);
  hipLaunchKernelGGL(k_small_linear_bwd, dim3((unsigned)nb, (unsigned)n), dim3(256), 0, st, R, K, rows_per_block, pp);
  IGCN_CHECK_LAUNCH("small_linear_bwd");
  for (int i = 0; i < n; ++i) {
    const int w = pp.C[i] * K + pp.C[i];
    const int rc = igcn_launch_reduce_rows_final(pp.partial[i], nb, w, w, dwb[i], st);     // dW | db in one pass
    if (rc) return rc;
  }
  return IGCN_OK;
}

extern "C" int igcn_small_linear_bwd(int64_t R, int K, int C, const float* x, const float* keep, const float* W,
                                     const float* dy, float* dx /* or NULL */,
                                     float* dwb /* [C*K + C]: dW, then db */, float* scratch, void* stream) {
  SmallLinPtrs pp = {};
  pp.x[0] = pp.x[1] = x; pp.keep[0] = pp.keep[1] = keep; pp.W[0] = pp.W[1] = W; pp.dy[0] = pp.dy[1] = dy;
  pp.dx[0] = pp.dx[1] = dx; pp.partial[0] = pp.partial[1] = scratch; pp.C[0] = pp.C[1] = C;
  return small_linear_bwd_launch(R, K, 1, pp, &dwb, (hipStream_t)stream);
}

extern "C" int igcn_small_linear_pair_bwd(int64_t R, int K, int C0, const float* x0, const float* keep0,
                                          const float* W0, const float* dy0, float* dx0, float* dwb0, float* scratch0,
                                          int C1, const float* x1, const float* keep1, const float* W1,
                                          const float* dy1, float* dx1, float* dwb1, float* scratch1, void* stream) {
  SmallLinPtrs pp = {};
  pp.x[0] = x0; pp.keep[0] = keep0; pp.W[0] = W0; pp.dy[0] = dy0; pp.dx[0] = dx0; pp.partial[0] = scratch0; pp.C[0] = C0;
  pp.x[1] = x1; pp.keep[1] = keep1; pp.W[1] = W1; pp.dy[1] = dy1; pp.dx[1] = dx1; pp.partial[1] = scratch1; pp.C[1] = C1;
  float* dwb[2] = {dwb0, dwb1};
  return small_linear_bwd_launch(R, K, 2, pp, dwb, (hipStream_t)stream);
}

// =================================================================================================
// global_mean_pool | global_max_pool | global_add_pool over the nodes of each graph, concatenated
// (kernel/sgcn_img_snp.py:230-235,246-252; PyG 2.0.2 global_*_pool = scatter(x, batch, reduce)).
// Uniform graphs of R nodes (the model's contract): x [G*R, D] -> out [G, 3D] = mean | max | add.
// One thread per (graph, column): the R rows are read with consecutive lanes on consecutive columns (coalesced),
// summed in node order (the reference's sequential scatter order); the max keeps the FIRST arg-max
// (torch-scatter's CPU scatter_max updates on strictly-greater only), saved for the backward.
// =================================================================================================
__global__ void __launch_bounds__(256)
k_graph_pool_fwd(int64_t G, int R, int D, const float* __restrict__ x, float* __restrict__ out,
                 int32_t* __restrict__ arg) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= G * D) return;
  const int64_t g = i / D;
  const int d = (int)(i - g * D);
  const float* p = x + g * R * D + d;
  float s = 0.f, m = p[0];
  int am = 0;
#pragma unroll 4
  for (int r = 0; r < R; ++r) {
    const float v = p[(int64_t)r * D];
    s += v;
    if (v > m || (v != v && !(m != m))) { m = v; am = r; }      // NaN propagates like torch's max
  }
  float* o = out + g * 3 * D;
  o[d] = s / (float)R;
  o[D + d] = m;
  o[2 * D + d] = s;
  arg[i] = am;
}

// dx[g*R + r, d] = dmean[g,d]/R + dadd[g,d] + (r == arg[g,d]) * dmax[g,d]
__global__ void __launch_bounds__(256)
k_graph_pool_bwd(int64_t G, int R, int D, const float* __restrict__ dout, const int32_t* __restrict__ arg,
                 float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= G * R * D) return;
  const int64_t row = i / D;
  const int d = (int)(i - row * D);
  const int64_t g = row / R;
  const int r = (int)(row - g * R);
  const float* o = dout + g * 3 * D;
  float v = o[d] / (float)R + o[2 * D + d];
  if (arg[g * D + d] == r) v += o[D + d];
  dx[i] = v;
}

extern "C" int igcn_graph_pool_fwd(int64_t n_graphs, int nodes_per_graph, int D, const float* x, float* out,
                                   int32_t* argmax, void* stream) {
  IGCN_REQUIRE(n_graphs > 0 && nodes_per_graph > 0 && D > 0, "graph_pool_fwd: bad sizes");
  hipLaunchKernelGGL(k_graph_pool_fwd, dim3((unsigned)igcn_cdiv(n_graphs * D, 256)), dim3(256), 0,
                     (hipStream_t)stream, n_graphs, nodes_per_graph, D, x, out, argmax);
  IGCN_CHECK_LAUNCH("graph_pool_fwd");
  return IGCN_OK;
}

extern "C" int igcn_graph_pool_bwd(int64_t n_graphs, int nodes_per_graph, int D, const float* dout,
                                   const int32_t* argmax, float* dx, void* stream) {
  IGCN_REQUIRE(n_graphs > 0 && nodes_per_graph > 0 && D > 0, "graph_pool_bwd: bad sizes");
  hipLaunchKernelGGL(k_graph_pool_bwd, dim3((unsigned)igcn_cdiv(n_graphs * nodes_per_graph * D, 256)), dim3(256), 0,
                     (hipStream_t)stream, n_graphs, nodes_per_graph, D, dout, argmax, dx);
  IGCN_CHECK_LAUNCH("graph_pool_bwd");
  return IGCN_OK;
}


// =================================================================================================
// Dropout masks of every site of a forward pass in ONE launch (the reference draws them site by site: nn.Dropout
// / nn.Dropout2d / F.dropout at kernel/go_model.py:104,113,128,136,143 and kernel/sgcn_img_snp.py:289,299 — a dozen
// library launches per step forward and backward).  out[i] = 0 with probability p(segment of i), else 1/(1-p): the
// {0, 1/(1-p)} factors the consumers multiply by (igcn_nodes_ln_*, igcn_node_linear_bn_*, igcn_bn1d_*,
// igcn_small_linear_* take them as `keep`).  Counter-based generator: a 32-bit integer hash of (index, stream counter);
// `state` on the device holds the counter and arrival words — the LAST workgroup to finish advances the counter, so
// every replay of a captured launch draws fresh masks without a host round trip.
// =================================================================================================
#include "dropout.h"
__global__ void __launch_bounds__(256)
k_dropout_masks(int64_t total, DropSegs segs, unsigned long long* __restrict__ state, float* __restrict__ out,
                DropCounters cnt) {
  dropout_masks_body(blockIdx.x, gridDim.x, total, segs, state, out, cnt);
}

extern "C" int igcn_dropout_state_words(void) { return DM_WORDS; }
extern "C" int igcn_dropout_max_segments(void) { return DM_MAXSEG; }

int igcn_dropout_job(DropJob& job, const char* who, int64_t total, int n_segments, const int64_t* seg_end, const float* seg_p,
                     void* state, float* out, int n_counters, int64_t* const* counters, int64_t counter_inc) {
  IGCN_REQUIRE(n_counters >= 0 && n_counters <= DM_MAXCNT && (n_counters == 0 || counters != nullptr),
               "%s: at most %d counters", who, DM_MAXCNT);
  job = DropJob{};
  job.cnt.n = n_counters;
  job.cnt.inc = counter_inc;
  for (int k = 0; k < n_counters; ++k) job.cnt.c[k] = (long long*)counters[k];
  IGCN_REQUIRE(total > 0 && n_segments >= 1 && n_segments <= DM_MAXSEG && state != nullptr &&
               ((uintptr_t)out & 15) == 0, "%s: 1..%d segments, 16-byte aligned output", who, DM_MAXSEG);
  job.sg.n = n_segments;
  for (int k = 0; k < n_segments; ++k) {
    IGCN_REQUIRE(seg_p[k] >= 0.f && seg_p[k] < 1.f && seg_end[k] <= total, "%s: bad segment %d", who, k);
    job.sg.end[k] = seg_end[k];
    job.sg.p[k] = seg_p[k];
  }
  IGCN_REQUIRE(job.sg.end[n_segments - 1] == total, "%s: the segments must cover [0, total)", who);
  for (int k = 0; k + 1 < n_segments; ++k)
    IGCN_REQUIRE(job.sg.end[k] % 4 == 0, "%s: interior segment ends must be multiples of 4", who);
  int64_t blocks = igcn_cdiv(total, 1024);
  job.blocks = (unsigned)(blocks > 2048 ? 2048 : blocks);
  job.total = total;
  job.state = (unsigned long long*)state;
  job.out = out;
  return IGCN_OK;
}

int igcn_dropout_launch(const DropJob& job, hipStream_t st) {
  hipLaunchKernelGGL(k_dropout_masks, dim3(job.blocks), dim3(256), 0, st, job.total, job.sg, job.state, job.out, job.cnt);
  IGCN_CHECK_LAUNCH("dropout_masks");
  return IGCN_OK;
}

extern "C" int igcn_dropout_masks(int64_t total, int n_segments, const int64_t* seg_end /*HOST*/,
                                  const float* seg_p /*HOST*/, void* state /*device uint64[2]*/, float* out,
                                  int n_counters, int64_t* const* counters /*HOST array of device pointers*/,
                                  int64_t counter_inc, void* stream) {
  DropJob job;
  const int rc = igcn_dropout_job(job, "dropout_masks", total, n_segments, seg_end, seg_p, state, out, n_counters, counters,
                                  counter_inc);
  if (rc) return rc;
  return igcn_dropout_launch(job, (hipStream_t)stream);
}
