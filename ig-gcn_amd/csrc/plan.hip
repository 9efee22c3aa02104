// Graph plan: stable grouping of the batch's edges by target and by source node (include/igcn.h).
// The sort is rocPRIM's LSD radix sort (stable), restricted to the bits a node id needs; everything
// else is hand-written.  Runs once per batch; every later kernel reuses its output.
#include <stdarg.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

static thread_local char g_err[512] = "";

void igcn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* igcn_last_error(void) { return g_err; }
extern "C" int igcn_version(void) { return 100; }

__global__ void k_reduce_rows(const float* __restrict__ partial, int64_t rows, int64_t ld, int n,
                              float* __restrict__ out, int accumulate) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  float t = 0.f;
  for (int64_t r = 0; r < rows; ++r) t += partial[r * ld + j];
  out[j] = accumulate ? out[j] + t : t;
}

// many rows, few columns: one 256-thread block per column, fixed reduction tree (deterministic)
__global__ void __launch_bounds__(256)
k_reduce_rows_par(const float* __restrict__ partial, int64_t rows, int64_t ld, float* __restrict__ out,
                  int accumulate) {
  __shared__ float red[16];
  const int j = blockIdx.x;
  float t = 0.f;
  for (int64_t r = threadIdx.x; r < rows; r += 256) t += partial[r * ld + j];
  t = block_sum_all(t, red);
  if (threadIdx.x == 0) out[j] = accumulate ? out[j] + t : t;
}

int igcn_launch_reduce_rows(const float* partial, int64_t rows, int64_t ld, int n, float* out, int accumulate,
                            hipStream_t st) {
  if (n <= 0) return IGCN_OK;
  if (rows > 32 && n <= 4096) {
    hipLaunchKernelGGL(k_reduce_rows_par, dim3((unsigned)n), dim3(256), 0, st, partial, rows, ld, out, accumulate);
    IGCN_CHECK_LAUNCH("reduce_rows_par");
    return IGCN_OK;
  }
  hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)igcn_cdiv(n, 64)), dim3(64), 0, st, partial, rows, ld, n, out,
                     accumulate);
  IGCN_CHECK_LAUNCH("reduce_rows");
  return IGCN_OK;
}

// ---- kernels -----------------------------------------------------------------------------------
// Coalesced read of the int64 edge_index rows; int32 copies + identity values for the sorts;
// loop_edge via atomicMax (max edge id == "last stored loop wins", order independent => deterministic).
__global__ void k_plan_split(int64_t n_nodes, int64_t n_edges, const int64_t* __restrict__ ei, int32_t* __restrict__ src32,
                             int32_t* __restrict__ dst32, int32_t* __restrict__ iota,
                             int32_t* __restrict__ loop_edge) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_edges) return;
  int32_t s = (int32_t)ei[k], d = (int32_t)ei[n_edges + k];
  src32[k] = s;
  dst32[k] = d;
  iota[k] = (int32_t)k;
  if (s == d && s >= 0 && s < n_nodes) atomicMax(&loop_edge[s], (int32_t)k);
}

__global__ void k_fill_i32(int64_t n, int32_t* p, int32_t v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ptr[i] = first position in the sorted key array whose key >= i   (i in [0, n_nodes]).
__global__ void k_plan_ptr(int64_t n_nodes, int64_t n_edges, const int32_t* __restrict__ sorted_keys,
                           int32_t* __restrict__ ptr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_nodes) return;
  int64_t lo = 0, hi = n_edges;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (sorted_keys[mid] < (int32_t)i) lo = mid + 1; else hi = mid;
  }
  ptr[i] = (int32_t)lo;
}

static int key_bits(int64_t n_nodes) {
  int b = 1;
  while (((int64_t)1 << b) < n_nodes) ++b;
  return b;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t sort_temp_bytes(int64_t n_nodes, int64_t n_edges) {
  size_t bytes = 0;
  int32_t* nul = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, nul, nul, (size_t)n_edges, 0u,
                                  (unsigned)key_bits(n_nodes), (hipStream_t)0);
  return bytes;
}

extern "C" size_t igcn_graph_plan_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
  if (n_edges <= 0) return 256;
  // iota [E] + sorted keys [E] + rocPRIM temp
  return 2 * align256((size_t)n_edges * 4) + align256(sort_temp_bytes(n_nodes, n_edges)) + 256;
}

extern "C" int igcn_graph_plan_build(int64_t n_nodes, int64_t n_edges, const int64_t* edge_index, int32_t* src32,
                                     int32_t* dst32, int32_t* tgt_ptr, int32_t* tgt_perm, int32_t* src_ptr,
                                     int32_t* src_perm, int32_t* loop_edge, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  IGCN_REQUIRE(n_nodes > 0 && n_nodes < ((int64_t)1 << 31) && n_edges >= 0 && n_edges < ((int64_t)1 << 31),
               "graph_plan_build: n_nodes=%lld n_edges=%lld out of int32 range", (long long)n_nodes,
               (long long)n_edges);
  IGCN_REQUIRE(workspace_bytes >= igcn_graph_plan_workspace_bytes(n_nodes, n_edges),
               "graph_plan_build: workspace too small");
  const int T = 256;
  hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)igcn_cdiv(n_nodes, T)), dim3(T), 0, st, n_nodes, loop_edge, -1);
  if (n_edges == 0) {
    hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)igcn_cdiv(n_nodes + 1, T)), dim3(T), 0, st, n_nodes + 1, tgt_ptr, 0);
    hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)igcn_cdiv(n_nodes + 1, T)), dim3(T), 0, st, n_nodes + 1, src_ptr, 0);
    IGCN_CHECK_LAUNCH("graph_plan_build(empty)");
    return IGCN_OK;
  }
  char* ws = (char*)workspace;
  int32_t* iota = (int32_t*)ws;
  int32_t* skeys = (int32_t*)(ws + align256((size_t)n_edges * 4));
  void* temp = ws + 2 * align256((size_t)n_edges * 4);
  size_t temp_bytes = sort_temp_bytes(n_nodes, n_edges);
  const unsigned bits = (unsigned)key_bits(n_nodes);

  hipLaunchKernelGGL(k_plan_split, dim3((unsigned)igcn_cdiv(n_edges, T)), dim3(T), 0, st, n_nodes, n_edges,
                     edge_index, src32, dst32, iota, loop_edge);
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, (const int32_t*)dst32, skeys, (const int32_t*)iota,
                                           tgt_perm, (size_t)n_edges, 0u, bits, st);
  if (e != hipSuccess) { igcn_set_error("graph_plan_build: sort(dst): %s", hipGetErrorString(e)); return IGCN_ERR_LAUNCH; }
  hipLaunchKernelGGL(k_plan_ptr, dim3((unsigned)igcn_cdiv(n_nodes + 1, T)), dim3(T), 0, st, n_nodes, n_edges, skeys,
                     tgt_ptr);
  e = rocprim::radix_sort_pairs(temp, temp_bytes, (const int32_t*)src32, skeys, (const int32_t*)iota, src_perm,
                                (size_t)n_edges, 0u, bits, st);
  if (e != hipSuccess) { igcn_set_error("graph_plan_build: sort(src): %s", hipGetErrorString(e)); return IGCN_ERR_LAUNCH; }
  hipLaunchKernelGGL(k_plan_ptr, dim3((unsigned)igcn_cdiv(n_nodes + 1, T)), dim3(T), 0, st, n_nodes, n_edges, skeys,
                     src_ptr);
  IGCN_CHECK_LAUNCH("graph_plan_build");
  return IGCN_OK;
}

// Plan of `copies` disjoint copies of the same batch (nodes g*N + i, edges g*E + k): used to run the plain and the
// masked forward pass of a train step as ONE block-diagonal problem without sorting again.
__global__ void k_plan_replicate(int64_t n, int64_t e, int copies, const int32_t* __restrict__ src32,
                                 const int32_t* __restrict__ dst32, const int32_t* __restrict__ tgt_ptr,
                                 const int32_t* __restrict__ tgt_perm, const int32_t* __restrict__ src_ptr,
                                 const int32_t* __restrict__ src_perm, const int32_t* __restrict__ loop_edge,
                                 int32_t* __restrict__ o_src32, int32_t* __restrict__ o_dst32,
                                 int32_t* __restrict__ o_tgt_ptr, int32_t* __restrict__ o_tgt_perm,
                                 int32_t* __restrict__ o_src_ptr, int32_t* __restrict__ o_src_perm,
                                 int32_t* __restrict__ o_loop_edge) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t tot_e = e * copies, tot_n = n * copies;
  if (i < tot_e) {
    const int64_t g = i / e, k = i - g * e;
    o_src32[i] = src32[k] + (int32_t)(g * n);
    o_dst32[i] = dst32[k] + (int32_t)(g * n);
    o_tgt_perm[i] = tgt_perm[k] + (int32_t)(g * e);
    o_src_perm[i] = src_perm[k] + (int32_t)(g * e);
  }
  if (i < tot_n) {
    const int64_t g = i / n, v = i - g * n;
    o_tgt_ptr[i] = tgt_ptr[v] + (int32_t)(g * e);
    o_src_ptr[i] = src_ptr[v] + (int32_t)(g * e);
    const int32_t le = loop_edge[v];
    o_loop_edge[i] = le >= 0 ? le + (int32_t)(g * e) : -1;
  }
  if (i == tot_n) {
    o_tgt_ptr[tot_n] = (int32_t)tot_e;
    o_src_ptr[tot_n] = (int32_t)tot_e;
  }
}

extern "C" int igcn_graph_plan_replicate(int64_t n_nodes, int64_t n_edges, int copies, const int32_t* src32,
                                         const int32_t* dst32, const int32_t* tgt_ptr, const int32_t* tgt_perm,
                                         const int32_t* src_ptr, const int32_t* src_perm, const int32_t* loop_edge,
                                         int32_t* o_src32, int32_t* o_dst32, int32_t* o_tgt_ptr, int32_t* o_tgt_perm,
                                         int32_t* o_src_ptr, int32_t* o_src_perm, int32_t* o_loop_edge,
                                         void* stream) {
  IGCN_REQUIRE(copies >= 1 && n_nodes > 0 && n_nodes * copies < ((int64_t)1 << 31) &&
               n_edges * copies < ((int64_t)1 << 31), "graph_plan_replicate: sizes out of int32 range");
  const int64_t work = (n_edges * copies > n_nodes * copies + 1) ? n_edges * copies : n_nodes * copies + 1;
  hipLaunchKernelGGL(k_plan_replicate, dim3((unsigned)igcn_cdiv(work, 256)), dim3(256), 0, (hipStream_t)stream,
                     n_nodes, n_edges, copies, src32, dst32, tgt_ptr, tgt_perm, src_ptr, src_perm, loop_edge, o_src32,
                     o_dst32, o_tgt_ptr, o_tgt_perm, o_src_ptr, o_src_perm, o_loop_edge);
  IGCN_CHECK_LAUNCH("graph_plan_replicate");
  return IGCN_OK;
}
