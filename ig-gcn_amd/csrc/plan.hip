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
#pragma unroll 8
  for (int64_t r = 0; r < rows; ++r) t += partial[r * ld + j];     // independent loads: several in flight
  out[j] = accumulate ? out[j] + t : t;
}

// many rows, few columns: one 256-thread block per column, fixed reduction tree (deterministic)
__global__ void __launch_bounds__(256)
k_reduce_rows_par(const float* __restrict__ partial, int64_t rows, int64_t ld, float* __restrict__ out,
                  int accumulate) {
  __shared__ float red[16];
  const int j = blockIdx.x;
  float t = 0.f;
#pragma unroll 4
  for (int64_t r = threadIdx.x; r < rows; r += 256) t += partial[r * ld + j];
  t = block_sum_all(t, red);
  if (threadIdx.x == 0) out[j] = accumulate ? out[j] + t : t;
}

// out[j] = sum_r partial[j * rows + r]: the summands of one output are CONTIGUOUS (coalesced), one block per output
__global__ void __launch_bounds__(1024)
k_reduce_contig(const float* __restrict__ partial, int64_t rows, float* __restrict__ out) {
  __shared__ float red[16];
  const float* p = partial + (int64_t)blockIdx.x * rows;
  float t = 0.f;
#pragma unroll 8
  for (int64_t r = threadIdx.x; r < rows; r += 1024) t += p[r];
  t = block_sum_all(t, red);
  if (threadIdx.x == 0) out[blockIdx.x] = t;
}

int igcn_launch_reduce_contig(const float* partial, int64_t rows, int n, float* out, hipStream_t st) {
  if (n <= 0) return IGCN_OK;
  hipLaunchKernelGGL(k_reduce_contig, dim3((unsigned)n), dim3(1024), 0, st, partial, rows, out);
  IGCN_CHECK_LAUNCH("reduce_contig");
  return IGCN_OK;
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

// ---- segmented build: one workgroup per graph of the batch ------------------------------------------------
// PyG batches are block diagonal: graph g owns the contiguous node range [node_ptr[g], node_ptr[g+1]) and the
// contiguous edge range [edge_ptr[g], edge_ptr[g+1]).  For small graphs (a 90-ROI brain graph has 270 edges) the
// whole stable grouping fits in LDS: integer histogram -> scan -> each node's thread walks the graph's edges in
// stored order and appends its own (so the order inside a group is the stored order: bit-identical to a stable
// sort), for targets and for sources.  One launch replaces two device-wide radix sorts (~14 launches).
#define SEG_MAXE 4096
#define SEG_MAXN 1024
__global__ void __launch_bounds__(256)
k_plan_segmented(int64_t n_nodes, int64_t n_edges, int n_graphs, const int64_t* __restrict__ ei,
                 const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ edge_ptr,
                 int32_t* __restrict__ src32, int32_t* __restrict__ dst32, int32_t* __restrict__ tgt_ptr,
                 int32_t* __restrict__ tgt_perm, int32_t* __restrict__ src_ptr, int32_t* __restrict__ src_perm,
                 int32_t* __restrict__ loop_edge, int32_t* __restrict__ status) {
  __shared__ int16_t ls[SEG_MAXE], ld[SEG_MAXE];        // local (graph-relative) endpoints
  __shared__ int32_t ct[SEG_MAXN + 1], cs[SEG_MAXN + 1], lp[SEG_MAXN];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int64_t nb = node_ptr[g], eb = edge_ptr[g];
  const int nn = (int)(node_ptr[g + 1] - nb), ne = (int)(edge_ptr[g + 1] - eb);
  if (nn > SEG_MAXN || ne > SEG_MAXE || nn < 0 || ne < 0) {       // host checked the maxima; refuse, don't corrupt
    if (tid == 0) atomicExch(status, 1);
    return;
  }
  for (int i = tid; i <= nn; i += 256) { ct[i] = 0; cs[i] = 0; }
  for (int i = tid; i < nn; i += 256) lp[i] = -1;
  __syncthreads();
  bool bad = false;
  for (int k = tid; k < ne; k += 256) {
    const int64_t s = ei[eb + k] - nb, d = ei[n_edges + eb + k] - nb;
    src32[eb + k] = (int32_t)(s + nb);
    dst32[eb + k] = (int32_t)(d + nb);
    if (s < 0 || s >= nn || d < 0 || d >= nn) { bad = true; ls[k] = 0; ld[k] = 0; continue; }
    ls[k] = (int16_t)s;
    ld[k] = (int16_t)d;
    atomicAdd(&ct[d + 1], 1);
    atomicAdd(&cs[s + 1], 1);
    if (s == d) atomicMax(&lp[s], k);
  }
  if (bad) atomicExch(status, 2);                                  // edge leaves its graph: not a PyG batch
  __syncthreads();
  if (tid < 64) {
    // exclusive -> inclusive scan of both histograms by one wave: every lane owns a run of consecutive entries
    // (a serial scan by one thread is a chain of ~2 nn dependent LDS round trips)
    const int per = (nn + 64) / 64, lo = tid * per, hi = min(nn + 1, lo + per);
    int st = 0, ss = 0;
    for (int i = lo; i < hi; ++i) { st += ct[i]; ss += cs[i]; }
    int pt = st, ps = ss;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int a = __shfl_up(pt, o, 64), b = __shfl_up(ps, o, 64);
      if (tid >= o) { pt += a; ps += b; }
    }
    int rt = pt - st, rs = ps - ss;                                 // sums of the lanes before this one
    for (int i = lo; i < hi; ++i) {
      rt += ct[i]; rs += cs[i];
      ct[i] = rt; cs[i] = rs;
    }
  }
  __syncthreads();
  for (int i = tid; i < nn; i += 256) {
    tgt_ptr[nb + i] = (int32_t)(eb + ct[i]);
    src_ptr[nb + i] = (int32_t)(eb + cs[i]);
    loop_edge[nb + i] = lp[i] >= 0 ? (int32_t)(eb + lp[i]) : -1;
  }
  // stable placement, one thread per edge: its slot inside its group = the number of EARLIER edges with the same
  // key.  All lanes read the same ld[j] / ls[j] (an LDS broadcast), so the scan is a divergence-free stream of
  // compare-and-count steps for both groupings at once.
  for (int k = tid; k < ne; k += 256) {
    const int16_t kd = ld[k], ks = ls[k];
    int rd = 0, rs = 0;
#pragma unroll 8
    for (int j = 0; j < k; ++j) {
      rd += (ld[j] == kd);
      rs += (ls[j] == ks);
    }
    tgt_perm[eb + ct[kd] + rd] = (int32_t)(eb + k);
    src_perm[eb + cs[ks] + rs] = (int32_t)(eb + k);
  }
  if (g == n_graphs - 1 && tid == 0) {
    tgt_ptr[n_nodes] = (int32_t)n_edges;
    src_ptr[n_nodes] = (int32_t)n_edges;
  }
}

extern "C" int igcn_graph_plan_build_segmented(int64_t n_nodes, int64_t n_edges, int n_graphs,
                                               const int64_t* edge_index, const int64_t* node_ptr,
                                               const int64_t* edge_ptr, int64_t max_nodes_per_graph,
                                               int64_t max_edges_per_graph, int32_t* src32, int32_t* dst32,
                                               int32_t* tgt_ptr, int32_t* tgt_perm, int32_t* src_ptr,
                                               int32_t* src_perm, int32_t* loop_edge, int32_t* status,
                                               void* stream) {
  IGCN_REQUIRE(n_graphs > 0 && n_nodes > 0 && n_nodes < ((int64_t)1 << 31) && n_edges < ((int64_t)1 << 31),
               "graph_plan_build_segmented: bad sizes");
  if (max_nodes_per_graph > SEG_MAXN || max_edges_per_graph > SEG_MAXE) {
    igcn_set_error("graph_plan_build_segmented: graphs too large for the LDS path (max %d nodes / %d edges per graph)",
                   SEG_MAXN, SEG_MAXE);
    return IGCN_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(k_plan_segmented, dim3(n_graphs), dim3(256), 0, (hipStream_t)stream, n_nodes, n_edges, n_graphs,
                     edge_index, node_ptr, edge_ptr, src32, dst32, tgt_ptr, tgt_perm, src_ptr, src_perm, loop_edge,
                     status);
  IGCN_CHECK_LAUNCH("graph_plan_build_segmented");
  return IGCN_OK;
}
