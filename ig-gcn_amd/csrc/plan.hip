// Graph plan: stable grouping of the batch's edges by target and by source node (include/igcn.h).
// Three builders with bit-identical output, all hand-written and hipGraph-capturable: one workgroup per graph in LDS
// (small graphs), a tiled one-pass counting sort per graph (<= 1024 nodes per graph, any number of edges) and a
// general LSD radix sort (arbitrary edge_index).  Runs once per batch; every later kernel reuses its output.
#include <stdarg.h>

#include "common.h"
#include "loss_final.h"

static thread_local char g_err[512] = "";

void igcn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* igcn_last_error(void) { return g_err; }
extern "C" int igcn_version(void) { return IGCN_ABI_VERSION; }

unsigned g_igcn_options = 0;
int g_igcn_gemm_bn_cap = 0;
int g_igcn_attn_chunk_rows = 0;
extern "C" int igcn_configure(unsigned options, int gemm_bn_cap, int attn_chunk_rows) {
  g_igcn_options = options;
  g_igcn_gemm_bn_cap = gemm_bn_cap;
  g_igcn_attn_chunk_rows = attn_chunk_rows;
  return IGCN_OK;
}

__global__ void k_reduce_rows(const float* __restrict__ partial, int64_t rows, int64_t ld, int n,
                              float* __restrict__ out, int accumulate) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  float t = 0.f;
#pragma unroll 8
  for (int64_t r = 0; r < rows; ++r) t += partial[r * ld + j];     // independent loads: several in flight
  out[j] = accumulate ? out[j] + t : t;
}

// many rows AND many columns (split-K / per-workgroup partial rows of wide gradients): a thread per column walking all
// rows is a chain of `rows` dependent-latency loads (128 rows: 16 batches of 8 = the whole 16 us of the launch).  Four
// row groups per column (rows r = g mod 4), combined in the fixed order ((s0 + s1) + s2) + s3: 64 columns x 4 groups per
// 256-thread workgroup.  Shared by the stand-alone and the deferred reduction so that both give the same bits.
#define RR_WIDE_ROWS 32
// Which form sums `rows` partial rows of n columns — ONE rule for the stand-alone launches, the deferred launch and its
// workgroup count.  0: a thread per column, rows in order (few rows).  Tall (> 32 rows): 1 = a wave per column
// (n < 16), 2 = 16 x 16 tiles (n <= 4096, or >= 256 rows whatever the width: with four row groups a thread would walk
// rows / 4 dependent-latency loads — 512 per-sample rows of a LayerNorm's affine partials = 16 batches of 8), 3 = four
// row groups per column (32 < rows < 256 and wide: fewer, fatter workgroups).
__host__ __device__ inline int rr_form(int64_t rows, int n) {
  if (!(rows > RR_WIDE_ROWS)) return 0;
  if (n < 16) return 1;
  return (n <= 4096 || rows >= 256) ? 2 : 3;
}
__device__ __forceinline__ void reduce_cols_grouped(const float* __restrict__ partial, int64_t rows, int64_t ld, int n,
                                                    float* __restrict__ out, int accumulate, int64_t block,
                                                    float (*lds)[64]) {
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t j = block * 64 + c;
  float t = 0.f;
  if (j < n) {
#pragma unroll 8
    for (int64_t r = g; r < rows; r += 4) t += partial[r * ld + j];
  }
  lds[g][c] = t;
  __syncthreads();
  if (g == 0 && j < n) {
    const float v = ((lds[0][c] + lds[1][c]) + lds[2][c]) + lds[3][c];
    out[j] = accumulate ? out[j] + v : v;
  }
}
__global__ void __launch_bounds__(256)
k_reduce_rows_grouped(const float* __restrict__ partial, int64_t rows, int64_t ld, int n, float* __restrict__ out,
                      int accumulate) {
  __shared__ float lds[4][64];
  reduce_cols_grouped(partial, rows, ld, n, out, accumulate, blockIdx.x, lds);
}

// many rows, few columns: one WAVE per column (four columns per 256-thread workgroup): lanes stride the rows, fixed
// xor tree — no LDS, no barrier.  [A workgroup per column with a block-wide tree: 5 000 one-load workgroups in the
// deferred launch of a train step, whose cost was their dispatch.]  Shared by the stand-alone and the deferred form.
__device__ __forceinline__ void reduce_col_wave(const float* __restrict__ partial, int64_t rows, int64_t ld, int n,
                                                float* __restrict__ out, int accumulate, int64_t block) {
  const int lane = threadIdx.x & 63;
  const int64_t j = block * 4 + (threadIdx.x >> 6);
  if (j >= n) return;                                 // wave-uniform
  float t = 0.f;
#pragma unroll 4
  for (int64_t r = lane; r < rows; r += 64) t += partial[r * ld + j];
  t = wave_sum(t);
  if (lane == 0) out[j] = accumulate ? out[j] + t : t;
}
__global__ void __launch_bounds__(256)
k_reduce_rows_par(const float* __restrict__ partial, int64_t rows, int64_t ld, int n, float* __restrict__ out,
                  int accumulate) {
  reduce_col_wave(partial, rows, ld, n, out, accumulate, blockIdx.x);
}

// many rows, 16..4096 columns: 16 columns x 16 row groups per 256-thread workgroup — a wave's load instruction covers
// 4 rows x 64 contiguous bytes (the wave-per-column form above touches 64 different cache lines per instruction and
// uses 4 bytes of each), a thread sums rows r = g mod 16 (8 loads in flight), the 16 group sums of a column are
// combined through LDS in a fixed pairwise tree.  Shared by the stand-alone and the deferred form.
// Which 16-column tile a workgroup takes.  A tile's rows are 64-byte segments, i.e. HALF of the 128-byte lines the L2
// fetches, and the hardware deals consecutive workgroups to the eight XCDs in turn: with tile = workgroup every line was
// fetched by two XCDs — k_multi_reduce read 92 MB for 53 MB of queued partials (PMC, profiles/r04_step_traffic_full.csv)
// at the HBM rate.  Within every 64 workgroups, the eight that land on XCD x (x, x + 8, ...) take eight CONSECUTIVE
// tiles: 7 of 8 shared lines now meet in one L2.  (Which workgroup sums a column does not change the sum.)
__device__ __forceinline__ int64_t tile16_of(int64_t block, int n) {
  const int64_t base = block & ~(int64_t)63, tiles = ((int64_t)n + 15) >> 4;
  if (base + 64 > tiles) return block;                  // the incomplete last group: as dealt
  const int o = (int)(block & 63);
  return base + ((o & 7) << 3) + (o >> 3);
}
__device__ __forceinline__ void reduce_cols_tile16(const float* __restrict__ partial, int64_t rows, int64_t ld, int n,
                                                   float* __restrict__ out, int accumulate, int64_t block,
                                                   float (*lds)[64]) {
  float* s = &lds[0][0];                               // [16 groups][16 columns]
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t j = tile16_of(block, n) * 16 + c;
  float t = 0.f;
  if (j < n) {
#pragma unroll 8
    for (int64_t r = g; r < rows; r += 16) t += partial[r * ld + j];
  }
  s[g * 16 + c] = t;
  __syncthreads();
  if (g == 0 && j < n) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = s[k * 16 + c];
#pragma unroll
    for (int w = 1; w < 16; w *= 2)
#pragma unroll
      for (int k = 0; k < 16; k += 2 * w) v[k] += v[k + w];
    out[j] = accumulate ? out[j] + v[0] : v[0];
  }
}
__global__ void __launch_bounds__(256)
k_reduce_rows_tile16(const float* __restrict__ partial, int64_t rows, int64_t ld, int n, float* __restrict__ out,
                     int accumulate) {
  __shared__ float lds[4][64];
  reduce_cols_tile16(partial, rows, ld, n, out, accumulate, blockIdx.x, lds);
}

// gridDim.y independent sums of the same shape (the slab sums of a batched split-K product)
__global__ void __launch_bounds__(256)
k_reduce_rows_tile16_batched(const float* __restrict__ partial, int64_t rows, int64_t ld, int n, float* __restrict__ out,
                             int64_t p_batch, int64_t o_batch) {
  __shared__ float lds[4][64];
  reduce_cols_tile16(partial + (int64_t)blockIdx.y * p_batch, rows, ld, n, out + (int64_t)blockIdx.y * o_batch, 0,
                     blockIdx.x, lds);
}

int igcn_launch_reduce_rows_batched(const float* partial, int64_t rows, int64_t ld, int n, float* out, int batch,
                                    int64_t p_batch, int64_t o_batch, hipStream_t st) {
  if (n <= 0 || batch <= 0) return IGCN_OK;
  hipLaunchKernelGGL(k_reduce_rows_tile16_batched, dim3((unsigned)igcn_cdiv(n, 16), (unsigned)batch), dim3(256), 0, st,
                     partial, rows, ld, n, out, p_batch, o_batch);
  IGCN_CHECK_LAUNCH("reduce_rows_batched");
  return IGCN_OK;
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
  const int form = rr_form(rows, n);
  if (form == 1 || form == 2) {
    if (form == 2) {
      hipLaunchKernelGGL(k_reduce_rows_tile16, dim3((unsigned)igcn_cdiv(n, 16)), dim3(256), 0, st, partial, rows, ld, n,
                         out, accumulate);
      IGCN_CHECK_LAUNCH("reduce_rows_tile16");
      return IGCN_OK;
    }
    hipLaunchKernelGGL(k_reduce_rows_par, dim3((unsigned)igcn_cdiv(n, 4)), dim3(256), 0, st, partial, rows, ld, n, out,
                       accumulate);
    IGCN_CHECK_LAUNCH("reduce_rows_par");
    return IGCN_OK;
  }
  if (form == 3) {
    hipLaunchKernelGGL(k_reduce_rows_grouped, dim3((unsigned)igcn_cdiv(n, 64)), dim3(256), 0, st, partial, rows, ld, n,
                       out, accumulate);
    IGCN_CHECK_LAUNCH("reduce_rows_grouped");
    return IGCN_OK;
  }
  hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)igcn_cdiv(n, 64)), dim3(64), 0, st, partial, rows, ld, n, out,
                     accumulate);
  IGCN_CHECK_LAUNCH("reduce_rows");
  return IGCN_OK;
}

// ---- deferred reductions -------------------------------------------------------------------------------------
// A backward pass ends ~30 of its kernels with a small "sum the block partials" launch whose output is a parameter
// gradient that nothing reads before the optimiser.  With igcn_reduce_defer(1) those launches are queued (the
// partial buffers stay alive on the caller's side) and igcn_reduce_flush performs all of them in ONE launch whose
// table travels by value in the kernel arguments — same arithmetic and summation order as the stand-alone kernels.
#define MRQ_MAX 48
struct ReduceEntry {
  const float* partial;
  float* out;
  int64_t rows, ld;
  int n;
  int nptr;                                          // > 0: the rows are SEPARATE buffers (partial, more[0..nptr-1]),
  const float* more[3];                              //      summed in that order (igcn_sum_n_final)
};
struct ReduceTable {
  ReduceEntry e[MRQ_MAX];
  int start[MRQ_MAX + 1];                            // first workgroup of every entry (flat grid: no idle workgroups)
  int count;
};

// the wide in-order form may take four columns per thread (same sums, column by column)
__host__ __device__ inline bool rq_wide_vec(const ReduceEntry& e) {
  return e.nptr == 0 && !(e.rows > 32) && e.n >= 1024 && (e.n & 3) == 0 && (e.ld & 3) == 0 &&
         (((uintptr_t)e.partial | (uintptr_t)e.out) & 15) == 0;
}

__global__ void __launch_bounds__(256) k_multi_reduce(ReduceTable t, int32_t* tick) {
  __shared__ float lds[4][64];
  if (tick && blockIdx.x == 0 && threadIdx.x == 0) *tick += 1;      // the optimiser's step counter (igcn_reduce_flush_tick)
  // flat grid: workgroup -> (entry, block inside the entry).  [A (max blocks) x (entries) grid launched 57 000
  // workgroups for 11 000 with work.]
  // entry = number of starts at or below blockIdx.x (start[] ascending, unused slots INT_MAX): the whole prefix table is
  // fetched from the kernel arguments in ONE batch of scalar loads and compared in registers.  [A binary search is six
  // DEPENDENT scalar loads before the entry itself can be fetched, a linear walk up to 40.]
  int ei = 0;
#pragma unroll
  for (int i = 1; i < MRQ_MAX; ++i) ei += (int)blockIdx.x >= t.start[i] ? 1 : 0;
  const ReduceEntry e = t.e[ei];
  const int64_t blk = (int64_t)blockIdx.x - t.start[ei];
  if (e.nptr == -2) {                                // the train step's loss value from its partial sums (loss_final.h):
    // partial = [rows][4] head-loss sums, more[0] = Gram partials [ld & 0xffffffff][4], more[1] = regulariser partials
    // [ld >> 32], more[2] = weights [10], out = loss | terms[7]; one workgroup
    loss_final_body(e.partial, (int)e.rows, e.more[0], (int)(e.ld & 0xffffffff), e.more[1], (int)(e.ld >> 32), e.more[2], e.out,
                    &lds[0][0]);
    return;
  }
  if (e.nptr < 0) {                                  // GO attention: parameter gradients from the block partials
    // (go_attn_finish_output, common.h — shared with go.hip's k_go_attn_bwd_finish; output j = blk, ld = FIN | FOUT << 8)
    go_attn_finish_output(e.partial, e.rows, (int)(e.ld & 255), (int)(e.ld >> 8), e.more[0], e.more[1], e.out, (int)blk,
                          &lds[0][0]);
    return;
  }
  if (e.nptr > 0) {                                  // a leaf's gradient = sum of its consumers' gradients (k_sum_n)
    const int64_t j4 = (blk * 256 + threadIdx.x) * 4;          // 16 bytes per lane (buffers 16-byte aligned)
    // more[] with CONSTANT indices only.  [A `for (k < nptr) ... e.more[k]` loop indexes the entry dynamically: the
    // compiler then keeps the whole entry addressable, and EVERY path of the kernel — the tiles, the row groups —
    // re-reads its fields around each load instead of holding them in registers: a 512 x 2400 tile sum took 13 us in
    // this launch and 4.7 us with this branch compiled out (tools/reduce_trace.py); same instructions otherwise.]
    const float* m0 = e.more[0];
    const float* m1 = e.more[1];
    const float* m2 = e.more[2];
    const int np = e.nptr;
    if (j4 + 3 < e.n) {
      float4 a = *reinterpret_cast<const float4*>(e.partial + j4);
      {
        const float4 b = *reinterpret_cast<const float4*>(m0 + j4);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      if (np > 1) {
        const float4 b = *reinterpret_cast<const float4*>(m1 + j4);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      if (np > 2) {
        const float4 b = *reinterpret_cast<const float4*>(m2 + j4);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      *reinterpret_cast<float4*>(e.out + j4) = a;
    } else {
      for (int64_t j = j4; j < e.n; ++j) {
        float s = e.partial[j] + m0[j];
        if (np > 1) s += m1[j];
        if (np > 2) s += m2[j];
        e.out[j] = s;
      }
    }
    return;
  }
  const int form = rr_form(e.rows, e.n);
  if (form == 3) {                                   // tall and wide: grouped rows (k_reduce_rows_grouped)
    reduce_cols_grouped(e.partial, e.rows, e.ld, e.n, e.out, 0, blk, lds);
    return;
  }
  if (form != 0) {                                   // tall: 16 x 16 tiles (k_reduce_rows_tile16), or a wave per column
    if (form == 2) reduce_cols_tile16(e.partial, e.rows, e.ld, e.n, e.out, 0, blk, lds);
    else reduce_col_wave(e.partial, e.rows, e.ld, e.n, e.out, 0, blk);
  } else if (rq_wide_vec(e)) {                       // wide, 16-byte rows: four columns per thread, rows in order
    const int64_t j = (blk * 256 + threadIdx.x) * 4;
    if (j >= e.n) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int64_t r = 0; r < e.rows; ++r) {
      const float4 v = *reinterpret_cast<const float4*>(e.partial + r * e.ld + j);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(e.out + j) = s;
  } else {                                           // wide: one thread per column, rows in order (k_reduce_rows)
    const int64_t j = blk * 256 + threadIdx.x;
    if (j >= e.n) return;
    float s = 0.f;
#pragma unroll 8
    for (int64_t r = 0; r < e.rows; ++r) s += e.partial[r * e.ld + j];
    e.out[j] = s;
  }
}

#include <map>
#include <mutex>
#include <vector>
// One queue per STREAM: a backward pass defers on the stream it runs on (the autograd engine replays the forward's
// stream on its own thread, so the key is the stream, not the thread) and flushes that stream's entries only — several
// trainers in one process, each on a stream of its own, do not see each other's partials.
struct DeferQueue {
  bool on = false;
  std::vector<ReduceEntry> q;
};
static std::mutex g_rq_mutex;
static std::map<hipStream_t, DeferQueue> g_rq;

extern "C" int igcn_reduce_defer(void* stream, int on) {
  std::lock_guard<std::mutex> lk(g_rq_mutex);
  DeferQueue& d = g_rq[(hipStream_t)stream];
  d.on = on != 0;
  if (!d.on && d.q.empty()) g_rq.erase((hipStream_t)stream);
  return IGCN_OK;
}

extern "C" int igcn_reduce_pending(void) {
  std::lock_guard<std::mutex> lk(g_rq_mutex);
  size_t n = 0;
  for (const auto& kv : g_rq) n += kv.second.q.size();
  return (int)n;
}

// deferred reductions waiting on ONE stream (igcn_stream_pending, gemm.hip)
int igcn_reduce_pending_on(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_rq_mutex);
  auto it = g_rq.find(st);
  return it == g_rq.end() ? 0 : (int)it->second.q.size();
}

__global__ void k_tick(int32_t* tick) { *tick += 1; }

// the queue of `st` when that stream is in defer mode, else NULL (caller holds the mutex)
static std::vector<ReduceEntry>* rq_of(hipStream_t st) {
  auto it = g_rq.find(st);
  return it != g_rq.end() && it->second.on ? &it->second.q : nullptr;
}

static int reduce_flush_locked(hipStream_t st, int32_t* tick = nullptr) {
  auto it = g_rq.find(st);
  std::vector<ReduceEntry> none;
  std::vector<ReduceEntry>& q = it != g_rq.end() ? it->second.q : none;
  if (igcn_opt(IGCN_OPT_DEBUG_REDUCE))
    for (const ReduceEntry& e : q)
      fprintf(stderr, "[igcn] deferred reduction: rows %lld x n %d (ld %lld)%s\n", (long long)e.rows, e.n,
              (long long)e.ld, e.nptr == -2 ? "  loss value" : e.nptr < 0 ? "  GO attention finish" : e.nptr > 0 ? "  separate buffers" : rr_form(e.rows, e.n) ? "  tree" : "  in-order");
  size_t done = 0;
  while (done < q.size()) {
    ReduceTable t = {};
    const int cnt = (int)(q.size() - done < MRQ_MAX ? q.size() - done : MRQ_MAX);
    int64_t total = 0;
    for (int i = 0; i < cnt; ++i) {
      const ReduceEntry& e = q[done + i];
      t.e[i] = e;
      const int64_t need = e.nptr == -2 ? 1 : e.nptr < 0 ? e.n : e.nptr > 0 ? igcn_cdiv(e.n, 1024)
                           : rr_form(e.rows, e.n) == 1 ? igcn_cdiv(e.n, 4)
                           : rr_form(e.rows, e.n) == 2 ? igcn_cdiv(e.n, 16)
                           : rr_form(e.rows, e.n) == 3 ? igcn_cdiv(e.n, 64)
                                                       : igcn_cdiv(e.n, rq_wide_vec(e) ? 1024 : 256);
      t.start[i] = (int)total;
      total += need;
    }
    for (int i = cnt; i <= MRQ_MAX; ++i) t.start[i] = 0x7fffffff;
    t.count = cnt;
    done += cnt;
    hipLaunchKernelGGL(k_multi_reduce, dim3((unsigned)total), dim3(256), 0, st, t, done == q.size() ? tick : nullptr);
    if (done == q.size()) tick = nullptr;
  }
  if (tick) hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, st, tick);          // nothing was queued: the tick alone
  q.clear();
  if (it != g_rq.end() && !it->second.on) g_rq.erase(it);
  IGCN_CHECK_LAUNCH("reduce_flush");
  return IGCN_OK;
}

extern "C" int igcn_reduce_flush(void* stream) {
  std::lock_guard<std::mutex> lk(g_rq_mutex);
  return reduce_flush_locked((hipStream_t)stream);
}

// The flush that ends a train step's backward also advances the optimiser's step counter (int32, device) — the one-thread
// launch in front of igcn_adam_step_multi otherwise; pair with igcn_adam_step*_ticked.
extern "C" int igcn_reduce_flush_tick(void* stream, int32_t* step_counter) {
  std::lock_guard<std::mutex> lk(g_rq_mutex);
  return reduce_flush_locked((hipStream_t)stream, step_counter);
}

int igcn_launch_reduce_rows_final(const float* partial, int64_t rows, int64_t ld, int n, float* out,
                                  hipStream_t st) {
  if (n <= 0) return IGCN_OK;
  {
    std::lock_guard<std::mutex> lk(g_rq_mutex);
    if (std::vector<ReduceEntry>* q = rq_of(st)) {
      q->push_back(ReduceEntry{partial, out, rows, ld, n, 0, {nullptr, nullptr, nullptr}});
      return IGCN_OK;
    }
  }
  return igcn_launch_reduce_rows(partial, rows, ld, n, out, 0, st);
}

// tools/reduce_bench.py: a final reduction by itself (queued while igcn_reduce_defer is on)
// out[j] = sum_r partial[r * ld + j], j < n, as a FINAL reduction: queued while the stream defers (igcn_reduce_defer),
// performed by a launch of its own otherwise — for partial rows a kernel left behind for a parameter gradient
extern "C" int igcn_reduce_rows_final(const float* partial, int64_t rows, int64_t ld, int n, float* out, void* stream) {
  IGCN_REQUIRE(partial && out && rows >= 1 && n >= 1 && ld >= n, "reduce_rows_final: bad arguments");
  return igcn_launch_reduce_rows_final(partial, rows, ld, n, out, (hipStream_t)stream);
}
extern "C" int igcn_debug_reduce_rows_final(const float* partial, int64_t rows, int64_t ld, int n, float* out, void* stream) {
  return igcn_launch_reduce_rows_final(partial, rows, ld, n, out, (hipStream_t)stream);
}

// the loss value of a train step from its partial sums (loss.hip: igcn_loss_final) as an entry of the flush: queued while
// the stream defers (returns 1), else not handled here (returns 0)
int igcn_queue_loss_final(const float* parts, int nparts, const float* gram, int gram_rows, const float* prob,
                          int prob_rows, const float* wts, float* out, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_rq_mutex);
  std::vector<ReduceEntry>* q = rq_of(st);
  if (!q) return 0;
  q->push_back(ReduceEntry{parts, out, (int64_t)nparts, (int64_t)gram_rows | ((int64_t)prob_rows << 32), 8, -2,
                           {gram, prob, wts}});
  return 1;
}

// GO attention backward: dparams [2 FOUT FIN + 3 FOUT] from gpart [(2 FOUT + 3) FIN][parts] as a FINAL reduction:
// queued while the stream defers (returns 1), else not handled here (returns 0: k_go_attn_bwd_finish).
int igcn_queue_go_finish(const float* gpart, int64_t parts, int fin, int fout, const float* w_inc, const float* w_s,
                         float* dparams, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_rq_mutex);
  std::vector<ReduceEntry>* q = rq_of(st);
  if (!q || fin > 64 || fout > 64) return 0;
  ReduceEntry e = {gpart, dparams, parts, (int64_t)(fin | (fout << 8)), 2 * fout * fin + 3 * fout, -1, {w_inc, w_s, nullptr}};
  q->push_back(e);
  return 1;
}

// out = parts[0] + parts[1] + ... (k separate buffers of numel floats) as a FINAL reduction: queued while the stream
// defers (returns 1), otherwise not handled here (returns 0: the caller launches k_sum_n).
int igcn_queue_sum_final(const float* const* parts, int k, int64_t numel, float* out, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_rq_mutex);
  std::vector<ReduceEntry>* q = rq_of(st);
  if (!q || k < 2 || k > 4 || numel > 0x7fffffff) return 0;
  uintptr_t al = (uintptr_t)out;
  for (int i = 0; i < k; ++i) al |= (uintptr_t)parts[i];
  if (al & 15) return 0;
  ReduceEntry e = {parts[0], out, (int64_t)k, 0, (int)numel, k - 1, {nullptr, nullptr, nullptr}};
  for (int i = 1; i < k; ++i) e.more[i - 1] = parts[i];
  q->push_back(e);
  return 1;
}

// ---- stable grouping by a hand-written counting / radix sort ------------------------------------------------
// No library sort: three kernels per pass, no scratch beyond the caller's workspace, nothing that cannot be captured
// into a hipGraph (round 1 used rocPRIM's onesweep radix sort, which faulted when replayed from inside the step graph).
//
// A pass orders the elements of every SEGMENT by a digit < ND (ND <= 1024), stably:
//   k_sort_hist    tile (<= ST_TILE consecutive elements of one segment) -> digit histogram hist[seg][tile][ND]
//   k_sort_scan    per segment: hist -> exclusive offsets over (digit, tile)  (+ the row pointers in one-pass mode)
//   k_sort_scatter tile re-read in order; every element's slot = offset of (digit, tile, wave) + its rank among the
//                  EARLIER elements of the wave's chunk with the same digit (wave ballots: no atomics, no sort)
// Both groupings (A: by target = dst, B: by source = src) run through the same launches.
//  * segmented one-pass mode (a PyG batch: graph g owns nodes [node_ptr[g], node_ptr[g+1]) and edges
//    [edge_ptr[g], edge_ptr[g+1]), at most 1024 nodes per graph, any number of edges): digit = node id inside the
//    graph, so ONE pass finishes the grouping and the digit totals ARE the row pointers;
//  * general mode (one segment = all edges): LSD passes of 10 bits over the node id, ping-pong through the workspace.
#define ST_TILE 4096
#define ST_WAVES 4
#define ST_MAXD 1024

struct SortArgs {
  int64_t n_nodes, n_edges;
  const int64_t* ei;            // first pass: int64 edge_index [2,E]; keys = its rows, values = edge ids
  const int64_t* node_ptr;      // segmented mode (else NULL: one segment)
  const int64_t* edge_ptr;
  const int32_t *keyA_in, *valA_in, *keyB_in, *valB_in;   // later passes: permuted keys / edge ids of the last pass
  int32_t *keyA_out, *valA_out, *keyB_out, *valB_out;     // key*_out may be NULL (one-pass mode)
  int32_t *src32, *dst32, *loop_edge, *status;            // first pass side outputs
  int32_t *histA, *histB;       // [segments][tiles][nd]
  int32_t *dbaseA, *dbaseB;     // [segments][nd]
  int32_t *ptrA, *ptrB;         // one-pass mode: tgt_ptr / src_ptr
  int tiles, nd, shift, first, onepass;
};

// the wave's chunk of the tile: elements [lo, hi) (absolute edge positions), segment offsets nb / eb
__device__ __forceinline__ void sort_tile_range(const SortArgs& a, int seg, int tile, int64_t& nb, int64_t& nn,
                                                int64_t& lo, int64_t& hi) {
  int64_t eb = 0, ee = a.n_edges;
  nb = 0;
  nn = a.n_nodes;
  if (a.node_ptr) {
    nb = a.node_ptr[seg];
    nn = a.node_ptr[seg + 1] - nb;
    eb = a.edge_ptr[seg];
    ee = a.edge_ptr[seg + 1];
  }
  lo = eb + (int64_t)tile * ST_TILE;
  hi = lo + ST_TILE < ee ? lo + ST_TILE : ee;
  if (lo > hi) lo = hi;
}

// digits (A, B) of element i; first pass also emits src32 / dst32 / loop_edge
__device__ __forceinline__ void sort_load(const SortArgs& a, int64_t i, int64_t nb, int64_t nn, int32_t& kA, int32_t& kB,
                                          int32_t& vA, int32_t& vB, int& dA, int& dB) {
  if (a.first) {
    const int64_t s = a.ei[i], d = a.ei[a.n_edges + i];
    kA = (int32_t)d;
    kB = (int32_t)s;
    vA = vB = (int32_t)i;
  } else {
    kA = a.keyA_in[i]; vA = a.valA_in[i];
    kB = a.keyB_in[i]; vB = a.valB_in[i];
  }
  if (a.onepass) {
    int64_t la = (int64_t)kA - nb, lb = (int64_t)kB - nb;
    if (la < 0 || la >= nn || lb < 0 || lb >= nn) {      // edge leaves its graph: not a PyG batch
      atomicExch(a.status, 2);
      la = la < 0 || la >= nn ? 0 : la;
      lb = lb < 0 || lb >= nn ? 0 : lb;
    }
    dA = (int)la;
    dB = (int)lb;
  } else {
    dA = (kA >> a.shift) & (a.nd - 1);
    dB = (kB >> a.shift) & (a.nd - 1);
  }
}

// lanes of the wave whose digit equals this lane's (restricted to `active`)
__device__ __forceinline__ unsigned long long sort_peers(int d, int nbits, unsigned long long active) {
  unsigned long long peers = active;
  for (int b = 0; b < nbits; ++b) {
    const bool bit = (d >> b) & 1;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

__device__ __forceinline__ int sort_nbits(int nd) {
  int b = 0;
  while ((1 << b) < nd) ++b;
  return b;
}

// per-wave digit counts of the tile into cnt[2][ST_WAVES][nd] (LDS, zeroed here)
__device__ __forceinline__ void sort_wave_hist(const SortArgs& a, int32_t* cnt, int64_t nb, int64_t nn, int64_t lo,
                                               int64_t hi, bool side_outputs) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nd = a.nd, nbits = sort_nbits(nd);
  for (int i = tid; i < 2 * ST_WAVES * nd; i += 256) cnt[i] = 0;
  __syncthreads();
  int32_t* cA = cnt + w * nd;
  int32_t* cB = cnt + (ST_WAVES + w) * nd;
  const int64_t wlo = lo + (int64_t)w * (ST_TILE / ST_WAVES);
  for (int64_t base = wlo; base < wlo + ST_TILE / ST_WAVES && base < hi; base += 64) {     // wave-uniform bounds
    const int64_t i = base + lane;
    const bool valid = i < hi;
    int32_t kA = 0, kB = 0, vA = 0, vB = 0;
    int dA = 0, dB = 0;
    if (valid) {
      sort_load(a, i, nb, nn, kA, kB, vA, vB, dA, dB);
      if (side_outputs) {
        a.src32[i] = kB;
        a.dst32[i] = kA;
        if (kA == kB && kA >= 0 && kA < a.n_nodes) atomicMax(&a.loop_edge[kA], (int32_t)i);   // last stored loop wins
      }
    }
    const unsigned long long active = __ballot(valid);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long pA = sort_peers(dA, nbits, active), pB = sort_peers(dB, nbits, active);
    if (valid && (pA & lt) == 0) cA[dA] += __popcll(pA);      // one leader per digit: no two lanes share an address
    if (valid && (pB & lt) == 0) cB[dB] += __popcll(pB);
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) k_sort_hist(SortArgs a) {
  extern __shared__ int32_t st_cnt[];
  const int seg = blockIdx.y, tile = blockIdx.x, nd = a.nd;
  int64_t nb, nn, lo, hi;
  sort_tile_range(a, seg, tile, nb, nn, lo, hi);
  if (a.onepass && nn > nd) {                       // the host checked the maximum; refuse, do not corrupt
    if (threadIdx.x == 0) atomicExch(a.status, 1);
    nn = nd;
  }
  sort_wave_hist(a, st_cnt, nb, nn, lo, hi, a.first != 0);
  int32_t* hA = a.histA + ((int64_t)seg * a.tiles + tile) * nd;
  int32_t* hB = a.histB + ((int64_t)seg * a.tiles + tile) * nd;
  for (int d = threadIdx.x; d < nd; d += 256) {
    int tA = 0, tB = 0;
#pragma unroll
    for (int w = 0; w < ST_WAVES; ++w) {
      tA += st_cnt[w * nd + d];
      tB += st_cnt[(ST_WAVES + w) * nd + d];
    }
    hA[d] = tA;
    hB[d] = tB;
  }
}

// one 1024-thread block per segment: thread d owns digit d.  hist[seg][t][d] becomes the number of elements of digit
// d in the EARLIER tiles of the segment; dbase[seg][d] = segment start + elements of smaller digits.
__global__ void __launch_bounds__(1024) k_sort_scan(SortArgs a, int n_segments) {
  __shared__ int32_t part[2][16];
  const int seg = blockIdx.x, d = threadIdx.x, nd = a.nd, lane = d & 63, w = d >> 6;
  int64_t eb = 0, nb = 0, nn = a.n_nodes;
  if (a.node_ptr) {
    eb = a.edge_ptr[seg];
    nb = a.node_ptr[seg];
    nn = a.node_ptr[seg + 1] - nb;
  }
  int totA = 0, totB = 0;
  if (d < nd) {
    int32_t* hA = a.histA + (int64_t)seg * a.tiles * nd + d;
    int32_t* hB = a.histB + (int64_t)seg * a.tiles * nd + d;
#pragma unroll 4
    for (int t = 0; t < a.tiles; ++t) {
      const int cA = hA[(int64_t)t * nd], cB = hB[(int64_t)t * nd];
      hA[(int64_t)t * nd] = totA;
      hB[(int64_t)t * nd] = totB;
      totA += cA;
      totB += cB;
    }
  }
  // exclusive scan of the digit totals over the block (wave scans + one pass over the 16 wave sums)
  int sA = totA, sB = totB;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int uA = __shfl_up(sA, o, 64), uB = __shfl_up(sB, o, 64);
    if (lane >= o) { sA += uA; sB += uB; }
  }
  if (lane == 63) { part[0][w] = sA; part[1][w] = sB; }
  __syncthreads();
  int offA = 0, offB = 0;
  for (int i = 0; i < w; ++i) { offA += part[0][i]; offB += part[1][i]; }
  const int exA = offA + sA - totA, exB = offB + sB - totB;
  if (d < nd) {
    a.dbaseA[(int64_t)seg * nd + d] = (int32_t)eb + exA;
    a.dbaseB[(int64_t)seg * nd + d] = (int32_t)eb + exB;
    if (a.onepass && d < nn) {
      a.ptrA[nb + d] = (int32_t)eb + exA;
      a.ptrB[nb + d] = (int32_t)eb + exB;
    }
  }
  if (a.onepass && seg == n_segments - 1 && d == 0) {
    a.ptrA[a.n_nodes] = (int32_t)a.n_edges;
    a.ptrB[a.n_nodes] = (int32_t)a.n_edges;
  }
}

__global__ void __launch_bounds__(256) k_sort_scatter(SortArgs a) {
  extern __shared__ int32_t st_cnt[];
  const int seg = blockIdx.y, tile = blockIdx.x, nd = a.nd, nbits = sort_nbits(nd);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int64_t nb, nn, lo, hi;
  sort_tile_range(a, seg, tile, nb, nn, lo, hi);
  if (a.onepass && nn > nd) nn = nd;
  sort_wave_hist(a, st_cnt, nb, nn, lo, hi, false);
  // per-wave counts -> first slot of (digit, wave): digit base + earlier tiles + earlier waves of this tile
  const int32_t* hA = a.histA + ((int64_t)seg * a.tiles + tile) * nd;
  const int32_t* hB = a.histB + ((int64_t)seg * a.tiles + tile) * nd;
  for (int d = tid; d < nd; d += 256) {
    int rA = a.dbaseA[(int64_t)seg * nd + d] + hA[d], rB = a.dbaseB[(int64_t)seg * nd + d] + hB[d];
#pragma unroll
    for (int k = 0; k < ST_WAVES; ++k) {
      const int cA = st_cnt[k * nd + d], cB = st_cnt[(ST_WAVES + k) * nd + d];
      st_cnt[k * nd + d] = rA;
      st_cnt[(ST_WAVES + k) * nd + d] = rB;
      rA += cA;
      rB += cB;
    }
  }
  __syncthreads();
  int32_t* cA = st_cnt + w * nd;
  int32_t* cB = st_cnt + (ST_WAVES + w) * nd;
  const int64_t wlo = lo + (int64_t)w * (ST_TILE / ST_WAVES);
  for (int64_t base = wlo; base < wlo + ST_TILE / ST_WAVES && base < hi; base += 64) {
    const int64_t i = base + lane;
    const bool valid = i < hi;
    int32_t kA = 0, kB = 0, vA = 0, vB = 0;
    int dA = 0, dB = 0;
    if (valid) sort_load(a, i, nb, nn, kA, kB, vA, vB, dA, dB);
    const unsigned long long active = __ballot(valid);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long pA = sort_peers(dA, nbits, active), pB = sort_peers(dB, nbits, active);
    int posA = 0, posB = 0;
    if (valid) {
      posA = cA[dA] + __popcll(pA & lt);
      posB = cB[dB] + __popcll(pB & lt);
    }
    // all lanes have read their running offsets (LDS operations of a wave execute in order) before the leaders
    // advance them
    if (valid && (pA & lt) == 0) cA[dA] += __popcll(pA);
    if (valid && (pB & lt) == 0) cB[dB] += __popcll(pB);
    if (valid) {
      a.valA_out[posA] = vA;
      a.valB_out[posB] = vB;
      if (a.keyA_out) { a.keyA_out[posA] = kA; a.keyB_out[posB] = kB; }
    }
  }
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

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static int pow2_digits(int64_t n) {        // smallest power of two >= n, at least 2
  int d = 2;
  while (d < n) d <<= 1;
  return d;
}

static int sort_launch_pass(const SortArgs& a, int n_segments, hipStream_t st) {
  const size_t lds = (size_t)2 * ST_WAVES * a.nd * sizeof(int32_t);
  dim3 grid((unsigned)a.tiles, (unsigned)n_segments);
  hipLaunchKernelGGL(k_sort_hist, grid, dim3(256), lds, st, a);
  hipLaunchKernelGGL(k_sort_scan, dim3((unsigned)n_segments), dim3(1024), 0, st, a, n_segments);
  hipLaunchKernelGGL(k_sort_scatter, grid, dim3(256), lds, st, a);
  IGCN_CHECK_LAUNCH("graph_plan sort pass");
  return IGCN_OK;
}

// ---- general build: arbitrary edge_index, LSD passes of 10 bits over the node id ------------------------------
static int general_passes(int64_t n_nodes) {
  int bits = 1;
  while (((int64_t)1 << bits) < n_nodes) ++bits;
  return (bits + 9) / 10;
}

extern "C" size_t igcn_graph_plan_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
  if (n_edges <= 0) return 256;
  const int64_t tiles = igcn_cdiv(n_edges, ST_TILE);
  const int passes = general_passes(n_nodes);
  // two histograms + two digit-base rows, then (multi-pass only) two ping-pong (key, value) buffer pairs per grouping
  size_t bytes = 2 * align256((size_t)tiles * ST_MAXD * 4) + 2 * align256((size_t)ST_MAXD * 4) + 256;
  bytes += (size_t)(passes > 1 ? 8 : 2) * align256((size_t)n_edges * 4);
  return bytes;
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
  const int64_t tiles = igcn_cdiv(n_edges, ST_TILE);
  const int passes = general_passes(n_nodes);
  char* ws = (char*)workspace;
  const size_t hsz = align256((size_t)tiles * ST_MAXD * 4), dsz = align256((size_t)ST_MAXD * 4);
  const size_t esz = align256((size_t)n_edges * 4);
  SortArgs a = {};
  a.n_nodes = n_nodes;
  a.n_edges = n_edges;
  a.ei = edge_index;
  a.src32 = src32;
  a.dst32 = dst32;
  a.loop_edge = loop_edge;
  a.histA = (int32_t*)ws;
  a.histB = (int32_t*)(ws + hsz);
  a.dbaseA = (int32_t*)(ws + 2 * hsz);
  a.dbaseB = (int32_t*)(ws + 2 * hsz + dsz);
  char* eb = ws + 2 * hsz + 2 * dsz;
  int32_t* buf[8];
  for (int i = 0; i < 8; ++i) buf[i] = (int32_t*)(eb + (size_t)(passes > 1 ? i : (i & 1)) * esz);
  a.tiles = (int)tiles;
  a.ptrA = tgt_ptr;
  a.ptrB = src_ptr;
  a.status = nullptr;
  a.onepass = 0;
  // keys of the final order go to buf[.] for the row-pointer search; values of the final pass are the permutations
  const int32_t *kA = nullptr, *vA = nullptr, *kB = nullptr, *vB = nullptr;
  int bits_left = 1;
  while (((int64_t)1 << bits_left) < n_nodes) ++bits_left;
  for (int p = 0; p < passes; ++p) {
    const bool last = p == passes - 1;
    const int pb = bits_left < 10 ? bits_left : 10;
    a.nd = 1 << pb;
    a.shift = 10 * p;
    a.first = p == 0;
    a.keyA_in = kA; a.valA_in = vA; a.keyB_in = kB; a.valB_in = vB;
    const int o = (p & 1) * 4;
    a.keyA_out = buf[o + 0];
    a.keyB_out = buf[o + 1];
    a.valA_out = last ? tgt_perm : buf[o + 2];
    a.valB_out = last ? src_perm : buf[o + 3];
    const int rc = sort_launch_pass(a, 1, st);
    if (rc) return rc;
    kA = a.keyA_out; vA = a.valA_out; kB = a.keyB_out; vB = a.valB_out;
    bits_left -= pb;
  }
  hipLaunchKernelGGL(k_plan_ptr, dim3((unsigned)igcn_cdiv(n_nodes + 1, T)), dim3(T), 0, st, n_nodes, n_edges, kA,
                     tgt_ptr);
  hipLaunchKernelGGL(k_plan_ptr, dim3((unsigned)igcn_cdiv(n_nodes + 1, T)), dim3(T), 0, st, n_nodes, n_edges, kB,
                     src_ptr);
  IGCN_CHECK_LAUNCH("graph_plan_build");
  return IGCN_OK;
}

// ---- tiled segmented build: a PyG batch of graphs with <= 1024 nodes and ANY number of edges each --------------
extern "C" size_t igcn_graph_plan_tiled_workspace_bytes(int n_graphs, int64_t max_nodes_per_graph,
                                                        int64_t max_edges_per_graph) {
  const int64_t tiles = igcn_cdiv(max_edges_per_graph > 0 ? max_edges_per_graph : 1, ST_TILE);
  const int nd = pow2_digits(max_nodes_per_graph);
  return 2 * align256((size_t)n_graphs * tiles * nd * 4) + 2 * align256((size_t)n_graphs * nd * 4) + 256;
}

extern "C" int igcn_graph_plan_build_tiled(int64_t n_nodes, int64_t n_edges, int n_graphs, const int64_t* edge_index,
                                           const int64_t* node_ptr, const int64_t* edge_ptr,
                                           int64_t max_nodes_per_graph, int64_t max_edges_per_graph, int32_t* src32,
                                           int32_t* dst32, int32_t* tgt_ptr, int32_t* tgt_perm, int32_t* src_ptr,
                                           int32_t* src_perm, int32_t* loop_edge, int32_t* status, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  IGCN_REQUIRE(n_graphs > 0 && n_nodes > 0 && n_nodes < ((int64_t)1 << 31) && n_edges > 0 &&
               n_edges < ((int64_t)1 << 31) && max_nodes_per_graph > 0 && max_edges_per_graph > 0,
               "graph_plan_build_tiled: bad sizes");
  if (max_nodes_per_graph > ST_MAXD) {
    igcn_set_error("graph_plan_build_tiled: more than %d nodes per graph (use igcn_graph_plan_build)", ST_MAXD);
    return IGCN_ERR_UNSUPPORTED;
  }
  IGCN_REQUIRE(workspace_bytes >= igcn_graph_plan_tiled_workspace_bytes(n_graphs, max_nodes_per_graph,
                                                                         max_edges_per_graph),
               "graph_plan_build_tiled: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)igcn_cdiv(n_nodes, 256)), dim3(256), 0, st, n_nodes, loop_edge, -1);
  const int64_t tiles = igcn_cdiv(max_edges_per_graph, ST_TILE);
  const int nd = pow2_digits(max_nodes_per_graph);
  char* ws = (char*)workspace;
  const size_t hsz = align256((size_t)n_graphs * tiles * nd * 4), dsz = align256((size_t)n_graphs * nd * 4);
  SortArgs a = {};
  a.n_nodes = n_nodes;
  a.n_edges = n_edges;
  a.ei = edge_index;
  a.node_ptr = node_ptr;
  a.edge_ptr = edge_ptr;
  a.valA_out = tgt_perm;
  a.valB_out = src_perm;
  a.src32 = src32;
  a.dst32 = dst32;
  a.loop_edge = loop_edge;
  a.status = status;
  a.histA = (int32_t*)ws;
  a.histB = (int32_t*)(ws + hsz);
  a.dbaseA = (int32_t*)(ws + 2 * hsz);
  a.dbaseB = (int32_t*)(ws + 2 * hsz + dsz);
  a.ptrA = tgt_ptr;
  a.ptrB = src_ptr;
  a.tiles = (int)tiles;
  a.nd = nd;
  a.shift = 0;
  a.first = 1;
  a.onepass = 1;
  return sort_launch_pass(a, n_graphs, st);
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
// `rep`: the plan of `copies` disjoint copies of the batch (what igcn_graph_plan_replicate derives from the plan: the
// two passes of a train step as one block-diagonal problem), written by the same workgroups — one launch less per step.
struct PlanRep { int copies; int32_t *src32, *dst32, *tgt_ptr, *tgt_perm, *src_ptr, *src_perm, *loop_edge; };
__device__ __forceinline__ void
plan_segmented_body(const int g, int64_t n_nodes, int64_t n_edges, int n_graphs, const int64_t* __restrict__ ei,
                    const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ edge_ptr,
                    int32_t* __restrict__ src32, int32_t* __restrict__ dst32, int32_t* __restrict__ tgt_ptr,
                    int32_t* __restrict__ tgt_perm, int32_t* __restrict__ src_ptr, int32_t* __restrict__ src_perm,
                    int32_t* __restrict__ loop_edge, int32_t* __restrict__ status, const PlanRep& rep) {
  const int32_t N32 = (int32_t)n_nodes, E32 = (int32_t)n_edges;
  __shared__ int16_t ls[SEG_MAXE], ld[SEG_MAXE];        // local (graph-relative) endpoints
  __shared__ int32_t ct[SEG_MAXN + 1], cs[SEG_MAXN + 1], lp[SEG_MAXN];
  const int tid = threadIdx.x;
  const int64_t nb = node_ptr[g], eb = edge_ptr[g];
  const int nn = (int)(node_ptr[g + 1] - nb), ne = (int)(edge_ptr[g + 1] - eb);
  // host checked the maxima; the offsets themselves come from device memory (a loader may have handed over garbage):
  // refuse anything that does not lie inside the batch, don't read or write out of bounds
  if (nn > SEG_MAXN || ne > SEG_MAXE || nn < 0 || ne < 0 || nb < 0 || eb < 0 || nb + nn > n_nodes ||
      eb + ne > n_edges) {
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
    for (int c = 0; c < rep.copies; ++c) {
      rep.src32[(int64_t)c * n_edges + eb + k] = (int32_t)(s + nb) + c * N32;
      rep.dst32[(int64_t)c * n_edges + eb + k] = (int32_t)(d + nb) + c * N32;
    }
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
    const int32_t tp = (int32_t)(eb + ct[i]), sp = (int32_t)(eb + cs[i]);
    const int32_t le = lp[i] >= 0 ? (int32_t)(eb + lp[i]) : -1;
    tgt_ptr[nb + i] = tp;
    src_ptr[nb + i] = sp;
    loop_edge[nb + i] = le;
    for (int c = 0; c < rep.copies; ++c) {
      rep.tgt_ptr[(int64_t)c * n_nodes + nb + i] = tp + c * E32;
      rep.src_ptr[(int64_t)c * n_nodes + nb + i] = sp + c * E32;
      rep.loop_edge[(int64_t)c * n_nodes + nb + i] = le >= 0 ? le + c * E32 : -1;
    }
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
    for (int c = 0; c < rep.copies; ++c) {
      rep.tgt_perm[(int64_t)c * n_edges + eb + ct[kd] + rd] = (int32_t)(eb + k) + c * E32;
      rep.src_perm[(int64_t)c * n_edges + eb + cs[ks] + rs] = (int32_t)(eb + k) + c * E32;
    }
  }
  if (g == n_graphs - 1 && tid == 0) {
    tgt_ptr[n_nodes] = (int32_t)n_edges;
    src_ptr[n_nodes] = (int32_t)n_edges;
    if (rep.copies > 0) {
      rep.tgt_ptr[(int64_t)rep.copies * n_nodes] = rep.copies * E32;
      rep.src_ptr[(int64_t)rep.copies * n_nodes] = rep.copies * E32;
    }
  }
}

__global__ void __launch_bounds__(256)
k_plan_segmented(int64_t n_nodes, int64_t n_edges, int n_graphs, const int64_t* __restrict__ ei,
                 const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ edge_ptr,
                 int32_t* __restrict__ src32, int32_t* __restrict__ dst32, int32_t* __restrict__ tgt_ptr,
                 int32_t* __restrict__ tgt_perm, int32_t* __restrict__ src_ptr, int32_t* __restrict__ src_perm,
                 int32_t* __restrict__ loop_edge, int32_t* __restrict__ status, const PlanRep rep) {
  plan_segmented_body((int)blockIdx.x, n_nodes, n_edges, n_graphs, ei, node_ptr, edge_ptr, src32, dst32, tgt_ptr, tgt_perm,
                      src_ptr, src_perm, loop_edge, status, rep);
}

// ---- a rider: the step's dropout masks drawn by extra workgroups of THIS launch ----------------------------------
// The plan build and the mask generation are the two launches of a train step that depend on nothing the step computes
// (edge_index / a counter), 11.7 and 12.3 us back to back.  As two ROLES of one grid — workgroups [0, n_graphs) build the
// plan, the rest draw the masks — they overlap; a second stream or a forked graph branch costs the replay 40-75 us on
// this part, a role costs nothing.  igcn_rider_dropout queues the job for a stream; the next per-graph plan
// build on that stream carries it; igcn_rider_flush launches a job nobody carried.
#include "dropout.h"
__global__ void __launch_bounds__(256)
k_plan_segmented_ride(int64_t n_nodes, int64_t n_edges, int n_graphs, const int64_t* __restrict__ ei,
                      const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ edge_ptr,
                      int32_t* __restrict__ src32, int32_t* __restrict__ dst32, int32_t* __restrict__ tgt_ptr,
                      int32_t* __restrict__ tgt_perm, int32_t* __restrict__ src_ptr, int32_t* __restrict__ src_perm,
                      int32_t* __restrict__ loop_edge, int32_t* __restrict__ status, const PlanRep rep,
                      int64_t d_total, const DropSegs d_segs, unsigned long long* __restrict__ d_state,
                      float* __restrict__ d_out, const DropCounters d_cnt) {
  if ((int)blockIdx.x < n_graphs) {
    plan_segmented_body((int)blockIdx.x, n_nodes, n_edges, n_graphs, ei, node_ptr, edge_ptr, src32, dst32, tgt_ptr,
                        tgt_perm, src_ptr, src_perm, loop_edge, status, rep);
  } else {
    dropout_masks_body(blockIdx.x - (unsigned)n_graphs, gridDim.x - (unsigned)n_graphs, d_total, d_segs, d_state, d_out,
                       d_cnt);
  }
}

static std::mutex g_rider_mutex;
static std::map<hipStream_t, DropJob> g_riders;

extern "C" int igcn_rider_dropout(void* stream, int64_t total, int n_segments, const int64_t* seg_end, const float* seg_p,
                                  void* state, float* out, int n_counters, int64_t* const* counters, int64_t counter_inc) {
  DropJob job;
  const int rc = igcn_dropout_job(job, "rider_dropout", total, n_segments, seg_end, seg_p, state, out, n_counters, counters,
                                  counter_inc);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_rider_mutex);
  if (g_riders.count((hipStream_t)stream)) {
    igcn_set_error("rider_dropout: this stream already has a job waiting (igcn_rider_flush it first)");
    return IGCN_ERR_BADARG;
  }
  g_riders[(hipStream_t)stream] = job;
  return IGCN_OK;
}

// a job still waiting on the stream is launched by itself; nothing waiting: nothing happens
extern "C" int igcn_rider_flush(void* stream) {
  DropJob job;
  {
    std::lock_guard<std::mutex> lk(g_rider_mutex);
    auto it = g_riders.find((hipStream_t)stream);
    if (it == g_riders.end()) return IGCN_OK;
    job = it->second;
    g_riders.erase(it);
  }
  return igcn_dropout_launch(job, (hipStream_t)stream);
}

// forget a job that is still waiting (a step that failed between queueing it and the launch that would have carried it:
// its output buffer may be gone — it must not be launched by a later flush)
void igcn_rider_dropout_cancel(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_rider_mutex);
  g_riders.erase(st);
}

int igcn_rider_dropout_waiting(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_rider_mutex);
  return (int)g_riders.count(st);
}

// (also taken by the front kernel of the image branch, csrc/sgcn_fused.hip, which builds the plan itself)
bool igcn_rider_dropout_take(hipStream_t st, DropJob& job);
static bool rider_take(hipStream_t st, DropJob& job) { return igcn_rider_dropout_take(st, job); }
bool igcn_rider_dropout_take(hipStream_t st, DropJob& job) {
  std::lock_guard<std::mutex> lk(g_rider_mutex);
  auto it = g_riders.find(st);
  if (it == g_riders.end()) return false;
  job = it->second;
  g_riders.erase(it);
  return true;
}

static int plan_build_segmented(int64_t n_nodes, int64_t n_edges, int n_graphs, const int64_t* edge_index,
                                const int64_t* node_ptr, const int64_t* edge_ptr, int64_t max_nodes_per_graph,
                                int64_t max_edges_per_graph, int32_t* src32, int32_t* dst32, int32_t* tgt_ptr,
                                int32_t* tgt_perm, int32_t* src_ptr, int32_t* src_perm, int32_t* loop_edge,
                                int32_t* status, const PlanRep& rep, void* stream) {
  IGCN_REQUIRE(n_graphs > 0 && n_nodes > 0 && n_nodes < ((int64_t)1 << 31) && n_edges < ((int64_t)1 << 31),
               "graph_plan_build_segmented: bad sizes");
  IGCN_REQUIRE(rep.copies >= 0 && n_nodes * (rep.copies > 0 ? rep.copies : 1) < ((int64_t)1 << 31) &&
                   n_edges * (rep.copies > 0 ? rep.copies : 1) < ((int64_t)1 << 31),
               "graph_plan_build_segmented: replica sizes out of int32 range");
  if (max_nodes_per_graph > SEG_MAXN || max_edges_per_graph > SEG_MAXE) {
    igcn_set_error("graph_plan_build_segmented: graphs too large for the LDS path (max %d nodes / %d edges per graph)",
                   SEG_MAXN, SEG_MAXE);
    return IGCN_ERR_UNSUPPORTED;
  }
  DropJob job;
  if (rider_take((hipStream_t)stream, job)) {
    hipLaunchKernelGGL(k_plan_segmented_ride, dim3((unsigned)n_graphs + job.blocks), dim3(256), 0, (hipStream_t)stream,
                       n_nodes, n_edges, n_graphs, edge_index, node_ptr, edge_ptr, src32, dst32, tgt_ptr, tgt_perm, src_ptr,
                       src_perm, loop_edge, status, rep, job.total, job.sg, job.state, job.out, job.cnt);
    IGCN_CHECK_LAUNCH("graph_plan_build_segmented (+ dropout rider)");
    return IGCN_OK;
  }
  hipLaunchKernelGGL(k_plan_segmented, dim3(n_graphs), dim3(256), 0, (hipStream_t)stream, n_nodes, n_edges, n_graphs,
                     edge_index, node_ptr, edge_ptr, src32, dst32, tgt_ptr, tgt_perm, src_ptr, src_perm, loop_edge,
                     status, rep);
  IGCN_CHECK_LAUNCH("graph_plan_build_segmented");
  return IGCN_OK;
}

extern "C" int igcn_graph_plan_build_segmented(int64_t n_nodes, int64_t n_edges, int n_graphs,
                                               const int64_t* edge_index, const int64_t* node_ptr,
                                               const int64_t* edge_ptr, int64_t max_nodes_per_graph,
                                               int64_t max_edges_per_graph, int32_t* src32, int32_t* dst32,
                                               int32_t* tgt_ptr, int32_t* tgt_perm, int32_t* src_ptr,
                                               int32_t* src_perm, int32_t* loop_edge, int32_t* status,
                                               void* stream) {
  const PlanRep none = {};
  return plan_build_segmented(n_nodes, n_edges, n_graphs, edge_index, node_ptr, edge_ptr, max_nodes_per_graph,
                              max_edges_per_graph, src32, dst32, tgt_ptr, tgt_perm, src_ptr, src_perm, loop_edge, status,
                              none, stream);
}

// The same build that also fills the plan of `copies` disjoint copies of the batch (the arrays
// igcn_graph_plan_replicate would derive: sizes copies * E, copies * N + 1, copies * N).
extern "C" int igcn_graph_plan_build_segmented_rep(int64_t n_nodes, int64_t n_edges, int n_graphs,
                                                   const int64_t* edge_index, const int64_t* node_ptr,
                                                   const int64_t* edge_ptr, int64_t max_nodes_per_graph,
                                                   int64_t max_edges_per_graph, int32_t* src32, int32_t* dst32,
                                                   int32_t* tgt_ptr, int32_t* tgt_perm, int32_t* src_ptr,
                                                   int32_t* src_perm, int32_t* loop_edge, int32_t* status, int copies,
                                                   int32_t* o_src32, int32_t* o_dst32, int32_t* o_tgt_ptr,
                                                   int32_t* o_tgt_perm, int32_t* o_src_ptr, int32_t* o_src_perm,
                                                   int32_t* o_loop_edge, void* stream) {
  IGCN_REQUIRE(copies >= 1 && o_src32 && o_dst32 && o_tgt_ptr && o_tgt_perm && o_src_ptr && o_src_perm && o_loop_edge,
               "graph_plan_build_segmented_rep: copies >= 1 and all seven replica arrays");
  const PlanRep rep = {copies, o_src32, o_dst32, o_tgt_ptr, o_tgt_perm, o_src_ptr, o_src_perm, o_loop_edge};
  return plan_build_segmented(n_nodes, n_edges, n_graphs, edge_index, node_ptr, edge_ptr, max_nodes_per_graph,
                              max_edges_per_graph, src32, dst32, tgt_ptr, tgt_perm, src_ptr, src_perm, loop_edge, status,
                              rep, stream);
}
