// Probe: can two INDEPENDENT kernels of one stream overlap when the second is launched with hipExtAnyOrderLaunch
// (AQL barrier bit cleared)?  Eager launches and a captured hipGraph.  Build: hipcc --offload-arch=gfx950 -O3 -o anyorder_probe anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// a latency-bound kernel: few workgroups, each a long dependent chain
__global__ void k_chain(float* out, int iters, float seed) {
  float v = seed + threadIdx.x;
  for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0000001f, 0.5f);
  if (v == 12345.f) out[blockIdx.x] = v;
}

int main() {
  float* buf;
  CK(hipMalloc(&buf, 1 << 20));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 20000, wgs = 128;          // half the chip's CUs, one wave each
  auto run = [&](int flags, int pairs, const char* what) -> int {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int p = 0; p < pairs; ++p) {
        hipExtLaunchKernelGGL(k_chain, dim3(wgs), dim3(64), 0, st, nullptr, nullptr, 0, buf, iters, 1.f);
        hipExtLaunchKernelGGL(k_chain, dim3(wgs), dim3(64), 0, st, nullptr, nullptr, flags, buf + 4096, iters, 2.f);
      }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("%-40s %8.1f us for %d pairs (%.1f us per pair)\n", what, ms * 1e3, pairs, ms * 1e3 / pairs);
    }
    return 0;
  };
  if (run(0, 10, "eager, in order")) return 1;
  if (run(hipExtAnyOrderLaunch, 10, "eager, second of a pair any-order")) return 1;
  // captured
  for (int flags : {0, (int)hipExtAnyOrderLaunch}) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < 10; ++p) {
      hipExtLaunchKernelGGL(k_chain, dim3(wgs), dim3(64), 0, st, nullptr, nullptr, 0, buf, iters, 1.f);
      hipExtLaunchKernelGGL(k_chain, dim3(wgs), dim3(64), 0, st, nullptr, nullptr, flags, buf + 4096, iters, 2.f);
    }
    hipError_t ec = hipStreamEndCapture(st, &g);
    if (ec != hipSuccess) { printf("capture with flags %d failed: %s\n", flags, hipGetErrorString(ec)); continue; }
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("graph replay, flags %d: %32.1f us for 10 pairs (%.1f us per pair)\n", flags, ms * 1e3, ms * 1e2);
    }
  }
  // a hand-built graph: 10 pairs, the two kernels of a pair depend on the previous pair only (fork / join per pair)
  {
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    std::vector<hipGraphNode_t> prev;
    float seed1 = 1.f, seed2 = 2.f;
    int it = iters;
    float* b2 = buf + 4096;
    for (int p = 0; p < 10; ++p) {
      hipGraphNode_t n1, n2;
      void* a1[] = {&buf, &it, &seed1};
      void* a2[] = {&b2, &it, &seed2};
      hipKernelNodeParams kp = {};
      kp.func = (void*)k_chain; kp.gridDim = dim3(wgs); kp.blockDim = dim3(64); kp.sharedMemBytes = 0; kp.kernelParams = a1; kp.extra = nullptr;
      CK(hipGraphAddKernelNode(&n1, g, prev.data(), prev.size(), &kp));
      kp.kernelParams = a2;
      CK(hipGraphAddKernelNode(&n2, g, prev.data(), prev.size(), &kp));
      prev = {n1, n2};
    }
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("hand-built graph, fork/join per pair: %18.1f us for 10 pairs (%.1f us per pair)\n", ms * 1e3, ms * 1e2);
    }
  }
  // single kernel reference
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int p = 0; p < 10; ++p) hipLaunchKernelGGL(k_chain, dim3(wgs), dim3(64), 0, st, buf, iters, 1.f);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 2) printf("one kernel alone: %37.1f us per launch\n", ms * 1e2);
  }
  return 0;
}
