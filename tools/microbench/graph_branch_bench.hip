// Do independent branches of one hipGraph run concurrently?  NCHAIN chains of 1000 dependent tiny kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k(const int* __restrict__ in, int* __restrict__ out, int n, int spin) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int v = in[i % n];
  for (int j = 0; j < spin; ++j) asm volatile("s_nop 7" ::: "memory");
  out[i] = v;
}

void run(int nchain, int wgs, int spin, bool graph) {
  int n = wgs * 256, iters = 1000;
  std::vector<int*> a(nchain), b(nchain);
  std::vector<hipStream_t> st(nchain);
  std::vector<hipEvent_t> ev(nchain);
  for (int c = 0; c < nchain; ++c) { CK(hipMalloc(&a[c], n * 4)); CK(hipMalloc(&b[c], n * 4)); CK(hipMemset(a[c], 0, n * 4)); CK(hipStreamCreateWithFlags(&st[c], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ev[c], hipEventDisableTiming)); }
  hipEvent_t fork; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  auto enqueue = [&]() {
    CK(hipEventRecord(fork, st[0]));
    for (int c = 1; c < nchain; ++c) CK(hipStreamWaitEvent(st[c], fork, 0));
    for (int i = 0; i < iters; ++i)
      for (int c = 0; c < nchain; ++c)
        hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, st[c], (i & 1) ? b[c] : a[c], (i & 1) ? a[c] : b[c], n, spin);
    for (int c = 1; c < nchain; ++c) { CK(hipEventRecord(ev[c], st[c])); CK(hipStreamWaitEvent(st[0], ev[c], 0)); }
  };
  hipGraphExec_t ge = nullptr;
  if (graph) {
    hipGraph_t g;
    CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
    enqueue();
    CK(hipStreamEndCapture(st[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, st[0]));
    if (graph) CK(hipGraphLaunch(ge, st[0])); else enqueue();
    CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("chains %d wgs %4d spin %4d %s: total %.2f ms = %.2f us per kernel-step (all chains)\n", nchain, wgs, spin, graph ? "graph" : "eager", best, best * 1e3 / iters);
}

int main() {
  for (int graph = 1; graph >= 0; --graph)
    for (int spin : {0, 200})
      for (int nchain : {1, 2, 4, 8}) run(nchain, 64, spin, graph);
  return 0;
}
