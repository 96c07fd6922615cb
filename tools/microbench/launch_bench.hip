// Microbenchmark: cost of a chain of tiny dependent kernels (the v1 "stage kernel" regime).
//  trips = number of dependent global-memory round trips inside each kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o launch_bench launch_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int TRIPS>
__global__ __launch_bounds__(256) void chain_kernel(const int* __restrict__ in, int* __restrict__ out, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int v = i;
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) v = in[(v + t) % n];   // dependent loads
  out[i] = v;
}

template <int TRIPS>
void run(int wgs, int iters, bool graph) {
  int n = wgs * 256;
  int *a, *b;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  hipGraphExec_t ge = nullptr;
  if (graph) {
    hipGraph_t g;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) { hipLaunchKernelGGL(chain_kernel<TRIPS>, dim3(wgs), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n); }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  }
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, st));
    if (graph) CK(hipGraphLaunch(ge, st));
    else for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(chain_kernel<TRIPS>, dim3(wgs), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("wgs %4d trips %d %s: %.2f us per kernel\n", wgs, TRIPS, graph ? "graph" : "eager", best * 1e3 / iters);
  CK(hipFree(a)); CK(hipFree(b));
}

int main() {
  for (int graph = 0; graph < 2; ++graph)
    for (int wgs : {32, 256, 1024}) {
      run<0>(wgs, 2000, graph); run<1>(wgs, 2000, graph); run<2>(wgs, 2000, graph); run<4>(wgs, 2000, graph);
    }
  return 0;
}
