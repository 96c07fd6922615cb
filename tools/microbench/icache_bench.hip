// Does straight-line code size cost time in tiny kernels (cold instruction cache at every launch)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const int* __restrict__ in, int* __restrict__ out, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int v = in[i % n];
  if (MODE == 1) asm volatile(".rept 1024\n s_nop 0\n .endr" ::: "memory");         // 4 KB straight line
  if (MODE == 2) asm volatile(".rept 4096\n s_nop 0\n .endr" ::: "memory");         // 16 KB straight line
  if (MODE == 3) for (int j = 0; j < 64; ++j) asm volatile(".rept 16\n s_nop 0\n .endr" ::: "memory");   // 1024 nops, looped
  if (MODE == 4) for (int j = 0; j < 256; ++j) asm volatile(".rept 16\n s_nop 0\n .endr" ::: "memory");  // 4096 nops, looped
  out[i] = v;
}

template <int MODE>
void run(int wgs, const char* name) {
  int n = wgs * 256, iters = 1000;
  int *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n);
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("wgs %4d %-28s %.2f us per kernel\n", wgs, name, best * 1e3 / iters);
}

int main() {
  for (int wgs : {32, 256}) {
    run<0>(wgs, "no padding");
    run<1>(wgs, "1024 nops straight (4 KB)");
    run<3>(wgs, "1024 nops looped");
    run<2>(wgs, "4096 nops straight (16 KB)");
    run<4>(wgs, "4096 nops looped");
  }
  return 0;
}
