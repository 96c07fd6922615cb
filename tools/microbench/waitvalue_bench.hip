// Does hipStreamWaitValue32 let a stream follow the progress of a kernel that is still running on another stream, and
// what does the hand-over cost?  (The teacher-forced chain is cut into launches of <= 64 steps only so that the caller's
// stream can follow it with events: each boundary costs the chain a gap of 6.5 us and a set-up of 11 us.)
// Kernel A on stream 1 publishes progress 1 .. N, one value every `period` us, into signal memory; stream 2 waits for value i
// (command-processor wait, no compute unit involved) and then runs a one-thread kernel that notes the time.
// hipcc --offload-arch=gfx950 -O3 -o tools/microbench/waitvalue_bench tools/microbench/waitvalue_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                    \
      return 1;                                                                \
    }                                                                          \
  } while (0)

__global__ void producer(unsigned* flag, unsigned long long* when, int n, int period_ticks) {
  if (threadIdx.x != 0) return;
  unsigned long long t = __builtin_amdgcn_s_memrealtime();
  for (int i = 1; i <= n; ++i) {
    while (__builtin_amdgcn_s_memrealtime() - t < (unsigned long long)period_ticks) __builtin_amdgcn_s_sleep(8);
    t = __builtin_amdgcn_s_memrealtime();
    when[i] = t;
    __threadfence_system();
    __hip_atomic_store(flag, (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void note(unsigned long long* when, int i) { when[i] = __builtin_amdgcn_s_memrealtime(); }

int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  if (!can) return 0;
  const int N = 32;
  unsigned* flag = nullptr;
  CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
  unsigned long long *wp = nullptr, *wc = nullptr;
  CK(hipMalloc((void**)&wp, (N + 1) * 8));
  CK(hipMalloc((void**)&wc, (N + 1) * 8));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(flag, 0, 8));
    CK(hipMemset(wp, 0, (N + 1) * 8));
    CK(hipMemset(wc, 0, (N + 1) * 8));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, s1, flag, wp, N, 5000);   // one value every 50 us
    for (int i = 1; i <= N; ++i) {
      CK(hipStreamWaitValue32(s2, flag, (unsigned)i, hipStreamWaitValueGte, 0xFFFFFFFFu));
      hipLaunchKernelGGL(note, dim3(1), dim3(1), 0, s2, wc, i);
    }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hp(N + 1), hc(N + 1);
    CK(hipMemcpy(hp.data(), wp, (N + 1) * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), wc, (N + 1) * 8, hipMemcpyDeviceToHost));
    double sum = 0, mx = 0;
    for (int i = 1; i <= N; ++i) {
      const double d = (double)(long long)(hc[i] - hp[i]) * 0.01;
      sum += d, mx = d > mx ? d : mx;
    }
    printf("rep %d: value published -> waiting stream's kernel runs: mean %.1f us, max %.1f us (first %.1f, last %.1f)\n", rep, sum / N, mx,
           (double)(long long)(hc[1] - hp[1]) * 0.01, (double)(long long)(hc[N] - hp[N]) * 0.01);
  }
  return 0;
}
