// Measures the shader clock a small kernel sees (s_memtime vs the 100 MHz s_memrealtime),
// the latency of one dependent global load chain, and MFMA f32 16x16x4 issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ void clock_kernel(unsigned long long* out, const int* chain, int n_fma, int n_ld, int n_mfma) {
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float a = threadIdx.x * 1e-3f;
  for (int i = 0; i < n_fma; ++i) a = fmaf(a, 1.0001f, 0.5f);
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int v = threadIdx.x & 63;
  for (int i = 0; i < n_ld; ++i) v = chain[v];
  unsigned long long c2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
  f32x4 acc = {0, 0, 0, 0};
  for (int i = 0; i < n_mfma; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 1.0f, acc, 0, 0, 0);
  unsigned long long c3 = __builtin_amdgcn_s_memtime(), r3 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    unsigned long long* o = out + blockIdx.x * 16;
    o[0] = c1 - c0; o[1] = r1 - r0; o[2] = c2 - c1; o[3] = r2 - r1; o[4] = c3 - c2; o[5] = r3 - r2;
    o[6] = (unsigned long long)(a + v + acc[0]);
  }
}

int main() {
  unsigned long long* out; int* chain;
  CK(hipMalloc(&out, 1024 * 16 * 8)); CK(hipMalloc(&chain, 4096 * 4));
  int h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (i * 67 + 13) & 4095;
  CK(hipMemcpy(chain, h, sizeof(h), hipMemcpyHostToDevice));
  for (int blocks : {1, 256, 1024}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(clock_kernel, dim3(blocks), dim3(64), 0, 0, out, chain, 20000, 200, 2000);
      CK(hipDeviceSynchronize());
    }
    unsigned long long o[16]; CK(hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost));
    printf("blocks %4d: fma: %.0f cyc in %.2f us -> %.2f GHz (%.2f cyc/fma) | load chain: %.1f ns/load (%.0f cyc) | mfma: %.1f cyc each, clock %.2f GHz\n", blocks,
           (double)o[0], o[1] * 0.01, o[0] / (o[1] * 10.0), (double)o[0] / 20000, o[3] * 10.0 / 200, (double)o[2] / 200, (double)o[4] / 2000, o[4] / (o[5] * 10.0));
  }
  return 0;
}
