// Diagnostic: variants of the length-regulator expand kernel (rows per workgroup, rows in flight per wave,
// store cache policy) at the BASELINE config-4 shape.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o lr_bench lr_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using v4f = __attribute__((ext_vector_type(4))) float;

template <int ROWS, int UNR, bool NT>
__global__ __launch_bounds__(256) void expand(const float* __restrict__ x, const int* __restrict__ cumsum, int T, int C4,
                                              float* __restrict__ out, int M) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* cs = reinterpret_cast<int*>(smem);
  int* src = cs + T;
  const int b = blockIdx.y, m0 = blockIdx.x * ROWS, tid = threadIdx.x;
  for (int j = tid; j < T; j += 256) cs[j] = cumsum[(size_t)b * T + j];
  __syncthreads();
  if (tid < ROWS) {
    const int m = m0 + tid;
    int lo = 0, hi = T;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cs[mid] > m) hi = mid; else lo = mid + 1; }
    src[tid] = lo;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const v4f* x4 = reinterpret_cast<const v4f*>(x) + (size_t)b * T * C4;
  v4f* o4 = reinterpret_cast<v4f*>(out) + (size_t)b * M * C4;
  const v4f z = {0.f, 0.f, 0.f, 0.f};
  for (int rr = wave * UNR; rr < ROWS; rr += 4 * UNR) {
    for (int c = lane; c < C4; c += 64) {
      v4f v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int m = m0 + rr + u;
        const int j = (m < M) ? src[rr + u] : T;
        v[u] = j < T ? x4[(size_t)j * C4 + c] : z;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int m = m0 + rr + u;
        if (m < M) { if (NT) __builtin_nontemporal_store(v[u], o4 + (size_t)m * C4 + c); else o4[(size_t)m * C4 + c] = v[u]; }
      }
    }
  }
}

template <int ROWS, int UNR, bool NT>
void run(const char* name, const float* x, const int* cs, int B, int T, int C, float* out, int M, double bytes) {
  dim3 grid((M + ROWS - 1) / ROWS, B);
  size_t lds = (size_t)(T + ROWS) * 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((expand<ROWS, UNR, NT>), grid, dim3(256), lds, 0, x, cs, T, C / 4, out, M);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((expand<ROWS, UNR, NT>), grid, dim3(256), lds, 0, x, cs, T, C / 4, out, M);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  printf("%-28s %7.2f us  %6.0f GB/s\n", name, best * 1e3 / 20, bytes / (best * 1e-3 / 20) / 1e9);
}

int main() {
  const int B = 64, T = 128, C = 512;
  std::mt19937 g(4); std::uniform_int_distribution<int> d(1, 11);
  std::vector<int> cs(B * T); int M = 0;
  for (int b = 0; b < B; ++b) { int run_ = 0; for (int j = 0; j < T; ++j) { run_ += d(g); cs[b * T + j] = run_; } M = run_ > M ? run_ : M; }
  float *x, *out; int* dcs;
  CK(hipMalloc(&x, (size_t)B * T * C * 4)); CK(hipMalloc(&out, (size_t)B * M * C * 4)); CK(hipMalloc(&dcs, cs.size() * 4));
  CK(hipMemset(x, 0, (size_t)B * T * C * 4)); CK(hipMemcpy(dcs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
  const double bytes = 4.0 * C * B * (T + M) + 4.0 * B * T;
  printf("M=%d bytes=%.1f MB\n", M, bytes / 1e6);
  run<32, 1, true>("rows32 unr1 nt (shipped)", x, dcs, B, T, C, out, M, bytes);
  run<32, 1, false>("rows32 unr1 plain", x, dcs, B, T, C, out, M, bytes);
  run<32, 2, true>("rows32 unr2 nt", x, dcs, B, T, C, out, M, bytes);
  run<32, 4, true>("rows32 unr4 nt", x, dcs, B, T, C, out, M, bytes);
  run<32, 8, true>("rows32 unr8 nt", x, dcs, B, T, C, out, M, bytes);
  run<64, 4, true>("rows64 unr4 nt", x, dcs, B, T, C, out, M, bytes);
  run<64, 8, true>("rows64 unr8 nt", x, dcs, B, T, C, out, M, bytes);
  run<16, 4, true>("rows16 unr4 nt", x, dcs, B, T, C, out, M, bytes);
  run<64, 4, false>("rows64 unr4 plain", x, dcs, B, T, C, out, M, bytes);
  run<128, 8, true>("rows128 unr8 nt", x, dcs, B, T, C, out, M, bytes);
  run<16, 4, false>("rows16 unr4 plain", x, dcs, B, T, C, out, M, bytes);
  run<16, 2, false>("rows16 unr2 plain", x, dcs, B, T, C, out, M, bytes);
  run<32, 4, false>("rows32 unr4 plain", x, dcs, B, T, C, out, M, bytes);
  run<32, 2, false>("rows32 unr2 plain", x, dcs, B, T, C, out, M, bytes);
  run<32, 8, false>("rows32 unr8 plain", x, dcs, B, T, C, out, M, bytes);
  run<8, 2, false>("rows8 unr2 plain", x, dcs, B, T, C, out, M, bytes);
  {  // 4x the batch: output 454 MB, larger than the 256 MB Infinity Cache
    const int B4 = 256; std::vector<int> cs4(B4 * T); for (int i = 0; i < B4 * T; ++i) cs4[i] = cs[i % (B * T)];
    float *x4, *out4; int* dcs4;
    CK(hipMalloc(&x4, (size_t)B4 * T * C * 4)); CK(hipMalloc(&out4, (size_t)B4 * M * C * 4)); CK(hipMalloc(&dcs4, cs4.size() * 4));
    CK(hipMemset(x4, 0, (size_t)B4 * T * C * 4)); CK(hipMemcpy(dcs4, cs4.data(), cs4.size() * 4, hipMemcpyHostToDevice));
    const double bytes4 = 4.0 * C * B4 * (T + M) + 4.0 * B4 * T;
    printf("B=256: bytes=%.1f MB\n", bytes4 / 1e6);
    run<32, 1, true>("B256 rows32 unr1 nt", x4, dcs4, B4, T, C, out4, M, bytes4);
    run<32, 1, false>("B256 rows32 unr1 plain", x4, dcs4, B4, T, C, out4, M, bytes4);
    run<16, 4, true>("B256 rows16 unr4 nt", x4, dcs4, B4, T, C, out4, M, bytes4);
    run<16, 4, false>("B256 rows16 unr4 plain", x4, dcs4, B4, T, C, out4, M, bytes4);
    run<32, 4, false>("B256 rows32 unr4 plain", x4, dcs4, B4, T, C, out4, M, bytes4);
  }
  return 0;
}
