// Ablation of the small-batch stage GEMM (pq-shaped: N=256, K=256, 64 rows): which part costs time?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using f32x4 = __attribute__((ext_vector_type(4))) float;

// MODE bits: 1 = MFMA, 2 = LDS cross-wave reduce, 4 = bias load in epilogue, 8 = (unused; loads are always guarded),
//            16 = interleaved X, 32 = skip W loads, 64 = skip X loads, 128 = rotate chunk order per block
template <int MODE, int NB, int U>
__global__ __launch_bounds__(256) void gemm_k(const float4* __restrict__ wt, int KC, const float* __restrict__ x, int ld,
                                              int rows, const float* __restrict__ bias, float* __restrict__ y, int ldy) {
  __shared__ f32x4 red[4][NB][64];
  const int tile = blockIdx.x, row0 = blockIdx.y * 16 * NB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, s = lane >> 4;
  f32x4 acc[NB];
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0, 0, 0, 0};
  const float4* wp = wt + (size_t)tile * KC * 64 + lane;
  const int n_it = (KC - wave + 3) >> 2;
  for (int i0 = 0; i0 < n_it; i0 += U) {
    float4 w[U], xv[U][NB];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int qi = (MODE & 128) ? (i0 + q + tile) % n_it : i0 + q;   // rotate the chunk order per block
      const int kc = wave + 4 * qi;
      const bool in = i0 + q < n_it;   // always guarded: chunks past KC are out of bounds
      w[q] = (in && !(MODE & 32)) ? wp[(size_t)kc * 64] : make_float4(1, 1, 1, 1);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int row = row0 + nb * 16 + j;
        const bool ok = in && row < rows && !(MODE & 64);
        const float* xa = (MODE & 16) ? x + ((size_t)(kc * 4 + s) * rows + row) * 4   // [K/4][rows][4] interleaved
                                          : x + (size_t)row * ld + kc * 16 + s * 4;        // row-major
        xv[q][nb] = ok ? *reinterpret_cast<const float4*>(xa) : make_float4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < U; ++q)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (MODE & 1) {
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q].x, xv[q][nb].x, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q].y, xv[q][nb].y, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q].z, xv[q][nb].z, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q].w, xv[q][nb].w, acc[nb], 0, 0, 0);
        } else {
          acc[nb][0] += w[q].x * xv[q][nb].x; acc[nb][1] += w[q].y * xv[q][nb].y;
          acc[nb][2] += w[q].z * xv[q][nb].z; acc[nb][3] += w[q].w * xv[q][nb].w;
        }
      }
  }
  if (MODE & 2) {
    for (int nb = 0; nb < NB; ++nb) red[wave][nb][lane] = acc[nb];
    __syncthreads();
    for (int idx = threadIdx.x; idx < NB * 64; idx += 256) {
      const int nb = idx >> 6, l = idx & 63;
      f32x4 v = (red[0][nb][l] + red[1][nb][l]) + (red[2][nb][l] + red[3][nb][l]);
      const int row = row0 + nb * 16 + (l & 15), n = tile * 16 + (l >> 4) * 4;
      if (MODE & 4) for (int e = 0; e < 4; ++e) v[e] += bias[n + e];
      if (row < rows) *reinterpret_cast<f32x4*>(y + (size_t)row * ldy + n) = v;
    }
  } else {
    if (wave == 0) for (int nb = 0; nb < NB; ++nb) {
      const int row = row0 + nb * 16 + j, n = tile * 16 + s * 4;
      f32x4 v = acc[nb];
      if (MODE & 4) for (int e = 0; e < 4; ++e) v[e] += bias[n + e];
      if (row < rows) *reinterpret_cast<f32x4*>(y + (size_t)row * ldy + n) = v;
    }
  }
}

template <int MODE, int NB, int U>
void run(const char* name, int N, int K, int rows) {
  int tiles = N / 16, KC = K / 16;
  float4* wt; float *x, *bias, *y;
  CK(hipMalloc(&wt, (size_t)tiles * KC * 64 * 16)); CK(hipMalloc(&x, (size_t)rows * K * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&y, (size_t)rows * N * 4));
  CK(hipMemset(wt, 0, (size_t)tiles * KC * 64 * 16)); CK(hipMemset(x, 0, (size_t)rows * K * 4)); CK(hipMemset(bias, 0, N * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipGraph_t g; hipGraphExec_t ge; int iters = 1000;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_k<MODE, NB, U>), dim3(tiles, (rows + 16 * NB - 1) / (16 * NB)), dim3(256), 0, st, wt, KC, x, K, rows, bias, y, N);
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("N %4d K %4d rows %3d  %-44s %.2f us\n", N, K, rows, name, best * 1e3 / iters);
}

int main() {
  for (auto [N, K] : std::vector<std::pair<int, int>>{{256, 256}, {2048, 1024}, {2048, 512}}) {
    run<0, 2, 8>("loads only (VALU sum, no LDS, no bias)", N, K, 64);
    run<16, 2, 8>("loads only, X interleaved [K/4][B][4]", N, K, 64);
    run<16 + 32, 2, 8>("X loads only (interleaved), no W", N, K, 64);
    run<16 + 64, 2, 8>("W loads only, no X", N, K, 64);
    run<16 + 32 + 64, 2, 8>("no loads at all", N, K, 64);
    run<16 + 32 + 128, 2, 8>("X loads only, rotated chunk order", N, K, 64);
    run<31 + 128, 2, 8>("full, X interleaved, rotated", N, K, 64);
    run<31, 2, 8>("full, X interleaved, U=8", N, K, 64);
    run<31, 2, 4>("full, X interleaved, U=4", N, K, 64);
    run<31, 4, 4>("full, X interleaved, NB=4 U=4", N, K, 64);
    run<31, 1, 8>("full, X interleaved, NB=1 U=8", N, K, 64);
    run<1, 2, 8>("+ MFMA", N, K, 64);
    run<3, 2, 8>("+ MFMA + LDS reduce", N, K, 64);
    run<7, 2, 8>("+ MFMA + LDS + bias", N, K, 64);
    run<15, 2, 8>("+ MFMA + LDS + bias + guarded loads (U=8)", N, K, 64);
    run<15, 2, 4>("same, U=4", N, K, 64);
    run<15, 2, 2>("same, U=2", N, K, 64);
    run<15, 4, 4>("NB=4 (all 64 rows per WG), U=4", N, K, 64);
    run<15, 1, 8>("NB=1 (16 rows per WG), U=8", N, K, 64);
  }
  return 0;
}
