// Microbenchmark: hand-overs BETWEEN a persistent kernel and a stream of per-step launches.
// Models the teacher-forced decoder step at B = 64, T = 128 with the attention as a persistent kernel (encoder memories
// in registers for the whole utterance) beside one launch per step that carries the GRU, the query projection and riders:
//
//   tick(k), stream 1:  G blocks (256): wait C(k-1) [16 rows x 256 dims = 2048 pairs] -> work -> publish H(k) [32 pairs each]
//                       Q blocks (64):  wait H(k)   [64 producers x 32 pairs]        -> work -> publish Q(k) [128 pairs each]
//                       riders (R blocks): stream 52 KB each from a cold buffer (the LSTM cells' weight fetch)
//   persistent P, stream 2 (256 blocks, row b = blk / 4, chunk c = blk % 4), for every k:
//                       wait Q(k) [row's 128 pairs from 16 producers] -> work -> publish R(k) [160 pairs] ->
//                       wait R(k) of the row's other 3 chunks [3 x 49 pairs] -> work -> publish C(k) [32 pairs]
//
// Mode "base": the same chain as two launches per step without a persistent kernel (tick A = G + Q + half the riders,
// tick B = P's work for one step + the other half), i.e. the current schedule.  "work" is a timed sleep, so the two modes
// differ only in what the hand-overs and launches cost.
// Build: hipcc --offload-arch=gfx950 -O3 -o crosskernel_bench crosskernel_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

using u32 = unsigned;
using u64 = unsigned long long;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kB = 64, kChunks = 4, kRG = 4;          // batch rows, attention chunks per row, 16-row groups
constexpr int kG = 256, kQ = 64, kP = kB * kChunks;   // blocks
constexpr u32 kSpin = 1u << 15;

__device__ __forceinline__ u32x4 ld_pair(const u64* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st_pair(u64* p, u32 tag, float a, float b) {
  const u32x4 q = {__float_as_uint(a), tag, __float_as_uint(b), tag};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ void work(int cycles) {
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
}

// Wait until the PER pairs of this thread carry `tag`; returns their sum (a true data dependency).  Bounded.
template <int PER>
__device__ __forceinline__ float wait_pairs(const u64* const (&addr)[PER], const bool (&act)[PER], u32 tag, u32* err) {
  float s = 0.0f;
  for (u32 spins = 0;; ++spins) {
    bool ok = true;
    s = 0.0f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      u32x4 v;
      asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(addr[i]) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (act[i]) ok &= v[1] == tag && v[3] == tag, s += __uint_as_float(v[0]) + __uint_as_float(v[2]);
    }
    if (__all(ok)) break;
    if ((spins & 63) == 63 && *(volatile u32*)err) break;   // somebody gave up: drain
    if (spins > kSpin) {
      if ((threadIdx.x & 63) == 0) atomicOr(err, 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  return s;
}

// Sentinel wait: lanes 0..n-1 of wave 0 poll one pair each (stride `stride` pairs apart), everyone else waits at the barrier.
__device__ __forceinline__ void sentinel(const u64* base, int n, int stride, u32 tag, u32* err) {
  if (threadIdx.x < 64) {
    const u64* a[1] = {base + (size_t)min((int)threadIdx.x, n - 1) * stride * 2};
    const bool act[1] = {(int)threadIdx.x < n};
    wait_pairs<1>(a, act, tag, err);
  }
  __syncthreads();
}

struct Bufs {
  u64 *H, *Q, *R, *C;   // pairs: H [4 rg][64 prod][32], Q [4 rg][16 qb][16 rows][8], R [64 rows][4 chunks][160], C [64 rows][4 chunks][32]
  u32* err;
  const float4* cold;   // riders' buffer
  float* sink;
  int wG, wQ, wP1, wP2; // work in cycles
  int rider_f4;         // float4 per rider thread
  int use_sentinel;
  u64* QS;              // persist2: query slices [64 rows][4 chunks][32 pairs]
};

__device__ __forceinline__ void g_block(const Bufs& p, int blk, int k) {
  const int rg = blk >> 6, tid = threadIdx.x;
  float s = 0.0f;
  if (k > 0) {
    const u64* base = p.C + (size_t)rg * 16 * kChunks * 32 * 2;
    if (p.use_sentinel) sentinel(base, 64, 32, (u32)k, p.err);
    const u64* a[8];
    bool act[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = base + (size_t)(tid + 256 * i) * 2, act[i] = true;
    s = wait_pairs<8>(a, act, (u32)k, p.err);
  }
  work(p.wG);
  if (tid < 32) st_pair(p.H + ((size_t)blk * 32 + tid) * 2, (u32)k + 1, s * 1e-30f + 1.0f, 2.0f);
}
__device__ __forceinline__ void q_block(const Bufs& p, int qb, int k) {
  const int rg = qb >> 4, tid = threadIdx.x;
  const u64* base = p.H + (size_t)rg * 64 * 32 * 2;
  if (p.use_sentinel) sentinel(base, 64, 32, (u32)k + 1, p.err);
  const u64* a[8];
  bool act[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = base + (size_t)(tid + 256 * i) * 2, act[i] = true;
  const float s = wait_pairs<8>(a, act, (u32)k + 1, p.err);
  work(p.wQ);
  if (tid < 128) st_pair(p.Q + ((size_t)qb * 128 + tid) * 2, (u32)k + 1, s * 1e-30f + 1.0f, 2.0f);
}
__device__ __forceinline__ void rider_block(const Bufs& p, int r, int k) {
  const float4* src = p.cold + (size_t)r * 256 * p.rider_f4 + threadIdx.x;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 0; i < p.rider_f4; ++i) {
    const float4 v = src[(size_t)i * 256];
    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.0f) p.sink[r] = (float)k;
}
__device__ __forceinline__ void p_step(const Bufs& p, int blk, int k) {
  const int row = blk >> 2, c = blk & 3, rg = row >> 4, r = row & 15, tid = threadIdx.x;
  const u32 tag = (u32)k + 1;
  float s;
  {  // Q(k): 16 producers x 8 pairs of this row
    const u64* a[1] = {p.Q + ((size_t)((rg * 16 + ((tid >> 3) & 15)) * 16 + r) * 8 + (tid & 7)) * 2};
    const bool act[1] = {tid < 128};
    s = wait_pairs<1>(a, act, tag, p.err);
  }
  __syncthreads();
  work(p.wP1);
  if (tid < 160) st_pair(p.R + ((size_t)blk * 160 + tid) * 2, tag, s * 1e-30f + 1.0f, 2.0f);
  {  // R(k) of the other three chunks: 17 pairs of statistics + 32 pairs of this block's context slice
    const int which = tid / 49, e = tid % 49, c2 = (c + 1 + which) & 3;
    const int idx = e < 17 ? e : 32 + 32 * c + (e - 17);
    const u64* a[1] = {p.R + ((size_t)(row * 4 + c2) * 160 + min(idx, 159)) * 2};
    const bool act[1] = {tid < 147};
    s += wait_pairs<1>(a, act, tag, p.err);
  }
  __syncthreads();
  work(p.wP2);
  if (tid < 32) st_pair(p.C + ((size_t)blk * 32 + tid) * 2, tag, s * 1e-30f + 1.0f, 2.0f);
}

// persist2: the query projection inside P.  Block (row, c) waits for the row's new state (128 pairs from the 64 G blocks of
// its row group), projects query dims [64c, 64c + 64), exchanges the slices inside the row, then continues as p_step.
__device__ __forceinline__ void p2_step(const Bufs& p, int blk, int k) {
  const int row = blk >> 2, c = blk & 3, rg = row >> 4, r = row & 15, tid = threadIdx.x;
  const u32 tag = (u32)k + 1;
  float s;
  {
    const u64* a[1] = {p.H + ((size_t)(rg * 64 + ((tid >> 1) & 63)) * 32 + r * 2 + (tid & 1)) * 2};
    const bool act[1] = {tid < 128};
    s = wait_pairs<1>(a, act, tag, p.err);
  }
  __syncthreads();
  work(p.wQ / 2);
  if (tid < 32) st_pair(p.QS + ((size_t)blk * 32 + tid) * 2, tag, s * 1e-30f + 1.0f, 2.0f);
  {
    const int which = tid >> 5, c2 = (c + 1 + which) & 3;
    const u64* a[1] = {p.QS + ((size_t)(row * 4 + c2) * 32 + (tid & 31)) * 2};
    const bool act[1] = {tid < 96};
    s += wait_pairs<1>(a, act, tag, p.err);
  }
  __syncthreads();
  work(p.wP1);
  if (tid < 160) st_pair(p.R + ((size_t)blk * 160 + tid) * 2, tag, s * 1e-30f + 1.0f, 2.0f);
  {
    const int which = tid / 49, e = tid % 49, c2 = (c + 1 + which) & 3;
    const int idx = e < 17 ? e : 32 + 32 * c + (e - 17);
    const u64* a[1] = {p.R + ((size_t)(row * 4 + c2) * 160 + min(idx, 159)) * 2};
    const bool act[1] = {tid < 147};
    s += wait_pairs<1>(a, act, tag, p.err);
  }
  __syncthreads();
  work(p.wP2);
  if (tid < 32) st_pair(p.C + ((size_t)blk * 32 + tid) * 2, tag, s * 1e-30f + 1.0f, 2.0f);
}
__global__ __launch_bounds__(256) void persistent2_kernel(Bufs p, int steps) {
  for (int k = 0; k < steps; ++k) {
    p2_step(p, blockIdx.x, k);
    if (*(volatile u32*)p.err) break;
  }
}
__global__ __launch_bounds__(256) void tick2_kernel(Bufs p, int k, int riders, int rider0) {
  const int blk = blockIdx.x;
  if (blk < kG) g_block(p, blk, k);
  else rider_block(p, rider0 + blk - kG, k);
}
__global__ __launch_bounds__(256) void tick_kernel(Bufs p, int k, int riders, int rider0) {
  const int blk = blockIdx.x;
  if (blk < kG) g_block(p, blk, k);
  else if (blk < kG + kQ) q_block(p, blk - kG, k);
  else rider_block(p, rider0 + blk - kG - kQ, k);
}
__global__ __launch_bounds__(256) void persistent_kernel(Bufs p, int steps) {
  for (int k = 0; k < steps; ++k) {
    p_step(p, blockIdx.x, k);
    if (*(volatile u32*)p.err) break;
  }
}
__global__ __launch_bounds__(256) void tickb_kernel(Bufs p, int k, int riders, int rider0) {
  const int blk = blockIdx.x;
  if (blk < kP) p_step(p, blk, k);
  else rider_block(p, rider0 + blk - kP, k);
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "persist";
  const int steps = argc > 2 ? atoi(argv[2]) : 1000;
  const int riders = argc > 3 ? atoi(argv[3]) : 512;
  const float scale = argc > 4 ? atof(argv[4]) : 1.0f;   // multiplies the modelled work
  Bufs p{};
  p.use_sentinel = argc > 5 ? atoi(argv[5]) : 1;
  CK(hipMalloc(&p.QS, 64 * 4 * 32 * 16)); CK(hipMemset(p.QS, 0, 64 * 4 * 32 * 16));
  const size_t nH = 4 * 64 * 32, nQ = 4 * 16 * 16 * 8, nR = 64 * 4 * 160, nC = 64 * 4 * 32;
  CK(hipMalloc(&p.H, nH * 16)); CK(hipMalloc(&p.Q, nQ * 16)); CK(hipMalloc(&p.R, nR * 16)); CK(hipMalloc(&p.C, nC * 16));
  CK(hipMemset(p.H, 0, nH * 16)); CK(hipMemset(p.Q, 0, nQ * 16)); CK(hipMemset(p.R, 0, nR * 16)); CK(hipMemset(p.C, 0, nC * 16));
  CK(hipMalloc(&p.err, 4)); CK(hipMemset(p.err, 0, 4));
  p.rider_f4 = 13;  // 256 threads x 13 x 16 B = 52 KB per rider block
  const size_t cold_bytes = (size_t)(riders > 0 ? riders : 1) * 256 * p.rider_f4 * 16;
  CK(hipMalloc((void**)&p.cold, cold_bytes)); CK(hipMemset((void*)p.cold, 0, cold_bytes));
  CK(hipMalloc(&p.sink, 4096 * 4));
  // modelled compute (cycles at ~2.4 GHz): GRU 1.2 us, query 0.6 us, attention energies 2.5 us, combine 0.4 us
  p.wG = (int)(2900 * scale), p.wQ = (int)(1400 * scale), p.wP1 = (int)(6000 * scale), p.wP2 = (int)(1000 * scale);
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemsetAsync(p.H, 0, nH * 16, s1)); CK(hipMemsetAsync(p.Q, 0, nQ * 16, s1)); CK(hipMemsetAsync(p.R, 0, nR * 16, s1)); CK(hipMemsetAsync(p.C, 0, nC * 16, s1));
    CK(hipMemsetAsync(p.QS, 0, 64 * 4 * 32 * 16, s1));
    CK(hipStreamSynchronize(s1));
    CK(hipEventRecord(e0, s1));
    if (!strcmp(mode, "persist2")) {
      persistent2_kernel<<<kP, 256, 0, s2>>>(p, steps);
      for (int k = 0; k < steps; ++k) tick2_kernel<<<kG + riders, 256, 0, s1>>>(p, k, riders, 0);
    } else if (!strcmp(mode, "persist")) {
      persistent_kernel<<<kP, 256, 0, s2>>>(p, steps);
      for (int k = 0; k < steps; ++k) tick_kernel<<<kG + kQ + riders, 256, 0, s1>>>(p, k, riders, 0);
    } else {
      for (int k = 0; k < steps; ++k) {
        tick_kernel<<<kG + kQ + riders / 2, 256, 0, s1>>>(p, k, riders / 2, 0);
        tickb_kernel<<<kP + riders - riders / 2, 256, 0, s1>>>(p, k, riders - riders / 2, riders / 2);
      }
    }
    CK(hipEventRecord(e1, s1));
    CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    u32 err = 0;
    CK(hipMemcpy(&err, p.err, 4, hipMemcpyDeviceToHost));
    printf("mode %s steps %d riders %d work x%.2f sentinel %d: %.2f us per step%s\n", mode, steps, riders, scale, p.use_sentinel, ms * 1e3 / steps, err ? "  TIMEOUT" : "");
    if (err) return 1;
  }
  return 0;
}
