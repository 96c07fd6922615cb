// Microbenchmark: the seam of a persistent, weights-resident recurrent kernel at SMALL batch.
// 256 workgroups (one per compute unit) run a chain of dependent stages; in every stage each workgroup
// publishes N/256 values as 8-byte {tag, value} granules (write-through stores) and then gathers all N
// granules of the stage into LDS (sc1 loads, re-swept until every tag matches), the value it publishes next
// depending on what it gathered.  N = 128 .. 4096 covers a 128..512-wide activation vector for 1..8 batch rows.
//   sweep 0: wave 0 alone sweeps (N/64 granules per lane)
//   sweep 1: all waves sweep (N/threads granules per thread)
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

using u32 = unsigned;
using u64 = unsigned long long;

constexpr int NWG = 256;
constexpr int kStageBufs = 8;
constexpr int SPIN_LIMIT = 1 << 20;

__device__ __forceinline__ u64 ld_g(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_g(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int THREADS, int SWEEP, int MAXPER>
__global__ __launch_bounds__(THREADS) void gather_kernel(u64* gran /*[kStageBufs][N]*/, int N, int stages, float* out, u32* timeout) {
  extern __shared__ float xs[];  // [N]
  __shared__ u32 sh_fail;
  __shared__ float red[THREADS / 64];
  const int wg = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) sh_fail = 0;
  __syncthreads();
  const int per_wg = N / NWG > 0 ? N / NWG : 1;     // values this workgroup publishes (N < 256: only the first N workgroups publish)
  const bool publisher = wg * per_wg < N;
  float carry = 0.0f;
  const int nsweep = SWEEP == 0 ? 64 : THREADS;
  for (int it = 0; it < stages; ++it) {
    const u32 epoch = it + 1;
    u64* g = gran + (size_t)(it % kStageBufs) * N;
    // publish: value depends on what was gathered in the previous stage
    if (publisher && tid < per_wg) {
      const float v = (float)((wg * per_wg + tid + it) & 255) + carry;
      st_g(g + wg * per_wg + tid, ((u64)epoch << 32) | __float_as_uint(v));
    }
    // gather
    if (tid < nsweep) {
      int spins = 0;
      u64 x[MAXPER];
      bool ok;
      do {
        ok = true;
#pragma unroll
        for (int i = 0; i < MAXPER; ++i) {
          const int idx = tid + i * nsweep;
          x[i] = ld_g(g + (idx < N ? idx : N - 1));
          ok &= (u32)(x[i] >> 32) == epoch;
        }
        if (++spins > SPIN_LIMIT) { sh_fail = 1; break; }
      } while (!__all(ok));
#pragma unroll
      for (int i = 0; i < MAXPER; ++i) {
        const int idx = tid + i * nsweep;
        if (idx < N) xs[idx] = __uint_as_float((u32)x[i]);
      }
    }
    __syncthreads();
    if (sh_fail) { if (tid == 0) atomicOr(timeout, 1u); break; }
    // "compute": block sum of the gathered vector, folded into the next published value (x 0 keeps it bounded
    // but still a true dependency)
    float a = 0.0f;
    for (int i = tid; i < N; i += THREADS) a += xs[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    float s = 0.0f;
    for (int w = 0; w < THREADS / 64; ++w) s += red[w];
    carry = s * 1e-30f;
    if (it == stages - 1 && tid == 0) out[wg] = s;
    __syncthreads();
  }
}

// Pipelined polling: every wave keeps DEPTH sweeps of ITS granules in flight, issued a fraction of a round trip apart,
// and checks them in order as they return, so that the memory side is sampled DEPTH times per round trip.
template <int THREADS, int PER, int DEPTH>
__global__ __launch_bounds__(THREADS) void gather_pipe_kernel(u64* gran, int N, int stages, float* out, u32* timeout, int gap) {
  extern __shared__ float xs[];
  __shared__ u32 sh_fail;
  __shared__ float red[THREADS / 64];
  const int wg = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) sh_fail = 0;
  __syncthreads();
  const int per_wg = N / NWG > 0 ? N / NWG : 1;
  const bool publisher = wg * per_wg < N;
  float carry = 0.0f;
  for (int it = 0; it < stages; ++it) {
    const u32 epoch = it + 1;
    u64* g = gran + (size_t)(it % kStageBufs) * N;
    if (publisher && tid < per_wg) {
      const float v = (float)((wg * per_wg + tid + it) & 255) + carry;
      st_g(g + wg * per_wg + tid, ((u64)epoch << 32) | __float_as_uint(v));
    }
    {
      u64 x[DEPTH][PER];
      auto issue = [&](int d) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int idx = tid + i * THREADS;
          x[d][i] = ld_g(g + (idx < N ? idx : N - 1));
        }
      };
      auto ready = [&](int d) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < PER; ++i) ok &= (u32)(x[d][i] >> 32) == epoch;
        return __all(ok);
      };
      auto stage_out = [&](int d) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int idx = tid + i * THREADS;
          if (idx < N) xs[idx] = __uint_as_float((u32)x[d][i]);
        }
      };
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        issue(d);
        if (d + 1 < DEPTH)
          for (int q = 0; q < gap; ++q) __builtin_amdgcn_s_sleep(2);
      }
      bool got = false;
      for (int spins = 0; !got && spins < SPIN_LIMIT; ++spins) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          if (!got) {
            if (ready(d)) { stage_out(d); got = true; }
            else issue(d);
          }
        }
      }
      if (!got) sh_fail = 1;
    }
    __syncthreads();
    if (sh_fail) { if (tid == 0) atomicOr(timeout, 1u); break; }
    float a = 0.0f;
    for (int i = tid; i < N; i += THREADS) a += xs[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    float s = 0.0f;
    for (int w = 0; w < THREADS / 64; ++w) s += red[w];
    carry = s * 1e-30f;
    if (it == stages - 1 && tid == 0) out[wg] = s;
    __syncthreads();
  }
}

template <int THREADS, int PER, int DEPTH>
double run_pipe(int N, int stages, int gap) {
  u64* gran; u32* tmo; float* out;
  const size_t bytes = (size_t)kStageBufs * N * 8;
  CK(hipMalloc(&gran, bytes)); CK(hipMalloc(&tmo, 4)); CK(hipMalloc(&out, NWG * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(gran, 0, bytes)); CK(hipMemset(tmo, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((gather_pipe_kernel<THREADS, PER, DEPTH>), dim3(NWG), dim3(THREADS), N * 4, 0, gran, N, stages, out, tmo, gap);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  u32 t; CK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
  if (t) printf("  !! TIMEOUT N %d\n", N);
  std::vector<float> h(NWG); CK(hipMemcpy(h.data(), out, NWG * 4, hipMemcpyDeviceToHost));
  double exp = 0; const int it = stages - 1;
  for (int i = 0; i < N; ++i) exp += (double)((i + it) & 255);
  int bad = 0;
  for (int w = 0; w < NWG; ++w) if (fabs(h[w] - exp) > 1e-3 * exp + 1) bad++;
  if (bad) printf("  check N %d: %d workgroups MISMATCH\n", N, bad);
  CK(hipFree(gran)); CK(hipFree(tmo)); CK(hipFree(out));
  return best * 1e3 / stages;
}

template <int THREADS, int SWEEP, int MAXPER>
double run(int N, int stages, bool check) {
  u64* gran; u32* tmo; float* out;
  const size_t bytes = (size_t)kStageBufs * N * 8;
  CK(hipMalloc(&gran, bytes)); CK(hipMalloc(&tmo, 4)); CK(hipMalloc(&out, NWG * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(gran, 0, bytes)); CK(hipMemset(tmo, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((gather_kernel<THREADS, SWEEP, MAXPER>), dim3(NWG), dim3(THREADS), N * 4, 0, gran, N, stages, out, tmo);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  u32 t; CK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
  if (t) printf("  !! TIMEOUT N %d\n", N);
  if (check) {
    std::vector<float> h(NWG); CK(hipMemcpy(h.data(), out, NWG * 4, hipMemcpyDeviceToHost));
    double exp = 0; const int it = stages - 1;
    for (int i = 0; i < N; ++i) exp += (double)((i + it) & 255);
    int bad = 0;
    for (int w = 0; w < NWG; ++w) if (fabs(h[w] - exp) > 1e-3 * exp + 1) bad++;
    if (bad) printf("  check N %d: %d workgroups MISMATCH (exp %.0f got %.0f)\n", N, bad, exp, h[0]);
  }
  CK(hipFree(gran)); CK(hipFree(tmo)); CK(hipFree(out));
  return best * 1e3 / stages;
}

// Placement of the granules: one value per workgroup (N = 256), slot(w) = perm(w) * STRIDE; the sweep reads every slot.
//   MODE 0: slot = w * STRIDE (a 128-B line is written by 16 / STRIDE workgroups of all XCDs)
//   MODE 1: slot = ((w % 8) * 32 + w / 8) * STRIDE (a line's writers share blockIdx % 8, i.e. an XCD under round-robin dispatch)
template <int STRIDE, int MODE>
__global__ __launch_bounds__(512) void gather_place_kernel(u64* gran, int stages, float* out, u32* timeout) {
  __shared__ float xs[256];
  __shared__ u32 sh_fail;
  __shared__ float red[8];
  const int wg = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) sh_fail = 0;
  __syncthreads();
  const int myslot = (MODE == 0 ? wg : ((wg & 7) * 32 + (wg >> 3))) * STRIDE;
  float carry = 0.0f;
  for (int it = 0; it < stages; ++it) {
    const u32 epoch = it + 1;
    u64* g = gran + (size_t)(it % kStageBufs) * 256 * STRIDE;
    if (tid == 0) st_g(g + myslot, ((u64)epoch << 32) | __float_as_uint((float)((wg + it) & 255) + carry));
    if (tid < 256) {
      int spins = 0;
      u64 x;
      do {
        x = ld_g(g + tid * STRIDE);
        if (++spins > SPIN_LIMIT) { sh_fail = 1; break; }
      } while (!__all((u32)(x >> 32) == epoch));
      xs[tid] = __uint_as_float((u32)x);
    }
    __syncthreads();
    if (sh_fail) { if (tid == 0) atomicOr(timeout, 1u); break; }
    float a = tid < 256 ? xs[tid] : 0.0f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    float s = 0.0f;
    for (int w = 0; w < 8; ++w) s += red[w];
    carry = s * 1e-30f;
    if (it == stages - 1 && tid == 0) out[wg] = s;
    __syncthreads();
  }
}

template <int STRIDE, int MODE>
double run_place(int stages) {
  u64* gran; u32* tmo; float* out;
  const size_t bytes = (size_t)kStageBufs * 256 * STRIDE * 8;
  CK(hipMalloc(&gran, bytes)); CK(hipMalloc(&tmo, 4)); CK(hipMalloc(&out, NWG * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(gran, 0, bytes)); CK(hipMemset(tmo, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((gather_place_kernel<STRIDE, MODE>), dim3(NWG), dim3(512), 0, 0, gran, stages, out, tmo);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  u32 t; CK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
  if (t) printf("  !! TIMEOUT place\n");
  CK(hipFree(gran)); CK(hipFree(tmo)); CK(hipFree(out));
  return best * 1e3 / stages;
}


// Participants: all-to-all seam among NW workgroups only (NW = 32 .. 256 of the 256 compute units), every workgroup
// publishing PAIRS 16-byte granule pairs ({tag, v0, tag, v1}, one write-through store each, from lanes 0 .. PAIRS-1 of one
// store instruction) into its own 64-byte slot (128-byte for PAIRS > 4) and polling every slot with 16-byte loads.
// What a stage-per-group layout of a persistent kernel would pay per hand-over.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
template <int PAIRS, int GAP = -1>
__global__ __launch_bounds__(512) void gather_group_kernel(u64* gran, int nw, int stages, float* out, u32* timeout) {
  __shared__ float xs[2 * 256 * 8];
  __shared__ u32 sh_fail;
  __shared__ float red[8];
  constexpr int SLOT = PAIRS > 4 ? 16 : 8;   // granules per slot
  const int wg = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) sh_fail = 0;
  __syncthreads();
  float carry = 0.0f;
  const int npoll = nw * PAIRS;   // 16-byte pieces to fetch
  for (int it = 0; it < stages; ++it) {
    const u32 epoch = it + 1;
    u64* g = gran + (size_t)(it % kStageBufs) * 256 * SLOT;
    if (tid < PAIRS) {
      const float v = (float)((wg + it) & 255) + carry;
      const u32x4 q = {__float_as_uint(v), epoch, __float_as_uint(v), epoch};
      asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(g + wg * SLOT + 2 * tid), "v"(q) : "memory");
    }
    for (int base = 0; base < npoll; base += 512) {
      const int i = base + tid;
      const bool mine = i < npoll;
      if (!__any(mine)) continue;
      const u64* a = g + (mine ? (i / PAIRS) * SLOT + 2 * (i % PAIRS) : 0);
      u32x4 q;
      int spins = 0;
      if constexpr (GAP < 0) {
        do {
          asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(a) : "memory");
          if (++spins > SPIN_LIMIT) { sh_fail = 1; break; }
        } while (!__all(!mine || (q.y == epoch && q.w == epoch)));
      } else {
        // two polls in flight, issued GAP sleep units apart: the memory side is sampled twice per round trip
        u32x4 qa, qb;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(qa) : "v"(a) : "memory");
        for (int z = 0; z < GAP; ++z) __builtin_amdgcn_s_sleep(1);
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(qb) : "v"(a) : "memory");
        for (;;) {
          asm volatile("s_waitcnt vmcnt(1)" : "+v"(qa)::"memory");
          if (__all(!mine || (qa.y == epoch && qa.w == epoch))) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(qb)::"memory"); q = qa; break; }
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(qa) : "v"(a) : "memory");
          asm volatile("s_waitcnt vmcnt(1)" : "+v"(qb)::"memory");
          if (__all(!mine || (qb.y == epoch && qb.w == epoch))) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(qa)::"memory"); q = qb; break; }
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(qb) : "v"(a) : "memory");
          if (++spins > SPIN_LIMIT) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(qa), "+v"(qb)::"memory"); sh_fail = 1; q = qa; break; }
        }
      }
      if (mine) xs[2 * i] = __uint_as_float(q.x), xs[2 * i + 1] = __uint_as_float(q.z);
    }
    __syncthreads();
    if (sh_fail) { if (tid == 0) atomicOr(timeout, 1u); break; }
    float a = 0.0f;
    for (int i = tid; i < 2 * npoll; i += 512) a += xs[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    float s2 = 0.0f;
    for (int w = 0; w < 8; ++w) s2 += red[w];
    carry = s2 * 1e-30f;
    if (it == stages - 1 && tid == 0) out[wg] = s2;
    __syncthreads();
  }
}

template <int PAIRS, int GAP = -1>
double run_group(int nw, int stages) {
  u64* gran; u32* tmo; float* out;
  const size_t bytes = (size_t)kStageBufs * 256 * 16 * 8;
  CK(hipMalloc(&gran, bytes)); CK(hipMalloc(&tmo, 4)); CK(hipMalloc(&out, NWG * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(gran, 0, bytes)); CK(hipMemset(tmo, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((gather_group_kernel<PAIRS, GAP>), dim3(nw), dim3(512), 0, 0, gran, nw, stages, out, tmo);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  u32 t; CK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
  if (t) printf("  !! TIMEOUT group\n");
  CK(hipFree(gran)); CK(hipFree(tmo)); CK(hipFree(out));
  return best * 1e3 / stages;
}


// Teams: 256 workgroups in 8 teams of 32, every team its own all-to-all (PAIRS granule pairs per workgroup).  XCD = 1: a
// team = the workgroups with the same blockIdx % 8 (dealt to one XCD: the hand-over stays in that XCD's L2); XCD = 0: a
// team = 32 consecutive block indices (spread over all eight XCDs).  `teams` of the 8 take part, the others exit.
// ST: 0 plain store, 1 sc0, 2 sc1; LD: 1 sc0 (past the CU's L1, may hit the XCD's L2), 2 sc1
template <int PAIRS, int XCD, int ST = 2, int LD = 2>
__global__ __launch_bounds__(512) void gather_team_kernel(u64* gran, int teams, int stages, float* out, u32* timeout) {
  __shared__ float xs[2 * 32 * 16];
  __shared__ u32 sh_fail;
  __shared__ float red[8];
  constexpr int SLOT = 2 * PAIRS;   // granules per slot
  const int tid = threadIdx.x;
  const int team = XCD ? (blockIdx.x & 7) : (blockIdx.x >> 5), wg = XCD ? (blockIdx.x >> 3) : (blockIdx.x & 31);
  if (team >= teams) return;
  if (tid == 0) sh_fail = 0;
  __syncthreads();
  float carry = 0.0f;
  constexpr int npoll = 32 * PAIRS;   // 16-byte pieces to fetch (<= 512)
  for (int it = 0; it < stages; ++it) {
    const u32 epoch = it + 1;
    u64* g = gran + ((size_t)(it % kStageBufs) * 8 + team) * 32 * SLOT;
    if (tid < PAIRS) {
      const float v = (float)((wg + it) & 255) + carry;
      const u32x4 q = {__float_as_uint(v), epoch, __float_as_uint(v), epoch};
      if constexpr (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(g + wg * SLOT + 2 * tid), "v"(q) : "memory");
      else if constexpr (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(g + wg * SLOT + 2 * tid), "v"(q) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(g + wg * SLOT + 2 * tid), "v"(q) : "memory");
    }
    {
      const int i = tid;
      const bool mine = i < npoll;
      if (__any(mine)) {
        const u64* a = g + 2 * (mine ? i : 0);
        u32x4 q;
        int spins = 0;
        do {
          if constexpr (LD == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(a) : "memory");
          else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(a) : "memory");
          if (++spins > (1 << 14)) { sh_fail = 1; break; }
        } while (!__all(!mine || (q.y == epoch && q.w == epoch)));
        if (mine) xs[2 * i] = __uint_as_float(q.x), xs[2 * i + 1] = __uint_as_float(q.z);
      }
    }
    __syncthreads();
    if (sh_fail) { if (tid == 0) atomicOr(timeout, 1u); break; }
    float a = 0.0f;
    for (int i = tid; i < 2 * npoll; i += 512) a += xs[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    float s2 = 0.0f;
    for (int w = 0; w < 8; ++w) s2 += red[w];
    carry = s2 * 1e-30f;
    if (it == stages - 1 && tid == 0) out[blockIdx.x] = s2;
    __syncthreads();
  }
}

template <int PAIRS, int XCD, int ST = 2, int LD = 2>
double run_team(int teams, int stages) {
  u64* gran; u32* tmo; float* out;
  const size_t bytes = (size_t)kStageBufs * 256 * 32 * 8;
  CK(hipMalloc(&gran, bytes)); CK(hipMalloc(&tmo, 4)); CK(hipMalloc(&out, NWG * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(gran, 0, bytes)); CK(hipMemset(tmo, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((gather_team_kernel<PAIRS, XCD, ST, LD>), dim3(256), dim3(512), 0, 0, gran, teams, stages, out, tmo);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  u32 t; CK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
  if (t) printf("  !! TIMEOUT team\n");
  CK(hipFree(gran)); CK(hipFree(tmo)); CK(hipFree(out));
  return best * 1e3 / stages;
}

template <int THREADS, int SWEEP>
double run_n(int N, int stages) {
  const int nsweep = SWEEP == 0 ? 64 : THREADS;
  const int per = (N + nsweep - 1) / nsweep;
  switch (per) {
    case 1: return run<THREADS, SWEEP, 1>(N, stages, true);
    case 2: return run<THREADS, SWEEP, 2>(N, stages, true);
    case 4: return run<THREADS, SWEEP, 4>(N, stages, true);
    case 8: return run<THREADS, SWEEP, 8>(N, stages, true);
    case 16: return run<THREADS, SWEEP, 16>(N, stages, true);
    case 32: return run<THREADS, SWEEP, 32>(N, stages, true);
    case 64: return run<THREADS, SWEEP, 64>(N, stages, true);
  }
  return -1.0;
}

int main(int argc, char** argv) {
  const int stages = 4000;
  if (argc > 1 && argv[1][0] == 'x') {   // teams of 32 workgroups: same XCD against spread
    printf("all-to-all within teams of 32 workgroups (256 launched), us per stage: team on one XCD | team spread over the XCDs\n");
    for (int teams : {1, 2, 8})
      printf("  %d teams: 8 pairs %.2f | %.2f   16 pairs %.2f | %.2f\n", teams, run_team<8, 1>(teams, stages), run_team<8, 0>(teams, stages),
             run_team<16, 1>(teams, stages), run_team<16, 0>(teams, stages));
    printf("team on one XCD, 16 pairs, store / load modifiers: us per stage (a TIMEOUT line = the hand-over was not seen)\n");
    for (int teams : {1, 8}) {
      printf("  %d teams: st plain ld sc0 %.2f\n", teams, run_team<16, 1, 0, 1>(teams, stages));
      printf("  %d teams: st sc0   ld sc0 %.2f\n", teams, run_team<16, 1, 1, 1>(teams, stages));
      printf("  %d teams: st sc1   ld sc0 %.2f\n", teams, run_team<16, 1, 2, 1>(teams, stages));
      printf("  %d teams: st plain ld sc1 %.2f\n", teams, run_team<16, 1, 0, 2>(teams, stages));
    }
    printf("spread teams (must NOT work with sc0 loads unless the L2s are coherent): st sc1 ld sc0 %.2f\n", run_team<16, 0, 2, 1>(1, 200));
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'g') {   // participants sweep only
    printf("all-to-all among NW workgroups, 16-byte granule pairs, one slot per workgroup: us per stage\n");
    for (int nw : {16, 32, 64, 128, 256})
      printf("  NW %3d: 1 pair %.2f   2 pairs %.2f   4 pairs %.2f   8 pairs %.2f\n", nw, run_group<1>(nw, stages), run_group<2>(nw, stages),
             run_group<4>(nw, stages), run_group<8>(nw, stages));
    printf("two polls in flight (8 pairs per workgroup), gap in s_sleep units between them: us per stage\n");
    for (int nw : {32, 64, 128, 256})
      printf("  NW %3d: gap0 %.2f  gap4 %.2f  gap8 %.2f  gap16 %.2f  gap32 %.2f\n", nw, run_group<8, 0>(nw, stages), run_group<8, 4>(nw, stages),
             run_group<8, 8>(nw, stages), run_group<8, 16>(nw, stages), run_group<8, 32>(nw, stages));
    return 0;
  }
  printf("us per stage (publish N/256 granules per workgroup, gather all N into LDS, block-sum), %d stages, best of 4\n", stages);
  for (int N : {128, 256, 512, 1024, 2048, 4096}) {
    const double a = run_n<256, 0>(N, stages);   // wave 0 sweeps
    const double b = run_n<256, 1>(N, stages);   // all 256 threads sweep
    const double c = run_n<512, 1>(N, stages);   // all 512 threads sweep
    const double d = run_n<512, 0>(N, stages);   // 512-thread workgroups, wave 0 sweeps
    printf("N %5d granules (%6d B): 256thr/wave0 %.2f   256thr/all %.2f   512thr/all %.2f   512thr/wave0 %.2f\n", N, N * 8, a, b, c, d);
  }
  printf("placement, N = 256 (one granule per workgroup), 4 polling waves: stride in granules x mode (0 = by workgroup, 1 = a line's writers share blockIdx %% 8)\n");
  printf("  mode 0: stride1 %.2f stride2 %.2f stride4 %.2f stride8 %.2f stride16 %.2f\n", run_place<1, 0>(stages), run_place<2, 0>(stages),
         run_place<4, 0>(stages), run_place<8, 0>(stages), run_place<16, 0>(stages));
  printf("  mode 1: stride1 %.2f stride2 %.2f stride4 %.2f stride8 %.2f stride16 %.2f\n", run_place<1, 1>(stages), run_place<2, 1>(stages),
         run_place<4, 1>(stages), run_place<8, 1>(stages), run_place<16, 1>(stages));
  printf("pipelined polls, 512 threads (one granule per thread at N = 512): depth x gap (s_sleep 2 units between the first issues)\n");
  for (int gap : {0, 2, 4, 8}) {
    printf("N 512 gap %d: depth1 %.2f depth2 %.2f depth3 %.2f depth4 %.2f   N 256 depth2 %.2f depth4 %.2f   N 1024(per 2) depth2 %.2f depth4 %.2f\n", gap,
           run_pipe<512, 1, 1>(512, stages, gap), run_pipe<512, 1, 2>(512, stages, gap), run_pipe<512, 1, 3>(512, stages, gap),
           run_pipe<512, 1, 4>(512, stages, gap), run_pipe<512, 1, 2>(256, stages, gap), run_pipe<512, 1, 4>(256, stages, gap),
           run_pipe<512, 2, 2>(1024, stages, gap), run_pipe<512, 2, 4>(1024, stages, gap));
  }
  return 0;
}
