// Microbenchmark: per-step cost of an in-launch all-gather among a GROUP of workgroups
// (the seam of a persistent recurrent kernel).  256 WGs x 256 threads, groups of GS WGs
// (group = blockIdx % (256/GS), so that under round-robin dispatch a group of 32 shares an XCD).
// Each step: every WG publishes SLICE floats, then reads all GS slices of its group.
//   variant 0: sc1 16-B payload stores + vmcnt(0) + barrier + agent atomic add on a group counter;
//              consumer: sc1 poll of the counter, barrier, sc1 16-B loads
//   variant 1: 8-byte {tag, value} granules (sc1 stores), consumer re-sweeps until all tags match
//   variant 2: like 0 but per-producer flags (sc1 store) polled by lanes 0..GS-1 of wave 0
// Build: hipcc --offload-arch=gfx950 -O3 -o seam_bench seam_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

using u32 = unsigned;
using u64 = unsigned long long;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) u32;

constexpr int NWG = 256;
constexpr int SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ u32 ld_sc1(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int VARIANT>
__global__ __launch_bounds__(256) void seam_kernel(float* buf /*[2][NWG][SLICE] or granules*/, u32* counters, u32* flags,
                                                    int GS, int SLICE, int steps, float* out, u32* timeout) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int ngroups = NWG / GS;
  const int grp = wg % ngroups, member = wg / ngroups;
  __shared__ u32 sh_fail;
  if (tid == 0) sh_fail = 0;
  __syncthreads();
  float acc = 0.f;
  const int total = GS * SLICE;  // floats gathered per step
  for (int it = 0; it < steps; ++it) {
    const int par = it & 1;
    const u32 epoch = it + 1;
    if (VARIANT == 0 || VARIANT == 2) {
      float* mybase = buf + ((size_t)par * NWG + (size_t)grp * GS + member) * SLICE;
      auto rs = __builtin_amdgcn_make_buffer_rsrc(mybase, 0, SLICE * 4, 0x00020000);
      if (tid * 4 < SLICE) {
        f32x4 v;
        for (int e = 0; e < 4; ++e) v[e] = (float)((it * 7 + wg * 3 + tid * 4 + e) & 1023);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, tid * 16, 0, 16 /*sc1*/);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (VARIANT == 0) {
        if (tid == 0) __hip_atomic_fetch_add(counters + grp * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
          int spins = 0;
          while (ld_sc1(counters + grp * 32) < epoch * GS) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { sh_fail = 1; break; }
          }
        }
      } else {
        if (tid == 0) __hip_atomic_store(flags + (size_t)(grp * GS + member) , epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 64) {
          int spins = 0;
          bool ok;
          do {
            ok = tid < GS ? ld_sc1(flags + (size_t)(grp * GS + tid)) >= epoch : true;
            if (++spins > SPIN_LIMIT) { sh_fail = 1; break; }
          } while (!__all(ok));
        }
      }
      __syncthreads();
      if (sh_fail) { if (tid == 0) atomicOr(timeout, 1u); break; }
      const float* gbase = buf + ((size_t)par * NWG + (size_t)grp * GS) * SLICE;
      auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)gbase, 0, total * 4, 0x00020000);
      for (int i = tid * 4; i < total; i += 1024) {
        u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rg, i * 4, 0, 16 /*sc1*/);
        f32x4 v = __builtin_bit_cast(f32x4, r);
        acc += v[0] + v[1] + v[2] + v[3];
      }
    } else {
      // granules: [par][grp][GS*SLICE] u64
      u64* g = reinterpret_cast<u64*>(buf) + ((size_t)par * NWG + (size_t)grp * GS) * SLICE;
      if (tid < SLICE) {
        float v = (float)((it * 7 + wg * 3 + tid) & 1023);
        __hip_atomic_store(g + member * SLICE + tid, ((u64)epoch << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (int i = tid; i < total; i += 256) {
        int spins = 0;
        u64 x;
        do {
          x = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (++spins > SPIN_LIMIT) { sh_fail = 1; break; }
        } while ((u32)(x >> 32) != epoch);
        acc += __uint_as_float((u32)x);
      }
      __syncthreads();   // keeps the WG's waves in the same step (needed before re-using parity buffers)
      if (sh_fail) { if (tid == 0) atomicOr(timeout, 1u); break; }
    }
  }
  out[(size_t)wg * 256 + tid] = acc;
}

template <int V>
double run(int GS, int SLICE, int steps, bool check) {
  float* buf; u32 *cnt, *flags, *tmo; float* out;
  size_t bufbytes = (size_t)2 * NWG * SLICE * 8;
  CK(hipMalloc(&buf, bufbytes)); CK(hipMalloc(&cnt, 8 * 32 * 4 * 32)); CK(hipMalloc(&flags, NWG * 4)); CK(hipMalloc(&tmo, 4));
  CK(hipMalloc(&out, NWG * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(buf, 0, bufbytes)); CK(hipMemset(cnt, 0, 8 * 32 * 4 * 32)); CK(hipMemset(flags, 0, NWG * 4)); CK(hipMemset(tmo, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(seam_kernel<V>, dim3(NWG), dim3(256), 0, 0, buf, cnt, flags, GS, SLICE, steps, out, tmo);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  u32 t; CK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
  if (t) printf("  !! TIMEOUT in variant %d\n", V);
  if (check) {
    std::vector<float> h(NWG * 256); CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    // expected: per WG sum over steps of all values in its group
    int ngroups = NWG / GS; int bad = 0;
    for (int wg = 0; wg < NWG; wg += 37) {
      int grp = wg % ngroups; double exp = 0, got = 0;
      for (int it = 0; it < steps; ++it) for (int m = 0; m < GS; ++m) { int w = m * ngroups + grp; for (int i = 0; i < SLICE; ++i) exp += (double)((it * 7 + w * 3 + i) & 1023); }
      for (int tdx = 0; tdx < 256; ++tdx) got += h[wg * 256 + tdx];
      if (fabs(exp - got) > 1e-3 * exp) { bad++; printf("  wg %d exp %.0f got %.0f\n", wg, exp, got); }
    }
    printf("  check variant %d GS %d SLICE %d: %s\n", V, GS, SLICE, bad ? "MISMATCH" : "ok");
  }
  CK(hipFree(buf)); CK(hipFree(cnt)); CK(hipFree(flags)); CK(hipFree(tmo)); CK(hipFree(out));
  return best * 1e3 / steps;  // us per step
}

int main() {
  const int steps = 2000;
  printf("us per step (all-gather within group), %d steps, best of 4\n", steps);
  for (int GS : {32, 64, 256})
    for (int SLICE : {128, 256}) {
      double a = run<0>(GS, SLICE, steps, true);
      double b = run<1>(GS, SLICE, steps, true);
      double c = run<2>(GS, SLICE, steps, true);
      printf("GS %3d SLICE %4d floats (gather %6d B): counter %.2f  granule %.2f  flags %.2f\n", GS, SLICE, GS * SLICE * 4, a, b, c);
    }
  return 0;
}
