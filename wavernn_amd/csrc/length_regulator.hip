// Length-regulator expand for gfx950: replaces the per-item repeat_interleave +
// pad_sequence loop of the reference (models/common_layers.py:17-24) with one
// prefix-sum kernel and one coalesced HBM-streaming row-copy kernel.
//
// Bound: HBM.  Algorithmic bytes = 4*C*B*(T + M) + 4*B*T (read x once, write the
// padded output once, read the durations).
#include <mutex>

#include "common.h"
#include "../../include/wrnn_mi355.h"

// Scratch of wrnn_lr_regulate_static (the allocation-free entry point): the per-token prefix sums that
// wrnn_lr_prepare / wrnn_lr_expand pass through a caller-allocated buffer.  One copy per GPU the library runs on,
// held until the process ends: 1 MiB.  In host code this NAME is only a handle for hipGetSymbolAddress; every use
// of the memory goes through the per-device pointer resolved from it (scratch_ptr below).
constexpr int kLrStaticMaxTokens = 1 << 18;  // B * T
static __device__ __align__(256) int32_t scratch[kLrStaticMaxTokens];

namespace {

using v4f = __attribute__((ext_vector_type(4))) float;

constexpr int kPrepThreads = 256;

// one workgroup per batch row: clamp negatives in place, round, inclusive scan
__global__ __launch_bounds__(kPrepThreads) void lr_prepare_kernel(float* __restrict__ dur, int T,
                                                                   int32_t* __restrict__ cumsum,
                                                                   int32_t* __restrict__ lens) {
  __shared__ int part[kPrepThreads];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int per = (T + kPrepThreads - 1) / kPrepThreads;
  const int lo = tid * per;
  const int hi = min(lo + per, T);
  float* d = dur + (size_t)b * T;
  int32_t* cs = cumsum + (size_t)b * T;
  int local = 0;
  for (int j = lo; j < hi; ++j) {
    float v = d[j];
    if (v < 0.0f) {
      v = 0.0f;
      d[j] = 0.0f;  // the reference mutates the caller's tensor (common_layers.py:18)
    }
    local += (int)(v + 0.5f);  // (dur + 0.5).long(): fp32 add, then truncate
  }
  part[tid] = local;
  __syncthreads();
  // Hillis-Steele inclusive scan over the 256 partials
  for (int off = 1; off < kPrepThreads; off <<= 1) {
    int add = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  int run = part[tid] - local;  // exclusive prefix of this thread's chunk
  for (int j = lo; j < hi; ++j) {
    run += (int)(d[j] + 0.5f);
    cs[j] = run;
  }
  if (tid == kPrepThreads - 1) lens[b] = part[tid];
}

// grid (ceil(M / ROWS), B); each workgroup writes ROWS output rows of one item.
// Lanes 0..ROWS-1 find the source token of each row by binary search over the item's prefix sums
// (staged in LDS); then the 4 waves stream the rows as float4, UNR rows in flight per wave.
// Two tunings, measured on the B=64 / B=256, T=128, C=512 shapes (tools/microbench/lr_bench.hip):
//   output fits the 256 MiB Infinity Cache: plain stores, 32 rows, 1 in flight   (7.2 TB/s algorithmic)
//   larger outputs: non-temporal stores, 16 rows, 4 in flight                     (6.5 TB/s; plain: 4.1-5.4)
template <int ROWS, int UNR, bool NT>
__global__ __launch_bounds__(256) void lr_expand_kernel(const float* __restrict__ x,
                                                         const int32_t* __restrict__ cumsum, int T,
                                                         int C4, float* __restrict__ out, int M) {
  static_assert(ROWS % (4 * UNR) == 0 && ROWS <= 64, "rows are dealt to 4 waves, UNR at a time");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* cs = reinterpret_cast<int32_t*>(smem);  // [T]
  int32_t* src = cs + T;                           // [ROWS]
  const int b = blockIdx.y;
  const int m0 = blockIdx.x * ROWS;
  const int tid = threadIdx.x;
  for (int j = tid; j < T; j += 256) cs[j] = cumsum[(size_t)b * T + j];
  __syncthreads();
  if (tid < ROWS) {
    const int m = m0 + tid;
    // first j with cs[j] > m (upper bound); T if none => padding row
    int lo = 0, hi = T;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (cs[mid] > m) hi = mid; else lo = mid + 1;
    }
    src[tid] = lo;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const v4f* x4 = reinterpret_cast<const v4f*>(x) + (size_t)b * T * C4;
  v4f* o4 = reinterpret_cast<v4f*>(out) + (size_t)b * M * C4;
  const v4f z = {0.f, 0.f, 0.f, 0.f};
  for (int rr = wave * UNR; rr < ROWS; rr += 4 * UNR) {
    if (m0 + rr >= M) break;
    for (int c = lane; c < C4; c += 64) {
      v4f v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int j = (m0 + rr + u < M) ? src[rr + u] : T;
        v[u] = j < T ? x4[(size_t)j * C4 + c] : z;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int m = m0 + rr + u;
        if (m < M) {
          if (NT) __builtin_nontemporal_store(v[u], o4 + (size_t)m * C4 + c);
          else o4[(size_t)m * C4 + c] = v[u];
        }
      }
    }
  }
}

constexpr size_t kCacheResidentBytes = (size_t)160 << 20;  // output sizes up to this use the cache-friendly tuning

// Device address of `scratch` on the CURRENT device, resolved once per device (each GPU holds its own copy of the
// array at its own address) and kept for the life of the process.
constexpr int kMaxDevices = 64;
int scratch_ptr(int32_t** out) {
  static std::mutex mu;
  static int32_t* ptr[kMaxDevices] = {};
  int dev = -1;
  WRNN_HIP(hipGetDevice(&dev));
  WRNN_REQUIRE(dev >= 0 && dev < kMaxDevices, "wrnn_lr_regulate_static: device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(mu);
  if (!ptr[dev]) {
    void* p = nullptr;
    WRNN_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    WRNN_REQUIRE(p != nullptr, "wrnn_lr_regulate_static: hipGetSymbolAddress returned a null address");
    ptr[dev] = static_cast<int32_t*>(p);
  }
  *out = ptr[dev];
  return 0;
}

}  // namespace

extern "C" int wrnn_lr_prepare(float* dur, int B, int T, int32_t* cumsum, int32_t* lens, void* stream) {
  WRNN_REQUIRE(B >= 0 && T >= 0, "wrnn_lr_prepare: negative size (B=%d, T=%d)", B, T);
  if (B == 0) return 0;
  WRNN_REQUIRE(((dur && cumsum) || T == 0) && lens, "wrnn_lr_prepare: null pointer");
  hipLaunchKernelGGL(lr_prepare_kernel, dim3(B), dim3(kPrepThreads), 0, (hipStream_t)stream, dur, T,
                     cumsum, lens);
  WRNN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wrnn_lr_expand(const float* x, const int32_t* cumsum, int B, int T, int C, float* out,
                              int M, void* stream) {
  WRNN_REQUIRE(B >= 0 && T >= 0 && C >= 0 && M >= 0, "wrnn_lr_expand: negative size");
  if (B == 0 || M == 0 || C == 0) return 0;
  WRNN_REQUIRE(C % 4 == 0, "wrnn_lr_expand: C=%d must be a multiple of 4", C);
  WRNN_REQUIRE(((x && cumsum) || T == 0) && out, "wrnn_lr_expand: null pointer");
  WRNN_REQUIRE(B <= 65535, "wrnn_lr_expand: B=%d exceeds grid.y limit", B);
  const size_t lds = (size_t)(T + 32) * sizeof(int32_t);
  WRNN_REQUIRE(lds <= 64 * 1024, "wrnn_lr_expand: T=%d too large for the LDS prefix table", T);
  const size_t out_bytes = (size_t)B * M * C * sizeof(float);
  if (out_bytes <= kCacheResidentBytes) {
    hipLaunchKernelGGL((lr_expand_kernel<32, 1, false>), dim3((M + 31) / 32, B), dim3(256), lds, (hipStream_t)stream, x,
                       cumsum, T, C / 4, out, M);
  } else {
    hipLaunchKernelGGL((lr_expand_kernel<16, 4, true>), dim3((M + 15) / 16, B), dim3(256), lds, (hipStream_t)stream, x,
                       cumsum, T, C / 4, out, M);
  }
  WRNN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wrnn_lr_regulate_static(const float* x, float* dur, int B, int T, int C, float* out, int M_cap,
                                       int32_t* lens, void* stream) {
  WRNN_REQUIRE(B >= 0 && T >= 0 && C >= 0 && M_cap >= 0, "wrnn_lr_regulate_static: negative size");
  if (B == 0) return 0;
  // (T == 0 runs the pair as well: lens = 0 and zero rows, as wrnn_lr_prepare + wrnn_lr_expand give)
  WRNN_REQUIRE((long)B * T <= kLrStaticMaxTokens,
               "wrnn_lr_regulate_static: B*T = %ld exceeds the static scratch of %d tokens; use wrnn_lr_prepare + "
               "wrnn_lr_expand with a caller-allocated prefix-sum buffer", (long)B * T, kLrStaticMaxTokens);
  WRNN_REQUIRE(((x && dur) || T == 0) && lens && (out || M_cap == 0 || C == 0), "wrnn_lr_regulate_static: null pointer");
  int32_t* cs = nullptr;
  if (scratch_ptr(&cs)) return 1;
  // every word of the scratch that the expand reads (B*T prefix sums) is rewritten by the prepare kernel of the
  // same call, on the same stream: nothing has to start at zero
  if (wrnn_lr_prepare(dur, B, T, cs, lens, stream)) return 1;
  return wrnn_lr_expand(x, cs, B, T, C, out, M_cap, stream);
}
