// Shared host-side helpers for libwrnn_mi355.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <string>

namespace wrnn {

inline std::string& last_error() {
  static thread_local std::string e;
  return e;
}

inline int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return 1;
}

#define WRNN_HIP(expr)                                                              \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess)                                                           \
      return ::wrnn::fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define WRNN_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) return ::wrnn::fail(__VA_ARGS__); \
  } while (0)

// ONE stream per device for every persistent launch whose workgroups wait for each other inside the launch and need one
// workgroup on every compute unit (chain_kernel, bilstm_resident_kernel): two such grids half-resident at the same time
// would wait for each other until their spin limits, so all of them, from every handle and host thread, are queued on this
// stream (stream order is exactly the order they need) and ordered against their callers' streams with events.  Hold the
// mutex while enqueueing.
inline std::mutex& persistent_mutex() {
  static std::mutex m;
  return m;
}
// (which = 1: a second such stream per device, for the grids that are built to be resident BESIDE a grid of the first one:
//  the decoder's LSTM stack next to the chain launches, taco_stack.h; grids of one stream still never overlap each other)
inline int persistent_stream(int device, hipStream_t* out, int which = 0) {
  static hipStream_t streams[2][64] = {};
  hipStream_t& s = streams[which & 1][device & 63];
  if (!s) {
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) return fail("hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
  }
  *out = s;
  return 0;
}

// Enqueue launch(ps) on the current device's persistent stream ps, after the work already on `st` and before what is
// enqueued on `st` next (two events per device, re-recorded per call: a wait refers to the record before it).
template <class F>
int on_persistent_stream(hipStream_t st, F&& launch) {
  static hipEvent_t ev[64][2] = {};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return fail("hipGetDevice failed");
  std::lock_guard<std::mutex> lock(persistent_mutex());
  hipStream_t ps = nullptr;
  if (persistent_stream(device, &ps)) return 1;
  hipEvent_t* e = ev[device & 63];
  for (int i = 0; i < 2; ++i)
    if (!e[i] && hipEventCreateWithFlags(&e[i], hipEventDisableTiming) != hipSuccess) return fail("hipEventCreateWithFlags failed");
  if (hipEventRecord(e[0], st) != hipSuccess || hipStreamWaitEvent(ps, e[0], 0) != hipSuccess) return fail("persistent stream: event hand-in failed");
  if (int rc = launch(ps)) return rc;
  if (hipEventRecord(e[1], ps) != hipSuccess || hipStreamWaitEvent(st, e[1], 0) != hipSuccess) return fail("persistent stream: event hand-back failed");
  return 0;
}

// check the launch itself (bad configuration); execution errors surface at the next sync
#define WRNN_LAUNCH_CHECK() WRNN_HIP(hipGetLastError())


// A bounded poll loop's "keep waiting?" after a sweep that did not find what it waits for.  The first 4096 sweeps cost one
// scalar compare; after that (the producer is not resident yet, or somebody has timed out) the wave looks at the launch's
// error word.  Past `limit` sweeps it raises the word itself.  Everything is wave-uniform on purpose: a per-lane read of the
// error word makes `gave_up` divergent and the compiler then wraps every sweep of every poll loop in dozens of scalar
// exec-mask instructions, which delays the detection of every hand-over (measured on the persistent free-running kernel:
// 9.6 -> 10.05 us per step).  The caller sleeps between sweeps as it sees fit.
__device__ __forceinline__ bool keep_polling(unsigned spins, unsigned limit, unsigned* err, bool& gave_up) {
  if (spins < 4096u) return true;
  const unsigned e = __builtin_amdgcn_readfirstlane(*(volatile unsigned*)err);
  if (e != 0u || spins > limit) {
    if (e == 0u && (threadIdx.x & 63) == 0) atomicOr(err, 1u);
    gave_up = true;
    return false;
  }
  return true;
}

}  // namespace wrnn
