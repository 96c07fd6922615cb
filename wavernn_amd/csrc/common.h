// Shared host-side helpers for libwrnn_mi355.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
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

// check the launch itself (bad configuration); execution errors surface at the next sync
#define WRNN_LAUNCH_CHECK() WRNN_HIP(hipGetLastError())

}  // namespace wrnn
