// Shared host-side helpers for libglnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "glnn_hip.h"

namespace glnn {

void set_error(const char* fmt, ...);

inline int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  set_error("%s", buf);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GLNN_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString(e));
  return GLNN_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define GLNN_REQUIRE(cond, ...)                                     \
  do {                                                              \
    if (!(cond)) return ::glnn::fail(GLNN_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

constexpr int kWave = 64;  // CDNA wavefront

}  // namespace glnn
