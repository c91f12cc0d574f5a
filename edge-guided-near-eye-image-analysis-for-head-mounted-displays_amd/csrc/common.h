// Shared host-side helpers for the C-ABI (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/egne_hip.h"

namespace egne {

char* err_buf();  // thread-local 512-byte buffer (defined in api.hip)

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(EGNE_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return EGNE_OK;
}

#define EGNE_REQUIRE(cond, ...) \
  do { if (!(cond)) return ::egne::fail(EGNE_ERR_ARG, __VA_ARGS__); } while (0)

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace egne
