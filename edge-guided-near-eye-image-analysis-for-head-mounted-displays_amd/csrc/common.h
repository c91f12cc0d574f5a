// Shared host-side helpers for the C-ABI (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/egne_hip.h"

// 256 bytes of zeros in device memory: invalid (out-of-image / padded-channel) lanes load from here with
// an UNCONDITIONAL load (pointer select) instead of a conditional load -- hipcc otherwise branches around
// each load and waits vmcnt(0) per element, which serialises the whole staging phase.
static __device__ __attribute__((aligned(256), used)) float egne_zero_page[64] = {0};

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

bool halo3_supported(const egne_conv_desc& d);   // conv_halo3.hip
int halo3_launch(const egne_conv_desc& d, hipStream_t st);

bool wgrad_halo_supported(const egne_conv_desc& d, long long gzs);   // wgrad_halo.hip
int wgrad_halo_splits(const egne_conv_desc& d);
int wgrad_halo_launch(const egne_conv_desc& d, const float* gz, long long gzs, int gzo, float* ws, hipStream_t st);
int wgrad_halo_f16_launch(const egne_conv_desc& d, const float* gz, long long gzs, int gzo, const unsigned* gz_dyn, float* ws, hipStream_t st);

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace egne
