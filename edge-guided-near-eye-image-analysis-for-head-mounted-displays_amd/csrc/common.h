// Shared host-side helpers for the C-ABI (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <unordered_map>
#include "../../include/egne_hip.h"

// 256 bytes of zeros in device memory: invalid (out-of-image / padded-channel) lanes load from here with
// an UNCONDITIONAL load (pointer select) instead of a conditional load -- hipcc otherwise branches around
// each load and waits vmcnt(0) per element, which serialises the whole staging phase.
static __device__ __attribute__((aligned(256), used)) float egne_zero_page[64] = {0};

// ---- storage types of activation tensors ------------------------------------------------------------------------------
// fp32 everywhere by default; training plans may keep activations and activation gradients in HBM as bf16 (BASELINE.json
// configs[2..4]: "bf16"), always with fp32 arithmetic / accumulation.  Kernels that take either are templates over the element
// type and read / write FOUR consecutive channels through ld4 / st4 (16 bytes of fp32, 8 bytes of bf16; round-to-nearest-even
// on store: v_cvt_pk_bf16_f32).  The C-ABI twin of such an entry point carries the suffix _bf16 (include/egne_hip.h).
typedef __bf16 egne_bf16;
typedef float egne_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 egne_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 egne_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ egne_f32x4 ld4(const float* p) { return *(const egne_f32x4*)p; }
__device__ __forceinline__ egne_f32x4 ld4(const egne_bf16* p) { return __builtin_convertvector(*(const egne_bf16x4*)p, egne_f32x4); }
__device__ __forceinline__ void st4(float* p, egne_f32x4 v) { *(egne_f32x4*)p = v; }
__device__ __forceinline__ void st4(egne_bf16* p, egne_f32x4 v) { *(egne_bf16x4*)p = __builtin_convertvector(v, egne_bf16x4); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const egne_bf16* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(egne_bf16* p, float v) { *p = (egne_bf16)v; }
// Widest aligned vector of a storage type: 16 bytes = 4 fp32 or 8 bf16 channels.  The HBM-heavy element-wise / reduction kernels
// move one such vector per lane and access (8-byte loads of 4 bf16 ran at 1.4-1.6 TB/s where 16-byte ones reach 4+).
template <typename T> struct egne_vt { static constexpr int N = 4; };
template <> struct egne_vt<egne_bf16> { static constexpr int N = 8; };
template <int N> struct egne_fv { float v[N]; };
__device__ __forceinline__ egne_fv<4> ldv(const float* p) {
  const egne_f32x4 t = *(const egne_f32x4*)p;
  return egne_fv<4>{{t[0], t[1], t[2], t[3]}};
}
__device__ __forceinline__ egne_fv<8> ldv(const egne_bf16* p) {
  typedef unsigned u4_ __attribute__((ext_vector_type(4)));
  const u4_ w = *(const u4_*)p;
  egne_fv<8> r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[2 * i] = __builtin_bit_cast(float, w[i] << 16);
    r.v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
  }
  return r;
}
__device__ __forceinline__ void stv(float* p, const egne_fv<4>& a) { *(egne_f32x4*)p = egne_f32x4{a.v[0], a.v[1], a.v[2], a.v[3]}; }
__device__ __forceinline__ void stv(egne_bf16* p, const egne_fv<8>& a) {
  const egne_f32x4 lo = {a.v[0], a.v[1], a.v[2], a.v[3]}, hi = {a.v[4], a.v[5], a.v[6], a.v[7]};
  const egne_bf16x4 l = __builtin_convertvector(lo, egne_bf16x4), h = __builtin_convertvector(hi, egne_bf16x4);
  *(egne_bf16x8*)p = egne_bf16x8{l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
}
// the same 16-byte vector kept PACKED (four dwords) until it is used: streaming kernels with several tensors and rows in flight hold
// their loads this way (eight bf16 channels unpacked are eight registers)
typedef unsigned egne_u32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ egne_u32x4 ldraw(const T* p) { return *(const egne_u32x4*)p; }
__device__ __forceinline__ egne_fv<4> unpackv(const float*, egne_u32x4 w) {
  return egne_fv<4>{{__builtin_bit_cast(float, w[0]), __builtin_bit_cast(float, w[1]), __builtin_bit_cast(float, w[2]), __builtin_bit_cast(float, w[3])}};
}
__device__ __forceinline__ egne_fv<8> unpackv(const egne_bf16*, egne_u32x4 w) {
  egne_fv<8> r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[2 * i] = __builtin_bit_cast(float, w[i] << 16);
    r.v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
  }
  return r;
}
template <int N> __device__ __forceinline__ egne_fv<N> ldf(const float* p) {      // N consecutive fp32 table entries
  egne_fv<N> r;
#pragma unroll
  for (int i = 0; i < N; i += 4) { const egne_f32x4 t = *(const egne_f32x4*)(p + i); r.v[i] = t[0]; r.v[i + 1] = t[1]; r.v[i + 2] = t[2]; r.v[i + 3] = t[3]; }
  return r;
}
template <int N> __device__ __forceinline__ egne_fv<N> fv_fill(float x) {
  egne_fv<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = x;
  return r;
}
// the all-zero page typed for either storage (invalid lanes load from it unconditionally)
template <typename T> __device__ __forceinline__ const T* zero_page() { return (const T*)egne_zero_page; }

// ---- f16 overflow of the split operands (round 5) ---------------------------------------------------------------------
// The split-f16 kernels scale an fp32 operand by a power of two calibrated on an earlier batch (engine.Plan: 32x of head-room)
// before they round it to f16.  A later batch whose raw activations exceed that head-room turns operands into +-inf, and every
// accumulator such an operand reaches becomes +-inf or NaN (inf * w, inf * 0, inf - inf): a NON-FINITE accumulator in an epilogue
// is the one reliable trace of it (activations and pools downstream can swallow it again).  Epilogues OR a per-lane test of what
// they store (v_cmp_class_f32: sNaN | qNaN | -inf | +inf, one vector instruction per value) and set the sticky device word
// egne_conv_desc.ovf_flag, which engine.Plan reads back behind every run (Plan.check_overflow: recalibrate and run again).
// What has to be tested (measured: testing every stored value cost 1.7 % of the B=64 inference step): an f16 operand that became
// inf contaminates EVERY output channel of EVERY output pixel whose receptive field holds it (inf * w is inf or NaN for any w,
// zero included).  So a kernel whose lanes hold pixels tests ONE channel per pixel, and a kernel whose lanes hold channels and
// whose registers walk the pixels of an image row tests the rows y % 3 == 1 and the last row of a 3x3 / dilation-1 convolution
// (any three consecutive rows, clipped to the image, contain one of them) -- egne_ovf_row.
#ifdef EGNE_NO_OVF_CHECK      // (A/B builds only: what the tests cost, scratch/ab_ovf.sh)
__device__ __forceinline__ bool egne_nonfinite(float) { return false; }
__device__ __forceinline__ bool egne_nonfinite64(double) { return false; }
#else
__device__ __forceinline__ bool egne_nonfinite(float v) { return __builtin_amdgcn_classf(v, 0x207); }
__device__ __forceinline__ bool egne_nonfinite64(double v) { return !__builtin_isfinite(v); }
#endif
__device__ __forceinline__ bool egne_ovf_row(int y, int H) { return y % 3 == 1 || y == H - 1; }
__device__ __forceinline__ void egne_ovf_commit(bool bad, unsigned* flag) {
  if (flag && bad) atomicOr(flag, 1u);
}

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

// Dynamic-LDS limit of ONE kernel (hipFuncAttributeMaxDynamicSharedMemorySize), raised once per kernel address and again if a later
// launch needs more.  Launch helpers that take the kernel as a generic-lambda / template argument share one function-pointer TYPE
// between all instantiations of a kernel template: a `static bool once` inside them is one flag for all of them, and only the
// first variant launched in the process would get its limit raised (round-5 advisor finding).
inline bool raise_lds(const void* kern, size_t bytes) {
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> have;
  std::lock_guard<std::mutex> g(mu);
  auto it = have.find(kern);
  if (it != have.end() && it->second >= bytes) return true;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  have[kern] = bytes;
  return true;
}

#define EGNE_REQUIRE(cond, ...) \
  do { if (!(cond)) return ::egne::fail(EGNE_ERR_ARG, __VA_ARGS__); } while (0)

bool halo3_supported(const egne_conv_desc& d);   // conv_halo3.hip
int halo3_launch(const egne_conv_desc& d, hipStream_t st);

bool wgrad_halo_supported(const egne_conv_desc& d, long long gzs);   // wgrad_halo.hip
int wgrad_halo_splits(const egne_conv_desc& d);
int wgrad_halo_launch(const egne_conv_desc& d, const float* gz, long long gzs, int gzo, float* ws, hipStream_t st);
int wgrad_halo_f16_launch(const egne_conv_desc& d, const float* gz, long long gzs, int gzo, const unsigned* gz_dyn, float* ws, hipStream_t st);

bool wgrad3x3_bf16_supported(const egne_conv_desc& d, long long gzs);   // wgrad_bf16.hip (bf16 tensors, bf16 MFMA)
int wgrad3x3_bf16_splits(const egne_conv_desc& d);
int wgrad3x3_bf16_launch(const egne_conv_desc& d, const egne_bf16* gz, long long gzs, int gzo, float* ws, hipStream_t st);
bool wgrad1x1_bf16_supported(const egne_conv_desc& d, long long gzs);   // 1x1 over raw bf16 slices, all block pairs in one workgroup
int wgrad1x1_bf16_splits(const egne_conv_desc& d);
int wgrad1x1_bf16_launch(const egne_conv_desc& d, const egne_bf16* gz, long long gzs, int gzo, float* ws, hipStream_t st);

int msdil_ps_launch(const egne_conv_desc& d, const void* fhi, const void* flo, float a_scale, float w_scale, const float* score_w,
                    const float* score_c, float* s0, float* s1, int accumulate, hipStream_t st);   // msblock_dil_ps_f16.hip

bool msdil1_wanted(const egne_conv_desc& d);      // msblock_dil1_f16.hip: plain f16 operands, ring-of-rows form
int msdil1_launch(const egne_conv_desc& d, const void* fhi, float a_scale, float w_scale, const float* score_w, const float* score_c,
                  float* s0, float* s1, int accumulate, hipStream_t st);

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace egne
