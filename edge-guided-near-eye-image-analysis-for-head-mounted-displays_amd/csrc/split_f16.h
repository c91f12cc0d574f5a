// fp32 -> (hi, lo) f16 split used by every split-f16 staging path: hi = f16(x * s), lo = f16(x * s - hi).
//
// Written with one-lane-value VALU instructions on purpose.  hipcc packs adjacent f32 multiplies / subtracts into
// v_pk_mul_f32 / v_pk_add_f32, and next to MFMAs a packed f32 instruction costs ~17 cycles of the SIMD's vector issue instead of
// the 2 x 4 of two plain ones (MI355X_MICROARCH.md, 'price of one filler beside MFMAs'): with a producer and a consumer wave
// sharing each SIMD the packed form took 15-20 % of the tile time of the role-split kernels.  Inline asm keeps the SLP
// vectoriser away; `s` must be wave-uniform (an SGPR).
#pragma once

namespace egne {

typedef float sp_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 sp_h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(float x0, float x1, float s, sp_h2& h, sp_h2& l) {
  float t0, t1, d0, d1;
  asm("v_mul_f32_e32 %0, %1, %2" : "=v"(t0) : "s"(s), "v"(x0));
  asm("v_mul_f32_e32 %0, %1, %2" : "=v"(t1) : "s"(s), "v"(x1));
  const sp_f32x2 t = {t0, t1};
  h = __builtin_convertvector(t, sp_h2);
  const float f0 = (float)h[0], f1 = (float)h[1];
  asm("v_sub_f32_e32 %0, %1, %2" : "=v"(d0) : "v"(t0), "v"(f0));
  asm("v_sub_f32_e32 %0, %1, %2" : "=v"(d1) : "v"(t1), "v"(f1));
  const sp_f32x2 d = {d0, d1};
  l = __builtin_convertvector(d, sp_h2);
}

}  // namespace egne
