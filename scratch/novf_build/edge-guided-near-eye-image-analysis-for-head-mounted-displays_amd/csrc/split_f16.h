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

// Device-side pre-scale (egne_conv_desc.dyn_scale): `dyn` holds the bit pattern of max |x| of the input; a_scale becomes the
// power of two that puts that maximum in [1024, 2048) (what engine._a_scale_for computes on the host for frozen networks)
// and out_scale takes its inverse.  Scalar loads and scalar ALU only; both stay in SGPRs.
__device__ __forceinline__ void dyn_scales(const unsigned* dyn, float& a_scale, float& out_scale) {
  if (dyn) {
    unsigned eb = (unsigned)__builtin_amdgcn_readfirstlane((int)*dyn) >> 23;
    if (eb) {
      eb = eb < 27u ? 27u : (eb > 227u ? 227u : eb);
      out_scale *= a_scale * __builtin_bit_cast(float, (eb - 10u) << 23);     // (the host folded ITS a_scale into out_scale)
      a_scale = __builtin_bit_cast(float, (264u - eb) << 23);
    }
  }
}

}  // namespace egne
