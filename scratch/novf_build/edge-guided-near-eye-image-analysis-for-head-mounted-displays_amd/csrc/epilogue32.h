// Epilogue of one 32 x 32 result block in the v_mfma_f32_32x32x16 layout, shared by the role-split kernels: the lane holds
// output channel n = n0 + (lane & 31) of the 16 pixels x = xl + c_r, c_r = (r & 3) + 8 * (r >> 2), xl = x0 + 4 * (lane >> 5),
// of one image row.  v = act(acc * os + bias) [* post_scale + post_shift] [+ residual]; optional fp64 partial sums of the stored
// values (InstanceNorm statistics of the consumer, egne_norm_stats_finish).
//
// The per-pixel byte offset goes into the instruction's SCALAR offset (c_r * pixel pitch: SALU work) and interior tiles carry no
// per-value range check, so that a value costs 3 vector instructions + its store (mul-add, mul, max) instead of ~25: with one
// producer and one consumer wave per SIMD the vector ALU, not the matrix pipe, was what these kernels were waiting for.
#pragma once
#include "common.h"

namespace egne {

typedef float epi_f32x16 __attribute__((ext_vector_type(16)));

struct EpiLane {          // per lane and 32-channel block, fixed for the launch
  float bias, post_scale, post_shift;
};

// MASK: edge tile (pixels past W, rows past H, channels past Cout_store are dropped: cm = valid pixels from xl, 0 for none);
// POST: post affine present; RES: residual present (a load per value); STATS: accumulate st_s / st_q.
template <bool MASK, bool POST, bool RES, bool STATS, bool CHECK = true>
__device__ __forceinline__ void epi_row32(const epi_f32x16& acc, const __amdgpu_buffer_rsrc_t rout, const __amdgpu_buffer_rsrc_t rres,
                                          int voff, int roff, int out_step, int res_step, int cm, float os, float slope,
                                          const EpiLane& k, double& st_s, double& st_q, bool& bad) {
  constexpr int OOBO = (int)0x80000000u;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = (r & 3) + 8 * (r >> 2);
    const bool ok = !MASK || c < cm;
    float v = acc[r] * os + k.bias;
    v = fmaxf(v, v * slope);
    if constexpr (POST) v = v * k.post_scale + k.post_shift;
    if constexpr (RES) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, ok ? roff : OOBO, c * res_step, 0));
    if constexpr (CHECK) bad |= egne_nonfinite(v);          // (egne_conv_desc.ovf_flag: an f16 operand left its range upstream of this value)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, ok ? voff : OOBO, c * out_step, 0);
    if constexpr (STATS) {
      const double vm = ok ? (double)v : 0.;
      st_s += vm; st_q += vm * vm;
    }
  }
}

// runtime (wave-uniform) selection of the specialisation.  `post` / `res`: post affine / residual present.  The convBlock head of
// ESF-Net (eval BatchNorm folded into a post affine, no residual; utils.py:1047-1049) is the most expensive launch of that network:
// its interior tiles take a form of their own -- the generic one issued a residual load per value from an empty resource.
template <bool STATS>
__device__ __forceinline__ void epi_row32_select(bool mask, bool post, bool res, const epi_f32x16& acc, const __amdgpu_buffer_rsrc_t rout,
                                                 const __amdgpu_buffer_rsrc_t rres, int voff, int roff, int out_step, int res_step, int cm,
                                                 float os, float slope, const EpiLane& k, double& st_s, double& st_q, bool& bad, bool chk) {
  // `chk` (wave-uniform): this image row is one the overflow test looks at (egne_ovf_row); the common interior forms skip it otherwise
  if (!chk && !mask && !post && !res) { epi_row32<false, false, false, STATS, false>(acc, rout, rres, voff, roff, out_step, res_step, cm, os, slope, k, st_s, st_q, bad); return; }
  if (!chk && !mask && !res) { epi_row32<false, true, false, STATS, false>(acc, rout, rres, voff, roff, out_step, res_step, cm, os, slope, k, st_s, st_q, bad); return; }
  if (!mask && !post && !res) epi_row32<false, false, false, STATS>(acc, rout, rres, voff, roff, out_step, res_step, cm, os, slope, k, st_s, st_q, bad);
  else if (!post && !res) epi_row32<true, false, false, STATS>(acc, rout, rres, voff, roff, out_step, res_step, cm, os, slope, k, st_s, st_q, bad);
  else if (!mask && !res) epi_row32<false, true, false, STATS>(acc, rout, rres, voff, roff, out_step, res_step, cm, os, slope, k, st_s, st_q, bad);
  else if (!res) epi_row32<true, true, false, STATS>(acc, rout, rres, voff, roff, out_step, res_step, cm, os, slope, k, st_s, st_q, bad);
  else epi_row32<true, true, true, STATS>(acc, rout, rres, voff, roff, out_step, res_step, cm, os, slope, k, st_s, st_q, bad);
}

}  // namespace egne
