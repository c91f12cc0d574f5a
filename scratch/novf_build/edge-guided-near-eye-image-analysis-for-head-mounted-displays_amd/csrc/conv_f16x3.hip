// Split-precision implicit-GEMM convolution: fp32 data, three f16 MFMAs per product, fp32 accumulate.
//
// gfx950 has no xf32/TF32 and its exact-fp32 MFMA runs at 1/16 of the f16 rate, so an fp32 direct
// convolution tops out at 157 TFLOP/s (1213 frames/s for this path).  Here every fp32
// operand is split on the fly into two halves, x = hi + lo with hi = f16(x), lo = f16(x - hi): 22 bits of
// significand, and  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  (the dropped lo*lo term is 2^-22 relative).
// Three v_mfma_f32_32x32x16_f16 cost 96 cycles for K=16 against 512 cycles of eight 32x32x2 fp32 MFMAs.
// Operands are pre-scaled by exact powers of two (weights at pack time, activations while staging) so that
// the lo halves stay in the f16 normal range; the epilogue undoes the scaling exactly.
//
// Used for the frozen BDCN only (single input slice, Cin % 32 == 0, no fused affine): measured error against a
// float64 convolution 4e-7..2e-6 (the fp32 CPU convolution: 2e-7..3e-7), edge map within 3e-6 of the reference;
// ESF-Net (training, gradients) stays on the exact-fp32 kernels.
#include "common.h"
#include "split_f16.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KC = 32;
constexpr int LDH = 40;          // LDS row pitch in halfs (80 B): conflict-free ds_read_b128

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == EGNE_ACT_RELU) return fmaxf(v, 0.f);
  if (act == EGNE_ACT_LEAKY) return v > 0.f ? v : 0.01f * v;
  return v;
}

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// 4 waves arranged WGM x WGN, each wave owns TM x TN MFMA tiles (32x32).  Shapes instantiated:
//   <2,2,2,2>  128 x 128  wide layers (VGG trunk)          <4,1,2,2>  256 x 64
//   <4,1,2,1>  256 x  32  the 32-channel MSBlock convs; GROUPED fuses its three dilated convs + 4-way sum
//
// Addressing (PMC: the first version issued 8.5 VALU + 4.6 SALU per MFMA, as much issue time as the MFMAs): every
// staged row keeps ONE byte offset relative to the first frame of the tile and a bit mask of the taps that fall
// inside the image; a K-step adds a scalar tap offset and selects 0x80000000 for padded lanes, which the
// buffer unit's range check turns into zeros.  Weights, residual and output use buffer instructions with
// lane-constant offsets and scalar step offsets as well.
template <int WGM, int WGN, int TM, int TN, bool GROUPED>
__global__ __launch_bounds__(256) void conv_f16x3_kernel(const egne_conv_desc p, const _Float16* __restrict__ whi,
                                                         const _Float16* __restrict__ wlo, float a_scale,
                                                         float out_scale, float* __restrict__ ksplit_ws) {
  egne::dyn_scales(p.dyn_scale, a_scale, out_scale);
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int AR = BM / 32;                  // A rows staged per thread (float4 each)
  constexpr int BI = (BN * 4 + 255) / 256;     // B 16-byte items per thread and array
  __shared__ __attribute__((aligned(16))) _Float16 lds[(2 * BM + 2 * BN) * LDH];
  _Float16* Ahi = lds;
  _Float16* Alo = Ahi + BM * LDH;
  _Float16* Bhi = Alo + BM * LDH;
  _Float16* Blo = Bhi + BN * LDH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;
  const long long M = (long long)p.B * p.Ho * p.Wo;
  const long long m0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int T = p.kh * p.kw;
  const egne_seg sg = p.seg[0];
  const int hw = p.Ho * p.Wo;
  const int b0 = (int)(m0 / hw);               // first frame of the tile; rows belong to b0 .. b0 + BM/hw + 1
  const int frame_px = p.H * p.W;
  const long long in_left = ((long long)p.B - b0) * frame_px * sg.pix_stride * 4;
  const __amdgpu_buffer_rsrc_t rin = make_rsrc(sg.ptr + (long long)b0 * frame_px * sg.pix_stride,
                                               (unsigned)(in_left < 0x7fffffffll ? in_left : 0x7fffffffll));
  const unsigned wbytes = (unsigned)p.ngroups * T * p.CoutP * p.Ktot * 2u;
  const __amdgpu_buffer_rsrc_t rwh = make_rsrc(whi, wbytes), rwl = make_rsrc(wlo, wbytes);

  const int col4 = tid & 7, rbase = tid >> 3;
  int roff[AR], pb[AR], pyx[AR];   // byte offset of the row's centre pixel (+ channel column), frame index (or -1), oy<<16|ox
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const long long m = m0 + rbase + 32 * i;
    const int b = (int)(m / hw);
    const int r = (int)(m - (long long)b * hw);
    const int oy = r / p.Wo, ox = r - oy * p.Wo;
    pb[i] = m < M ? b : -1; pyx[i] = (oy << 16) | ox;
    roff[i] = ((((b - b0) * p.H + oy) * p.W + ox) * (int)sg.pix_stride + sg.ch_off + col4 * 4) * 4;
  }
  // taps of group g that read inside the image, one bit per tap and row
  unsigned tapmask[AR];
  auto make_masks = [&](int g) {
    const int dil = p.dil[g];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      unsigned mk = 0;
      for (int ky = 0; ky < p.kh; ++ky)
        for (int kx = 0; kx < p.kw; ++kx) {
          const int iy = (pyx[i] >> 16) + (ky - p.pad_h) * dil, ix = (pyx[i] & 0xffff) + (kx - p.pad_w) * dil;
          if (pb[i] >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mk |= 1u << (ky * p.kw + kx);
        }
      tapmask[i] = mk;
    }
  };
  make_masks(0);

  int boff[BI];                // weights: lane-constant byte offset inside one (group, tap) block
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int item = tid + 256 * j;
    const int row = item >> 2, piece = item & 3;
    boff[j] = row < BN ? ((n0 + row) * p.Ktot + piece * 8) * 2 : (int)OOB;
  }

  u32x4 ra[AR];
  u32x4 rbh[BI], rbl[BI];   // 8 halfs each (loaded as 16 B)
  unsigned okmask = 0;
  int st_c = 0;
  int ky_n = 0, kx_n = 0;   // tap coordinates of the step being loaded (no per-step division)
  // egne_conv_desc.f16_products == 1 (wave-uniform): plain f16 operands -- no lo halves loaded, derived, stored or multiplied (the frame
  // tails of the deep trunk layers of the edge network next to a bf16-storage training plan)
  const bool np1 = p.f16_products == 1;
  auto load_step = [&](int g, int tap, int c0) {
    const int dil = p.dil[g];
    const int tapoff = (((ky_n - p.pad_h) * p.W + (kx_n - p.pad_w)) * dil * (int)sg.pix_stride + c0) * 4;
    const unsigned cbad = c0 + col4 * 4 < sg.Cp ? 0u : OOB;   // channel tail of a slice whose width is not a multiple of 32
    okmask = 0;
    st_c = c0 + col4 * 4;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const bool ok = ((tapmask[i] >> tap) & 1u) && !cbad;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? roff[i] + tapoff : (int)OOB, 0, 0);   // tapoff may be negative: not an soffset
      okmask |= (ok ? 1u : 0u) << i;
    }
    const int wstep = (((g * T + tap) * p.CoutP) * p.Ktot + c0) * 2;
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      rbh[j] = __builtin_amdgcn_raw_buffer_load_b128(rwh, boff[j], wstep, 0);
      if (!np1) rbl[j] = __builtin_amdgcn_raw_buffer_load_b128(rwl, boff[j], wstep, 0);
    }
  };
  const float slope_in = sg.act_in == EGNE_ACT_RELU ? 0.f : (sg.act_in == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  const float slope_out = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  auto store_step = [&]() {
    if (!GROUPED && sg.scale) {   // fused InstanceNorm affine (+ activation) of the consumer, zero padding applied after it
      const bool same = pb[0] == pb[AR - 1] && pb[0] >= 0 && st_c < sg.Cp;
      f32x4 sc0 = {0.f, 0.f, 0.f, 0.f}, sh0 = {0.f, 0.f, 0.f, 0.f};
      if (same) {
        sc0 = *(const f32x4*)(sg.scale + (long long)pb[0] * sg.Cp + st_c);
        sh0 = *(const f32x4*)(sg.shift + (long long)pb[0] * sg.Cp + st_c);
      }
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const bool ok = (okmask >> i) & 1u;
        f32x4 sc = sc0, sh = sh0;
        if (!same) {
          sc = *(const f32x4*)(ok ? sg.scale + (long long)pb[i] * sg.Cp + st_c : egne_zero_page);
          sh = *(const f32x4*)(ok ? sg.shift + (long long)pb[i] * sg.Cp + st_c : egne_zero_page);
        } else if (!ok) {
          sc = (f32x4)(0.f); sh = (f32x4)(0.f);
        }
        f32x4 v = __builtin_bit_cast(f32x4, ra[i]) * sc + sh;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope_in);
        ra[i] = __builtin_bit_cast(u32x4, v);
      }
    }
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const f32x4 v = __builtin_bit_cast(f32x4, ra[i]);
      h2 h0, h1, l0, l1;                        // x * a_scale = hi + lo, plain (unpacked) VALU: split_f16.h
      egne::split2(v[0], v[1], a_scale, h0, l0);
      egne::split2(v[2], v[3], a_scale, h1, l1);
      const h4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
      const int o = (rbase + 32 * i) * LDH + col4 * 4;
      *(h4*)&Ahi[o] = hi;
      if (!np1) *(h4*)&Alo[o] = lo;
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      const int item = tid + 256 * j;
      if (BN * 4 % 256 == 0 || (item >> 2) < BN) {
        const int o = (item >> 2) * LDH + (item & 3) * 8;
        *(u32x4*)&Bhi[o] = rbh[j];
        if (!np1) *(u32x4*)&Blo[o] = rbl[j];
      }
    }
  };

  f32x16 acc[TM][TN];
  f32x16 res[GROUPED ? TM : 1][GROUPED ? TN : 1];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      acc[a][b] = (f32x16)(0.f);
      if (GROUPED) res[a][b] = (f32x16)(0.f);
    }

  const int nchunk = (sg.Cp + KC - 1) / KC;
  const int nsteps_all = T * nchunk * p.ngroups;
  // split-K launches (gridDim.z > 1, one group): block z accumulates steps [z * n / Z, (z + 1) * n / Z) and leaves its scaled partial
  // sums in ksplit_ws[z][M][CoutP]; splitk_finish_k adds them up and applies the epilogue
  const int Z = GROUPED ? 1 : (int)gridDim.z, zi = GROUPED ? 0 : (int)blockIdx.z;
  const int step0 = (int)((long long)nsteps_all * zi / Z), nsteps = (int)((long long)nsteps_all * (zi + 1) / Z);
  int g = 0, tap = step0 % T, c0 = (step0 / T) * KC;
  ky_n = tap / p.kw; kx_n = tap - ky_n * p.kw;
  load_step(0, tap, c0);
  store_step();
  __syncthreads();

  const int arow = (wm * TM * 32 + li) * LDH + lh * 8;
  const int brow = (wn * TN * 32 + li) * LDH + lh * 8;
  for (int step = step0; step < nsteps; ++step) {
    int ng = g, ntap = tap + 1, nc0 = c0;
    if (++kx_n == p.kw) { kx_n = 0; ++ky_n; }
    if (ntap == T) {
      ntap = 0; ky_n = 0; kx_n = 0; nc0 += KC;
      if (nc0 >= sg.Cp) { nc0 = 0; ++ng; if (GROUPED && ng < p.ngroups) make_masks(ng); }
    }
    const bool more = step + 1 < nsteps;
    if (more) load_step(ng, ntap, nc0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        ah[t] = *(const h8*)&Ahi[arow + t * 32 * LDH + ks * 16];
        if (!np1) al[t] = *(const h8*)&Alo[arow + t * 32 * LDH + ks * 16];
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        bh[t] = *(const h8*)&Bhi[brow + t * 32 * LDH + ks * 16];
        if (!np1) bl[t] = *(const h8*)&Blo[brow + t * 32 * LDH + ks * 16];
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          if (!np1) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
          }
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
        }
    }
    if (GROUPED && (!more || ng != g)) {   // end of a dilation group: res += act(acc + bias_g)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + (wn * TN + tn) * 32 + li;
        const float bv = (p.bias && n < p.Cout_store) ? p.bias[g * p.CoutP + n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[tm][tn][r] * out_scale + bv;
            res[tm][tn][r] += fmaxf(v, v * slope_out);
          }     // (a non-finite term keeps the sum non-finite: the final store below tests it)
          acc[tm][tn] = (f32x16)(0.f);
        }
      }
    }
    __syncthreads();
    if (more) store_step();
    __syncthreads();
    g = ng; tap = ntap; c0 = nc0;
  }

  // ---- epilogue: lane holds channel n of 16 rows (pixels) m = mrow + c_r, c_r = (r&3) + 8*(r>>2) + 4*lh ----
  const long long left = M - m0;                                    // rows of this tile inside the tensor
  if (!GROUPED && Z > 1) {
    float* wz = ksplit_ws + ((long long)zi * M + m0) * p.CoutP;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n0 + (wn * TN + tn) * 32 + li;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int mrow = (wm * TM + tm) * 32 + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mr = mrow + (r & 3) + 8 * (r >> 2);
          if (mr < left) wz[(long long)mr * p.CoutP + n] = acc[tm][tn][r] * out_scale;      // (splitk_finish_k tests the sum)
        }
      }
    }
    return;
  }
  const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out + m0 * p.out_pix_stride, (unsigned)((left < BM ? left : BM) * p.out_pix_stride * 4));
  const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.residual ? p.residual + m0 * p.res_pix_stride : nullptr,
                                                p.residual ? (unsigned)((left < BM ? left : BM) * p.res_pix_stride * 4) : 0u);
  const int ostep = (int)p.out_pix_stride * 4, rstep = (int)p.res_pix_stride * 4;
  bool bad = false;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + (wn * TN + tn) * 32 + li;
    const bool nok = n < p.Cout_store;
    const float bv = (!GROUPED && p.bias && nok) ? p.bias[n] : 0.f;
    float ps = 1.f, pt = 0.f;
    if (p.post_scale && nok) { ps = p.post_scale[n]; pt = p.post_shift[n]; }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int mrow = (wm * TM + tm) * 32 + 4 * lh;
      const unsigned o0 = nok ? (unsigned)(mrow * ostep + (p.out_ch_off + n) * 4) : OOB;   // rows past M: range check
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = 0.f;
      if (p.residual) {
        const unsigned r0 = nok ? (unsigned)(mrow * rstep + (p.res_ch_off + n) * 4) : OOB;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, (int)(r0 + ((r & 3) + 8 * (r >> 2)) * rstep), 0, 0));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v;
        if (GROUPED) v = res[tm][tn][r];
        else { v = acc[tm][tn][r] * out_scale + bv; v = fmaxf(v, v * slope_out); }
        v = v * ps + pt + rv[r];
        if (tn == 0) bad |= egne_nonfinite(v);         // (every output channel of a contaminated pixel is contaminated: one block per wave)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, (int)(o0 + ((r & 3) + 8 * (r >> 2)) * ostep), 0, 0);
      }
    }
  }
  egne_ovf_commit(bad, p.ovf_flag);
}

// out[m][n] = epilogue(sum_z ws[z][m][n]): bias, activation, post affine, residual -- the tail of a split-K launch.
// One thread per (row, 4 channels).
__global__ __launch_bounds__(256) void splitk_finish_k(const egne_conv_desc p, const float* __restrict__ ws, int Z, long long M) {
  const int nq = p.CoutP / 4;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * nq) return;
  const long long m = i / nq;
  const int n = (int)(i - m * nq) * 4;
  if (n >= p.Cout_store) return;
  f32x4 a = *(const f32x4*)(ws + m * p.CoutP + n);
  for (int z = 1; z < Z; ++z) a += *(const f32x4*)(ws + ((long long)z * M + m) * p.CoutP + n);
  const float slope = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  float* o = p.out + m * p.out_pix_stride + p.out_ch_off + n;
  const float* rs = p.residual ? p.residual + m * p.res_pix_stride + p.res_ch_off + n : nullptr;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (n + e >= p.Cout_store) break;
    float v = a[e] + (p.bias ? p.bias[n + e] : 0.f);
    v = fmaxf(v, v * slope);
    if (p.post_scale) v = v * p.post_scale[n + e] + p.post_shift[n + e];
    if (rs) v += rs[e];
    o[e] = v;
    egne_ovf_commit(egne_nonfinite(v), p.ovf_flag);
  }
}

// OIHW fp32 -> two f16 arrays [tap][CoutP][Ktot] holding hi / lo of w * wscale (zero padded)
__global__ void pack_weight_f16x2_k(const float* __restrict__ w, int Cout, int Cin, int T, int CoutP, int Ktot,
                                    float wscale, _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
  const long long total = (long long)T * CoutP * Ktot;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Ktot);
    long long q = i / Ktot;
    const int n = (int)(q % CoutP);
    const int t = (int)(q / CoutP);
    float v = (n < Cout && k < Cin) ? w[((long long)n * Cin + k) * T + t] * wscale : 0.f;
    const _Float16 h = (_Float16)v;
    hi[i] = h;
    lo[i] = (_Float16)(v - (float)h);
  }
}

}  // namespace

extern "C" int egne_pack_conv_weight_f16x2(const float* w_oihw, int Cout, int Cin, int kh, int kw, int CoutP, int Ktot,
                                           float wscale, void* whi, void* wlo, void* stream) {
  EGNE_REQUIRE(w_oihw && whi && wlo && Cout > 0 && Cin > 0 && CoutP >= Cout && CoutP % 32 == 0 && Ktot >= Cin && Ktot % 32 == 0,
               "pack_f16x2: bad sizes Cout %d Cin %d CoutP %d Ktot %d", Cout, Cin, CoutP, Ktot);
  long long total = (long long)kh * kw * CoutP * Ktot, g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pack_weight_f16x2_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, kh * kw, CoutP,
                     Ktot, wscale, (_Float16*)whi, (_Float16*)wlo);
  return egne::check_launch("egne_pack_conv_weight_f16x2");
}

// Same descriptor as egne_conv2d_fwd (d->w unused; CoutP = row count of the f16 pack: a multiple of 128 selects
// the 128x128 tile, otherwise the 256x32 tile; ngroups = 3 runs the fused MSBlock branch, weights packed
// group after group with ONE common w_scale).  One input slice without fused affine, Cp % 32 == 0, stride 1,
// zero padding.  a_scale / w_scale: exact power-of-two pre-scales (w_scale must match the pack).
namespace {

// Small problems (few output tiles, a long K loop: the 30x40 and 15x20 levels at one or two frames) leave most of the chip idle
// and each workgroup latency bound on its own K loop.  They run on smaller tiles (64 x 64, 128 x 32) and with the K range split
// over gridDim.z workgroups whose partial sums go through a workspace.  Returns Z (1 = no split) and the tile choice.
struct SmallPlan { int small, Z, wide; long long ws_floats; };   // wide: 128 x 128 tiles instead of 64 x 64

SmallPlan small_plan(const egne_conv_desc& d) {
  static const int target = [] { const char* e = getenv("EGNE_SMALL_WGS"); return e ? atoi(e) : 640; }();
  static const int wide_ok = [] { const char* e = getenv("EGNE_SMALL_WIDE"); return e ? atoi(e) : 1; }();
  static const int minsteps = [] { const char* e = getenv("EGNE_SMALL_MINSTEPS"); return e ? atoi(e) : 6; }();
  SmallPlan sp{0, 1, 0, 0};
  if (d.ngroups != 1) return sp;
  const long long M = (long long)d.B * d.Ho * d.Wo;
  const int nsteps = d.kh * d.kw * ((d.seg[0].Cp + KC - 1) / KC);
  long long tiles;      // of the standard tile choice
  if (d.CoutP % 128 == 0) tiles = ((M + 127) / 128) * (d.CoutP / 128);
  else if (d.CoutP % 64 == 0) tiles = ((M + 255) / 256) * (d.CoutP / 64);
  else tiles = ((M + 255) / 256) * (d.CoutP / 32);
  if (tiles >= 192) return sp;
  sp.small = 1;
  // wide layers with a deep K loop keep the 128 x 128 tile (half the operand bytes per flop: at 64 x 64 the launch is bound by the L2)
  sp.wide = wide_ok && d.CoutP % 128 == 0 && nsteps >= 8 * minsteps && tiles * (nsteps / minsteps) >= target;
  const long long t2 = sp.wide ? tiles : (d.CoutP % 64 == 0 ? ((M + 63) / 64) * (d.CoutP / 64) : ((M + 127) / 128) * (d.CoutP / 32));
  int Z = (int)((target + t2 - 1) / t2);
  if (Z > nsteps / minsteps) Z = nsteps / minsteps;
  if (Z > 16) Z = 16;
  if (Z < 1) Z = 1;
  sp.Z = Z;
  sp.ws_floats = Z > 1 ? (long long)Z * M * d.CoutP : 0;
  return sp;
}

int f16x3_impl(const egne_conv_desc* dp, const void* whi, const void* wlo, float a_scale, float w_scale, float* ws, long long ws_floats,
               bool allow_small, void* stream) {
  EGNE_REQUIRE(dp && whi && wlo, "conv_f16x3: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.nseg == 1 && d.ngroups >= 1 && d.ngroups <= EGNE_MAXGROUP && d.stride == 1 && d.pad_mode == 0 &&
               (d.seg[0].scale == nullptr) == (d.seg[0].shift == nullptr) && (d.ngroups == 1 || !d.seg[0].scale),
               "conv_f16x3: unsupported descriptor");
  EGNE_REQUIRE(d.seg[0].Cp % 8 == 0 && (d.seg[0].Cp + 31) / 32 * 32 == d.Ktot && d.CoutP % 32 == 0,
               "conv_f16x3: Cp %d Ktot %d (must be Cp rounded up to 32) CoutP %d", d.seg[0].Cp, d.Ktot, d.CoutP);
  EGNE_REQUIRE(d.seg[0].ptr && ((uintptr_t)d.seg[0].ptr & 15) == 0 && d.seg[0].ch_off % 4 == 0 && d.seg[0].pix_stride % 4 == 0,
               "conv_f16x3: input alignment");
  EGNE_REQUIRE(((uintptr_t)whi & 15) == 0 && ((uintptr_t)wlo & 15) == 0 && d.out && d.Cout_store <= d.CoutP &&
               d.out_ch_off + d.Cout_store <= d.out_pix_stride, "conv_f16x3: weights / output");
  EGNE_REQUIRE(a_scale > 0.f && w_scale > 0.f, "conv_f16x3: scales");
  for (int g = 0; g < d.ngroups; ++g) {
    const int dd = d.dil[g];
    EGNE_REQUIRE(dd >= 1, "conv_f16x3: dilation");
    const int ho = d.H + 2 * d.pad_h * dd - dd * (d.kh - 1), wo = d.W + 2 * d.pad_w * dd - dd * (d.kw - 1);
    EGNE_REQUIRE(ho == d.Ho && wo == d.Wo, "conv_f16x3: output %dx%d inconsistent with geometry", d.Ho, d.Wo);
  }
  const long long M = (long long)d.B * d.Ho * d.Wo;
  const float os = 1.0f / (a_scale * w_scale);
  hipStream_t st = (hipStream_t)stream;
  const _Float16* h = (const _Float16*)whi;
  const _Float16* l = (const _Float16*)wlo;
  static const int big = [] { const char* e = getenv("EGNE_SPLIT_BIG"); return e ? atoi(e) : 0; }();
  const SmallPlan sp = allow_small ? small_plan(d) : SmallPlan{0, 1, 0, 0};
  if (sp.small) {
    EGNE_REQUIRE(sp.Z == 1 || (ws && ws_floats >= sp.ws_floats && ((uintptr_t)ws & 15) == 0), "conv_f16x3: split-K workspace too small (%lld floats, need %lld)",
                 ws_floats, sp.ws_floats);
    if (sp.wide) {
      dim3 grid((unsigned)((M + 127) / 128), (unsigned)(d.CoutP / 128), (unsigned)sp.Z);
      hipLaunchKernelGGL((conv_f16x3_kernel<2, 2, 2, 2, false>), grid, dim3(256), 0, st, d, h, l, a_scale, os, ws);
    } else if (d.CoutP % 64 == 0) {
      dim3 grid((unsigned)((M + 63) / 64), (unsigned)(d.CoutP / 64), (unsigned)sp.Z);
      hipLaunchKernelGGL((conv_f16x3_kernel<2, 2, 1, 1, false>), grid, dim3(256), 0, st, d, h, l, a_scale, os, ws);
    } else {
      dim3 grid((unsigned)((M + 127) / 128), (unsigned)(d.CoutP / 32), (unsigned)sp.Z);
      hipLaunchKernelGGL((conv_f16x3_kernel<4, 1, 1, 1, false>), grid, dim3(256), 0, st, d, h, l, a_scale, os, ws);
    }
    if (sp.Z > 1) {
      const long long items = M * (d.CoutP / 4);
      hipLaunchKernelGGL(splitk_finish_k, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, d, ws, sp.Z, M);
    }
  } else if (d.CoutP % 128 == 0 && d.ngroups == 1 && big && M >= 256 * 512) {
    dim3 grid((unsigned)((M + 255) / 256), (unsigned)(d.CoutP / 128));
    hipLaunchKernelGGL((conv_f16x3_kernel<2, 2, 4, 2, false>), grid, dim3(256), 0, st, d, h, l, a_scale, os, (float*)nullptr);
  } else if (d.CoutP % 128 == 0 && d.ngroups == 1) {
    dim3 grid((unsigned)((M + 127) / 128), (unsigned)(d.CoutP / 128));
    hipLaunchKernelGGL((conv_f16x3_kernel<2, 2, 2, 2, false>), grid, dim3(256), 0, st, d, h, l, a_scale, os, (float*)nullptr);
  } else if (d.CoutP % 64 == 0 && d.ngroups == 1) {
    dim3 grid((unsigned)((M + 255) / 256), (unsigned)(d.CoutP / 64));
    hipLaunchKernelGGL((conv_f16x3_kernel<4, 1, 2, 2, false>), grid, dim3(256), 0, st, d, h, l, a_scale, os, (float*)nullptr);
  } else {
    dim3 grid((unsigned)((M + 255) / 256), (unsigned)(d.CoutP / 32));
    if (d.ngroups > 1) hipLaunchKernelGGL((conv_f16x3_kernel<4, 1, 2, 1, true>), grid, dim3(256), 0, st, d, h, l, a_scale, os, (float*)nullptr);
    else hipLaunchKernelGGL((conv_f16x3_kernel<4, 1, 2, 1, false>), grid, dim3(256), 0, st, d, h, l, a_scale, os, (float*)nullptr);
  }
  return egne::check_launch("egne_conv2d_f16x3_fwd");
}

}  // namespace

extern "C" int egne_conv2d_f16x3_fwd(const egne_conv_desc* dp, const void* whi, const void* wlo, float a_scale,
                                     float w_scale, void* stream) {
  return f16x3_impl(dp, whi, wlo, a_scale, w_scale, nullptr, 0, false, stream);
}

// Workspace (floats) the small-problem form of the same convolution needs for this descriptor: > 0 split-K, 0 small tiles without a
// split, -1 the problem is not small (egne_conv2d_f16x3_small_fwd then runs the standard launch).
extern "C" int64_t egne_conv2d_f16x3_small_workspace_floats(const egne_conv_desc* dp) {
  if (!dp) return -1;
  const SmallPlan sp = small_plan(*dp);
  return sp.small ? sp.ws_floats : -1;
}

extern "C" int egne_conv2d_f16x3_small_fwd(const egne_conv_desc* dp, const void* whi, const void* wlo, float a_scale, float w_scale,
                                           float* ws, int64_t ws_floats, void* stream) {
  return f16x3_impl(dp, whi, wlo, a_scale, w_scale, ws, ws_floats, true, stream);
}

