// 1x1 convolution over a concatenation of raw NHWC slices as an LDS-staged split-f16 GEMM (fp32 tensors, three
// v_mfma_f32_32x32x16_f16 per product, fp32 accumulate; numerics: conv_f16x3.hip).
//
// For the 1x1 convs whose K (up to ~550 channels: decoder conv11 / conv21 at 30x40 and 60x80, models/RITnet_v2.py:84,86)
// or Cout (96: dense block 3) do not fit the streaming kernel's LDS weight image (conv1x1_f16.hip): M = B*H*W pixels,
// N = Cout, K = the slices one after the other, each padded to a multiple of 32 channels.  Register-staged A (buffer
// loads relative to the tile's first row, one resource per slice; converted to hi / lo while it is written to LDS),
// pre-split f16 weights [CoutP][Ktot] staged through LDS, 128x128 or 256x64 tiles, two barriers per 32-channel step --
// the structure of conv_f16x3_kernel without taps, padding masks or groups.
#include "common.h"
#include "split_f16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned OOB = 0x80000000u;
constexpr int KC = 32, LDH = 40;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

template <int WGM, int WGN, int TM, int TN>
__global__ __launch_bounds__(256) void conv1x1_ms_f16x3_kernel(const egne_conv_desc p, const _Float16* __restrict__ whi,
                                                               const _Float16* __restrict__ wlo, float a_scale, float out_scale,
                                                               long long M) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int AR = BM / 32;
  constexpr int BI = (BN * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) _Float16 lds[(2 * BM + 2 * BN) * LDH];
  _Float16* Ahi = lds;
  _Float16* Alo = Ahi + BM * LDH;
  _Float16* Bhi = Alo + BM * LDH;
  _Float16* Blo = Bhi + BN * LDH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;
  const long long m0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const long long left = M - m0;
  const long long rows = left < BM ? left : BM;
  const int col4 = tid & 7, rbase = tid >> 3;
  const unsigned wbytes = (unsigned)p.CoutP * p.Ktot * 2u;
  const __amdgpu_buffer_rsrc_t rwh = make_rsrc(whi, wbytes), rwl = make_rsrc(wlo, wbytes);
  int boff[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int item = tid + 256 * j;
    const int row = item >> 2, piece = item & 3;
    boff[j] = row < BN ? ((n0 + row) * p.Ktot + piece * 8) * 2 : (int)OOB;
  }

  u32x4 ra[AR], rbh[BI], rbl[BI];
  int seg = 0, c0 = 0, kofs = 0;          // slice / first channel / K offset of the step being loaded
  auto load_step = [&]() {
    const egne_seg sg = p.seg[seg];
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(sg.ptr + m0 * sg.pix_stride, (unsigned)(rows * sg.pix_stride * 4));   // rows past M: zeros
    const int ps4 = (int)sg.pix_stride * 4;
    const int coff = c0 + col4 * 4 < sg.Cp ? (sg.ch_off + c0 + col4 * 4) * 4 : (int)OOB;     // channel tail of the slice
#pragma unroll
    for (int i = 0; i < AR; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, (rbase + 32 * i) * ps4 + coff, 0, 0);
    const int wstep = (kofs + c0) * 2;
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      rbh[j] = __builtin_amdgcn_raw_buffer_load_b128(rwh, boff[j], wstep, 0);
      rbl[j] = __builtin_amdgcn_raw_buffer_load_b128(rwl, boff[j], wstep, 0);
    }
  };
  auto advance = [&]() {
    c0 += KC;
    if (c0 >= p.seg[seg].Cp) { kofs += (p.seg[seg].Cp + 31) / 32 * 32; c0 = 0; ++seg; }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const f32x4 v = __builtin_bit_cast(f32x4, ra[i]);
      h2 h0, h1, l0, l1;                        // x * a_scale = hi + lo, plain (unpacked) VALU: split_f16.h
      egne::split2(v[0], v[1], a_scale, h0, l0);
      egne::split2(v[2], v[3], a_scale, h1, l1);
      const h4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
      const int o = (rbase + 32 * i) * LDH + col4 * 4;
      *(h4*)&Ahi[o] = hi;
      *(h4*)&Alo[o] = lo;
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      const int item = tid + 256 * j;
      if (BN * 4 % 256 == 0 || (item >> 2) < BN) {
        const int o = (item >> 2) * LDH + (item & 3) * 8;
        *(u32x4*)&Bhi[o] = rbh[j];
        *(u32x4*)&Blo[o] = rbl[j];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = (f32x16)(0.f);

  int nsteps = 0;
  for (int s = 0; s < p.nseg; ++s) nsteps += (p.seg[s].Cp + KC - 1) / KC;
  load_step();
  advance();
  store_step();
  __syncthreads();
  const int arow = (wm * TM * 32 + li) * LDH + lh * 8;
  const int brow = (wn * TN * 32 + li) * LDH + lh * 8;
  for (int step = 0; step < nsteps; ++step) {
    const bool more = step + 1 < nsteps;
    if (more) { load_step(); advance(); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        ah[t] = *(const h8*)&Ahi[arow + t * 32 * LDH + ks * 16];
        al[t] = *(const h8*)&Alo[arow + t * 32 * LDH + ks * 16];
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        bh[t] = *(const h8*)&Bhi[brow + t * 32 * LDH + ks * 16];
        bl[t] = *(const h8*)&Blo[brow + t * 32 * LDH + ks * 16];
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
        }
    }
    __syncthreads();
    if (more) store_step();
    __syncthreads();
  }

  // epilogue: lane holds channel n of 16 rows m = mrow + c_r, c_r = (r&3) + 8*(r>>2) + 4*lh
  const float slope = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out + m0 * p.out_pix_stride, (unsigned)(rows * p.out_pix_stride * 4));
  const int ostep = (int)p.out_pix_stride * 4;
  bool bad = false;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + (wn * TN + tn) * 32 + li;
    const bool nok = n < p.Cout_store;
    const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int mrow = (wm * TM + tm) * 32 + 4 * lh;
      const unsigned o0 = nok ? (unsigned)(mrow * ostep + (p.out_ch_off + n) * 4) : OOB;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[tm][tn][r] * out_scale + bv;
        if (tn == 0) bad |= egne_nonfinite(v);         // (every output channel of a contaminated pixel is contaminated: one block per wave)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(v, v * slope)), rout,
                                              (int)(o0 + ((r & 3) + 8 * (r >> 2)) * ostep), 0, 0);
      }
    }
  }
  egne_ovf_commit(bad, p.ovf_flag);
}

// OIHW (kh = kw = 1) fp32 -> two f16 arrays [CoutP][Ktot]: hi / lo of w[n][kmap[k]] * wscale (kmap[k] = -1: padding column)
__global__ void pack_w1x1_map_k(const float* __restrict__ w, int Cout, int Cin, const int* __restrict__ kmap, int CoutP, int Ktot,
                                float wscale, _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
  const long long total = (long long)CoutP * Ktot;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Ktot), n = (int)(i / Ktot);
    const int ci = kmap[k];
    const float v = (n < Cout && ci >= 0) ? w[(long long)n * Cin + ci] * wscale : 0.f;
    const _Float16 h = (_Float16)v;
    hi[i] = h;
    lo[i] = (_Float16)(v - (float)h);
  }
}

}  // namespace

extern "C" int egne_pack_conv1x1_weight_f16x2_map(const float* w_oihw, int Cout, int Cin, const int32_t* kmap, int CoutP, int Ktot,
                                                  float wscale, void* whi, void* wlo, void* stream) {
  EGNE_REQUIRE(w_oihw && kmap && whi && wlo && Cout > 0 && Cin > 0 && CoutP >= Cout && CoutP % 32 == 0 && Ktot % 32 == 0 && wscale > 0.f,
               "pack_conv1x1_f16x2_map: bad sizes Cout %d Cin %d CoutP %d Ktot %d", Cout, Cin, CoutP, Ktot);
  long long total = (long long)CoutP * Ktot, g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pack_w1x1_map_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, kmap, CoutP, Ktot, wscale,
                     (_Float16*)whi, (_Float16*)wlo);
  return egne::check_launch("egne_pack_conv1x1_weight_f16x2_map");
}

// 1x1 / stride 1 / no padding over up to EGNE_MAXSEG raw slices (no fused affine, residual or post affine).  d->Ktot = sum of
// the slice widths rounded up to 32 each, d->CoutP = rows of the pack: a multiple of 128 selects the 128x128 tile, otherwise
// (a multiple of 64) the 256x64 tile.
extern "C" int egne_conv1x1_ms_f16x3_fwd(const egne_conv_desc* dp, const void* whi, const void* wlo, float a_scale, float w_scale,
                                         void* stream) {
  EGNE_REQUIRE(dp && whi && wlo, "conv1x1_ms_f16: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0 && d.ngroups == 1 && d.Ho == d.H && d.Wo == d.W &&
               d.nseg >= 1 && d.nseg <= EGNE_MAXSEG && !d.residual && !d.post_scale, "conv1x1_ms_f16: unsupported descriptor");
  int ktot = 0;
  for (int s = 0; s < d.nseg; ++s) {
    const egne_seg& g = d.seg[s];
    EGNE_REQUIRE(g.ptr && !g.scale && !g.shift && g.act_in == EGNE_ACT_NONE && g.Cp % 8 == 0 && g.ch_off % 4 == 0 && g.pix_stride % 4 == 0 &&
                 ((uintptr_t)g.ptr & 15) == 0 && g.ch_off + g.Cp <= g.pix_stride && g.pix_stride * 1024 < (1ll << 31), "conv1x1_ms_f16: slice %d", s);
    ktot += (g.Cp + 31) / 32 * 32;
  }
  EGNE_REQUIRE(ktot == d.Ktot && d.CoutP % 64 == 0 && d.Cout_store <= d.CoutP && d.out && d.out_ch_off + d.Cout_store <= d.out_pix_stride &&
               d.out_pix_stride * 1024 < (1ll << 31), "conv1x1_ms_f16: Ktot %d (expected %d) / output", d.Ktot, ktot);
  EGNE_REQUIRE(((uintptr_t)whi & 15) == 0 && ((uintptr_t)wlo & 15) == 0 && a_scale > 0.f && w_scale > 0.f, "conv1x1_ms_f16: weights / scales");
  const long long M = (long long)d.B * d.H * d.W;
  const float os = 1.0f / (a_scale * w_scale);
  hipStream_t st = (hipStream_t)stream;
  const _Float16* h = (const _Float16*)whi;
  const _Float16* l = (const _Float16*)wlo;
  if (d.CoutP % 128 == 0) {
    dim3 grid((unsigned)((M + 127) / 128), (unsigned)(d.CoutP / 128));
    hipLaunchKernelGGL((conv1x1_ms_f16x3_kernel<2, 2, 2, 2>), grid, dim3(256), 0, st, d, h, l, a_scale, os, M);
  } else {
    dim3 grid((unsigned)((M + 255) / 256), (unsigned)(d.CoutP / 64));
    hipLaunchKernelGGL((conv1x1_ms_f16x3_kernel<4, 1, 2, 2>), grid, dim3(256), 0, st, d, h, l, a_scale, os, M);
  }
  return egne::check_launch("egne_conv1x1_ms_f16x3_fwd");
}
