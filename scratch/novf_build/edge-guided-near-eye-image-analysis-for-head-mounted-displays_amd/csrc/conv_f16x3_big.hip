// Split-f16 implicit-GEMM convolution, deep variant for the wide VGG-trunk layers (Cout % 256 == 0 or 128, Cin % 32 == 0):
// 256 x BN tile, 8 waves (wave tile 128 x 64), K step = 32 channels of one tap, TWO LDS stages, ONE barrier per K step.
//
// The 128x128 kernel (conv_f16x3.hip) pays two barriers per 24 MFMAs and tops out at ~285 TFLOP/s algorithmic
// (~860 TFLOP/s of f16 MFMA, the ceiling of that structure).  Here
//   * the weights are packed as ready-made LDS images (hi | lo granules, 128 B per output channel and K step, XOR
//     swizzle applied at pack time) and staged by LDS-DMA (`buffer_load_dwordx4 ... lds`): no registers, no VALU;
//   * the activations stay fp32 in HBM: the loads of step k+1 are issued before the 48 MFMAs of step k, converted to
//     hi / lo afterwards and written to the other LDS stage (buffer addressing with per-row tap masks as in the
//     128x128 kernel);
//   * the product is computed transposed (weights as the A operand), so a lane ends up with 4 consecutive output
//     channels of one pixel and stores 16 bytes.
// LDS image of a stage: row r (pixel or output channel) = 128 bytes = 4 granules of 8 channels, granule = [hi x8 | lo x8];
// the 16-byte chunk c of row r sits at chunk c ^ (((r >> 1) + 6) & 7): conflict free for the ds_read_b128 patterns of BOTH MFMA
// shapes (32 rows x 2 chunk columns for 32x32x16, 16 rows x 4 chunk columns for 16x16x32).
//
// M16 computes the same tile with v_mfma_f32_16x16x32_f16 (one instruction per 32-channel step and 16 x 16 block): the same
// FLOP per cycle, but these kernels run at the power limit and the chip holds a higher clock on this shape (MI355X DVFS).
#include "common.h"
#include "split_f16.h"
#include <cstdlib>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

namespace {

constexpr unsigned OOB = 0x80000000u;
constexpr int BM = 256, ROWB = 128;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// NB = 32-channel blocks per wave along N (2 -> BN = 256 with 4 waves along N, 1 -> BN = 128)
// NP = products per multiply: 3 (hi hi + hi lo + lo hi) or 1 (hi hi: plain f16 operands; egne_conv_desc.f16_products) -- the lo halves
// are then neither derived, stored nor read (the weight image keeps its layout: the lo chunks of a row are simply not touched)
template <int NB, bool M16, int NP = 3>
__global__ __launch_bounds__(512) void conv_f16x3_big_kernel(const egne_conv_desc p, const char* __restrict__ wimg, float a_scale,
                                                             float out_scale) {
  constexpr int BN = 128 * NB;
  constexpr int STAGE = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;               // 2 x 4 waves; wave tile 128 x (32 * NB)
  const long long M = (long long)p.B * p.Ho * p.Wo;
  const long long m0 = (long long)blockIdx.x * BM;
  const int ntile = blockIdx.y;
  const int T = p.kh * p.kw;
  const egne_seg sg = p.seg[0];
  const int hw = p.Ho * p.Wo, frame_px = p.H * p.W;
  const int b0 = (int)(m0 / hw);
  const long long in_left = ((long long)p.B - b0) * frame_px * sg.pix_stride * 4;
  const __amdgpu_buffer_rsrc_t rin = make_rsrc(sg.ptr + (long long)b0 * frame_px * sg.pix_stride,
                                               (unsigned)(in_left < 0x7fffffffll ? in_left : 0x7fffffffll));
  const int nchunk = sg.Cp >> 5;
  const int nsteps = T * nchunk;
  // weight images: [ntile][step = chunk*T + tap][BN rows x 128 B]
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(wimg + (long long)ntile * nsteps * (BN * ROWB), (unsigned)(nsteps * BN * ROWB));

  // ---- A staging: thread -> 4 items (row = (tid>>3) + 64*i, float4 column c4 = tid&7) ----
  const int c4 = tid & 7;
  int roff[4];
  unsigned tapmask[4];
  {
    const int dil = p.dil[0];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long m = m0 + (tid >> 3) + 64 * i;
      const int b = (int)(m / hw);
      const int r = (int)(m - (long long)b * hw);
      const int oy = r / p.Wo, ox = r - oy * p.Wo;
      roff[i] = ((((b - b0) * p.H + oy) * p.W + ox) * (int)sg.pix_stride + sg.ch_off + c4 * 4) * 4;
      unsigned mk = 0;
      for (int ky = 0; ky < p.kh; ++ky)
        for (int kx = 0; kx < p.kw; ++kx) {
          const int iy = oy + (ky - p.pad_h) * dil, ix = ox + (kx - p.pad_w) * dil;
          if (m < M && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mk |= 1u << (ky * p.kw + kx);
        }
      tapmask[i] = mk;
    }
  }
  // LDS destination of item i: row r, granule c4>>1, half (c4&1): hi at chunk 2g, lo at chunk 2g+1, 8 bytes at (c4&1)*8
  int ldst_hi[4], ldst_lo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (tid >> 3) + 64 * i, g = c4 >> 1, sw = ((r >> 1) + 6) & 7;
    ldst_hi[i] = r * ROWB + (((2 * g) ^ sw) << 4) + (c4 & 1) * 8;
    ldst_lo[i] = r * ROWB + (((2 * g + 1) ^ sw) << 4) + (c4 & 1) * 8;
  }
  // weight DMA: BN rows x 128 B = BN/8 wave instructions of 1 KB, NB*2 per wave; the image is already swizzled: linear copy
  const int wvoff = lane * 16;

  u32x4 ra[4];
  int ky_n = 0, kx_n = 0, c0_n = 0;     // coordinates of the step being loaded
  auto load_a = [&](int tap) {
    const int dil = p.dil[0];
    const int tapoff = (((ky_n - p.pad_h) * p.W + (kx_n - p.pad_w)) * dil * (int)sg.pix_stride + c0_n) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = (tapmask[i] >> tap) & 1u;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? roff[i] + tapoff : (int)OOB, 0, 0);
    }
  };
  auto dma_b = [&](int stage, int step) {
    char* base = lds + stage * STAGE + BM * ROWB;
#pragma unroll
    for (int j = 0; j < 2 * NB; ++j) {
      const int blk = wave * 2 * NB + j;      // 1-KB block of the image
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(base + blk * 1024), 16, wvoff, step * (BN * ROWB) + blk * 1024, 0, 0);
    }
  };
  auto store_a = [&](int stage) {
    char* base = lds + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = __builtin_bit_cast(f32x4, ra[i]);
      if constexpr (NP == 1) {
        const f32x2 t0 = {v[0] * a_scale, v[1] * a_scale}, t1 = {v[2] * a_scale, v[3] * a_scale};
        const h2 h0 = __builtin_convertvector(t0, h2), h1 = __builtin_convertvector(t1, h2);
        const h4 hi = {h0[0], h0[1], h1[0], h1[1]};
        *(h4*)(base + ldst_hi[i]) = hi;
        continue;
      }
      h2 h0, h1, l0, l1;                        // x * a_scale = hi + lo, plain (unpacked) VALU: split_f16.h
      egne::split2(v[0], v[1], a_scale, h0, l0);
      egne::split2(v[2], v[3], a_scale, h1, l1);
      const h4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
      *(h4*)(base + ldst_hi[i]) = hi;
      *(h4*)(base + ldst_lo[i]) = lo;
    }
  };
  auto advance = [&]() {
    if (++kx_n == p.kw) { kx_n = 0; ++ky_n; }
    if (ky_n == p.kh) { ky_n = 0; c0_n += 32; }
  };

  // 32x32x16: acc[4][NB] blocks of 32 x 32; 16x16x32: acc[8][2 * NB] blocks of 16 x 16 (same 64 * NB registers)
  constexpr int MB = M16 ? 16 : 32, NTM = 128 / MB, NTN = 32 * NB / MB;
  using acc_t = std::conditional_t<M16, f32x4, f32x16>;
  acc_t acc[NTM][NTN];
#pragma unroll
  for (int a = 0; a < NTM; ++a)
#pragma unroll
    for (int b = 0; b < NTN; ++b) acc[a][b] = (acc_t)(0.f);

  const int lr = lane & (MB - 1), kg = lane / MB;       // row inside a block, 8-channel group (2 or 4 of them)
  int aoffs[NTM], boffs[NTN];
#pragma unroll
  for (int t = 0; t < NTM; ++t) aoffs[t] = (wm * 128 + t * MB + lr) * ROWB;
#pragma unroll
  for (int t = 0; t < NTN; ++t) boffs[t] = BM * ROWB + (wn * 32 * NB + t * MB + lr) * ROWB;
  const int sw = ((lr >> 1) + 6) & 7;

  // prologue: step 0 into stage 0
  load_a(0);
  dma_b(0, 0);
  store_a(0);
  advance();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int tap_n = 1 % T;
  for (int step = 0; step < nsteps; ++step) {
    const int st = step & 1;
    const bool more = step + 1 < nsteps;
    if (more) {                     // loads of step+1 in flight during the MFMAs of this step
      load_a(tap_n);
      dma_b(st ^ 1, step + 1);
    }
    const char* base = lds + st * STAGE;
    if constexpr (M16) {
      const int ch = ((2 * kg) ^ sw) << 4, cl = ((2 * kg + 1) ^ sw) << 4;
      h8 bh[NTN], bl[NTN];
#pragma unroll
      for (int t = 0; t < NTN; ++t) {
        bh[t] = *(const h8*)(base + boffs[t] + ch);
        if constexpr (NP == 3) bl[t] = *(const h8*)(base + boffs[t] + cl);
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        h8 ah[4], al[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          ah[t] = *(const h8*)(base + aoffs[half * 4 + t] + ch);
          if constexpr (NP == 3) al[t] = *(const h8*)(base + aoffs[half * 4 + t] + cl);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int tn = 0; tn < NTN; ++tn) {
            acc_t& c = acc[half * 4 + t][tn];
            if constexpr (NP == 3) {
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[tn], al[t], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[tn], ah[t], c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[tn], ah[t], c, 0, 0, 0);
          }
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ((2 * (2 * ks + kg)) ^ sw) << 4, cl = ((2 * (2 * ks + kg) + 1) ^ sw) << 4;
      h8 ah[4], al[4], bh[NB], bl[NB];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ah[t] = *(const h8*)(base + aoffs[t] + ch);
        al[t] = *(const h8*)(base + aoffs[t] + cl);
      }
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        bh[t] = *(const h8*)(base + boffs[t] + ch);
        bl[t] = *(const h8*)(base + boffs[t] + cl);
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < NB; ++tn) {
          if constexpr (!M16) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[tn], al[tm], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[tn], ah[tm], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[tn], ah[tm], acc[tm][tn], 0, 0, 0);
          }
        }
    }
    }
    if (more) {
      store_a(st ^ 1);              // the other stage was last read in step-1: every wave passed the barrier since
      advance();
      tap_n = tap_n + 1 == T ? 0 : tap_n + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: transposed product, lane = pixel lr of the block; 32x32x16: channels n = 32*blk + 8*j + 4*kg + e (register
  //      4*j + e); 16x16x32: channels n = 16*blk + 4*kg + e (register e) ----
  const float slope = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  const long long left = M - m0;
  const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out + m0 * p.out_pix_stride, (unsigned)((left < BM ? left : BM) * p.out_pix_stride * 4));
  constexpr int NJ = M16 ? 1 : 4;
  bool bad = false;
#pragma unroll
  for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = ntile * BN + wn * 32 * NB + tn * MB + (M16 ? 4 * kg : 8 * j + 4 * kg);
      const bool nok = n < p.Cout_store;
      const f32x4 bv = (p.bias && nok) ? *(const f32x4*)(p.bias + n) : (f32x4)(0.f);
#pragma unroll
      for (int tm = 0; tm < NTM; ++tm) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = acc[tm][tn][4 * j + e] * out_scale + bv[e];
          v[e] = fmaxf(t, t * slope);
        }
        if (tn == 0 && j == 0) bad |= egne_nonfinite(v[0]);       // lane = pixel: one channel per pixel (common.h)
        const int row = wm * 128 + tm * MB + lr;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout,
                                               nok ? (row * (int)p.out_pix_stride + p.out_ch_off + n) * 4 : (int)OOB, 0, 0);
      }
    }
  egne_ovf_commit(bad, p.ovf_flag);
}

// OIHW fp32 -> LDS images [ntile][step = chunk*T + tap][BN rows][128 B]: row j = output channel ntile*BN + j, granule g =
// channels chunk*32 + 8g .. +7 as [hi x8 | lo x8], 16-byte chunk c stored at chunk c ^ (((j >> 1) + 6) & 7)
__global__ void pack_weight_f16img_k(const float* __restrict__ w, int Cout, int Cin, int T, int BN, int ntiles, int nchunk,
                                     float wscale, _Float16* __restrict__ out) {
  const long long total = (long long)ntiles * nchunk * T * BN * 64;     // halfs
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), pc = (int)((i >> 3) & 7);
    long long q = i >> 6;
    const int j = (int)(q % BN); q /= BN;
    const int step = (int)(q % (nchunk * T));
    const int nt = (int)(q / (nchunk * T));
    const int chunk = step / T, tap = step - chunk * T;
    const int lc = pc ^ (((j >> 1) + 6) & 7), g = lc >> 1, hl = lc & 1;
    const int n = nt * BN + j, ci = chunk * 32 + 8 * g + e;
    const float v = (n < Cout && ci < Cin) ? w[((long long)n * Cin + ci) * T + tap] * wscale : 0.f;
    const _Float16 h = (_Float16)v;
    out[i] = hl ? (_Float16)(v - (float)h) : h;
  }
}

}  // namespace

extern "C" int egne_pack_conv_weight_f16img(const float* w_oihw, int Cout, int Cin, int kh, int kw, int BN, int Ktot, float wscale,
                                            void* wimg, void* stream) {
  EGNE_REQUIRE(w_oihw && wimg && Cout > 0 && Cin > 0 && (BN == 128 || BN == 256) && Ktot >= Cin && Ktot % 32 == 0 && wscale > 0.f,
               "pack_f16img: bad sizes Cout %d Cin %d BN %d Ktot %d", Cout, Cin, BN, Ktot);
  const int ntiles = (Cout + BN - 1) / BN, nchunk = Ktot / 32;
  long long total = (long long)ntiles * nchunk * kh * kw * BN * 64, g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(pack_weight_f16img_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, kh * kw, BN, ntiles,
                     nchunk, wscale, (_Float16*)wimg);
  return egne::check_launch("egne_pack_conv_weight_f16img");
}

// Same descriptor as egne_conv2d_f16x3_fwd: one input slice without fused affine, Cp % 32 == 0, stride 1, zero padding,
// one group, no residual / post affine; d->CoutP = Cout rounded up to BN (128 or 256: CoutP % 256 == 0 selects 256).
extern "C" int egne_conv2d_f16x3_big_fwd(const egne_conv_desc* dp, const void* wimg, float a_scale, float w_scale, void* stream) {
  EGNE_REQUIRE(dp && wimg, "conv_f16x3_big: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.nseg == 1 && d.ngroups == 1 && d.stride == 1 && d.pad_mode == 0 && !d.seg[0].scale && !d.seg[0].shift && !d.residual &&
               !d.post_scale && d.kh * d.kw <= 32, "conv_f16x3_big: unsupported descriptor");
  const egne_seg& g = d.seg[0];
  EGNE_REQUIRE(g.ptr && g.Cp % 32 == 0 && g.Cp == d.Ktot && ((uintptr_t)g.ptr & 15) == 0 && g.ch_off % 4 == 0 && g.pix_stride % 4 == 0,
               "conv_f16x3_big: input slice (Cp %d Ktot %d)", g.Cp, d.Ktot);
  EGNE_REQUIRE(d.CoutP % 128 == 0 && d.Cout_store <= d.CoutP && d.Cout_store % 4 == 0 && d.out && ((uintptr_t)d.out & 15) == 0 &&
               d.out_ch_off % 4 == 0 && d.out_pix_stride % 4 == 0 && d.out_ch_off + d.Cout_store <= d.out_pix_stride &&
               d.out_pix_stride * 1024 < (1ll << 31) && (!d.bias || ((uintptr_t)d.bias & 15) == 0), "conv_f16x3_big: output");
  const int dd = d.dil[0];
  EGNE_REQUIRE(dd >= 1 && d.H + 2 * d.pad_h * dd - dd * (d.kh - 1) == d.Ho && d.W + 2 * d.pad_w * dd - dd * (d.kw - 1) == d.Wo,
               "conv_f16x3_big: output %dx%d inconsistent with geometry", d.Ho, d.Wo);
  EGNE_REQUIRE(((uintptr_t)wimg & 15) == 0 && a_scale > 0.f && w_scale > 0.f, "conv_f16x3_big: weights / scales");
  const long long M = (long long)d.B * d.Ho * d.Wo;
  const float os = 1.0f / (a_scale * w_scale);
  hipStream_t st = (hipStream_t)stream;
  static bool once = [] {
    return hipFuncSetAttribute((const void*)conv_f16x3_big_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess &&
           hipFuncSetAttribute((const void*)conv_f16x3_big_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess &&
           hipFuncSetAttribute((const void*)conv_f16x3_big_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess &&
           hipFuncSetAttribute((const void*)conv_f16x3_big_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
  }();
  static const bool m16 = !(getenv("EGNE_BIG_M16") && atoi(getenv("EGNE_BIG_M16")) == 0);     // MFMA shape (header)
  if (!once) return egne::fail(EGNE_ERR_LAUNCH, "conv_f16x3_big: cannot raise the dynamic LDS limit");
  if (d.f16_products == 1) {       // plain f16 operands (the edge network next to a bf16-storage training plan): 16x16x32 shape only
    static bool once1 = hipFuncSetAttribute((const void*)conv_f16x3_big_kernel<2, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess &&
                        hipFuncSetAttribute((const void*)conv_f16x3_big_kernel<1, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
    if (!once1) return egne::fail(EGNE_ERR_LAUNCH, "conv_f16x3_big: cannot raise the dynamic LDS limit");
    if (d.CoutP % 256 == 0)
      hipLaunchKernelGGL((conv_f16x3_big_kernel<2, true, 1>), dim3((unsigned)((M + BM - 1) / BM), (unsigned)(d.CoutP / 256)), dim3(512), 2 * (BM + 256) * ROWB, st, d, (const char*)wimg, a_scale, os);
    else
      hipLaunchKernelGGL((conv_f16x3_big_kernel<1, true, 1>), dim3((unsigned)((M + BM - 1) / BM), (unsigned)(d.CoutP / 128)), dim3(512), 2 * (BM + 128) * ROWB, st, d, (const char*)wimg, a_scale, os);
    return egne::check_launch("egne_conv2d_f16x3_big_fwd");
  }
  if (d.CoutP % 256 == 0) {
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)(d.CoutP / 256));
    if (m16) hipLaunchKernelGGL((conv_f16x3_big_kernel<2, true>), grid, dim3(512), 2 * (BM + 256) * ROWB, st, d, (const char*)wimg, a_scale, os);
    else hipLaunchKernelGGL((conv_f16x3_big_kernel<2, false>), grid, dim3(512), 2 * (BM + 256) * ROWB, st, d, (const char*)wimg, a_scale, os);
  } else {
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)(d.CoutP / 128));
    if (m16) hipLaunchKernelGGL((conv_f16x3_big_kernel<1, true>), grid, dim3(512), 2 * (BM + 128) * ROWB, st, d, (const char*)wimg, a_scale, os);
    else hipLaunchKernelGGL((conv_f16x3_big_kernel<1, false>), grid, dim3(512), 2 * (BM + 128) * ROWB, st, d, (const char*)wimg, a_scale, os);
  }
  return egne::check_launch("egne_conv2d_f16x3_big_fwd");
}
