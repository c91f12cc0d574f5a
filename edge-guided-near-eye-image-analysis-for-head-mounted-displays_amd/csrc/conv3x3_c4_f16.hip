// First layers (Cin <= 4: vgg16_c.py:66 conv1_1 on 3 channels, utils.py:1047 convBlock conv1 on 1-2 channels) on the
// split-f16 path, streaming: the 9 taps are folded into K (K = 9 taps x 4 channels, padded to 48 = three K=16 steps) and
// NOTHING is staged -- a lane's MFMA operand for one step is "8 consecutive K" = the 4-channel vectors of TWO taps of its
// pixel, i.e. two 16-byte buffer loads (taps outside the image carry the out-of-range offset and load zeros).  Weight
// fragments (48 x Cout, hi and lo) live in registers, every wave walks 32-pixel blocks on its own, the product is computed
// transposed so that a lane stores 16 bytes.  The layer is a pure store stream (128-256 B per pixel).
#include "common.h"
#include "split_f16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void split8(const u32x4 a, const u32x4 b, float s, h8& hi, h8& lo) {
  const f32x4 va = __builtin_bit_cast(f32x4, a), vb = __builtin_bit_cast(f32x4, b);
  const float x[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h2 h, l;
    egne::split2(x[2 * q], x[2 * q + 1], s, h, l);     // plain (unpacked) VALU: split_f16.h
    hi[2 * q] = h[0]; hi[2 * q + 1] = h[1];
    lo[2 * q] = l[0]; lo[2 * q + 1] = l[1];
  }
}

// fhi / flo: [3 steps][TN][64 lanes][8 halfs]: lane (n%32, h) of block tn holds K slots 16*s + 8*h + j = tap 4s + 2h + (j>>2), channel j&3
template <int TN>
__global__ __launch_bounds__(256) void conv3x3_c4_f16_kernel(const egne_conv_desc p, const _Float16* __restrict__ fhi,
                                                             const _Float16* __restrict__ flo, float a_scale, float out_scale,
                                                             long long M, int nblocks) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kq = lane >> 5;
  const egne_seg sg = p.seg[0];
  h8 wh[3][TN], wl[3][TN];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      wh[s][tn] = *(const h8*)(fhi + ((s * TN + tn) * 64 + lane) * 8);
      wl[s][tn] = *(const h8*)(flo + ((s * TN + tn) * 64 + lane) * 8);
    }
  const float slope = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  f32x4 bias[TN][4], ps[TN][4], pt[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = tn * 32 + 8 * j + 4 * kq;
      const bool nok = n < p.Cout_store;
      bias[tn][j] = (p.bias && nok) ? *(const f32x4*)(p.bias + n) : (f32x4)(0.f);
      ps[tn][j] = (p.post_scale && nok) ? *(const f32x4*)(p.post_scale + n) : (f32x4)(1.f);
      pt[tn][j] = (p.post_scale && nok) ? *(const f32x4*)(p.post_shift + n) : (f32x4)(0.f);
    }
  const int hw = p.H * p.W;
  const int ps4 = (int)sg.pix_stride * 4;
  const unsigned frame_in = (unsigned)hw * (unsigned)ps4;

  for (int blk = blockIdx.x * 4 + wave; blk < nblocks; blk += gridDim.x * 4) {
    const long long m = (long long)blk * 32 + li;
    const int b = (int)(m / hw);
    const int r = (int)(m - (long long)b * hw);
    const int y = r / p.W, x = r - y * p.W;
    // the block may straddle two frames: per-lane frame in the offset, resource over the rest of the tensor from the block's first frame
    const int b0 = (int)(((long long)blk * 32) / hw);
    // planar: the input is [B][C][H][W] (NCHW, C = seg.Cp <= 4 planes) read in place: pix_stride 1, one dword per tap and plane
    const bool planar = sg.pix_stride == 1;
    const int fstride = planar ? hw * sg.Cp : hw * (int)sg.pix_stride;         // floats per frame
    const long long left = ((long long)p.B - b0) * fstride * 4;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(sg.ptr + (long long)b0 * fstride, (unsigned)(left < 0x7fffffffll ? left : 0x7fffffffll));
    const int base = planar ? ((b - b0) * fstride + r) * 4 : (((b - b0) * hw + r) * (int)sg.pix_stride + sg.ch_off) * 4;
    (void)frame_in;
    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tn] = (f32x16)(0.f);
    u32x4 xa[3], xb[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int t0 = 4 * s + 2 * kq, t1 = t0 + 1;            // the lane's two taps of this step (taps >= 9: padding)
      const int dy0 = t0 / 3 - 1, dx0 = t0 % 3 - 1, dy1 = t1 / 3 - 1, dx1 = t1 % 3 - 1;
      const bool ok0 = m < M && t0 < 9 && (unsigned)(y + dy0) < (unsigned)p.H && (unsigned)(x + dx0) < (unsigned)p.W;
      const bool ok1 = m < M && t1 < 9 && (unsigned)(y + dy1) < (unsigned)p.H && (unsigned)(x + dx1) < (unsigned)p.W;
      if (planar) {
        const int o0 = ok0 ? base + (dy0 * p.W + dx0) * 4 : (int)OOB, o1 = ok1 ? base + (dy1 * p.W + dx1) * 4 : (int)OOB;
        u32x4 a = {0u, 0u, 0u, 0u}, c = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
          if (ch < sg.Cp) {
            a[ch] = __builtin_amdgcn_raw_buffer_load_b32(rin, o0, ch * hw * 4, 0);
            c[ch] = __builtin_amdgcn_raw_buffer_load_b32(rin, o1, ch * hw * 4, 0);
          }
        xa[s] = a; xb[s] = c;
      } else {
        xa[s] = __builtin_amdgcn_raw_buffer_load_b128(rin, ok0 ? base + (dy0 * p.W + dx0) * ps4 : (int)OOB, 0, 0);
        xb[s] = __builtin_amdgcn_raw_buffer_load_b128(rin, ok1 ? base + (dy1 * p.W + dx1) * ps4 : (int)OOB, 0, 0);
      }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      h8 ah, al;
      split8(xa[s], xb[s], a_scale, ah, al);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s][tn], al, acc[tn], 0, 0, 0);
        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s][tn], ah, acc[tn], 0, 0, 0);
        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s][tn], ah, acc[tn], 0, 0, 0);
      }
    }
    const long long m0 = (long long)blk * 32;
    const long long rows = M - m0 < 32 ? M - m0 : 32;
    // out_split = 2: the output is stored as f16 halves of v * out_split_scale (channel order kept): the input format of
    // conv3x3_rw_f16.hip's F16IN form -- the edge network's conv1_1 next to a bf16-storage training plan
    const bool o16 = p.out_split == 2;
    const int oesz = o16 ? 2 : 4;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc((char*)p.out + m0 * p.out_pix_stride * oesz, (unsigned)(rows * p.out_pix_stride * oesz));
    bool bad = false;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = tn * 32 + 8 * j + 4 * kq;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = acc[tn][4 * j + e] * out_scale + bias[tn][j][e];
          v[e] = fmaxf(t, t * slope) * ps[tn][j][e] + pt[tn][j][e];
        }
        if (tn == 0 && j == 0) bad |= egne_nonfinite(v[0]);       // lane = pixel: one channel per pixel (common.h)
        if (o16) {
          const f32x2 u0 = {v[0] * p.out_split_scale, v[1] * p.out_split_scale}, u1 = {v[2] * p.out_split_scale, v[3] * p.out_split_scale};
          const h2 h0 = __builtin_convertvector(u0, h2), h1 = __builtin_convertvector(u1, h2);
          if (tn == 0 && j == 0) bad |= egne_nonfinite((float)h0[0]);
          const u32x2 two = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
          __builtin_amdgcn_raw_buffer_store_b64(two, ro, n < p.Cout_store ? (li * (int)p.out_pix_stride + p.out_ch_off + n) * 2 : (int)OOB, 0, 0);
        } else
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro,
                                               n < p.Cout_store ? (li * (int)p.out_pix_stride + p.out_ch_off + n) * 4 : (int)OOB, 0, 0);
      }
    egne_ovf_commit(bad, p.ovf_flag);
  }
}

// OIHW [Cout][Cin<=4][3][3] fp32 -> hi / lo f16 fragments [3][CoutP/32][lane = h*32 + n%32][8]
__global__ void pack_c4_f16_k(const float* __restrict__ w, int Cout, int Cin, int NT, float wscale, _Float16* __restrict__ hi,
                              _Float16* __restrict__ lo) {
  const int total = 3 * NT * 512;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i & 7, nn = (i >> 3) & 31, h = (i >> 8) & 1;
    const int q = i >> 9;
    const int tn = q % NT, s = q / NT;
    const int n = tn * 32 + nn, tap = 4 * s + 2 * h + (j >> 2), c = j & 3;
    const float v = (n < Cout && tap < 9 && c < Cin) ? w[((long long)n * Cin + c) * 9 + tap] * wscale : 0.f;
    const _Float16 hh = (_Float16)v;
    hi[i] = hh;
    lo[i] = (_Float16)(v - (float)hh);
  }
}

}  // namespace

extern "C" int egne_pack_conv3x3_c4_weight_f16(const float* w_oihw, int Cout, int Cin, int CoutP, float wscale, void* fhi, void* flo,
                                               void* stream) {
  EGNE_REQUIRE(w_oihw && fhi && flo && Cout > 0 && Cin > 0 && Cin <= 4 && (CoutP == 32 || CoutP == 64) && Cout <= CoutP && wscale > 0.f,
               "pack_conv3x3_c4_f16: bad sizes Cout %d Cin %d CoutP %d", Cout, Cin, CoutP);
  hipLaunchKernelGGL(pack_c4_f16_k, dim3(6), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, CoutP / 32, wscale, (_Float16*)fhi,
                     (_Float16*)flo);
  return egne::check_launch("egne_pack_conv3x3_c4_weight_f16");
}

// Same layers as egne_conv3x3_smallcin_fwd (3x3 / stride 1 / pad 1, logical Cin <= 4, input slice of >= 4 channels, Cout <= 64),
// split-f16 arithmetic, streaming.  d->CoutP = 32 or 64 (rows of the pack).
extern "C" int egne_conv3x3_smallcin_f16_fwd(const egne_conv_desc* dp, const void* fhi, const void* flo, float a_scale, float w_scale,
                                             void* stream) {
  EGNE_REQUIRE(dp && fhi && flo, "conv_smallcin_f16: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad_h == 1 && d.pad_w == 1 && d.pad_mode == 0 && d.ngroups == 1 &&
               d.dil[0] == 1 && d.nseg == 1 && d.Ho == d.H && d.Wo == d.W && !d.residual, "conv_smallcin_f16: geometry not supported");
  const egne_seg& g = d.seg[0];
  const bool planar = g.pix_stride == 1;      // [B][Cp][H][W] (NCHW, Cp <= 4 planes) read in place
  EGNE_REQUIRE(g.ptr && !g.scale && ((uintptr_t)g.ptr & 15) == 0 &&
               (planar ? (g.Cp >= 1 && g.Cp <= 4 && g.ch_off == 0 && 2ll * d.H * d.W * g.Cp * 4 < (1ll << 31))
                       : (g.Cp >= 4 && g.ch_off % 4 == 0 && g.pix_stride % 4 == 0 && 2ll * d.H * d.W * g.pix_stride * 4 < (1ll << 31))),
               "conv_smallcin_f16: input slice");
  EGNE_REQUIRE((d.CoutP == 32 || d.CoutP == 64) && d.Cout_store <= d.CoutP && d.Cout_store % 4 == 0 && d.out && ((uintptr_t)d.out & 15) == 0 &&
               d.out_ch_off % 4 == 0 && d.out_pix_stride % 4 == 0 && d.out_ch_off + d.Cout_store <= d.out_pix_stride &&
               d.out_pix_stride * 128 < (1ll << 31) && (!d.bias || ((uintptr_t)d.bias & 15) == 0) &&
               (!d.post_scale || (((uintptr_t)d.post_scale & 15) == 0 && ((uintptr_t)d.post_shift & 15) == 0)), "conv_smallcin_f16: output");
  EGNE_REQUIRE(((uintptr_t)fhi & 15) == 0 && ((uintptr_t)flo & 15) == 0 && a_scale > 0.f && w_scale > 0.f, "conv_smallcin_f16: weights / scales");
  EGNE_REQUIRE(d.out_split == 0 || (d.out_split == 2 && d.out_split_scale > 0.f && !d.post_scale), "conv_smallcin_f16: f16 output (out_split = 2) needs a positive scale and no post affine");
  const long long M = (long long)d.B * d.H * d.W;
  const long long nb = (M + 31) / 32;
  EGNE_REQUIRE(nb < (1ll << 31), "conv_smallcin_f16: too many pixels");
  long long gx = (nb + 3) / 4;
  if (gx > 256 * 8) gx = 256 * 8;
  const float os = 1.0f / (a_scale * w_scale);
  hipStream_t st = (hipStream_t)stream;
  if (d.CoutP == 32)
    hipLaunchKernelGGL((conv3x3_c4_f16_kernel<1>), dim3((unsigned)gx), dim3(256), 0, st, d, (const _Float16*)fhi, (const _Float16*)flo, a_scale, os,
                       M, (int)nb);
  else
    hipLaunchKernelGGL((conv3x3_c4_f16_kernel<2>), dim3((unsigned)gx), dim3(256), 0, st, d, (const _Float16*)fhi, (const _Float16*)flo, a_scale, os,
                       M, (int)nb);
  return egne::check_launch("egne_conv3x3_smallcin_f16_fwd");
}
