// Weight gradient of the 3x3 / stride 1 / pad 1 single-slice convolutions (most of ESF-Net: utils.py:1047-1048,
// models/RITnet_v2.py:57-62,85-87) with an LDS-resident halo tile, exact fp32 on v_mfma_f32_32x32x2_f32.
//
//   gw[co][ci][tap] = sum over pixels  gz[pixel][co] * x[pixel + tap][ci]        (gz: gradient w.r.t. the pre-activation output)
//
// The generic kernel (backward.hip) runs one (tap, 32 channels) column tile per workgroup and re-reads gz and x from
// L2 for every tap: 8 FLOP per byte moved.  Here a workgroup owns one (32 output channels, 32 input channels) block
// for ALL 9 taps and walks 8 x 32 pixel tiles grid-stride: per tile the x halo (10 x 34 pixels) and the gz tile are
// staged ONCE (buffer loads, out-of-image lanes read zeros through the range check; the consumer's fused
// InstanceNorm affine + activation is applied while staging, as in the forward kernels), and the 9 taps are LDS
// address offsets: 288 MFMAs per wave and tile against 75 KB staged (72 FLOP per HBM byte).  The 9 x 16 accumulators
// stay in registers across tiles; at the end the four waves are summed through LDS and the block writes its partial
// to ws[split][tap][CoutP][Ktot], the layout egne_conv2d_wgrad's deterministic reduce expects.
#include "common.h"
#include "split_f16.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

namespace {

constexpr unsigned OOB = 0x80000000u;
constexpr int TH = 8, TW = 32, HW_ = TW + 2, HH_ = TH + 2, NPX = HH_ * HW_;   // 340 halo pixels
constexpr int NX = (NPX * 8 + 255) / 256;                                     // 11 x items (float4) per thread
constexpr int NG = TH * TW * 8 / 256;                                         // 8 gz items per thread

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_halo_kernel(const egne_conv_desc p, const float* __restrict__ gz, long long gzs,
                                                                    int gzo, int tiles_x, int tiles_y, int ntiles, int nchunk,
                                                                    float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;                 // [340][32]
  float* Gs = lds + NPX * 32;      // [256][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ct = blockIdx.y / nchunk, cc = blockIdx.y - ct * nchunk;
  const int co0 = ct * 32, c0 = cc * 32;
  const egne_seg sg = p.seg[0];
  const int c4 = tid & 7;
  const unsigned frame_x = (unsigned)p.H * p.W * (unsigned)sg.pix_stride * 4u;
  const unsigned frame_g = (unsigned)p.H * p.W * (unsigned)gzs * 4u;
  const bool cok = c0 + c4 * 4 < sg.Cp, gok = co0 + c4 * 4 < p.Cout_store;

  // tile-independent staging constants
  int hyx[NX];
  {
    int px = tid >> 3;
    int hy = px / HW_, hx = px - hy * HW_;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const bool in = tid + 256 * i < NPX * 8 && cok;
      hyx[i] = in ? (hy << 16) | hx : 0x7fff7fff;
      hx += 32;
      if (hx >= HW_) { hx -= HW_; ++hy; }
    }
  }
  const int goff0 = ((tid >> 3) * (int)gzs + c4 * 4) * 4, grow = p.W * (int)gzs * 4;   // gz item i: tile row i, column tid>>3
  const int xs4 = (int)sg.pix_stride * 4;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x16)(0.f);
  const float slope_in = sg.act_in == EGNE_ACT_RELU ? 0.f : (sg.act_in == EGNE_ACT_LEAKY ? 0.01f : 1.f);

  u32x4 sx[NX], sgz[NG];   // live one after the other
  unsigned xmask = 0;     // bit i: item i of the staged x tile is inside the image
  int stage_b = 0;
  auto load_tile = [&](int t) {
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    stage_b = b;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(sg.ptr + (long long)b * p.H * p.W * sg.pix_stride, frame_x);
    const int xbase = (((y0 - 1) * p.W + x0 - 1) * (int)sg.pix_stride + sg.ch_off + c0) * 4;
    xmask = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const unsigned iy = (unsigned)(y0 - 1 + (hyx[i] >> 16)), ix = (unsigned)(x0 - 1 + (hyx[i] & 0xffff));
      const bool ok = iy < (unsigned)p.H && ix < (unsigned)p.W;
      const int xo = ((hyx[i] >> 16) * p.W + (hyx[i] & 0xffff)) * xs4 + c4 * 16;
      sx[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xbase + xo : (int)OOB, 0, 0);
      xmask |= (ok ? 1u : 0u) << i;
    }
  };
  auto load_gz = [&](int t) {
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(gz + (long long)b * p.H * p.W * gzs, frame_g);
    const int gbase = ((y0 * p.W + x0) * (int)gzs + gzo + co0) * 4;
    const bool colok = gok && x0 + (tid >> 3) < p.W;
#pragma unroll
    for (int i = 0; i < NG; ++i)
      sgz[i] = __builtin_amdgcn_raw_buffer_load_b128(rg, (colok && y0 + i < p.H) ? gbase + goff0 : (int)OOB, i * grow, 0);
  };
  auto store_tile = [&]() {
    if (sg.scale) {
      f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (cok) {
        sc = *(const f32x4*)(sg.scale + (long long)stage_b * sg.Cp + c0 + c4 * 4);
        sh = *(const f32x4*)(sg.shift + (long long)stage_b * sg.Cp + c0 + c4 * 4);
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        f32x4 v = __builtin_bit_cast(f32x4, sx[i]) * sc + sh;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope_in);
        if (!((xmask >> i) & 1u)) v = (f32x4)(0.f);
        sx[i] = __builtin_bit_cast(u32x4, v);
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i)
      if (i < NX - 1 || tid + 256 * i < NPX * 8) *(u32x4*)&Xs[((tid >> 3) + 32 * i) * 32 + c4 * 4] = sx[i];
  };
  auto store_gz = [&]() {
#pragma unroll
    for (int i = 0; i < NG; ++i) *(u32x4*)&Gs[((tid >> 3) + 32 * i) * 32 + c4 * 4] = sgz[i];
  };

  // No register prefetch across the MFMA phase (144 accumulators leave no room at 2 waves per SIMD): the two
  // workgroups of a CU overlap each other's staging and matrix phases instead.
  int t = blockIdx.x;
  while (t < ntiles) {
    load_tile(t);
    __syncthreads();          // every wave has finished the MFMAs of the previous tile
    store_tile();
    load_gz(t);
    store_gz();
    __syncthreads();
    const int tn = t + gridDim.x;
    // wave w: tile rows 2w, 2w+1; K = pixel pairs (2s + lh)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * wave + rr;
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const float a = Gs[(row * 32 + 2 * s + lh) * 32 + li];
        float b[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) b[ky * 3 + kx] = Xs[((row + ky) * HW_ + 2 * s + lh + kx) * 32 + li];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[tp], acc[tp], 0, 0, 0);
      }
    }
    t = tn;
  }

  // cross-wave sum through LDS, one tap at a time; lane holds column k = li of rows co = (r&3) + 8*(r>>2) + 4*lh
  float* red = lds;   // [4][16][64]
  const long long tapsz = (long long)p.CoutP * p.Ktot;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[tp][r];
    __syncthreads();
    if (wave == 0) {
      float* dst = ws + ((long long)blockIdx.x * 9 + tp) * tapsz;
      const int k = c0 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) +
                        (red[(2 * 16 + r) * 64 + lane] + red[(3 * 16 + r) * 64 + lane]);
        const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (k < sg.Cp) dst[(long long)co * p.Ktot + k] = v;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// The same weight gradient on split-f16 products (3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulation; training
// plans, engine.TRAIN_SPLIT).  Same tiling, same staging loads, same partial layout.  Differences:
//   * both tiles are stored in LDS as ONE dword per element = (hi | lo << 16), hi = f16(v*s), lo = f16(v*s - hi): the split is
//     paid once per staged element (not per tap), the LDS footprint is that of the fp32 kernel;
//   * K = 16 consecutive pixels of a tile row: lane (li, lh) gathers the 8 dwords of pixels 16ks + 8lh + 0..7 of its channel
//     (channel-minor layout kept, so a tap is still just an address offset - any alignment) and v_perm_b32 sorts them into the
//     hi and lo operands; per kernel row the three kx taps share 10 gathered dwords;
//   * pre-scales: x by xa (16 for inputs normalised on load, else the forward launch's device word p.dyn_scale), gz by the
//     word act_bwd_bias_absmax wrote; the partials are scaled back when they are written.
// 27 MFMAs (864 cycles) per K step against 38 ds_read_b32 + ~80 v_perm: MFMA time per tile is 5.3x below the fp32 kernel's
// (108 x 32 vs 288 x 64 cycles), which leaves the kernel bound by its staging / HBM (2.5 GB per 240x320x32 layer and step).
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pack_hl(float v, float s) {
  const _Float16 h = (_Float16)(v * s);
  const _Float16 l = (_Float16)(v * s - (float)h);
  return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}

__device__ __forceinline__ void sort_hl(const unsigned* d, f16x8& hi, f16x8& lo) {      // d[0..7]: (hi | lo << 16) of 8 pixels
  u32x4 h, l;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    h[m] = __builtin_amdgcn_perm(d[2 * m + 1], d[2 * m], 0x05040100u);
    l[m] = __builtin_amdgcn_perm(d[2 * m + 1], d[2 * m], 0x07060302u);
  }
  hi = __builtin_bit_cast(f16x8, h);
  lo = __builtin_bit_cast(f16x8, l);
}

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_halo_f16_kernel(const egne_conv_desc p, const float* __restrict__ gz, long long gzs,
                                                                        int gzo, const unsigned* __restrict__ gz_dyn, int tiles_x,
                                                                        int tiles_y, int ntiles, int nchunk, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* Xs = (unsigned*)lds;                 // [340][32] packed (hi | lo << 16)
  unsigned* Gs = (unsigned*)lds + NPX * 32;      // [256][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ct = blockIdx.y / nchunk, cc = blockIdx.y - ct * nchunk;
  const int co0 = ct * 32, c0 = cc * 32;
  const egne_seg sg = p.seg[0];
  const int c4 = tid & 7;
  const unsigned frame_x = (unsigned)p.H * p.W * (unsigned)sg.pix_stride * 4u;
  const unsigned frame_g = (unsigned)p.H * p.W * (unsigned)gzs * 4u;
  const bool cok = c0 + c4 * 4 < sg.Cp, gok = co0 + c4 * 4 < p.Cout_store;
  float xa = 16.f, xo_ = 1.f / 16.f, ga = 1.f, go_ = 1.f;
  egne::dyn_scales(p.dyn_scale, xa, xo_);
  egne::dyn_scales(gz_dyn, ga, go_);
  const float out_scale = xo_ * go_;

  int hyx[NX];
  {
    int px = tid >> 3;
    int hy = px / HW_, hx = px - hy * HW_;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const bool in = tid + 256 * i < NPX * 8 && cok;
      hyx[i] = in ? (hy << 16) | hx : 0x7fff7fff;
      hx += 32;
      if (hx >= HW_) { hx -= HW_; ++hy; }
    }
  }
  const int goff0 = ((tid >> 3) * (int)gzs + c4 * 4) * 4, grow = p.W * (int)gzs * 4;
  const int xs4 = (int)sg.pix_stride * 4;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x16)(0.f);
  const float slope_in = sg.act_in == EGNE_ACT_RELU ? 0.f : (sg.act_in == EGNE_ACT_LEAKY ? 0.01f : 1.f);

  u32x4 sx[NX], sgz[NG];
  unsigned xmask = 0;
  int stage_b = 0;
  auto load_tile = [&](int t) {
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    stage_b = b;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(sg.ptr + (long long)b * p.H * p.W * sg.pix_stride, frame_x);
    const int xbase = (((y0 - 1) * p.W + x0 - 1) * (int)sg.pix_stride + sg.ch_off + c0) * 4;
    xmask = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const unsigned iy = (unsigned)(y0 - 1 + (hyx[i] >> 16)), ix = (unsigned)(x0 - 1 + (hyx[i] & 0xffff));
      const bool ok = iy < (unsigned)p.H && ix < (unsigned)p.W;
      const int xo = ((hyx[i] >> 16) * p.W + (hyx[i] & 0xffff)) * xs4 + c4 * 16;
      sx[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xbase + xo : (int)OOB, 0, 0);
      xmask |= (ok ? 1u : 0u) << i;
    }
  };
  auto load_gz = [&](int t) {
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(gz + (long long)b * p.H * p.W * gzs, frame_g);
    const int gbase = ((y0 * p.W + x0) * (int)gzs + gzo + co0) * 4;
    const bool colok = gok && x0 + (tid >> 3) < p.W;
#pragma unroll
    for (int i = 0; i < NG; ++i)
      sgz[i] = __builtin_amdgcn_raw_buffer_load_b128(rg, (colok && y0 + i < p.H) ? gbase + goff0 : (int)OOB, i * grow, 0);
  };
  auto store_tile = [&]() {
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (sg.scale && cok) {
      sc = *(const f32x4*)(sg.scale + (long long)stage_b * sg.Cp + c0 + c4 * 4);
      sh = *(const f32x4*)(sg.shift + (long long)stage_b * sg.Cp + c0 + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      f32x4 v = __builtin_bit_cast(f32x4, sx[i]);
      if (sg.scale) {
        v = v * sc + sh;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope_in);
        if (!((xmask >> i) & 1u)) v = (f32x4)(0.f);
      }
      u32x4 q;
#pragma unroll
      for (int e = 0; e < 4; ++e) q[e] = pack_hl(v[e], xa);
      if (i < NX - 1 || tid + 256 * i < NPX * 8) *(u32x4*)&Xs[((tid >> 3) + 32 * i) * 32 + c4 * 4] = q;
    }
  };
  auto store_gz = [&]() {
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const f32x4 v = __builtin_bit_cast(f32x4, sgz[i]);
      u32x4 q;
#pragma unroll
      for (int e = 0; e < 4; ++e) q[e] = pack_hl(v[e], ga);
      *(u32x4*)&Gs[((tid >> 3) + 32 * i) * 32 + c4 * 4] = q;
    }
  };

  int t = blockIdx.x;
  while (t < ntiles) {
    load_tile(t);
    __syncthreads();          // every wave has finished the MFMAs of the previous tile
    store_tile();
    load_gz(t);               // (after the x items are out of their registers: 144 accumulators leave no room for both)
    store_gz();
    __syncthreads();
    // wave w: tile rows 2w, 2w+1; K step = 16 pixels of the row, lane (li, lh): pixels 16ks + 8lh + j, j = 0..7
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {          // (rolled: one K step's operands live at a time beside the 144 accumulators)
      const int row = 2 * wave + (it >> 1);
      {
        const int xk = 16 * (it & 1) + 8 * lh;
        unsigned ad[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ad[j] = Gs[(row * 32 + xk + j) * 32 + li];
        f16x8 ah, al;
        sort_hl(ad, ah, al);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          unsigned bd[10];
#pragma unroll
          for (int j = 0; j < 10; ++j) bd[j] = Xs[((row + ky) * HW_ + xk + j) * 32 + li];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            f16x8 bh, bl;
            sort_hl(bd + kx, bh, bl);
            const int tp = ky * 3 + kx;
            acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[tp], 0, 0, 0);
            acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[tp], 0, 0, 0);
            acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[tp], 0, 0, 0);
          }
        }
      }
    }
    t += gridDim.x;
  }

  float* red = lds;   // [4][16][64]
  const long long tapsz = (long long)p.CoutP * p.Ktot;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[tp][r];
    __syncthreads();
    if (wave == 0) {
      float* dst = ws + ((long long)blockIdx.x * 9 + tp) * tapsz;
      const int k = c0 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) +
                        (red[(2 * 16 + r) * 64 + lane] + red[(3 * 16 + r) * 64 + lane]);
        const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (k < sg.Cp) dst[(long long)co * p.Ktot + k] = v * out_scale;
      }
    }
  }
}

}  // namespace

namespace egne {

bool wgrad_halo_supported(const egne_conv_desc& d, long long gzs) {
  static const bool off = [] { const char* e = getenv("EGNE_WGRAD_HALO"); return e && e[0] == '0'; }();
  return !off && d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad_h == 1 && d.pad_w == 1 && d.pad_mode == 0 && d.ngroups == 1 &&
         d.nseg == 1 && d.dil[0] == 1 && d.Ho == d.H && d.Wo == d.W && d.W >= 16 &&
         (long long)d.H * d.W * d.seg[0].pix_stride * 4 < (1ll << 31) && (long long)d.H * d.W * gzs * 4 < (1ll << 31);
}

int wgrad_halo_splits(const egne_conv_desc& d) {
  const int tiles = ((d.W + TW - 1) / TW) * ((d.H + TH - 1) / TH) * d.B;
  const int ny = (d.CoutP / 32) * ((d.seg[0].Cp + 31) / 32);
  int gx = (512 + ny - 1) / ny;
  if (gx > tiles) gx = tiles;
  return gx < 1 ? 1 : gx;
}

int wgrad_halo_launch(const egne_conv_desc& d, const float* gz, long long gzs, int gzo, float* ws, hipStream_t st) {
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y * d.B;
  const int nchunk = (d.seg[0].Cp + 31) / 32;
  const int gx = wgrad_halo_splits(d);
  const size_t lds = (size_t)(NPX + TH * TW) * 32 * sizeof(float);
  static bool once = [] {
    return hipFuncSetAttribute((const void*)conv3x3_wgrad_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
  }();
  if (!once) return fail(EGNE_ERR_LAUNCH, "wgrad_halo: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL(conv3x3_wgrad_halo_kernel, dim3(gx, (d.CoutP / 32) * nchunk), dim3(256), lds, st, d, gz, gzs, gzo, tiles_x, tiles_y,
                     ntiles, nchunk, ws);
  return check_launch("conv3x3_wgrad_halo");
}

// split-f16 form; gz_dyn: device word with the bit pattern of max|gz| (egne_act_bwd_bias_absmax).  The caller has checked
// wgrad_halo_supported and that x carries a usable pre-scale (normalised on load, or d.dyn_scale set by the forward launch).
int wgrad_halo_f16_launch(const egne_conv_desc& d, const float* gz, long long gzs, int gzo, const unsigned* gz_dyn, float* ws,
                          hipStream_t st) {
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y * d.B;
  const int nchunk = (d.seg[0].Cp + 31) / 32;
  const int gx = wgrad_halo_splits(d);
  const size_t lds = (size_t)(NPX + TH * TW) * 32 * sizeof(float);
  static bool once = [] {
    return hipFuncSetAttribute((const void*)conv3x3_wgrad_halo_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
  }();
  if (!once) return fail(EGNE_ERR_LAUNCH, "wgrad_halo_f16: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL(conv3x3_wgrad_halo_f16_kernel, dim3(gx, (d.CoutP / 32) * nchunk), dim3(256), lds, st, d, gz, gzs, gzo, gz_dyn,
                     tiles_x, tiles_y, ntiles, nchunk, ws);
  return check_launch("conv3x3_wgrad_halo_f16");
}

}  // namespace egne
