// k x k (k <= 7) / stride 1 / zero-padded convolution with a NARROW output (<= 8 channels) over one bf16 slice of 32-128 channels, on
// v_mfma_f32_16x16x32_bf16 with an LDS-resident halo (training plans with bf16 activation storage).
//
// The layer this exists for: the data gradient of the StyleEncoder's first convolution (RITnet_v2.py:95, utils.py:1051-1149: a
// reflect-padded 7x7 from the 3 softmax channels to 64) -- as an ordinary convolution (engine.TransposedLayer) it maps 64 channels
// back to 3 (8 padded) over 49 taps.  On the generic implicit GEMM that is 98 K steps per tile, each re-gathering a 128 x 32 operand
// tile from memory for two MFMAs per wave of which three quarters multiply padding columns: 4.4 ms per 64 frames, 22 TFLOP/s.
// Here a workgroup owns an 8 x 32 block of output pixels: the (8 + k - 1) x (32 + k - 1) halo of all input channels is staged ONCE
// (144-byte pixel pitch for 64 channels: conflict-free ds_read_b128 over 16 pixels), the taps are address offsets into it, the
// weights ([tap][32-channel chunk][8 output rows][32] bf16, rounded from the fp32 pack) stay in LDS for the whole launch, and the
// product is transposed (weights as the 16-row A operand, rows 8-15 zero) so that a lane ends with 4 consecutive output channels of
// one pixel.  Eight waves, one output row each; the next tile's halo is requested before the current tile's MFMAs.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int TW = 32, TH = 8, NT = 512;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// K: kernel size (kh = kw = K), NCH: 32-channel chunks of the input slice
template <int K, int NCH>
__global__ __launch_bounds__(NT)
void conv_narrow_bf16_kernel(const egne_conv_desc p, int tiles_x, int tiles_y, int ntiles) {
  constexpr int HW_ = TW + K - 1, HH_ = TH + K - 1, NPX = HH_ * HW_;
  constexpr int PITCH = NCH * 32 + 8;                                  // halfs per halo pixel: 16 bytes past a multiple of 64
  constexpr int NI = (NPX * NCH * 4 + NT - 1) / NT;                    // 16-byte items per thread and tile
  constexpr int T = K * K;
  extern __shared__ __attribute__((aligned(16))) egne_bf16 lds[];
  egne_bf16* const halo = lds;                                          // [NPX][PITCH]
  egne_bf16* const wl = lds + NPX * PITCH;                              // [T][NCH][8][32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // = output row of the tile
  const int l15 = lane & 15, kg = lane >> 4;
  const egne_seg sg = p.seg[0];
  const egne_bf16* const xin = (const egne_bf16*)sg.ptr;
  const unsigned frame_in = (unsigned)p.H * p.W * (unsigned)sg.pix_stride * 2u;
  const unsigned frame_out = (unsigned)p.Ho * p.Wo * (unsigned)p.out_pix_stride * 2u;

  // weights: fp32 pack [tap][CoutP][Ktot] -> bf16 [tap][chunk][8][32]
  for (int e = tid; e < T * NCH * 8 * 32; e += NT) {
    const int k = e & 31, n = (e >> 5) & 7, r = e >> 8, ch = r % NCH, tap = r / NCH;
    const int c = ch * 32 + k;
    wl[e] = (egne_bf16)((n < p.Cout_store && c < sg.Cp) ? p.w[((long long)tap * p.CoutP + n) * p.Ktot + c] : 0.f);
  }

  struct Tile { int b, y0, x0; };
  auto decode = [&](int t) {
    Tile r;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    r.b = t; r.y0 = ty * TH; r.x0 = tx * TW;
    return r;
  };
  u32x4 st[NI];
  auto issue = [&](int t, bool on) {
    const Tile tl = decode(t);
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(xin + (long long)tl.b * p.H * p.W * sg.pix_stride, frame_in);
#pragma unroll
    for (int I = 0; I < NI; ++I) {
      const int it = tid + NT * I, px = it / (NCH * 4), pc = it - px * (NCH * 4);
      const int hy = px / HW_, hx = px - hy * HW_;
      const int y = tl.y0 - p.pad_h + hy, x = tl.x0 - p.pad_w + hx;        // input pixel of halo position (hy, hx)
      const bool ok = on && px < NPX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W && pc * 8 < sg.Cp;
      st[I] = __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? (int)((((long long)y * p.W + x) * sg.pix_stride + sg.ch_off + pc * 8) * 2) : (int)OOB, 0, 0);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int I = 0; I < NI; ++I) {
      const int it = tid + NT * I, px = it / (NCH * 4), pc = it - px * (NCH * 4);
      if (px < NPX) *(u32x4*)&halo[px * PITCH + pc * 8] = st[I];
    }
  };

  const f32x4 bq = (p.bias && kg < 2) ? *(const f32x4*)(p.bias + 4 * kg) : (f32x4)(0.f);
  const float slope = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  int t = blockIdx.x;
  issue(t, t < ntiles);
  for (; t < ntiles; t += gridDim.x) {
    __syncthreads();                 // every wave is done with the previous tile's halo (and, the first time, the weights are written)
    stage();
    __syncthreads();
    const int tn = t + gridDim.x;
    issue(tn < ntiles ? tn : t, tn < ntiles);
    f32x4 acc[2] = {(f32x4)(0.f), (f32x4)(0.f)};
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int tap = ky * K + kx;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          // A: weights of (tap, chunk): row l15 (output channel; rows 8-15 are zero), k-group kg
          egne_bf16x8 a = *(const egne_bf16x8*)&wl[((tap * NCH + ch) * 8 + (l15 & 7)) * 32 + kg * 8];
          if (l15 >= 8) a = (egne_bf16x8)(egne_bf16)0.f;
#pragma unroll
          for (int ph = 0; ph < 2; ++ph) {
            const int q = (wave + ky) * HW_ + ph * 16 + l15 + kx;
            const egne_bf16x8 b = *(const egne_bf16x8*)&halo[q * PITCH + ch * 32 + kg * 8];
            acc[ph] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[ph], 0, 0, 0);
          }
        }
      }
    }
    // lane (l15, kg) holds output channels 4 kg .. 4 kg + 3 of pixel (y0 + wave, x0 + 16 ph + l15); channels >= 8 are padding rows
    const Tile tl = decode(t);
    const __amdgpu_buffer_rsrc_t rout = make_rsrc((egne_bf16*)p.out + (long long)tl.b * p.Ho * p.Wo * p.out_pix_stride, frame_out);
    const int y = tl.y0 + wave;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int x = tl.x0 + ph * 16 + l15;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u = acc[ph][e] + bq[e];
        v[e] = fmaxf(u, u * slope);
      }
      const bool ok = y < p.Ho && x < p.Wo && 4 * kg < p.Cout_store;
      const int off = ok ? (int)((((long long)y * p.Wo + x) * p.out_pix_stride + p.out_ch_off + 4 * kg) * 2) : (int)OOB;
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, __builtin_convertvector(v, egne_bf16x4)), rout, off, 0, 0);
    }
  }
}

template <int K, int NCH>
int launch_narrow(const egne_conv_desc& d, hipStream_t st) {
  const int tiles_x = (d.Wo + TW - 1) / TW, tiles_y = (d.Ho + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y * d.B;
  constexpr size_t lds = ((size_t)(TH + K - 1) * (TW + K - 1) * (NCH * 32 + 8) + (size_t)K * K * NCH * 8 * 32) * sizeof(egne_bf16);
  static_assert(lds <= 163840, "LDS budget");
  static bool once = hipFuncSetAttribute((const void*)conv_narrow_bf16_kernel<K, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
  if (!once) return egne::fail(EGNE_ERR_LAUNCH, "conv_narrow_bf16: cannot raise the dynamic LDS limit to %zu", lds);
  hipLaunchKernelGGL((conv_narrow_bf16_kernel<K, NCH>), dim3(ntiles < 256 ? ntiles : 256), dim3(NT), lds, st, d, tiles_x, tiles_y, ntiles);
  return egne::check_launch("egne_conv_narrow_bf16_fwd");
}

}  // namespace

extern "C" int egne_conv_narrow_bf16_supported(const egne_conv_desc* dp) {
  if (!dp) return 0;
  const egne_conv_desc& d = *dp;
  const egne_seg& g = d.seg[0];
  return d.dtype == 1 && d.kh == d.kw && (d.kh == 7 || d.kh == 5 || d.kh == 3) && d.stride == 1 && d.pad_mode == 0 && d.ngroups == 1 && d.nseg == 1 &&
         d.dil[0] == 1 && d.Ho == d.H + 2 * d.pad_h - d.kh + 1 && d.Wo == d.W + 2 * d.pad_w - d.kw + 1 && (g.Cp == 32 || g.Cp == 64) && d.Ktot == g.Cp &&
         !g.scale && g.act_in == EGNE_ACT_NONE && g.ch_off % 8 == 0 && g.pix_stride % 8 == 0 && ((uintptr_t)g.ptr & 15) == 0 &&
         d.Cout_store <= 8 && d.Cout_store % 4 == 0 && d.out_ch_off % 4 == 0 && d.out_pix_stride % 4 == 0 && ((uintptr_t)d.out & 7) == 0 &&
         !d.residual && !d.post_scale && d.w && (long long)d.H * d.W * g.pix_stride * 2 < (1ll << 31) &&
         (long long)d.Ho * d.Wo * d.out_pix_stride * 2 < (1ll << 31);
}

// d: the convolution exactly as egne_conv2d_fwd takes it (dtype 1, w = the fp32 flat pack [tap][CoutP][Ktot]); see the _supported test.
extern "C" int egne_conv_narrow_bf16_fwd(const egne_conv_desc* dp, void* stream) {
  EGNE_REQUIRE(dp && egne_conv_narrow_bf16_supported(dp), "conv_narrow_bf16: descriptor not supported");
  const egne_conv_desc& d = *dp;
  hipStream_t st = (hipStream_t)stream;
  const bool two = d.seg[0].Cp == 64;
  if (d.kh == 7) return two ? launch_narrow<7, 2>(d, st) : launch_narrow<7, 1>(d, st);
  if (d.kh == 5) return two ? launch_narrow<5, 2>(d, st) : launch_narrow<5, 1>(d, st);
  return two ? launch_narrow<3, 2>(d, st) : launch_narrow<3, 1>(d, st);
}
