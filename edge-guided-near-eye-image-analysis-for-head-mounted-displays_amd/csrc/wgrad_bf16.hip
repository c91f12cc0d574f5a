// Weight gradient of the 3x3 "same" convolutions over one bf16 slice on v_mfma_f32_32x32x16_bf16 (training plans with bf16
// activation storage; backward of models/RITnet_v2.py:57-62,85-87 and utils.py:1047-1048 under train.py:285-286):
//
//   dW[tap][co][ci] = sum over pixels  gz[n, y, x, co] * xin[n, y + dy, x + dx, ci]
//
// A workgroup owns one (32 co, 32 ci) block for ALL nine taps and walks 8 x 32 pixel tiles (every nsplit-th one): the gz tile
// [256 px][32 co] and the x halo [340 px][32 ci] are staged once per tile as plain 64-byte pixel rows of bf16, and both MFMA
// operands -- "eight consecutive PIXELS of one channel" -- come out of those images with ds_read_b64_tr_b16 (the 4 x 16
// transposing LDS read of gfx950: conflict-free on 64-byte rows), so a tap is nothing but an address offset into the halo.
// Wave w of eight contracts tile row w: 2 k-steps x 9 taps = 18 MFMAs per tile into 9 x 16 accumulator registers (with four waves
// of two rows each the 144 accumulators + 40 staging registers spilled).
// The layer is HBM-bound by a factor of ~6 (1.26 GB per 240x320x32 layer and 64 frames against 0.18 TFLOP), so the loop is a plain
// "load tile t+1 into registers, contract tile t, swap" with two barriers per tile.
// The optional per-(n, c) affine + activation of the forward launch (InstanceNorm fused on load) is applied while staging x.
// Partials: ws[split][tap][CoutP][Ktot] fp32, every element written (the caller reduces them into the OIHW gradient).
#include "common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int TW = 32, TH = 8, HWd = TW + 2, HHd = TH + 2, NPX = HHd * HWd;      // 340 halo pixels
constexpr int GPX = TW * TH;                                                      // 256 gz pixels
constexpr int NT = 512;                                                           // threads: eight waves, one tile row each
constexpr int NG = GPX * 4 / NT, NX = (NPX * 4 + NT - 1) / NT;                   // 16-byte items per thread: 2 + 3
constexpr unsigned OOB = 0x80000000u;
typedef __attribute__((address_space(3))) egne_bf16x4* lds_bf4_ptr;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 unpack4(unsigned a, unsigned b) {
  const u32x4 w = {a << 16, a & 0xffff0000u, b << 16, b & 0xffff0000u};
  return __builtin_bit_cast(f32x4, w);
}

__global__ __launch_bounds__(NT)
void wgrad3x3_bf16_kernel(const egne_conv_desc p, const egne_bf16* __restrict__ gz, long long gzs, int gzo, int nsplit, int nco,
                          int tiles_x, int tiles_y, int ntiles, float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) egne_bf16 lds[(GPX + NPX) * 32];        // 37.25 KB; reused for the final cross-wave sum
  egne_bf16* const Gi = lds;
  egne_bf16* const Xi = lds + GPX * 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  const egne_seg sg = p.seg[0];
  const egne_bf16* const xin = (const egne_bf16*)sg.ptr;
  const int cb = blockIdx.y % nco, kb = blockIdx.y / nco;
  const int co0 = cb * 32, ci0 = kb * 32;
  const int piece = tid & 3;                              // this thread's 8-channel group of a staged pixel (item = tid + 256 I: same piece)
  const bool gch_ok = co0 + piece * 8 < p.Cout_store, xch_ok = ci0 + piece * 8 < sg.Cp;
  const float slope_in = sg.act_in == EGNE_ACT_RELU ? 0.f : (sg.act_in == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  const unsigned frame_g = (unsigned)H * W * (unsigned)gzs * 2u, frame_x = (unsigned)H * W * (unsigned)sg.pix_stride * 2u;

  struct Tile { int b, y0, x0; };
  auto decode = [&](int t) {
    Tile r;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    r.b = t; r.y0 = ty * TH; r.x0 = tx * TW;
    return r;
  };
  u32x4 rg[NG], rx[NX];
  f32x4 asc[2] = {(f32x4)(1.f), (f32x4)(1.f)}, ash[2] = {(f32x4)(0.f), (f32x4)(0.f)};
  auto issue = [&](int t) {
    const Tile tl = decode(t);
    const __amdgpu_buffer_rsrc_t rgz = make_rsrc(gz + (long long)tl.b * H * W * gzs, frame_g);
    const __amdgpu_buffer_rsrc_t rxi = make_rsrc(xin + (long long)tl.b * H * W * sg.pix_stride, frame_x);
#pragma unroll
    for (int I = 0; I < NG; ++I) {
      const int px = (tid >> 2) + (NT / 4) * I, ty = px >> 5, tx = px & 31;
      const int y = tl.y0 + ty, x = tl.x0 + tx;
      const bool ok = gch_ok && y < H && x < W;
      rg[I] = __builtin_amdgcn_raw_buffer_load_b128(rgz, ok ? (int)(((long long)(y * W + x) * gzs + gzo + co0 + piece * 8) * 2) : (int)OOB, 0, 0);
    }
#pragma unroll
    for (int I = 0; I < NX; ++I) {
      const int q = (tid >> 2) + (NT / 4) * I, hy = q / HWd, hx = q - hy * HWd;
      const int y = tl.y0 - 1 + hy, x = tl.x0 - 1 + hx;
      const bool ok = xch_ok && q < NPX && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      rx[I] = __builtin_amdgcn_raw_buffer_load_b128(rxi, ok ? (int)(((long long)(y * W + x) * sg.pix_stride + sg.ch_off + ci0 + piece * 8) * 2) : (int)OOB, 0, 0);
    }
    if (sg.scale) {
      const int c = ci0 + piece * 8;
      const float* zs = xch_ok ? sg.scale + (long long)tl.b * sg.Cp + c : egne_zero_page;
      const float* zh = xch_ok ? sg.shift + (long long)tl.b * sg.Cp + c : egne_zero_page;
      asc[0] = *(const f32x4*)zs; asc[1] = *(const f32x4*)(zs + 4);
      ash[0] = *(const f32x4*)zh; ash[1] = *(const f32x4*)(zh + 4);
    }
  };
  auto stage = [&](int t) {       // registers -> LDS (plain [pixel][32] rows); the fused affine of the forward launch applied to x
    const Tile tl = decode(t);
#pragma unroll
    for (int I = 0; I < NG; ++I) *(u32x4*)&Gi[((tid >> 2) + (NT / 4) * I) * 32 + piece * 8] = rg[I];
#pragma unroll
    for (int I = 0; I < NX; ++I) {
      const int q = (tid >> 2) + (NT / 4) * I;
      if (I < NX - 1 || q < NPX) {
        u32x4 raw = rx[I];
        if (sg.scale) {
          const int hy = q / HWd, hx = q - hy * HWd;
          const int y = tl.y0 - 1 + hy, x = tl.x0 - 1 + hx;
          f32x4 v0 = unpack4(raw[0], raw[1]), v1 = unpack4(raw[2], raw[3]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t0 = v0[e] * asc[0][e] + ash[0][e], t1 = v1[e] * asc[1][e] + ash[1][e];
            v0[e] = fmaxf(t0, t0 * slope_in); v1[e] = fmaxf(t1, t1 * slope_in);
          }
          if (!(xch_ok && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)) { v0 = (f32x4)(0.f); v1 = (f32x4)(0.f); }
          const u32x2 p0 = __builtin_bit_cast(u32x2, __builtin_convertvector(v0, egne_bf16x4));
          const u32x2 p1 = __builtin_bit_cast(u32x2, __builtin_convertvector(v1, egne_bf16x4));
          raw = u32x4{p0[0], p0[1], p1[0], p1[1]};
        }
        *(u32x4*)&Xi[q * 32 + piece * 8] = raw;
      }
    }
  };

  // transposing-read address of this lane: 16-lane group g = lane >> 4 takes channels 16 (g & 1) .. + 15 and the pixel octet
  // h = g >> 1 of a 16-pixel k-step; lane 4 q + p of the group supplies pixel q, channels 4 p .. 4 p + 3 (cdna guide, T10)
  const int g16 = lane >> 4, i16 = lane & 15;
  const int lbase = ((8 * (g16 >> 1) + (i16 >> 2)) * 32 + 16 * (g16 & 1) + 4 * (i16 & 3));      // elements
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x16)(0.f);

  int t = blockIdx.x;
  if (t < ntiles) issue(t);
  for (; t < ntiles; t += nsplit) {
    __syncthreads();                 // every wave is done with the previous tile's images
    stage(t);
    __syncthreads();
    if (t + nsplit < ntiles) issue(t + nsplit);       // next tile's loads fly during this tile's MFMAs
    {
      const int ty = wave;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const egne_bf16* ga = Gi + lbase + (ty * 32 + 16 * s) * 32;
        const egne_bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)ga);
        const egne_bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(ga + 4 * 32));
        const egne_bf16x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int dy = tap / 3 - 1, dx = tap % 3 - 1;
          const egne_bf16* xa = Xi + lbase + ((ty + 1 + dy) * HWd + 16 * s + 1 + dx) * 32;
          const egne_bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)xa);
          const egne_bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(xa + 4 * 32));
          const egne_bf16x8 bq = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc[tap], 0, 0, 0);
        }
      }
    }
  }
  // cross-wave sum through LDS, one tap at a time (32 KB), then the workgroup's 32 x 32 block of every tap goes to its partial:
  // lane holds column k = lane & 31 of rows co = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* const red = (float*)lds;        // [8 waves][16][64]
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[tap][r];
    __syncthreads();
    float* dst = ws + ((long long)blockIdx.x * 9 + tap) * (long long)p.CoutP * p.Ktot;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + NT * j, r = e >> 6, l2 = e & 63;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += red[(w * 16 + r) * 64 + l2];
      const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * (l2 >> 5), k = ci0 + (l2 & 31);
      if (k < p.Ktot && co < p.CoutP) dst[(long long)co * p.Ktot + k] = v;
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// The same weight gradient for layers of 64+ channels on both sides (round 6).  With one (32 co, 32 ci) pair per workgroup a 64 -> 64
// layer staged every gz tile and every x halo twice (four workgroups per tile), an eight-wave workgroup spent a tile's time on 18 MFMAs
// per wave between two barriers and a single-buffered image, and the mid-resolution layers ran at 300-570 TFLOP/s.  Here a workgroup owns a
// (64 co, 64 ci) QUAD of pairs for all nine taps:
//   * both 32-channel halves of the gz tile and of the x halo are staged ONCE per tile, by LDS-DMA (buffer_load_dwordx4 ... lds: a wave
//     instruction writes 16 pixels x 64 bytes of one image, no staging registers, no vector ALU), into one of TWO image sets -- tile
//     t + 1 lands while tile t is contracted, one barrier per tile;
//   * wave w contracts pair (w & 3) over four tile rows (w >> 2): the x fragment of halo row r serves the taps (dy, tile row) with
//     row + 1 + dy = r, so it is read once for up to three MFMAs -- 36 x-fragment reads + 8 resident gz fragments per tile and wave for
//     72 MFMAs (the pair-per-workgroup form: 18 + 2 for 18);
//   * the two waves of a pair add their blocks through LDS at the end.
// Layers whose input is normalised on load (egne_seg.scale: conv1 of the down blocks) keep the form above (the DMA cannot apply the affine).
constexpr int WX = 352;                                   // halo pixels rounded up to whole DMA pieces of 16
constexpr int WSET = (2 * GPX + 2 * WX) * 32;             // bf16 elements of one image set: [gz half 0 | gz half 1 | x half 0 | x half 1]
constexpr int WNP = 2 * (GPX / 16) + 2 * (WX / 16);       // 76 DMA pieces per tile
constexpr int WPI = (WNP + 7) / 8;                        // pieces per wave (10; the last round is partial)
typedef __attribute__((address_space(3))) void* lds_vptr;

__global__ __launch_bounds__(NT)
void wgrad3x3_bf16_wide_kernel(const egne_conv_desc p, const egne_bf16* __restrict__ gz, long long gzs, int gzo, int nsplit, int nqco,
                               int nco, int nkc, int tiles_x, int tiles_y, int ntiles, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) egne_bf16 wlds[];      // two image sets; reused for the final cross-wave sum
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  const egne_seg sg = p.seg[0];
  const egne_bf16* const xin = (const egne_bf16*)sg.ptr;
  const int qc = blockIdx.y % nqco, qk = blockIdx.y / nqco;
  const int co0 = qc * 64, ci0 = qk * 64;
  const unsigned frame_g = (unsigned)H * W * (unsigned)gzs * 2u, frame_x = (unsigned)H * W * (unsigned)sg.pix_stride * 2u;

  struct Tile { int b, y0, x0; };
  auto decode = [&](int t) {
    Tile r;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    r.b = t; r.y0 = ty * TH; r.x0 = tx * TW;
    return r;
  };
  // this wave's DMA pieces j = wave + 8 I: tile-invariant byte offset relative to the tile's first pixel, and the column it reads
  // (the only coordinate that needs a test: rows outside the frame fall outside the buffer resource, channels past the tensor are baked in)
  int rel[WPI], col[WPI];
#pragma unroll
  for (int I = 0; I < WPI; ++I) {
    const int j = wave + 8 * I, pc = lane & 3;
    if (j < 2 * (GPX / 16)) {
      const int half = j / (GPX / 16), px = (j % (GPX / 16)) * 16 + (lane >> 2);
      const int c = co0 + half * 32 + pc * 8;
      col[I] = px & 31;
      rel[I] = c < p.Cout_store ? (int)((((long long)(px >> 5) * W + (px & 31)) * gzs + gzo + c) * 2) : (int)OOB;
    } else if (j < WNP) {
      const int jj = j - 2 * (GPX / 16), half = jj / (WX / 16), q = (jj % (WX / 16)) * 16 + (lane >> 2);
      const int hy = q / HWd, hx = q - hy * HWd, c = ci0 + half * 32 + pc * 8;
      col[I] = hx - 1;
      rel[I] = (q < NPX && c < sg.Cp) ? (int)((((long long)(hy - 1) * W + (hx - 1)) * sg.pix_stride + sg.ch_off + c) * 2) : (int)OOB;
    } else {
      col[I] = 0; rel[I] = (int)OOB;
    }
  }
  auto issue = [&](int t, int set) {
    const Tile tl = decode(t);
    const __amdgpu_buffer_rsrc_t rgz = make_rsrc(gz + (long long)tl.b * H * W * gzs, frame_g);
    const __amdgpu_buffer_rsrc_t rxi = make_rsrc(xin + (long long)tl.b * H * W * sg.pix_stride, frame_x);
    const int bg = (int)(((long long)tl.y0 * W + tl.x0) * gzs * 2), bx = (int)(((long long)tl.y0 * W + tl.x0) * sg.pix_stride * 2);
    char* const base = (char*)(wlds + set * WSET);
#pragma unroll
    for (int I = 0; I < WPI; ++I) {
      const int j = wave + 8 * I;
      if (j < WNP) {                                                     // (wave-uniform)
        const bool isg = j < 2 * (GPX / 16);
        const bool ok = rel[I] != (int)OOB && (unsigned)(tl.x0 + col[I]) < (unsigned)W;
        const int off = ok ? rel[I] + (isg ? bg : bx) : (int)OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(isg ? rgz : rxi, (lds_vptr)(base + j * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  // pair of this wave and its four tile rows
  const int pr = wave & 3, hh = wave >> 2;
  const int ch_c = pr & 1, ch_k = pr >> 1;
  const bool pair_ok = qc * 2 + ch_c < nco && qk * 2 + ch_k < nkc;     // (wave-uniform: the transposing reads run with all lanes enabled)
  const int g16 = lane >> 4, i16 = lane & 15;
  const int lbase = ((8 * (g16 >> 1) + (i16 >> 2)) * 32 + 16 * (g16 & 1) + 4 * (i16 & 3));      // elements (transposing read, see above)
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x16)(0.f);

  int t = blockIdx.x, set = 0;
  if (t < ntiles) issue(t, 0);
  for (; t < ntiles; t += nsplit, set ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile t have landed ...
    __syncthreads();                                       // ... and so have everyone's; every wave is done with the other set
    if (t + nsplit < ntiles) issue(t + nsplit, set ^ 1);
    if (pair_ok) {
      const egne_bf16* const Gi = wlds + set * WSET + ch_c * (GPX * 32) + lbase;
      const egne_bf16* const Xi = wlds + set * WSET + 2 * GPX * 32 + ch_k * (WX * 32) + lbase;
      // the wave's eight gz fragments (four rows x two 16-pixel k-steps) stay in registers for the tile
      egne_bf16x8 af[4][2];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const egne_bf16* ga = Gi + ((hh * 4 + r) * 32 + 16 * s2) * 32;
          const egne_bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)ga);
          const egne_bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(ga + 4 * 32));
          af[r][s2] = egne_bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        }
      // 36 x fragments: halo row hr (six per wave) x kernel column x k-step, each for the taps dy with tile row r = hr - 1 - dy in 0..3
      auto loadx = [&](int hr, int dx, int s2) {
        const egne_bf16* xa = Xi + ((hh * 4 + hr) * HWd + 16 * s2 + 1 + dx) * 32;
        const egne_bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)xa);
        const egne_bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(xa + 4 * 32));
        return egne_bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      };
      egne_bf16x8 bq = loadx(0, -1, 0);
#pragma unroll
      for (int it = 0; it < 36; ++it) {
        const int hr = it / 6, dx = (it % 6) / 2 - 1, s2 = it & 1;
        const egne_bf16x8 bcur = bq;
        if (it + 1 < 36) bq = loadx((it + 1) / 6, ((it + 1) % 6) / 2 - 1, (it + 1) & 1);
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
          const int r = hr - 1 - dy;
          if (r >= 0 && r < 4) acc[(dy + 1) * 3 + dx + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[r][s2], bcur, acc[(dy + 1) * 3 + dx + 1], 0, 0, 0);
        }
      }
    }
  }
  // the two waves of a pair (rows 0-3 / 4-7) add their blocks through LDS, one tap at a time; the pair's 32 x 32 block of every tap goes to
  // the workgroup's partial: lane holds column k = lane & 31 of rows co = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* const red = (float*)wlds;         // [8 waves][16][64]
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    __syncthreads();
    if (hh == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(pr * 16 + r) * 64 + lane] = acc[tap][r];
    }
    __syncthreads();
    if (hh == 0 && pair_ok) {
      float* dst = ws + ((long long)blockIdx.x * 9 + tap) * (long long)p.CoutP * p.Ktot;
      const int k = ci0 + ch_k * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + ch_c * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (k < p.Ktot && co < p.CoutP) dst[(long long)co * p.Ktot + k] = acc[tap][r] + red[(pr * 16 + r) * 64 + lane];
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// 1x1 / stride 1 weight gradient over raw bf16 slices with every (32 co, 32 k) block pair of the layer in ONE workgroup (the bf16
// form of conv1x1_wgrad_allpairs_kernel, backward.hip): a workgroup walks its pixel range in chunks of CH pixels (64 with up to 16
// tiles, 128 up to 8, 256 up to 4: always 64 KB of loads in flight per workgroup -- with 64-pixel chunks the 4-tile layers of the
// full-resolution dense block had 16 KB in flight between two barriers and ran at 2.2 TB/s), stages all gz
// tiles and all x tiles of the chunk once as [64 px][32 ch] bf16 rows (16-byte loads, 4 neighbouring lanes = the 64 contiguous
// bytes of a pixel's block), and its eight waves share the pairs: wave w owns pairs w, w + 8, ... for ALL pixels -- no cross-wave
// reduction, every HBM byte read once.  Operands by ds_read_b64_tr_b16 as above.  Partials: ws[split][CoutP][Ktot].
// ------------------------------------------------------------------------------------------------------------------------
constexpr int W1_MAXT = 16;      // tiles (gz + x) of a chunk
constexpr int W1_NI = 8;         // 16-byte items per thread and chunk: tiles x CH / 128
struct W1Tab { short seg[W1_MAXT]; short c0[W1_MAXT]; short kofs[W1_MAXT]; };

template <int PPW, int W1_CH>
__global__ __launch_bounds__(512)
void wgrad1x1_bf16_kernel(const egne_conv_desc p, const egne_bf16* __restrict__ gz, long long gzs, int gzo, int nsplit, int nco, int nkc,
                          W1Tab tab, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) egne_bf16 w1lds[];     // [nco + nkc][CH px][32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = nco + nkc, npairs = nco * nkc;
  const long long M = (long long)p.B * p.Ho * p.Wo;
  const long long per = ((M + nsplit - 1) / nsplit + W1_CH - 1) / W1_CH * W1_CH;
  const long long m_begin = (long long)blockIdx.x * per, m_end = m_begin + per < M ? m_begin + per : M;
  const int piece = tid & 3;                                            // item I of a thread: linear index tid + 512 I over (tile, pixel, 8-channel group)
  // Everything about an item but the chunk it is read from is fixed for the launch: its tensor, channel group and pixel within the chunk.
  // Round 5 derived all of it again per item and chunk -- a dynamically indexed egne_seg out of the kernel arguments (scalar loads), a
  // 64-bit product for the chunk's base, a buffer resource -- 16-38 scalar instructions per MFMA (profiles/r05_pmc_kernels_train.txt).
  // Now: one 64-bit address per item, advanced by a constant per chunk; lanes past the range or the tensor read a zero page.
  u32x4 rv[W1_NI];
  const char* ia[W1_NI];          // address of the item in the chunk being requested next
  int istep[W1_NI];               // bytes per chunk of W1_CH pixels (0: the item reads nothing)
#pragma unroll
  for (int I = 0; I < W1_NI; ++I) {
    const int lin = tid + 512 * I, j = lin / (4 * W1_CH), px = (lin % (4 * W1_CH)) >> 2;
    const bool isg = j < nco, on = j < nt;
    const egne_seg& sg = p.seg[(isg || !on) ? 0 : tab.seg[j]];
    const int c = (isg ? 32 * j : (on ? tab.c0[j] : 0)) + 8 * piece;
    const bool cok = on && (isg ? c < p.Cout_store : c < sg.Cp);
    const long long stride = isg ? gzs : sg.pix_stride;
    const egne_bf16* base = isg ? gz : (const egne_bf16*)sg.ptr;
    ia[I] = (const char*)(base + (m_begin + px) * stride + (isg ? gzo : sg.ch_off) + c);
    istep[I] = cok ? (int)(W1_CH * stride * 2) : 0;
  }
  auto issue = [&](long long mc) {
    const int rows = (int)(m_end - mc < W1_CH ? m_end - mc : W1_CH);
#pragma unroll
    for (int I = 0; I < W1_NI; ++I) {
      if (512 * I < nt * 4 * W1_CH) {                 // (uniform; the lanes past the last tile read nothing)
        const int px = ((tid + 512 * I) % (4 * W1_CH)) >> 2;
        const bool ok = istep[I] != 0 && px < rows;   // pixels past the range read zeros
        rv[I] = *(const u32x4*)(ok ? ia[I] : (const char*)egne_zero_page);
        ia[I] += istep[I];
      }
    }
  };
  const int g16 = lane >> 4, i16 = lane & 15;
  const int lbase = ((8 * (g16 >> 1) + (i16 >> 2)) * 32 + 16 * (g16 & 1) + 4 * (i16 & 3));      // elements (transposing read, see above)
  f32x16 acc[PPW];
  int aofs[PPW], bofs[PPW];       // LDS images of this wave's pairs (no division per pair and chunk)
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    acc[i] = (f32x16)(0.f);
    const int q = wave + 8 * i, kc = q / nco, ct = q - kc * nco;
    aofs[i] = (ct * W1_CH) * 32 + lbase;
    bofs[i] = ((nco + kc) * W1_CH) * 32 + lbase;
  }
  if (m_begin < m_end) issue(m_begin);
  for (long long mc = m_begin; mc < m_end; mc += W1_CH) {
    __syncthreads();                 // every wave is done with the previous chunk's tiles
#pragma unroll
    for (int I = 0; I < W1_NI; ++I) {
      const int lin = tid + 512 * I, j = lin / (4 * W1_CH), px = (lin % (4 * W1_CH)) >> 2;
      if (j < nt) *(u32x4*)&w1lds[(j * W1_CH + px) * 32 + piece * 8] = rv[I];
    }
    __syncthreads();
    if (mc + W1_CH < m_end) issue(mc + W1_CH);       // next chunk's loads fly during this chunk's MFMAs
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      if (wave + 8 * i < npairs) {    // (wave-uniform: the transposing reads below run with all lanes enabled)
        const egne_bf16* As = w1lds + aofs[i];
        const egne_bf16* Bs = w1lds + bofs[i];
#pragma unroll
        for (int s = 0; s < W1_CH / 16; ++s) {
          const egne_bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(As + 16 * s * 32));
          const egne_bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(As + (16 * s + 4) * 32));
          const egne_bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(Bs + 16 * s * 32));
          const egne_bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(Bs + (16 * s + 4) * 32));
          const egne_bf16x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          const egne_bf16x8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
      }
    }
  }
  // lane holds column k = lane & 31 of rows co = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of its pairs' blocks
  float* dst = ws + (long long)blockIdx.x * (long long)p.CoutP * p.Ktot;
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int q = wave + 8 * i;
    if (q < npairs) {
      const int kc = q / nco, ct = q - kc * nco;
      const int j = nco + kc;
      const int k = tab.c0[j] + (lane & 31);
      if (k < p.seg[tab.seg[j]].Cp) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          dst[(long long)co * p.Ktot + tab.kofs[j] + k] = acc[i][r];
        }
      }
    }
  }
}

}  // namespace

namespace egne {

bool wgrad3x3_bf16_supported(const egne_conv_desc& d, long long gzs) {
  static const bool off = [] { const char* e = getenv("EGNE_WGRAD_BF16"); return e && e[0] == '0'; }();
  if (off || d.dtype != 1) return false;
  if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad_h != 1 || d.pad_w != 1 || d.pad_mode != 0 || d.ngroups != 1 || d.nseg != 1 ||
      d.dil[0] != 1 || d.H != d.Ho || d.W != d.Wo) return false;
  const egne_seg& g = d.seg[0];
  if (!g.ptr || g.Cp % 8 || g.ch_off % 8 || g.pix_stride % 8 || ((uintptr_t)g.ptr & 15) || gzs % 8 || d.out_ch_off % 8 || d.Cout_store % 8) return false;
  if ((long long)d.H * d.W * g.pix_stride * 2 >= (1ll << 31) || (long long)d.H * d.W * gzs * 2 >= (1ll << 31)) return false;
  if (d.Ktot != g.Cp || d.CoutP % 32 || d.W < 16) return false;
  return true;
}

static int npairs_of(const egne_conv_desc& d) { return (d.CoutP / 32) * ((d.Ktot + 31) / 32); }

// the quad-per-workgroup form: layers with two or more 32-channel blocks on BOTH sides whose input is read raw (no affine on load)
static bool wgrad3x3_wide(const egne_conv_desc& d) {
  static const bool off = [] { const char* e = getenv("EGNE_WGRAD3_WIDE"); return e && e[0] == '0'; }();
  return !off && d.CoutP >= 64 && d.Ktot > 32 && !d.seg[0].scale;
}
static int nquads_of(const egne_conv_desc& d) { return ((d.CoutP / 32 + 1) / 2) * (((d.Ktot + 31) / 32 + 1) / 2); }

int wgrad3x3_bf16_splits(const egne_conv_desc& d) {
  const long long tiles = (long long)((d.W + TW - 1) / TW) * ((d.H + TH - 1) / TH) * d.B;
  long long ns = 256 / (wgrad3x3_wide(d) ? nquads_of(d) : npairs_of(d));          // one 8-wave workgroup per CU
  if (ns < 1) ns = 1;
  if (ns > tiles) ns = tiles;
  return (int)ns;
}

bool wgrad1x1_bf16_supported(const egne_conv_desc& d, long long gzs) {
  static const bool off = [] { const char* e = getenv("EGNE_WGRAD_BF16"); return e && e[0] == '0'; }();
  if (off || d.dtype != 1) return false;
  if (d.kh != 1 || d.kw != 1 || d.stride != 1 || d.pad_h != 0 || d.pad_w != 0 || d.ngroups != 1 || d.H != d.Ho || d.W != d.Wo) return false;
  int nkc = 0;
  for (int s = 0; s < d.nseg; ++s) {
    const egne_seg& g = d.seg[s];
    if (!g.ptr || g.scale || g.act_in != EGNE_ACT_NONE || g.Cp % 8 || g.ch_off % 8 || g.pix_stride % 8 || ((uintptr_t)g.ptr & 15) ||
        g.pix_stride * 256 * 2 >= (1ll << 31)) return false;
    nkc += (g.Cp + 31) / 32;
  }
  const int nco = d.CoutP / 32;
  if (gzs % 8 || d.out_ch_off % 8 || d.Cout_store % 8 || gzs * 256 * 2 >= (1ll << 31)) return false;
  // wide layers (the 1x1 convolutions of the decoder's lower levels: up to 15 channel tiles in, 5 out) run as several launches, each
  // over a group of input tiles with all output tiles: at most 64 block pairs and W1_MAXT staged tiles per launch
  if (nco > 12 || (long long)d.B * d.Ho * d.Wo < 4096) return false;
  (void)nkc;
  return true;
}

// input tiles per launch and the pixel chunk that goes with the staged tile count
static int w1_group(const egne_conv_desc& d) {
  const int nco = d.CoutP / 32;
  int nkc = 0;
  for (int s = 0; s < d.nseg; ++s) nkc += (d.seg[s].Cp + 31) / 32;
  int g = 64 / nco;
  if (g > W1_MAXT - nco) g = W1_MAXT - nco;
  return g < nkc ? g : nkc;
}
static int w1_chunk(const egne_conv_desc& d) { const int nt = d.CoutP / 32 + w1_group(d); return nt <= 4 ? 256 : (nt <= 8 ? 128 : 64); }

int wgrad1x1_bf16_splits(const egne_conv_desc& d) {
  const long long M = (long long)d.B * d.Ho * d.Wo;
  const int ch = w1_chunk(d);
  long long ns = (M + ch * 4 - 1) / (ch * 4);      // at least four chunks per workgroup
  if (ns > 512) ns = 512;
  return (int)(ns < 1 ? 1 : ns);
}

int wgrad1x1_bf16_launch(const egne_conv_desc& d, const egne_bf16* gz, long long gzs, int gzo, float* ws, hipStream_t st) {
  if (gzo % 8 || ((uintptr_t)gz & 15)) return fail(EGNE_ERR_ARG, "wgrad1x1_bf16: gz slice must start on a multiple of 8 channels (offset %d)", gzo);
  const int nco = d.CoutP / 32, nsplit = wgrad1x1_bf16_splits(d), group = w1_group(d), ch = w1_chunk(d);
  // every 32-channel input tile: (slice, first channel, column of the weight gradient)
  short tseg[64], tc0[64], tk[64];
  int nkc = 0, kofs = 0;
  for (int s = 0; s < d.nseg; ++s) {
    for (int c0 = 0; c0 < d.seg[s].Cp; c0 += 32, ++nkc) {
      if (nkc >= 64) return fail(EGNE_ERR_ARG, "wgrad1x1_bf16: more than 64 input tiles");
      tseg[nkc] = (short)s; tc0[nkc] = (short)c0; tk[nkc] = (short)kofs;
    }
    kofs += d.seg[s].Cp;
  }
  // (ws arrives zero-filled -- channels beyond a slice's blocks are not written -- and the reduction clears what it reads)
  for (int k0 = 0; k0 < nkc; k0 += group) {        // one launch per group of input tiles (gz is staged again by each)
    const int nk = nkc - k0 < group ? nkc - k0 : group, ppw = (nco * nk + 7) / 8;
    W1Tab tab{};
    for (int i = 0; i < nk; ++i) { tab.seg[nco + i] = tseg[k0 + i]; tab.c0[nco + i] = tc0[k0 + i]; tab.kofs[nco + i] = tk[k0 + i]; }
    const size_t lds = (size_t)(nco + nk) * ch * 32 * sizeof(egne_bf16);
    auto go = [&](auto kern) -> int {
      const bool raised = egne::raise_lds((const void*)kern, 64 * 1024);
      if (!raised) return fail(EGNE_ERR_LAUNCH, "wgrad1x1_bf16: cannot raise the dynamic LDS limit");
      hipLaunchKernelGGL(kern, dim3(nsplit), dim3(512), lds, st, d, gz, gzs, gzo, nsplit, nco, nk, tab, ws);
      return check_launch("egne_conv2d_wgrad (1x1, bf16)");
    };
    int rc;
    if (ch == 256) rc = go(wgrad1x1_bf16_kernel<1, 256>);                               // <= 4 tiles: <= 4 pairs
    else if (ch == 128) rc = ppw <= 1 ? go(wgrad1x1_bf16_kernel<1, 128>) : go(wgrad1x1_bf16_kernel<2, 128>);     // <= 8 tiles: <= 16 pairs
    else rc = ppw <= 2 ? go(wgrad1x1_bf16_kernel<2, 64>) : ppw <= 4 ? go(wgrad1x1_bf16_kernel<4, 64>) : go(wgrad1x1_bf16_kernel<8, 64>);
    if (rc != 0) return rc;
  }
  return 0;
}

int wgrad3x3_bf16_launch(const egne_conv_desc& d, const egne_bf16* gz, long long gzs, int gzo, float* ws, hipStream_t st) {
  if (gzo % 8 || ((uintptr_t)gz & 15)) return fail(EGNE_ERR_ARG, "wgrad3x3_bf16: gz slice must start on a multiple of 8 channels (offset %d)", gzo);
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH, ntiles = tiles_x * tiles_y * d.B;
  const int nsplit = wgrad3x3_bf16_splits(d), nco = d.CoutP / 32;
  if (wgrad3x3_wide(d)) {
    const int nkc = (d.Ktot + 31) / 32, nqco = (nco + 1) / 2;
    const size_t lds = (size_t)2 * WSET * sizeof(egne_bf16);
    if (!raise_lds((const void*)wgrad3x3_bf16_wide_kernel, lds)) return fail(EGNE_ERR_LAUNCH, "wgrad3x3_bf16 (wide): cannot raise the dynamic LDS limit to %zu", lds);
    hipLaunchKernelGGL(wgrad3x3_bf16_wide_kernel, dim3(nsplit, nquads_of(d)), dim3(NT), lds, st, d, gz, gzs, gzo, nsplit, nqco, nco, nkc, tiles_x, tiles_y,
                       ntiles, ws);
    return check_launch("egne_conv2d_wgrad (3x3, bf16, wide)");
  }
  hipLaunchKernelGGL(wgrad3x3_bf16_kernel, dim3(nsplit, npairs_of(d)), dim3(NT), 0, st, d, gz, gzs, gzo, nsplit, nco, tiles_x, tiles_y, ntiles, ws);
  return check_launch("egne_conv2d_wgrad (3x3, bf16)");
}

}  // namespace egne
