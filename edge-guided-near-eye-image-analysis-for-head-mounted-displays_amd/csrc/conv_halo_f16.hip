// 3x3 "same" convolution with an LDS-resident halo tile on the split-f16 MFMA path (fp32 tensors, three
// v_mfma_f32_32x32x16_f16 per product, fp32 accumulate; see conv_f16x3.hip for the numerics).
//
// With the matrix work 5x cheaper, the narrow (Cout = 32 / 64) full-resolution layers of ESF-Net and of the
// BDCN MSBlocks are bound by re-gathering their input 9 times from L2.  As in conv_halo.hip a workgroup owns
// an 8 x 32 block of output pixels: the (8+2d) x (32+2d) halo is fetched ONCE per 32 channels, converted to
// hi/lo halves while it is written to LDS (80-B pixel pitch, conflict-free ds_read_b128), and the 9 taps
// are address offsets into it.  Weight fragments ([tap][k/16][n/32][lane][8 halfs], hi and lo) come from L2
// through a 4-slot register ring; workgroups walk tiles grid-stride with the next halo prefetched.
#include "common.h"
#include "split_f16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KC = 32, LDH = 40, TW = 32;

__device__ __forceinline__ float act1(float v, int act) {
  if (act == EGNE_ACT_RELU) return fmaxf(v, 0.f);
  if (act == EGNE_ACT_LEAKY) return v > 0.f ? v : 0.01f * v;
  return v;
}

// LAT = true: "lattice" mode for large dilations.  A dilation-S conv is S*S independent ordinary 3x3 convs on
// the sub-lattices (phase_y + S*i, phase_x + S*j); a tile is 8 x 32 LATTICE points, its halo is one lattice
// step wide (D = 1), so a dilation-12 conv stages the same 340-pixel halo as a dilation-1 conv.  Pixels are
// 128-B channel vectors, so the strided gather still moves full cache lines.
//
// Instruction budget: the matrix work of a tile is only 54 MFMAs per 32x32 block, so every VALU / SALU
// instruction around it counts (PMC: the first version issued 12 VALU + 6 SALU per MFMA and ran VALU-bound).
// All global traffic therefore goes through BUFFER instructions on per-frame resources: the halo gather, the
// weight ring, the residual and the output use 32-bit byte offsets that are tile-independent per lane plus one
// scalar per tile, and out-of-image / padded lanes carry the offset 0x80000000, which the range check of the
// buffer unit turns into a zero load or a dropped store -- no per-lane 64-bit address math and no selects.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// NW = waves along N: <WM=2, WN, NW=1> gives every wave 2 tile rows x all 32*WN channels; <WM=4, WN=1, NW=2> gives a
// wave 4 tile rows x 32 of the 64 channels -- same accumulators, but each weight fragment fetched from L1/L2 feeds
// twice as many MFMAs (PMC: the weight ring of the <2,2,1> shape ran the vector L1 at 80 % of its 64 B/clk).
// NP: products per multiply (egne_conv_desc.f16_products): 3 = hi hi + hi lo + lo hi; 1 = hi hi only (plain f16 operands: no lo halves
// derived, stored, fetched or multiplied -- half the weight fragments each wave pulls from L2)
template <int WM, int WN, int D, bool LAT, int NW, int PF = 0, bool TP = false, bool POOL = false, int NP = 3>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_f16_kernel(const egne_conv_desc p, const _Float16* __restrict__ fhi,
                                                               const _Float16* __restrict__ flo, float a_scale,
                                                               float out_scale, int tiles_x, int tiles_y, int ntiles) {
  egne::dyn_scales(p.dyn_scale, a_scale, out_scale);
  // (vH, vW, rstep, cstep): the image as the kernel walks it.  Normal: (H, W, W, 1).  TP = transposed ("tall tiles", for maps
  // that 8 x 32 tiles fit badly): (W, H, 1, W) -- the 32-long MFMA rows then run along image y.  A template parameter: as
  // run-time values the extra address arithmetic cost the normal path 5 %.
  const int vH = TP ? p.W : p.H, vW = TP ? p.H : p.W, rstep = TP ? 1 : p.W, cstep = TP ? p.W : 1;
  constexpr int TH = (4 / NW) * WM;
  constexpr int d = D;
  constexpr int HWd = TW + 2 * d, HHd = TH + 2 * d, npx = HHd * HWd;
  constexpr int nitems = npx * 8;
  constexpr int NI = (nitems + 255) / 256;
  static_assert(HWd >= 32, "one wrap per 32-pixel step");
  extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
  _Float16* Ahi = ldsh;
  _Float16* Alo = ldsh + npx * LDH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wrow = wave / NW;                // wave's row group inside the tile
  const int nt0 = (blockIdx.y * NW + wave % NW) * WN;
  const int NT = p.CoutP >> 5, KT16 = p.Ktot >> 4;
  const egne_seg sg = p.seg[0];
  const int Cp = sg.Cp;
  const int c4 = tid & 7;
  const int S = LAT ? p.dil[0] : 1;          // lattice step
  const unsigned frame_in = (unsigned)p.H * p.W * (unsigned)sg.pix_stride * 4u;
  const unsigned frame_out = (unsigned)p.H * p.W * (unsigned)p.out_pix_stride * 4u;
  const unsigned frame_res = (unsigned)p.H * p.W * (unsigned)p.res_pix_stride * 4u;

  // tile-independent per-item constants: halo coordinates (in image pixels, lattice step applied) and byte offset
  int hyx[NI], roff[NI];
  {
    int px = tid >> 3;
    int hy = px / HWd, hx = px - hy * HWd;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bool in = tid + 256 * i < nitems;
      hyx[i] = in ? ((S * hy) << 16) | (S * hx) : 0x7fff7fff;
      roff[i] = ((S * hy * rstep + S * hx * cstep) * (int)sg.pix_stride + c4 * 4) * 4;
      hx += 32;
      if (hx >= HWd) { hx -= HWd; ++hy; }
    }
  }
  const int lofs0 = (tid >> 3) * LDH + c4 * 4;   // LDS slot of item 0; item i is 32 pixels further

  struct Tile { int b, y0, x0, py, px; };    // y0/x0 in lattice units, (py, px) = lattice phase
  auto tile_of = [&](int t) {
    Tile r;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    r.py = 0; r.px = 0;
    if (LAT) { r.px = t % S; t /= S; r.py = t % S; t /= S; }
    r.b = t; r.y0 = ty * TH; r.x0 = tx * TW;
    return r;
  };

  unsigned goff[NI];
  __amdgpu_buffer_rsrc_t rin = make_rsrc(sg.ptr, 0);
  int stage_b = 0;
  auto map_tile = [&](const Tile& tl) {
    const int ybase = tl.py + S * (tl.y0 - d), xbase = tl.px + S * (tl.x0 - d);
    const int tbase = ((ybase * rstep + xbase * cstep) * (int)sg.pix_stride + sg.ch_off) * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned iy = (unsigned)(ybase + (hyx[i] >> 16)), ix = (unsigned)(xbase + (hyx[i] & 0xffff));
      goff[i] = (iy < (unsigned)vH && ix < (unsigned)vW) ? (unsigned)(tbase + roff[i]) : OOB;
    }
    rin = make_rsrc(sg.ptr + (long long)tl.b * p.H * p.W * sg.pix_stride, frame_in);
    stage_b = tl.b;
  };

  u32x4 st[NI];
  f32x4 st_sc = {1.f, 1.f, 1.f, 1.f}, st_sh = {0.f, 0.f, 0.f, 0.f};
  unsigned st_cmask = 0;
  auto load_chunk = [&](int c0) {
    const bool cok = c0 + c4 * 4 < Cp;
    st_cmask = cok ? 0u : OOB;
    if (sg.scale) {
      st_sc = *(const f32x4*)(cok ? sg.scale + (long long)stage_b * Cp + c0 + c4 * 4 : egne_zero_page);
      st_sh = *(const f32x4*)(cok ? sg.shift + (long long)stage_b * Cp + c0 + c4 * 4 : egne_zero_page);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) st[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)(goff[i] | st_cmask), c0 * 4, 0);
  };
  // activations as max(v, slope*v): slope 0 = ReLU, 0.01 = LeakyReLU, 1 = identity (no per-element branches)
  const float slope_in = sg.act_in == EGNE_ACT_RELU ? 0.f : (sg.act_in == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  const float slope_out = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  auto store_chunk = [&]() {
    if (sg.scale) {     // fused InstanceNorm affine (+ activation) of the consumer; zero padding applied after it
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        f32x4 v = __builtin_bit_cast(f32x4, st[i]) * st_sc + st_sh;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope_in);
        if ((goff[i] | st_cmask) & OOB) v = (f32x4)(0.f);
        st[i] = __builtin_bit_cast(u32x4, v);
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (i < NI - 1 || tid + 256 * i < nitems) {
        const f32x4 v = __builtin_bit_cast(f32x4, st[i]);
        // x*a_scale = hi + lo, two elements per (packed) instruction
        const int o = lofs0 + i * 32 * LDH;
        if constexpr (NP == 1) {
          float t0, t1, t2, t3;
          asm("v_mul_f32_e32 %0, %1, %2" : "=v"(t0) : "s"(a_scale), "v"(v[0]));
          asm("v_mul_f32_e32 %0, %1, %2" : "=v"(t1) : "s"(a_scale), "v"(v[1]));
          asm("v_mul_f32_e32 %0, %1, %2" : "=v"(t2) : "s"(a_scale), "v"(v[2]));
          asm("v_mul_f32_e32 %0, %1, %2" : "=v"(t3) : "s"(a_scale), "v"(v[3]));
          const egne::sp_f32x2 u0 = {t0, t1}, u1 = {t2, t3};
          const h2 h0 = __builtin_convertvector(u0, h2), h1 = __builtin_convertvector(u1, h2);
          const h4 hi = {h0[0], h0[1], h1[0], h1[1]};
          *(h4*)&Ahi[o] = hi;
          continue;
        }
        h2 h0, h1, l0, l1;                        // x * a_scale = hi + lo, plain (unpacked) VALU: split_f16.h
        egne::split2(v[0], v[1], a_scale, h0, l0);
        egne::split2(v[2], v[3], a_scale, h1, l1);
        const h4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
        *(h4*)&Ahi[o] = hi;
        *(h4*)&Alo[o] = lo;
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int n = 0; n < WN; ++n) acc[a][n] = (f32x16)(0.f);

  // fragment-order weights: element ((tap*KT16 + k16)*NT + nt)*512 + lane*8 (halfs); byte offsets below
  const unsigned wbytes = 9u * (unsigned)p.Ktot * (unsigned)p.CoutP * 2u;
  const __amdgpu_buffer_rsrc_t rwh = make_rsrc(fhi, wbytes), rwl = make_rsrc(flo, wbytes);
  const int stride_k16 = NT * 1024, stride_tap = KT16 * NT * 1024;
  // transposed walk: the LDS tap (ky, kx) is the image tap (kx, ky) -> fetch the transposed 3x3 weight
  auto wtap = [&](int t) { return TP ? (t % 3) * 3 + t / 3 : t; };
  const int wlane = lane * 16;

  // epilogue constants: lane -> channel n, first x of the lane inside a tile row
  const int out_step = S * cstep * (int)p.out_pix_stride * 4, res_step = S * cstep * (int)p.res_pix_stride * 4;

  int t = blockIdx.x;
  if (t >= ntiles) return;
  Tile cur = tile_of(t);
  map_tile(cur);
  load_chunk(0);
  int c0 = 0;
  const int abase = (wrow * WM * HWd + li) * LDH + lh * 8;
  while (true) {
    __syncthreads();
    store_chunk();
    __syncthreads();
    const bool last_chunk = c0 + KC >= Cp;
    const int tnext = t + gridDim.x;
    // PF: where the halo prefetch of the next (tile, chunk) is issued.  vmcnt retires in order, so a prefetch in
    // front of the weight ring (PF 0) makes the ring's first wait cover the HBM latency of the halo as well.
    auto prefetch = [&]() {
      if (!last_chunk) {
        load_chunk(c0 + KC);
      } else if (tnext < ntiles) {
        const Tile nx = tile_of(tnext);
        map_tile(nx);
        load_chunk(0);
      }
    };
    if (PF == 0) prefetch();
    const int wchunk = nt0 * 1024 + (c0 >> 4) * stride_k16;
    // register ring: slot = (tap % RT) * 2 + ks holds the fragments of (tap, ks); refilled RT taps ahead
    constexpr int RT = (WN == 1 && NW == 1) ? 2 : 1;
    // a last chunk with <= 16 real channels (38 -> 64, 76 -> 96, 100 -> 100: ESF-Net's growth 1.2 leaves such tails): its second
    // 16-channel k-step multiplies zeros by zeros -- skipped, fragments and MFMAs alike (wave-uniform)
    const bool half = Cp - c0 <= 16;
    u32x4 qh[2 * RT][WN], ql[2 * RT][WN];
#pragma unroll
    for (int s = 0; s < 2 * RT; ++s)
#pragma unroll
      for (int tn = 0; tn < WN; ++tn) {
        const int o = wchunk + wtap(s >> 1) * stride_tap + (s & 1) * stride_k16 + tn * 1024;
        const int wl = (half && (s & 1)) ? (int)OOB : wlane;
        qh[s][tn] = __builtin_amdgcn_raw_buffer_load_b128(rwh, wl, o, 0);
        if constexpr (NP == 3) ql[s][tn] = __builtin_amdgcn_raw_buffer_load_b128(rwl, wl, o, 0);
      }
    constexpr int TAP_UNROLL = NW == 2 ? 1 : 9;   // the NW = 2 shape only fits 2 waves per SIMD with the tap loop rolled
#pragma unroll TAP_UNROLL
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int aoff = abase + (ky * d * HWd + kx * d) * LDH;
      if (PF == 1 && tap == 1) prefetch();      // behind the ring's first refills
      if (PF == 2 && tap == 4) prefetch();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks == 1 && half) continue;
        const int slot = (tap % RT) * 2 + ks;
        h8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
        for (int tm = 0; tm < WM; ++tm) {
          ah[tm] = *(const h8*)&Ahi[aoff + tm * HWd * LDH + ks * 16];
          if constexpr (NP == 3) al[tm] = *(const h8*)&Alo[aoff + tm * HWd * LDH + ks * 16];
        }
#pragma unroll
        for (int tn = 0; tn < WN; ++tn) {
          bh[tn] = __builtin_bit_cast(h8, qh[slot][tn]);
          if constexpr (NP == 3) bl[tn] = __builtin_bit_cast(h8, ql[slot][tn]);
        }
        if (tap + RT < 9) {
#pragma unroll
          for (int tn = 0; tn < WN; ++tn) {
            const int o = wchunk + wtap(tap + RT) * stride_tap + ks * stride_k16 + tn * 1024;
            qh[slot][tn] = __builtin_amdgcn_raw_buffer_load_b128(rwh, wlane, o, 0);
            if constexpr (NP == 3) ql[slot][tn] = __builtin_amdgcn_raw_buffer_load_b128(rwl, wlane, o, 0);
          }
        }
#pragma unroll
        for (int tm = 0; tm < WM; ++tm)
#pragma unroll
          for (int tn = 0; tn < WN; ++tn) {
            if constexpr (NP == 3) {
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
            }
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
          }
      }
    }
    if (!last_chunk) { c0 += KC; continue; }

    // ---- epilogue: lane holds channel n of 16 pixels x = x_lane + S*c_r, c_r = (r&3) + 8*(r>>2), of tile row tm ----
    {
      const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out + (long long)cur.b * p.H * p.W * p.out_pix_stride, frame_out);
      const __amdgpu_buffer_rsrc_t rres =
          make_rsrc(p.residual ? p.residual + (long long)cur.b * p.H * p.W * p.res_pix_stride : nullptr, p.residual ? frame_res : 0u);
      const int xl = cur.px + S * (cur.x0 + 4 * lh);
      const int cmax = xl < vW ? (vW - xl + S - 1) / S : 0;      // c_r < cmax  <=>  x < W
#pragma unroll
      for (int tn = 0; tn < WN; ++tn) {
        const int n = (nt0 + tn) * 32 + li;
        const bool nok = n < p.Cout_store;
        const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
        float ps = 1.f, pt = 0.f;
        if (p.post_scale && nok) { ps = p.post_scale[n]; pt = p.post_shift[n]; }
        double st_s = 0., st_q = 0.;         // sum / sum of squares of this wave's stored values of channel n (stats_ws)
        [[maybe_unused]] float pool_keep[16];   // (POOL) stored values of the wave's first row, -inf where nothing was stored
#pragma unroll
        for (int tm = 0; tm < WM; ++tm) {
          const int y = cur.py + S * (cur.y0 + wrow * WM + tm);
          const int cm = (nok && y < vH) ? cmax : 0;
          const int pix = y * rstep + xl * cstep;
          const unsigned o0 = (unsigned)((pix * (int)p.out_pix_stride + p.out_ch_off + n) * 4);
          float rv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = 0.f;
          if (p.residual) {     // all residual loads of the tile row first (one wait), then the stores
            const unsigned r0 = (unsigned)((pix * (int)p.res_pix_stride + p.res_ch_off + n) * 4);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int c = (r & 3) + 8 * (r >> 2);
              rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, (int)(c < cm ? r0 + c * res_step : OOB), 0, 0));
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2);
            float v = acc[tm][tn][r] * out_scale + bv;
            v = fmaxf(v, v * slope_out) * ps + pt + rv[r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, (int)(c < cm ? o0 + c * out_step : OOB), 0, 0);
            const double vm = c < cm ? (double)v : 0.;
            st_s += vm; st_q += vm * vm;
            if constexpr (POOL) {
              // second output: 2x2 / stride-2 ceil-mode max pooling of the STORED values (vgg16_c.py:70-78 pool behind the convolution).
              // A wave owns the row pair (y, y + 1), y even, and a lane the pixel pairs (x, x + 1), x even, of its channel: the
              // four values of a window sit in this lane's registers of the two rows.
              const float vp = c < cm ? v : -INFINITY;
              if (tm == 0) pool_keep[r] = vp;
              else pool_keep[r] = fmaxf(pool_keep[r], vp);
            }
          }
          acc[tm][tn] = (f32x16)(0.f);
        }
        if constexpr (POOL) {
          if (p.pool_out) {
            const int y = cur.y0 + wrow * WM, Hp = (p.H + 1) >> 1, Wp = (p.W + 1) >> 1;
            const __amdgpu_buffer_rsrc_t rpool = make_rsrc(p.pool_out + (long long)cur.b * Hp * Wp * p.pool_pix_stride,
                                                           (unsigned)Hp * Wp * (unsigned)p.pool_pix_stride * 4u);
            const int xl0 = cur.x0 + 4 * lh;
            const unsigned q0 = (unsigned)((((y >> 1) * Wp + (xl0 >> 1)) * (int)p.pool_pix_stride + p.pool_ch_off + n) * 4);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const int c = (r & 3) + 8 * (r >> 2);
              const float m = fmaxf(pool_keep[r], pool_keep[r + 1]);
              const bool ok = nok && y < p.H && xl0 + c < p.W;           // the window's first pixel exists (ceil mode clips the rest)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), rpool, (int)(ok ? q0 + (c >> 1) * (int)p.pool_pix_stride * 4 : OOB), 0, 0);
            }
          }
        }
        // (egne_conv_desc.ovf_flag is NOT tested here: this kernel sits at its register limit -- any form of the test, per value, per
        //  selected row or once per tile on the fp64 sum of squares, added 50+ spill instructions and 4-6 % to its launches.  A value
        //  it stores non-finite is caught one launch later: every tensor this kernel writes in the two networks is read raw by a
        //  kernel that does test -- the trunk / MSBlock / dense-block convolution behind it -- or enters InstanceNorm statistics,
        //  which turn the whole tensor NaN for those; engine.Plan.overflowed reads the word after the run either way.)
        if (!LAT && NW == 1 && p.stats_ws) {     // one chunk = this wave's rows of this tile (fixed order: deterministic; TP: tiles of the transposed walk)
          st_s += __shfl_xor(st_s, 32); st_q += __shfl_xor(st_q, 32);
          if (lh == 0 && n < p.Cout_store) {
            const int tile_in_frame = (cur.y0 / TH) * tiles_x + cur.x0 / TW;
            double2* w = (double2*)p.stats_ws + ((long long)cur.b * p.stats_nchunk + tile_in_frame * 4 + wave) * p.Cout_store + n;
            *w = make_double2(st_s, st_q);
          }
        }
      }
    }
    t = tnext;
    if (t >= ntiles) break;
    cur = tile_of(t);
    c0 = 0;
  }
}

// OIHW fp32 -> hi / lo f16 in fragment order [tap][Ktot/16][CoutP/32][lane = h*32 + n%32][8]: k = 16*k16 + 8*h + j
__global__ void pack_weight_f16frag_k(const float* __restrict__ w, int Cout, int Cin, int T, int CoutP, int Ktot, float wscale,
                                      _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
  const long long total = (long long)T * CoutP * Ktot;
  const int NT = CoutP >> 5;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), nn = (int)((i >> 3) & 31), h = (int)((i >> 8) & 1);
    long long q = i >> 9;
    const int nt = (int)(q % NT); q /= NT;
    const int k16 = (int)(q % (Ktot >> 4));
    const int t = (int)(q / (Ktot >> 4));
    const int n = nt * 32 + nn, k = k16 * 16 + h * 8 + j;
    const float v = (n < Cout && k < Cin) ? w[((long long)n * Cin + k) * T + t] * wscale : 0.f;
    const _Float16 hh = (_Float16)v;
    hi[i] = hh;
    lo[i] = (_Float16)(v - (float)hh);
  }
}

template <int WM, int WN, int D, bool LAT, int NW = 1, int PF = 0, bool TP = false, int NP = 3>
int launch_hf_tp(const egne_conv_desc& d, const _Float16* fhi, const _Float16* flo, float a_scale, float os, hipStream_t st) {
  constexpr int TH = (4 / NW) * WM;
  const int S = LAT ? d.dil[0] : 1;
  const int vH = TP ? d.W : d.H, vW = TP ? d.H : d.W;
  const int lw = (vW + S - 1) / S, lh_ = (vH + S - 1) / S;     // lattice extent (largest phase)
  const int tiles_x = (lw + TW - 1) / TW, tiles_y = (lh_ + TH - 1) / TH;
  const size_t lds = (size_t)2 * (TH + 2 * D) * (TW + 2 * D) * LDH * sizeof(_Float16);
  const int ntiles = tiles_x * tiles_y * d.B * S * S, ny = d.CoutP / (32 * WN * NW);
  int gx = (256 * 2 + ny - 1) / ny;
  if (gx > ntiles) gx = ntiles;
  hipLaunchKernelGGL((conv3x3_halo_f16_kernel<WM, WN, D, LAT, NW, PF, TP, false, NP>), dim3(gx, ny), dim3(256), lds, st, d, fhi, flo, a_scale, os, tiles_x,
                     tiles_y, ntiles);
  return egne::check_launch("egne_conv3x3_halo_f16_fwd");
}

// Walk the image transposed when 8 x 32 tiles fit it better that way (dilation 8 on 240x320: 30 x 40 lattice points per
// phase = 8 wide tiles at 59 % fill or 5 tall tiles at 94 %; plain 60x80 and 30x40 maps likewise).  D = 1 shapes only.
template <int WM, int WN, int D, bool LAT, int NW = 1, int PF = 0, int NP = 3>
int launch_hf(const egne_conv_desc& d, const _Float16* fhi, const _Float16* flo, float a_scale, float os, hipStream_t st) {
  constexpr int TH = (4 / NW) * WM;
  const int S = LAT ? d.dil[0] : 1;
  auto ntile = [&](int vh, int vw) { return ((((vw + S - 1) / S) + TW - 1) / TW) * ((((vh + S - 1) / S) + TH - 1) / TH); };
  static const bool tall_ok = [] { const char* e = getenv("EGNE_SHALO_TALL"); return !e || e[0] != '0'; }();
  if (D == 1 && NW == 1 && PF == 0 && tall_ok && ntile(d.W, d.H) < ntile(d.H, d.W))
    return launch_hf_tp<WM, WN, (D == 1 && NW == 1 && PF == 0 ? D : 1), LAT, (D == 1 && NW == 1 && PF == 0 ? NW : 1), 0, true, NP>(d, fhi, flo, a_scale, os, st);
  return launch_hf_tp<WM, WN, D, LAT, NW, PF, false, NP>(d, fhi, flo, a_scale, os, st);
}

}  // namespace

extern "C" int egne_pack_conv_weight_f16frag(const float* w_oihw, int Cout, int Cin, int kh, int kw, int CoutP, int Ktot,
                                             float wscale, void* fhi, void* flo, void* stream) {
  EGNE_REQUIRE(w_oihw && fhi && flo && Cout > 0 && Cin > 0 && CoutP >= Cout && CoutP % 32 == 0 && Ktot >= Cin && Ktot % 32 == 0,
               "pack_f16frag: bad sizes Cout %d Cin %d CoutP %d Ktot %d", Cout, Cin, CoutP, Ktot);
  long long total = (long long)kh * kw * CoutP * Ktot, g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pack_weight_f16frag_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, kh * kw, CoutP,
                     Ktot, wscale, (_Float16*)fhi, (_Float16*)flo);
  return egne::check_launch("egne_pack_conv_weight_f16frag");
}

// 3x3 / stride 1 / pad 1 / dilation 1-2 / one input slice (fused affine allowed) / CoutP 32 or 64; Ktot = slice
// width rounded up to 32; weights in the fragment order of egne_pack_conv_weight_f16frag.
extern "C" int egne_conv3x3_halo_f16_fwd(const egne_conv_desc* dp, const void* fhi, const void* flo, float a_scale,
                                         float w_scale, void* stream) {
  EGNE_REQUIRE(dp && fhi && flo, "conv_halo_f16: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad_mode == 0 && d.ngroups == 1 && d.nseg == 1 && d.pad_h == 1 &&
               d.pad_w == 1 && d.dil[0] >= 1 && d.dil[0] <= 32 && d.Ho == d.H && d.Wo == d.W, "conv_halo_f16: geometry not supported");
  const egne_seg& g = d.seg[0];
  EGNE_REQUIRE(g.ptr && g.Cp % 8 == 0 && (g.Cp + 31) / 32 * 32 == d.Ktot && g.ch_off % 4 == 0 && g.pix_stride % 4 == 0 &&
               ((uintptr_t)g.ptr & 15) == 0 && (g.scale == nullptr) == (g.shift == nullptr), "conv_halo_f16: input slice");
  EGNE_REQUIRE(d.CoutP % 32 == 0 && d.Cout_store <= d.CoutP && d.out && d.out_ch_off + d.Cout_store <= d.out_pix_stride,
               "conv_halo_f16: CoutP %d", d.CoutP);
  EGNE_REQUIRE(((uintptr_t)fhi & 15) == 0 && ((uintptr_t)flo & 15) == 0 && a_scale > 0.f && w_scale > 0.f, "conv_halo_f16: weights / scales");
  {
    // tiles of the walk the launcher will choose (transposed when that takes fewer 8 x 32 tiles: launch_hf), four chunks per tile
    const int t_n = ((d.W + 31) / 32) * ((d.H + 7) / 8), t_t = ((d.H + 31) / 32) * ((d.W + 7) / 8);
    static const bool tall_on = [] { const char* e = getenv("EGNE_SHALO_TALL"); return !e || e[0] != '0'; }();
    const int tiles = (tall_on && t_t < t_n) ? t_t : t_n;
    EGNE_REQUIRE(!d.stats_ws || (d.dil[0] == 1 && ((uintptr_t)d.stats_ws & 15) == 0 && d.stats_nchunk == tiles * 4),
                 "conv_halo_f16: stats_ws needs dilation 1 and stats_nchunk = tiles * 4 (%d tiles per frame)", tiles);
  }
  EGNE_REQUIRE((long long)d.H * d.W * g.pix_stride * 4 < (1ll << 31) && (long long)d.H * d.W * d.out_pix_stride * 4 < (1ll << 31) &&
               (!d.residual || (long long)d.H * d.W * d.res_pix_stride * 4 < (1ll << 31)), "conv_halo_f16: frame too large for 32-bit byte offsets");
  const float os = 1.0f / (a_scale * w_scale);
  hipStream_t st = (hipStream_t)stream;
  const _Float16* h = (const _Float16*)fhi;
  const _Float16* l = (const _Float16*)flo;
  const bool w2 = d.CoutP % 64 == 0;   // wider layers: several 64-wide N tiles along grid.y, each re-stages the halo
  // opt-in: measured equal to the <2,2,1> shape (the rolled tap loop gives back what the halved weight traffic gains)
  static const int nw2 = [] { const char* e = getenv("EGNE_SHALO_NW2"); return e ? atoi(e) : 0; }();
  if (nw2 && w2) {
    if (d.dil[0] == 1) return launch_hf<4, 1, 1, false, 2>(d, h, l, a_scale, os, st);
    if (d.dil[0] == 2) return launch_hf<4, 1, 2, false, 2>(d, h, l, a_scale, os, st);
    return launch_hf<4, 1, 1, true, 2>(d, h, l, a_scale, os, st);
  }
  if (d.pool_out) {
    // second output (2x2 / stride-2 ceil-mode max pooling of the stored values): plain walk, dilation 1, a monotonic activation, no post
    // affine / residual / statistics -- the engine asks for it only then
    EGNE_REQUIRE(d.dil[0] == 1 && !d.post_scale && !d.residual && !d.stats_ws && ((uintptr_t)d.pool_out & 3) == 0 &&
                 d.pool_ch_off + d.Cout_store <= d.pool_pix_stride &&
                 (long long)((d.H + 1) / 2) * ((d.W + 1) / 2) * d.pool_pix_stride * 4 < (1ll << 31), "conv_halo_f16: pooled second output");
    constexpr size_t lds = (size_t)2 * (8 + 2) * (TW + 2) * LDH * sizeof(_Float16);
    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + 7) / 8, ntiles = tiles_x * tiles_y * d.B;
    const int ny = d.CoutP / (w2 ? 64 : 32);
    int gx = (256 * 2 + ny - 1) / ny;
    if (gx > ntiles) gx = ntiles;
    if (w2 && d.f16_products == 1) hipLaunchKernelGGL((conv3x3_halo_f16_kernel<2, 2, 1, false, 1, 0, false, true, 1>), dim3(gx, ny), dim3(256), lds, st, d, h, l, a_scale, os, tiles_x, tiles_y, ntiles);
    else if (w2) hipLaunchKernelGGL((conv3x3_halo_f16_kernel<2, 2, 1, false, 1, 0, false, true>), dim3(gx, ny), dim3(256), lds, st, d, h, l, a_scale, os, tiles_x, tiles_y, ntiles);
    else hipLaunchKernelGGL((conv3x3_halo_f16_kernel<2, 1, 1, false, 1, 0, false, true>), dim3(gx, ny), dim3(256), lds, st, d, h, l, a_scale, os, tiles_x, tiles_y, ntiles);
    return egne::check_launch("egne_conv3x3_halo_f16_fwd");
  }
  static const int pf = [] { const char* e = getenv("EGNE_SHALO_PF"); return e ? atoi(e) : 0; }();
  if (d.dil[0] == 1 && pf == 1) return w2 ? launch_hf<2, 2, 1, false, 1, 1>(d, h, l, a_scale, os, st) : launch_hf<2, 1, 1, false, 1, 1>(d, h, l, a_scale, os, st);
  if (d.dil[0] == 1 && pf == 2) return w2 ? launch_hf<2, 2, 1, false, 1, 2>(d, h, l, a_scale, os, st) : launch_hf<2, 1, 1, false, 1, 2>(d, h, l, a_scale, os, st);
  // plain f16 operands (egne_conv_desc.f16_products = 1): the dilation-1 shapes of the edge network (MSBlock convolutions of stages 3-5, conv2_2)
  if (d.dil[0] == 1 && d.f16_products == 1) return w2 ? launch_hf<2, 2, 1, false, 1, 0, 1>(d, h, l, a_scale, os, st) : launch_hf<2, 1, 1, false, 1, 0, 1>(d, h, l, a_scale, os, st);
  if (d.dil[0] == 1) return w2 ? launch_hf<2, 2, 1, false>(d, h, l, a_scale, os, st) : launch_hf<2, 1, 1, false>(d, h, l, a_scale, os, st);
  if (d.dil[0] == 2) return w2 ? launch_hf<2, 2, 2, false>(d, h, l, a_scale, os, st) : launch_hf<2, 1, 2, false>(d, h, l, a_scale, os, st);
  // larger dilations: lattice mode (the dilation-S conv as S*S ordinary convs on sub-lattices)
  return w2 ? launch_hf<2, 2, 1, true>(d, h, l, a_scale, os, st) : launch_hf<2, 1, 1, true>(d, h, l, a_scale, os, st);
}
