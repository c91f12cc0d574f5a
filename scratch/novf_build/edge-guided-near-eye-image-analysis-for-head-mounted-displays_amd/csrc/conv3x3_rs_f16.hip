// 3x3 "same" convolution on the split-f16 MFMA path with FIXED WAVE ROLES (fp32 tensors, three v_mfma_f32_32x32x16_f16
// per product, fp32 accumulate; numerics: conv_f16x3.hip): the narrow-input layers of the pipeline -- vgg16_c.py:66-69
// (conv1_2, conv2_1), bdcn_new.py:50 (MSBlock conv of stage 1), models/RITnet_v2.py:57 (down-block conv1 behind its
// InstanceNorm), utils.py:1047-1048 (decoder convBlock) -- i.e. one input slice of <= 64 channels, 32 / 64 / 128 outputs.
//
// conv_halo_f16.hip runs staging (fp32 -> hi / lo conversion, ~4 VALU per MFMA) and the matrix work in the SAME waves and
// reaches ~40 % of the split-f16 MFMA rate on these layers.  Here a workgroup is 8 waves, two LDS halo images:
//   producers (waves 0-3)  gather the (TH+2) x 34 halo of tile i+1 (16 bytes per lane, eight or sixteen lanes per pixel),
//             optional fused InstanceNorm affine + activation of the consumer (zero padding applied AFTER it, as the
//             reference does), convert to hi / lo, write image (i+1)&1; the loads of tile i+3 / i+2 go out item by item
//             into the registers each conversion frees, so two tiles of loads are always in flight per lane;
//   consumers (waves 4-7)  9 taps x all input chunks from image i&1, weights through a register ring from L2 (32 -> 32:
//             resident in registers), epilogue (bias, activation, eval-BatchNorm affine, residual, InstanceNorm partial
//             sums for the next layer) -- nothing but LDS reads, MFMAs and the epilogue in these waves.
// One s_barrier per tile; tiles dealt so that the workgroups of one XCD work on neighbouring tiles.
//
// The consumer loop is software-pipelined by hand: the LDS operand rows of step j+1 are requested before the MFMAs of step j
// are issued (hipcc otherwise sinks each ds_read next to its use and every step pays the LDS latency), and the weight ring
// runs on across tiles.  Measured on 64 -> 64 at 240x320x64 (s_memtime stamps): 15.7k -> 11.8k cycles per tile against 6.9k
// of MFMA issue -- and the clock the chip holds falls from 1.87 to 1.55 GHz as the MFMA duty rises: these layers are limited
// by power, not by issue.  M16 builds the same loop on v_mfma_f32_16x16x32_f16, on which the chip holds a higher clock for
// the same FLOP per cycle (MI355X DVFS give-back, shape effect); pixel pitch 48 halfs keeps its ds_read_b128 pattern
// conflict-free (40 for the 32x32x16 pattern).
#include "common.h"
#include "split_f16.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
__device__ int g_rs_prio = 0;

constexpr int TW = 32, HWd = TW + 2;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// NCH: 32-channel chunks of the input slice (1 or 2); WN: 32-wide output tiles (1, 2 or 4); TH: tile rows (8 with one chunk,
// 4 with two: what two LDS images leave room for).  Consumer wave -> (rows, output tiles): NSPLIT waves share a row group.
// DBG 32: diagnostic build that stamps s_memtime / s_memrealtime around the consumer loop into stats_ws (no statistics).
template <int NCH, int TH, bool M16>
struct RsGeom {
  static constexpr int LDH = M16 ? 48 : 40;             // halfs per pixel of a 32-channel chunk
  static constexpr int NPX = (TH + 2) * HWd;
  static constexpr int PADW = (16 - (NPX * LDH / 2) % 32 + 32) % 32;      // dwords: chunk 1 starts 16 banks away from chunk 0
  static constexpr int CHS = NPX * LDH + 2 * PADW;      // halfs per chunk
  static constexpr int IMG = 2 * NCH * CHS;             // halfs per image: [hi | lo][NCH][NPX][LDH]
  static constexpr size_t LDS_BYTES = (size_t)2 * IMG * sizeof(_Float16);
  static_assert(CHS % 8 == 0 && LDS_BYTES <= 163840);
};

#ifndef RINGX
#define RINGX 1
#endif
// TPO: transposed product (weight fragment as the A operand): a lane ends up with 16 channels (4 runs of 4) of ONE pixel and stores
// 16 bytes per instruction -- 8 stores per 32 x 32 block instead of 32.  Beside MFMAs a vector-memory instruction costs the wave
// ~100 cycles of issue whatever its width (measured: 32 dword stores per tile = 3.6 k of 12.2 k cycles); needs all 32 * WN channels
// stored and no statistics (channel sums would need cross-lane reductions).
template <int NCH, int WN, int TH, bool M16, int DBG = 0, bool TPO = false>
__global__ __launch_bounds__(512)
void conv3x3_rs_kernel(const egne_conv_desc p, const _Float16* __restrict__ fhi, const _Float16* __restrict__ flo, float a_scale,
                       float out_scale, int tiles_x, int tiles_y, int ntiles) {
  egne::dyn_scales(p.dyn_scale, a_scale, out_scale);
  using Geo = RsGeom<NCH, TH, M16>;
  constexpr int LDH = Geo::LDH, NPX = Geo::NPX, CHS = Geo::CHS, IMG = Geo::IMG;
  constexpr int PPP = 8 * NCH;                          // 16-byte pieces per pixel
  constexpr int NI = (NPX * PPP + 255) / 256;           // items per producer lane and tile
  extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  const egne_seg sg = p.seg[0];

  const int per = gridDim.x >> 3;
  auto tile_at = [&](int i) { return (gridDim.x & 7) ? (int)blockIdx.x + i * (int)gridDim.x : ((i * 8 + ((int)blockIdx.x & 7)) * per + ((int)blockIdx.x >> 3)); };
  struct Tile { int b, y0, x0; };
  auto decode = [&](int t) {
    Tile r;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    r.b = t; r.y0 = ty * TH; r.x0 = tx * TW;
    return r;
  };
  int nmine = 0;
  while (tile_at(nmine) < ntiles) ++nmine;
  const int nloop = (nmine + 1) & ~1;                   // both roles run an even number of steps (register buffer = step parity)

  if (wave < 4) {
    // =================================================================== producers: halo -> hi / lo image
    const int piece = tid % PPP, pg = tid / PPP;        // 16-byte piece of the pixel, pixel group
    constexpr int PPI = 256 / PPP;                      // pixels per item round
    const int chunk = piece >> 3, pc = piece & 7;
    const float slope_in = sg.act_in == EGNE_ACT_RELU ? 0.f : (sg.act_in == EGNE_ACT_LEAKY ? 0.01f : 1.f);
    const bool cok = piece * 4 < sg.Cp;                 // channels past the slice (padding up to 32 * NCH) read zeros
    u32x4 st[2][NI];
    auto issue1 = [&](const Tile& tl, bool on, auto bc, auto ic) {
      constexpr int BUF = decltype(bc)::value, I = decltype(ic)::value;
      const __amdgpu_buffer_rsrc_t r = make_rsrc(sg.ptr + (long long)tl.b * H * W * sg.pix_stride, (unsigned)H * W * (unsigned)sg.pix_stride * 4u);
      int pq = pg;
      asm volatile("" : "+v"(pq));                      // opaque: no hoisting of the per-item coordinates out of the tile loop
      const int px = pq + PPI * I;
      const int hy = px / HWd, hx = px - hy * HWd;
      const int y = tl.y0 - 1 + hy, x = tl.x0 - 1 + hx;
      const bool ok = on && cok && px < NPX && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      st[BUF][I] = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? ((y * W + x) * (int)sg.pix_stride + sg.ch_off + piece * 4) * 4 : (int)OOB, 0, 0);
    };
    // fused affine: per-(frame, channel) coefficients of the tile converted in a step are requested one step EARLIER, ahead of
    // that step's halo loads in the (in-order) vector memory queue, so waiting for them never covers younger loads
    f32x4 asc[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}}, ash[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto load_aff = [&](const Tile& tl, auto bc) {
      constexpr int BUF = decltype(bc)::value;
      if (sg.scale && nmine > 0) {       // (a workgroup without tiles decodes a frame past the batch: no table row to read)
        const float* zs = cok ? sg.scale + (long long)tl.b * sg.Cp + piece * 4 : egne_zero_page;
        const float* zh = cok ? sg.shift + (long long)tl.b * sg.Cp + piece * 4 : egne_zero_page;
        asc[BUF] = *(const f32x4*)zs;
        ash[BUF] = *(const f32x4*)zh;
      }
    };
    auto convert1 = [&](const Tile& tl, _Float16* img, auto bc, auto ic) {
      constexpr int BUF = decltype(bc)::value, I = decltype(ic)::value;
      const int px = pg + PPI * I;
      if (I < NI - 1 || px < NPX) {
        f32x4 v = __builtin_bit_cast(f32x4, st[BUF][I]);
        if (sg.scale) {      // fused InstanceNorm affine (+ activation) of the consumer; zero padding applied after it
          const int hy = px / HWd, hx = px - hy * HWd;
          const int y = tl.y0 - 1 + hy, x = tl.x0 - 1 + hx;
          v = v * asc[BUF] + ash[BUF];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope_in);
          if (!((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)) v = (f32x4)(0.f);
        }
        h2 h0, h1, l0, l1;                        // x * a_scale = hi + lo, plain (unpacked) VALU: split_f16.h
        egne::split2(v[0], v[1], a_scale, h0, l0);
        egne::split2(v[2], v[3], a_scale, h1, l1);
        const h4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
        const int o = chunk * CHS + px * LDH + pc * 4;
        *(h4*)&img[o] = hi;
        *(h4*)&img[NCH * CHS + o] = lo;
      }
    };
    // step s (tile s): convert tile s+1 out of register buffer (s+1)&1 and refill every freed register with tile s+3
    auto step = [&](int s, auto bc) {
      constexpr int BUF = decltype(bc)::value;          // = (s + 1) & 1
      const bool c_on = s + 1 < nmine, i_on = s + 3 < nmine;
      const Tile tc = decode(tile_at(c_on ? s + 1 : 0)), ti = decode(tile_at(i_on ? s + 3 : 0));
      load_aff(decode(tile_at(s + 2 < nmine ? s + 2 : 0)), std::integral_constant<int, BUF ^ 1>{});
      _Float16* img = ldsh + ((s + 1) & 1) * IMG;
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (([&] {
          if (!(DBG & 4)) { if (c_on) convert1(tc, img, bc, std::integral_constant<int, Is>{}); }
          if (!(DBG & 16)) issue1(ti, i_on, bc, std::integral_constant<int, Is>{});
        }()), ...);
      }(std::make_integer_sequence<int, NI>{});
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    {   // prologue: tiles 0 and 1 requested, tile 0 converted (its registers refilled with tile 2)
      const Tile t0 = decode(tile_at(0)), t1 = decode(tile_at(nmine > 1 ? 1 : 0)), t2 = decode(tile_at(nmine > 2 ? 2 : 0));
      load_aff(t0, B0{});
      load_aff(t1, B1{});
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (issue1(t0, nmine > 0, B0{}, std::integral_constant<int, Is>{}), ...);
        (issue1(t1, nmine > 1, B1{}, std::integral_constant<int, Is>{}), ...);
      }(std::make_integer_sequence<int, NI>{});
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (([&] {
          if (nmine > 0) convert1(t0, ldsh, B0{}, std::integral_constant<int, Is>{});
          issue1(t2, nmine > 2, B0{}, std::integral_constant<int, Is>{});
        }()), ...);
      }(std::make_integer_sequence<int, NI>{});
    }
    lds_barrier();
    for (int s = 0; s < nloop; s += 2) {
      step(s, B1{});
      lds_barrier();
      step(s + 1, B0{});
      lds_barrier();
    }
  } else {
    // =================================================================== consumers: 9 taps from the LDS image
    if (g_rs_prio) __builtin_amdgcn_s_setprio(1);
    constexpr int NSPLIT = (TH == 4 && WN >= 2) ? 2 : 1;
    constexpr int WNW = WN / NSPLIT, WMW = TH * NSPLIT / 4;
    const int cw = wave - 4;
    const int row0 = (cw / NSPLIT) * WMW, nt0 = (cw % NSPLIT) * WNW;
    const unsigned frame_out = (unsigned)H * W * (unsigned)p.out_pix_stride * 4u;
    const unsigned frame_res = (unsigned)H * W * (unsigned)p.res_pix_stride * 4u;
    const float slope_out = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
    constexpr int KT16 = NCH * 2, NT2 = WN;
    const unsigned w2bytes = 9u * (unsigned)(KT16 * 16) * (unsigned)(NT2 * 32) * 2u;
    const __amdgpu_buffer_rsrc_t rwh = make_rsrc(fhi, w2bytes), rwl = make_rsrc(flo, w2bytes);
    constexpr int stride_k16 = NT2 * 1024, stride_tap = KT16 * NT2 * 1024;   // bytes: fragment (tap, k16, nt) of the 32x32x16 pack
    const int out_step = (int)p.out_pix_stride * 4, res_step = (int)p.res_pix_stride * 4;
    const bool st_on = !(DBG & 32) && p.stats_ws != nullptr;
    // lane -> (pixel within the MFMA row block, 8-channel group) for the operand reads, (output channel, pixel group) for the result
    constexpr int MB = M16 ? 16 : 32;                    // pixels / channels per MFMA block
    constexpr int NMH = 32 / MB;                         // blocks per 32
    const int lm = lane & (MB - 1), kg = lane / MB;
    float bvs[WNW][NMH], pss[WNW][NMH], pts[WNW][NMH];   // per-lane output channel constants
#pragma unroll
    for (int tn = 0; tn < WNW; ++tn)
#pragma unroll
      for (int nh = 0; nh < NMH; ++nh) {
        const int n = (nt0 + tn) * 32 + nh * MB + lm;
        const bool nok = n < p.Cout_store;
        bvs[tn][nh] = (p.bias && nok) ? p.bias[n] : 0.f;
        pss[tn][nh] = (p.post_scale && nok) ? p.post_scale[n] : 1.f;
        pts[tn][nh] = (p.post_scale && nok) ? p.post_shift[n] : 0.f;
      }

    // One step: 16 (32 with M16) input channels of one tap.  The weights are the same for every tile, so their register ring runs
    // on across tiles (a fragment is requested NR steps before its use, never at a tile start); the operand rows of step j + 1
    // are read from LDS before the MFMAs of step j are issued.
    constexpr int SPT = M16 ? 1 : 2;                     // steps per (tap, chunk)
    constexpr int NS = NCH * 9 * SPT;                    // steps per tile
    constexpr bool WREG = NCH == 1 && WNW == 1;          // 32 -> 32: all weight fragments stay in registers
    constexpr int NR = WREG ? NS : ((M16 && WNW == 2) ? 2 : (NCH == 2 && WNW == 1 ? RINGX : 1) * 3 * SPT);      // ring slots (NS % NR == 0)
    static_assert(NS % NR == 0);
    // M16 reads the 32x32x16 pack too: the B operand of a 16x16x32 block (32 k x 16 n) is, for k-group kg, 16 bytes of
    // fragment (tap, 2 * chunk + kg / 2, nt) at lane position (kg & 1) * 32 + nh * 16 + lm
    const int wlane = M16 ? (kg >> 1) * stride_k16 + ((kg & 1) * 32 + lm) * 16 + nt0 * 1024 : lane * 16 + nt0 * 1024;
    auto w_off = [&](int j, int tn, int nh) {
      const int ch = j / (9 * SPT), tap = (j % (9 * SPT)) / SPT, ks = j % SPT;
      return (ch * 2 + ks) * stride_k16 + tap * stride_tap + tn * 1024 + nh * 256;
    };
    auto a_off = [&](int j) {
      const int ch = j / (9 * SPT), tap = (j % (9 * SPT)) / SPT, ks = j % SPT;
      return ch * CHS + ((tap / 3) * HWd + tap % 3) * LDH + ks * 16;
    };
    const int abase = (row0 * HWd + lm) * LDH + kg * 8;
    u32x4 qh[NR][WNW][NMH], ql[NR][WNW][NMH];
#pragma unroll
    for (int s = 0; s < NR; ++s)
#pragma unroll
      for (int tn = 0; tn < WNW; ++tn)
#pragma unroll
        for (int nh = 0; nh < NMH; ++nh) {
          qh[s][tn][nh] = __builtin_amdgcn_raw_buffer_load_b128(rwh, wlane, w_off(s, tn, nh), 0);
          ql[s][tn][nh] = __builtin_amdgcn_raw_buffer_load_b128(rwl, wlane, w_off(s, tn, nh), 0);
        }
    lds_barrier();
    unsigned long long t0 = 0, r0 = 0;
    if constexpr (DBG & 32) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    using acc_t = std::conditional_t<M16, f32x4, f32x16>;
    unsigned long long ph_mma = 0, ph_epi = 0, ph_bar = 0, tp = t0;
    auto phase = [&](unsigned long long& accum) {
      if constexpr (DBG & 32) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        accum += t - tp; tp = t;
      }
    };

    // ---- deferred epilogue: the results of tile i-1 are scaled, activated and stored value by value BETWEEN the MFMAs of tile i
    // (one consumer wave per SIMD: an epilogue after the loop would leave the matrix pipe idle for its whole length).
    // 32x32x16: lane holds channel lm of pixels x = 4 * kg + (r & 3) + 8 * (r >> 2), r < 16;
    // 16x16x32: lane holds channel lm of pixels x = 16 * mh + 4 * kg + r, r < 4.
    constexpr int NRV = M16 ? 4 : 16;
    constexpr int NV = WNW * NMH * WMW * NMH * NRV;       // values per lane and tile, order (tn, nh, tm, mh, r)
    constexpr bool DEFER = NV <= 32 && !(NCH == 1 && WNW == 1);                     // wider shapes have no registers for a second result set: epilogue in place
    acc_t prev[WMW][NMH][WNW][NMH];
#pragma unroll
    for (int a = 0; a < WMW * NMH * WNW * NMH; ++a) (&prev[0][0][0][0])[a] = (acc_t)(0.f);
    __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out, 0u), rres = make_rsrc(nullptr, 0u);
    int pbase[WMW], pcm[WMW];                            // previous tile: first pixel of the lane per row, valid pixels from it
    bool pchk[WMW];                                      // ... and whether the row is one the overflow test looks at (egne_ovf_row)
#pragma unroll
    for (int tm = 0; tm < WMW; ++tm) { pbase[tm] = 0; pcm[tm] = 0; pchk[tm] = false; }
    int pvo[WMW][WNW][NMH], pvr[WMW][WNW][NMH];            // previous tile: byte offsets of (row, channel) in the output / residual
    bool pedge = true;                                   // ... and whether it needs per-value range checks (nothing stored before the first tile)
    const bool full_epi = p.post_scale != nullptr || p.residual != nullptr;
#pragma unroll
    for (int a = 0; a < WMW * WNW * NMH; ++a) { (&pvo[0][0][0])[a] = (int)OOB; (&pvr[0][0][0])[a] = (int)OOB; }
    int pchunk = -1;                                     // no previous tile yet: nothing to write
    bool ovf_bad = false;                                // a non-finite value was stored (egne_conv_desc.ovf_flag)
    double st_s[WNW][NMH], st_q[WNW][NMH];
    int n4[WNW][NMH];
    bool nokv[WNW][NMH];
#pragma unroll
    for (int tn = 0; tn < WNW; ++tn)
#pragma unroll
      for (int nh = 0; nh < NMH; ++nh) {
        const int n = (nt0 + tn) * 32 + nh * MB + lm;
        n4[tn][nh] = n;
        nokv[tn][nh] = n < p.Cout_store;
      }
    auto epi_value = [&](auto vc) {
      constexpr int V = decltype(vc)::value;
      constexpr int r = V % NRV, mh = (V / NRV) % NMH, tm = (V / (NRV * NMH)) % WMW, nh = (V / (NRV * NMH * WMW)) % NMH,
                    tn = V / (NRV * NMH * WMW * NMH);
      constexpr int c = M16 ? mh * 16 + r : (r & 3) + 8 * (r >> 2);
      constexpr bool first = r == 0 && mh == 0 && tm == 0, last = r == NRV - 1 && mh == NMH - 1 && tm == WMW - 1;
      // interior tiles: the lane's byte offset of the row is fixed at hand-over and the pixel step goes into the instruction's
      // scalar offset -- 3 VALU + the store per value; edge tiles mask value by value
      bool ok = true;
      int vo = pvo[tm][tn][nh], vr = pvr[tm][tn][nh];
      if (pedge) {
        ok = nokv[tn][nh] && c < pcm[tm];
        vo = ok ? vo : (int)OOB; vr = ok ? vr : (int)OOB;
      }
      float v = prev[tm][mh][tn][nh][r] * out_scale + bvs[tn][nh];
      v = fmaxf(v, v * slope_out);
      if (full_epi) {
        v = v * pss[tn][nh] + pts[tn][nh];
        v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, vr, c * res_step, 0));
      }
      if (pchk[tm]) ovf_bad |= egne_nonfinite(v);          // lane = channel: the rows egne_ovf_row names (wave-uniform)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, vo, c * out_step, 0);
      if (st_on) {
        if constexpr (first) { st_s[tn][nh] = 0.; st_q[tn][nh] = 0.; }
        const double vm = ok ? (double)v : 0.;
        st_s[tn][nh] += vm; st_q[tn][nh] += vm * vm;
        if constexpr (last) {                 // one chunk = this wave's rows of the tile (fixed order: deterministic)
          double a = st_s[tn][nh], b = st_q[tn][nh];
#pragma unroll
          for (int m = MB; m < 64; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
          if (kg == 0 && nokv[tn][nh] && pchunk >= 0)
            ((double2*)p.stats_ws)[(long long)pchunk * p.Cout_store + n4[tn][nh]] = make_double2(a, b);
        }
      }
    };
    auto epi_step = [&](auto jc) {            // the values handed to step j: V * NS / NV == j
      constexpr int J = decltype(jc)::value;
      [&]<int... Vs>(std::integer_sequence<int, Vs...>) {
        (([&] { if constexpr (Vs * NS / NV == J || (NV > NS && Vs / ((NV + NS - 1) / NS) == J)) epi_value(std::integral_constant<int, Vs>{}); }()), ...);
      }(std::make_integer_sequence<int, NV>{});
    };
    // ---- TPO: lane = pixel lm of the row, registers 4 j + e = channels 32 tn + 8 j + 4 kg + e: one 16-byte store per (row, tn, j)
    constexpr int NGRP = WMW * WNW * 4;                  // store groups per lane and tile
    f32x4 tb4[TPO ? WNW : 1][4];                         // bias of the lane's 16 channels per output tile
    int tvo[WMW], tvr[WMW];                              // previous tile: byte offset of the lane's pixel per row (OOB: outside the image)
    if constexpr (TPO) {
#pragma unroll
      for (int tn = 0; tn < WNW; ++tn)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          tb4[tn][j] = p.bias ? *(const f32x4*)(p.bias + (nt0 + tn) * 32 + 8 * j + 4 * kg) : (f32x4)(0.f);
#pragma unroll
      for (int tm = 0; tm < WMW; ++tm) { tvo[tm] = (int)OOB; tvr[tm] = (int)OOB; }
    }
    // the values are finished in place at hand-over (tpo_finish); the deferred part is the bare store: computing next to the store
    // would reuse its data registers group after group, and overwriting the source of a store in flight costs a vmcnt(0)
    auto tpo_finish = [&](auto gc) {
      constexpr int Gi = decltype(gc)::value, j = Gi % 4, tm = (Gi / 4) % WMW, tn = Gi / (4 * WMW);
      constexpr int choff = (tn * 32 + 8 * j) * 4;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = prev[tm][0][tn][0][4 * j + e] * out_scale + tb4[tn][j][e];
        v[e] = fmaxf(t, t * slope_out);
      }
      if (full_epi) {                                    // rare in these layers: post affine / residual straight from memory
        const int n = (nt0 + tn) * 32 + 8 * j + 4 * kg;
        if (p.post_scale) {
          const f32x4 ps = *(const f32x4*)(p.post_scale + n), pt = *(const f32x4*)(p.post_shift + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] * ps[e] + pt[e];
        }
        if (p.residual) {
          const f32x4 rv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, tvr[tm], choff, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += rv[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) prev[tm][0][tn][0][4 * j + e] = v[e];
      if constexpr (tn == 0 && j == 0) ovf_bad |= egne_nonfinite(v[0]);      // lane = pixel: one channel per pixel (common.h)
    };
    auto tpo_group = [&](auto gc) {
      constexpr int Gi = decltype(gc)::value, j = Gi % 4, tm = (Gi / 4) % WMW, tn = Gi / (4 * WMW);
      constexpr int choff = (tn * 32 + 8 * j) * 4;       // immediate offset of the run inside the pixel
      const f32x4 v = {prev[tm][0][tn][0][4 * j], prev[tm][0][tn][0][4 * j + 1], prev[tm][0][tn][0][4 * j + 2], prev[tm][0][tn][0][4 * j + 3]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout, tvo[tm], choff, 0);
    };
    auto tpo_step = [&](auto jc) {            // the groups handed to step j
      constexpr int J = decltype(jc)::value;
      [&]<int... Gs>(std::integer_sequence<int, Gs...>) {
        (([&] { if constexpr (Gs * NS / NGRP == J) tpo_group(std::integral_constant<int, Gs>{}); }()), ...);
      }(std::make_integer_sequence<int, NGRP>{});
    };
    bool have_prev = false;

    for (int i = 0; i < nloop; ++i) {
      if (i < nmine) {
        const Tile tl = decode(tile_at(i));
        const _Float16* Thi = ldsh + (i & 1) * IMG + abase;
        const _Float16* Tlo = Thi + NCH * CHS;
        acc_t acc[WMW][NMH][WNW][NMH];
#pragma unroll
        for (int a = 0; a < WMW * NMH * WNW * NMH; ++a) (&acc[0][0][0][0])[a] = (acc_t)(0.f);
        h8 ah[2][WMW][NMH], al[2][WMW][NMH];
#pragma unroll
        for (int tm = 0; tm < WMW; ++tm)
#pragma unroll
          for (int mh = 0; mh < NMH; ++mh) {
            ah[0][tm][mh] = *(const h8*)&Thi[a_off(0) + (tm * HWd + mh * MB) * LDH];
            al[0][tm][mh] = *(const h8*)&Tlo[a_off(0) + (tm * HWd + mh * MB) * LDH];
          }
        [&]<int... Js>(std::integer_sequence<int, Js...>) {
          (([&] {
            constexpr int j = Js;
            if constexpr (j + 1 < NS && !(DBG & 2)) {
#pragma unroll
              for (int tm = 0; tm < WMW; ++tm)
#pragma unroll
                for (int mh = 0; mh < NMH; ++mh) {
                  ah[(j + 1) & 1][tm][mh] = *(const h8*)&Thi[a_off(j + 1) + (tm * HWd + mh * MB) * LDH];
                  al[(j + 1) & 1][tm][mh] = *(const h8*)&Tlo[a_off(j + 1) + (tm * HWd + mh * MB) * LDH];
                }
            }
            h8 bh[WNW][NMH], bl[WNW][NMH];
#pragma unroll
            for (int tn = 0; tn < WNW; ++tn)
#pragma unroll
              for (int nh = 0; nh < NMH; ++nh) {
                bh[tn][nh] = __builtin_bit_cast(h8, qh[j % NR][tn][nh]);
                bl[tn][nh] = __builtin_bit_cast(h8, ql[j % NR][tn][nh]);
                if constexpr (!WREG && !(DBG & 1)) {
                  qh[j % NR][tn][nh] = __builtin_amdgcn_raw_buffer_load_b128(rwh, wlane, w_off((j + NR) % NS, tn, nh), 0);
                  ql[j % NR][tn][nh] = __builtin_amdgcn_raw_buffer_load_b128(rwl, wlane, w_off((j + NR) % NS, tn, nh), 0);
                }
              }
            if constexpr (TPO) { if constexpr (DEFER && !(DBG & 8)) tpo_step(std::integral_constant<int, j>{}); }
            else if constexpr (DEFER && !(DBG & 8)) epi_step(std::integral_constant<int, j>{});
            __builtin_amdgcn_sched_barrier(0);          // the reads, refills and stores above are issued before this step's MFMAs
#pragma unroll
            for (int tm = 0; tm < WMW; ++tm)
#pragma unroll
              for (int mh = 0; mh < NMH; ++mh)
#pragma unroll
                for (int tn = 0; tn < WNW; ++tn)
#pragma unroll
                  for (int nh = 0; nh < NMH; ++nh) {
                    acc_t& c = acc[tm][mh][tn][nh];
                    if constexpr (M16) {
                      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[j & 1][tm][mh], bh[tn][nh], c, 0, 0, 0);
                      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[j & 1][tm][mh], bl[tn][nh], c, 0, 0, 0);
                      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[j & 1][tm][mh], bh[tn][nh], c, 0, 0, 0);
                    } else if constexpr (TPO) {
                      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[tn][nh], al[j & 1][tm][mh], c, 0, 0, 0);
                      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[tn][nh], ah[j & 1][tm][mh], c, 0, 0, 0);
                      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[tn][nh], ah[j & 1][tm][mh], c, 0, 0, 0);
                    } else {
                      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j & 1][tm][mh], bh[tn][nh], c, 0, 0, 0);
                      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j & 1][tm][mh], bl[tn][nh], c, 0, 0, 0);
                      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j & 1][tm][mh], bh[tn][nh], c, 0, 0, 0);
                    }
                  }
            __builtin_amdgcn_sched_barrier(0);
          }()), ...);
        }(std::make_integer_sequence<int, NS>{});
        phase(ph_mma);
        // second output: 2x2 / stride 2 / ceil-mode max pooling of the activated result -- the wave owns both rows of a pooling
        // window (row0 and y0 are even) and the lane both of its columns (registers r, r + 1); act(max) = max(act): monotonic
        if constexpr (WMW == 2 && TPO) {
          if (p.pool_out) {                                // lane = pixel: the column partner is lane ^ 1 (DPP quad_perm), the row partner register tm ^ 1
            const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
            const __amdgpu_buffer_rsrc_t rpo = make_rsrc(p.pool_out + (long long)tl.b * Hp * Wp * p.pool_pix_stride,
                                                         (unsigned)Hp * Wp * (unsigned)p.pool_pix_stride * 4u);
            const int x = tl.x0 + lm, y = tl.y0 + row0;
            const bool y1 = y + 1 < H, x1 = x + 1 < W;
            const int poff = (!(lm & 1) && x < W && y < H) ? (((y >> 1) * Wp + (x >> 1)) * (int)p.pool_pix_stride + p.pool_ch_off + nt0 * 32 + 4 * kg) * 4 : (int)OOB;
#pragma unroll
            for (int tn = 0; tn < WNW; ++tn)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float m = acc[0][0][tn][0][4 * j + e];
                  if (y1) m = fmaxf(m, acc[1][0][tn][0][4 * j + e]);
                  const float q = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xf, 0xf, false));   // lane ^ 1
                  if (x1) m = fmaxf(m, q);
                  const float t = m * out_scale + tb4[tn][j][e];
                  v[e] = fmaxf(t, t * slope_out);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rpo, poff, (tn * 32 + 8 * j) * 4, 0);
              }
          }
        }
        if constexpr (WMW == 2 && !M16 && !TPO) {
          if (p.pool_out) {
            const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
            const __amdgpu_buffer_rsrc_t rpo = make_rsrc(p.pool_out + (long long)tl.b * Hp * Wp * p.pool_pix_stride,
                                                         (unsigned)Hp * Wp * (unsigned)p.pool_pix_stride * 4u);
            const int xl = tl.x0 + 4 * kg, y = tl.y0 + row0;
            const bool y1 = y + 1 < H;
#pragma unroll
            for (int tn = 0; tn < WNW; ++tn) {
              const int n = (nt0 + tn) * 32 + lm;
              const bool nok = n < p.Cout_store && y < H;
#pragma unroll
              for (int r = 0; r < 16; r += 2) {
                const int c = (r & 3) + 8 * (r >> 2), x = xl + c;
                float m = acc[0][0][tn][0][r];
                if (x + 1 < W) m = fmaxf(m, acc[0][0][tn][0][r + 1]);
                if (y1) {
                  m = fmaxf(m, acc[1][0][tn][0][r]);
                  if (x + 1 < W) m = fmaxf(m, acc[1][0][tn][0][r + 1]);
                }
                float v = m * out_scale + bvs[tn][0];
                v = fmaxf(v, v * slope_out);
                const int off = (nok && x < W) ? ((((y >> 1) * Wp + (x >> 1)) * (int)p.pool_pix_stride + p.pool_ch_off + n) * 4) : (int)OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rpo, off, 0, 0);
              }
            }
          }
        }
        // hand the tile over to the deferred epilogue
#pragma unroll
        for (int a = 0; a < WMW * NMH * WNW * NMH; ++a) (&prev[0][0][0][0])[a] = (&acc[0][0][0][0])[a];
        rout = make_rsrc(p.out + (long long)tl.b * H * W * p.out_pix_stride, frame_out);
        rres = make_rsrc(p.residual ? p.residual + (long long)tl.b * H * W * p.res_pix_stride : nullptr, p.residual ? frame_res : 0u);
        const int xl = tl.x0 + 4 * kg;
#pragma unroll
        for (int tm = 0; tm < WMW; ++tm) {
          const int y = tl.y0 + row0 + tm;
          pbase[tm] = y * W + xl;
          pcm[tm] = (y < H && xl < W) ? W - xl : 0;
          pchk[tm] = egne_ovf_row(y, H);
#pragma unroll
          for (int tn = 0; tn < WNW; ++tn)
#pragma unroll
            for (int nh = 0; nh < NMH; ++nh) {
              pvo[tm][tn][nh] = (pbase[tm] * (int)p.out_pix_stride + p.out_ch_off + n4[tn][nh]) * 4;
              pvr[tm][tn][nh] = (pbase[tm] * (int)p.res_pix_stride + p.res_ch_off + n4[tn][nh]) * 4;
            }
        }
        pedge = tl.x0 + TW > W || tl.y0 + TH > H || (nt0 + WNW) * 32 > p.Cout_store;       // wave-uniform
        if constexpr (TPO) {
#pragma unroll
          for (int tm = 0; tm < WMW; ++tm) {
            const int y = tl.y0 + row0 + tm, x = tl.x0 + lm;
            const bool okp = y < H && x < W;
            tvo[tm] = okp ? ((y * W + x) * (int)p.out_pix_stride + p.out_ch_off + nt0 * 32 + 4 * kg) * 4 : (int)OOB;
            tvr[tm] = okp ? ((y * W + x) * (int)p.res_pix_stride + p.res_ch_off + nt0 * 32 + 4 * kg) * 4 : (int)OOB;
          }
        }
        pchunk = tl.b * p.stats_nchunk + ((tl.y0 / TH) * tiles_x + tl.x0 / TW) * (4 / NSPLIT) + cw / NSPLIT;
        if constexpr (TPO) [&]<int... Gs>(std::integer_sequence<int, Gs...>) { (tpo_finish(std::integral_constant<int, Gs>{}), ...); }(std::make_integer_sequence<int, NGRP>{});
        have_prev = DEFER;
        if constexpr (!DEFER) {
          if constexpr (TPO) [&]<int... Gs>(std::integer_sequence<int, Gs...>) { (tpo_group(std::integral_constant<int, Gs>{}), ...); }(std::make_integer_sequence<int, NGRP>{});
          else [&]<int... Vs>(std::integer_sequence<int, Vs...>) { (epi_value(std::integral_constant<int, Vs>{}), ...); }(std::make_integer_sequence<int, NV>{});
        }
        phase(ph_epi);
      }
      lds_barrier();
      phase(ph_bar);
    }
    if (have_prev) {                            // the last tile's results
      if constexpr (TPO) [&]<int... Gs>(std::integer_sequence<int, Gs...>) { (tpo_group(std::integral_constant<int, Gs>{}), ...); }(std::make_integer_sequence<int, NGRP>{});
      else [&]<int... Vs>(std::integer_sequence<int, Vs...>) { (epi_value(std::integral_constant<int, Vs>{}), ...); }(std::make_integer_sequence<int, NV>{});
    }
    egne_ovf_commit(ovf_bad, p.ovf_flag);
    if constexpr (DBG & 32) {
      const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
      if (lane == 0 && p.stats_ws) {
        unsigned long long* o = (unsigned long long*)p.stats_ws + ((long long)blockIdx.x * 4 + cw) * 8;
        o[0] = t1 - t0; o[1] = r1 - r0; o[2] = nmine; o[3] = ph_mma; o[4] = ph_epi; o[5] = ph_bar;
      }
    }
  }
}

template <int NCH, int WN, int TH, bool M16, int DBG = 0, bool TPO = false>
int launch_rs(const egne_conv_desc& d, const _Float16* fhi, const _Float16* flo, float a_scale, float os, hipStream_t st) {
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y * d.B;
  constexpr size_t lds = RsGeom<NCH, TH, M16>::LDS_BYTES;
  static bool once = hipFuncSetAttribute((const void*)conv3x3_rs_kernel<NCH, WN, TH, M16, DBG, TPO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
  if (!once) return egne::fail(EGNE_ERR_LAUNCH, "conv3x3_rs: cannot raise the dynamic LDS limit to %zu", lds);
  int gx = 256;
  if (gx > ntiles) gx = ntiles;
  hipLaunchKernelGGL((conv3x3_rs_kernel<NCH, WN, TH, M16, DBG, TPO>), dim3(gx), dim3(512), lds, st, d, fhi, flo, a_scale, os, tiles_x, tiles_y, ntiles);
  return egne::check_launch("egne_conv3x3_rs_f16_fwd");
}

template <bool M16>
int dispatch_rs(const egne_conv_desc& d, const _Float16* h, const _Float16* l, float a_scale, float os, hipStream_t st, int dbg, bool tpo) {
  if constexpr (!M16) {
    if (tpo) {                  // transposed product, 16-byte stores: shapes with one output tile per consumer wave
      // (32 -> 32 stays on the per-channel layout: with all 36 weight fragments resident in registers its transposed build came out
      //  with one bias register overwritten in eight lanes -- the lanes hipcc spills scalar registers to; found by the network tests)
      if (d.Ktot == 64 && d.CoutP == 32) return launch_rs<2, 1, 4, false, 0, true>(d, h, l, a_scale, os, st);
      if (d.Ktot == 64 && d.CoutP == 64) {
        if (dbg == 32) return launch_rs<2, 2, 4, false, 32, true>(d, h, l, a_scale, os, st);
        if (dbg == 33) return launch_rs<2, 2, 4, false, 33, true>(d, h, l, a_scale, os, st);
        return launch_rs<2, 2, 4, false, 0, true>(d, h, l, a_scale, os, st);
      }
    }
  }
  if (d.Ktot == 32) {
    if (d.CoutP == 32) return launch_rs<1, 1, 8, M16>(d, h, l, a_scale, os, st);
    if (d.CoutP == 64) return launch_rs<1, 2, 8, false>(d, h, l, a_scale, os, st);      // 16x16x32 build of this shape spills
    return egne::fail(EGNE_ERR_ARG, "conv3x3_rs: 32 -> 128 is not built (register budget)");
  }
  if (d.CoutP == 32) return launch_rs<2, 1, 4, M16>(d, h, l, a_scale, os, st);
  if (d.CoutP == 64) {
    switch (dbg) {
      case 32: return launch_rs<2, 2, 4, M16, 32>(d, h, l, a_scale, os, st);
      case 33: return launch_rs<2, 2, 4, M16, 33>(d, h, l, a_scale, os, st);
      case 40: return launch_rs<2, 2, 4, M16, 40>(d, h, l, a_scale, os, st);
      default: return launch_rs<2, 2, 4, M16>(d, h, l, a_scale, os, st);
    }
  }
  return launch_rs<2, 4, 4, M16>(d, h, l, a_scale, os, st);
}

}  // namespace

// Same descriptor and weight pack as egne_conv3x3_halo_f16_fwd (one input slice with optional fused affine, 3x3 / pad 1 /
// dilation 1, Ktot = slice width rounded up to 32 and <= 64, CoutP = 32, 64 or 128 (128 with Ktot 64 only), stats_ws allowed
// with stats_nchunk = ceil(W/32) * ceil(H/TH) * rows-groups; TH = 8 for Ktot 32, 4 for Ktot 64).
// Environment: EGNE_RS_M16=1 selects the 16x16x32 MFMA shape (higher clock, more cycles: +-2 % overall, see the header); EGNE_RS_DBG=32 the clock-stamp build of the 64 -> 64 shape
// (stats_ws then receives {cycles, 100 MHz ticks, tiles} per consumer wave instead of statistics).
extern "C" int egne_conv3x3_rs_f16_fwd(const egne_conv_desc* dp, const void* fhi, const void* flo, float a_scale, float w_scale,
                                       void* stream) {
  EGNE_REQUIRE(dp && fhi && flo, "conv3x3_rs: null pointer");
  const egne_conv_desc& d = *dp;
  static const int dbg = getenv("EGNE_RS_DBG") ? atoi(getenv("EGNE_RS_DBG")) : 0;
  static const bool m16 = getenv("EGNE_RS_M16") && atoi(getenv("EGNE_RS_M16")) == 1;
  EGNE_REQUIRE(d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad_mode == 0 && d.ngroups == 1 && d.nseg == 1 && d.pad_h == 1 &&
               d.pad_w == 1 && d.dil[0] == 1 && d.Ho == d.H && d.Wo == d.W, "conv3x3_rs: geometry not supported");
  const egne_seg& g = d.seg[0];
  EGNE_REQUIRE(g.ptr && g.Cp % 8 == 0 && (g.Cp + 31) / 32 * 32 == d.Ktot && (d.Ktot == 32 || d.Ktot == 64) && g.ch_off % 4 == 0 &&
               g.pix_stride % 4 == 0 && ((uintptr_t)g.ptr & 15) == 0 && (g.scale == nullptr) == (g.shift == nullptr), "conv3x3_rs: input slice");
  EGNE_REQUIRE((d.CoutP == 32 || d.CoutP == 64 || (d.CoutP == 128 && d.Ktot == 64)) && d.Cout_store <= d.CoutP && d.out &&
               d.out_ch_off + d.Cout_store <= d.out_pix_stride, "conv3x3_rs: CoutP %d", d.CoutP);
  EGNE_REQUIRE(((uintptr_t)fhi & 15) == 0 && ((uintptr_t)flo & 15) == 0 && a_scale > 0.f && w_scale > 0.f, "conv3x3_rs: weights / scales");
  EGNE_REQUIRE((long long)d.H * d.W * g.pix_stride * 4 < (1ll << 31) && (long long)d.H * d.W * d.out_pix_stride * 4 < (1ll << 31) &&
               (!d.residual || (long long)d.H * d.W * d.res_pix_stride * 4 < (1ll << 31)), "conv3x3_rs: frame too large for 32-bit byte offsets");
  const int th = d.Ktot == 32 ? 8 : 4, rg = (th == 4 && d.CoutP >= 64) ? 2 : 4;
  EGNE_REQUIRE(!d.pool_out || (!m16 && !d.post_scale && d.Ktot == 64 && d.CoutP >= 64 && d.pool_ch_off + d.Cout_store <= d.pool_pix_stride &&
                                (long long)((d.H + 1) / 2) * ((d.W + 1) / 2) * d.pool_pix_stride * 4 < (1ll << 31)),
               "conv3x3_rs: the pooled output needs a shape with two rows per wave (Ktot 64, CoutP >= 64), no post affine");
  EGNE_REQUIRE((dbg & 32) || !d.stats_ws || (((uintptr_t)d.stats_ws & 15) == 0 && d.stats_nchunk == ((d.W + 31) / 32) * ((d.H + th - 1) / th) * rg),
               "conv3x3_rs: stats_nchunk must be tiles * %d for this shape", rg);
  const float os = 1.0f / (a_scale * w_scale);
  hipStream_t st = (hipStream_t)stream;
  const _Float16 *h = (const _Float16*)fhi, *l = (const _Float16*)flo;
  static const bool tpo_on = !(getenv("EGNE_RS_TPO") && atoi(getenv("EGNE_RS_TPO")) == 0);
  const bool tpo = tpo_on && !m16 && (!d.stats_ws || (dbg & 32)) && d.Cout_store == d.CoutP && (!d.bias || ((uintptr_t)d.bias & 15) == 0) &&
                   ((uintptr_t)d.out & 15) == 0 && d.out_pix_stride % 4 == 0 && d.out_ch_off % 4 == 0 &&
                   (!d.residual || (((uintptr_t)d.residual & 15) == 0 && d.res_pix_stride % 4 == 0 && d.res_ch_off % 4 == 0)) &&
                   (!d.pool_out || (((uintptr_t)d.pool_out & 15) == 0 && d.pool_pix_stride % 4 == 0 && d.pool_ch_off % 4 == 0)) &&
                   (!d.post_scale || (((uintptr_t)d.post_scale & 15) == 0 && ((uintptr_t)d.post_shift & 15) == 0));
  return m16 ? dispatch_rs<true>(d, h, l, a_scale, os, st, dbg, false) : dispatch_rs<false>(d, h, l, a_scale, os, st, dbg, tpo);
}

extern "C" int egne_rs_debug_prio(int on) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_rs_prio), &on, sizeof(int)) == hipSuccess ? 0 : -2;
}
