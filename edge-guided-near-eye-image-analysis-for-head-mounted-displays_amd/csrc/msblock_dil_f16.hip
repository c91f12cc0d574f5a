// The dilated branch of a BDCN MSBlock as ONE launch: out = o + sum_g relu(conv3x3_{dil g}(o) + b_g), g = 0..2
// (bdcn_new.py:51-54: o1 = relu(conv1(o)), o2 = relu(conv2(o)), o3 = relu(conv3(o)), out = o + o1 + o2 + o3; 32 -> 32
// channels, dilations 4 / 8 / 12), on the split-f16 MFMA path (fp32 tensors, three v_mfma_f32_32x32x16_f16 per product,
// fp32 accumulate; numerics: conv_f16x3.hip).
//
// Round 1 ran the three dilations as three "lattice" launches that accumulate through HBM (o read three times, the running
// sum read twice and written three times: 9 tensor passes, HBM-bound at 3.5 TB/s).  Here a workgroup owns an 8 x 32 pixel
// tile for all three dilations and the sum lives in registers: o is read once (plus halo) and out written once.
//
// A dilated 3x3 needs rows y + (ky-1)*d: for each (dilation, ky) the 8 x (32 + 2d) STRIP of o that holds the three kx taps
// is staged in LDS as hi / lo halves and used for 3 taps x 2 k-steps; 9 strips per tile, two strip buffers.  Eight waves
// with fixed roles (as conv_fused_1x1_3x3_f16.hip):
//   producers (waves 0-3)  gather strip s+1 (16 bytes per lane, eight lanes per pixel: whole 128-byte pixels), two strips
//             in flight ahead of the one being converted, out-of-image pixels carry the out-of-range offset and load the
//             zero padding; convert to hi / lo, write buffer (s+1)&1;
//   consumers (waves 4-7)  two tile rows each: 36 MFMAs per strip from buffer s&1 with the weight fragments of the strip's
//             three taps arriving one strip ahead through a register ring; after the third strip of a dilation the
//             accumulators get bias + ReLU and join the running sum; after the ninth the exact fp32 o is added and stored.
// One s_barrier per strip.  The conversion work (each element is split 9 times) sits in waves that do nothing else.
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int LDH = 40, TW = 32, TH = 8;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int D0, int D1, int D2>
__global__ __launch_bounds__(512)
void msblock_dil_kernel(const egne_conv_desc p, const _Float16* __restrict__ fhi, const _Float16* __restrict__ flo, float a_scale,
                        float out_scale, int tiles_x, int tiles_y, int ntiles) {
  constexpr int DMAX = D2 > D1 ? (D2 > D0 ? D2 : D0) : (D1 > D0 ? D1 : D0);
  constexpr int SWMAX = TW + 2 * DMAX, NPXMAX = TH * SWMAX;
  constexpr int BUFH = 2 * NPXMAX * LDH;                 // halfs per strip buffer: [hi | lo][NPXMAX][LDH]
  constexpr int NIMAX = (NPXMAX * 8 + 255) / 256;        // 16-byte items per producer lane and strip
  constexpr int NS = 9, NBUF = 3, DIST = 2;              // strips per tile; register buffers of the producers' prefetch
  static_assert(NS % NBUF == 0, "register buffer of a strip must not depend on the tile");
  extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
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
  auto dil_of = [](int g) constexpr { return g == 0 ? D0 : (g == 1 ? D1 : D2); };

  if (wave < 4) {
    // =================================================================== producers: strips of o -> hi / lo in LDS
    const int ptid = tid;                                // 0..255
    const int piece = ptid & 7;
    u32x4 st[NBUF][NIMAX];
    auto issue = [&](const Tile& tl, bool on, auto sc) {
      constexpr int S = decltype(sc)::value, BUF = S % NBUF, g = S / 3, ky = S % 3;
      constexpr int d = dil_of(g), SW = TW + 2 * d, nitems = TH * SW * 8, NI = (nitems + 255) / 256;
      const __amdgpu_buffer_rsrc_t r = make_rsrc(sg.ptr + (long long)tl.b * H * W * sg.pix_stride, (unsigned)H * W * (unsigned)sg.pix_stride * 4u);
      const int ybase = tl.y0 + (ky - 1) * d, xbase = tl.x0 - d;
#pragma unroll
      for (int i = 0; i < NIMAX; ++i) {
        const int px = (ptid >> 3) + 32 * i;             // pixel of the strip, row-major over TH x SW
        const int rr = px / SW, cc = px - rr * SW;
        const int y = ybase + rr, x = xbase + cc;
        const bool ok = on && i < NI && px < TH * SW && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const int off = ok ? ((y * W + x) * (int)sg.pix_stride + sg.ch_off + piece * 4) * 4 : (int)OOB;
        st[BUF][i] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
      }
    };
    auto convert = [&](_Float16* buf, auto sc) {
      constexpr int S = decltype(sc)::value, BUF = S % NBUF, g = S / 3;
      constexpr int d = dil_of(g), SW = TW + 2 * d, nitems = TH * SW * 8, NI = (nitems + 255) / 256;
      _Float16* Shi = buf;
      _Float16* Slo = buf + NPXMAX * LDH;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int px = (ptid >> 3) + 32 * i;
        if (i < NI - 1 || px < TH * SW) {
          const f32x4 v = __builtin_bit_cast(f32x4, st[BUF][i]);
          const f32x2 x0 = {v[0] * a_scale, v[1] * a_scale}, x1 = {v[2] * a_scale, v[3] * a_scale};
          const h2 h0 = __builtin_convertvector(x0, h2), h1 = __builtin_convertvector(x1, h2);
          const h2 l0 = __builtin_convertvector(x0 - __builtin_convertvector(h0, f32x2), h2);
          const h2 l1 = __builtin_convertvector(x1 - __builtin_convertvector(h1, f32x2), h2);
          const h4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
          const int o = px * LDH + piece * 4;
          *(h4*)&Shi[o] = hi;
          *(h4*)&Slo[o] = lo;
        }
      }
    };
    // 9 strips per tile: the LDS buffer of a strip follows the GLOBAL strip counter q = 9 i + s (buffer q & 1)
    auto produce_strip = [&](int i, const Tile& tl, const Tile& nx, bool nx_on, auto sc) {
      constexpr int S = decltype(sc)::value;
      if constexpr (S + DIST < NS) issue(tl, true, std::integral_constant<int, S + DIST>{});
      else issue(nx, nx_on, std::integral_constant<int, S + DIST - NS>{});
      convert(ldsh + ((9 * i + S) & 1) * BUFH, sc);
    };

    // schedule: strip q is converted into its buffer during consumer step q-1 (the first one before the first barrier)
    if (nmine > 0) {
      const Tile t0 = decode(tile_at(0));
      issue(t0, true, std::integral_constant<int, 0>{});
      issue(t0, true, std::integral_constant<int, 1>{});
    }
    for (int i = 0; i < nmine; ++i) {
      const Tile tl = decode(tile_at(i));
      const bool nx_on = i + 1 < nmine;
      const Tile nx = decode(tile_at(nx_on ? i + 1 : i));
      [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
        ((produce_strip(i, tl, nx, nx_on, std::integral_constant<int, Ss>{}), lds_barrier()), ...);
      }(std::make_integer_sequence<int, NS>{});
    }
    lds_barrier();          // matches the consumers' last step
  } else {
    // =================================================================== consumers: 3 taps x 2 k-steps per strip
    const int cw = wave - 4;
    const unsigned frame_out = (unsigned)H * W * (unsigned)p.out_pix_stride * 4u;
    const unsigned frame_res = (unsigned)H * W * (unsigned)p.res_pix_stride * 4u;
    // fragment-order weights per group: [tap][k16 = 2][lane][8] halfs -> 1 KB per (tap, k16), 18 KB per group
    const unsigned wbytes = 3u * 9u * 2u * 1024u;
    const __amdgpu_buffer_rsrc_t rwh = make_rsrc(fhi, wbytes), rwl = make_rsrc(flo, wbytes);
    const int wlane = lane * 16;
    const int out_step = (int)p.out_pix_stride * 4, res_step = (int)p.res_pix_stride * 4;
    // weight ring: slot f holds the (kx, ks) = (f >> 1, f & 1) fragments of the CURRENT strip and is refilled with the
    // same fragment of the NEXT strip right after it has been read: six steps (36 MFMAs) of look-ahead in 48 registers
    u32x4 qh[6], ql[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      qh[f] = __builtin_amdgcn_raw_buffer_load_b128(rwh, wlane, f * 1024, 0);
      ql[f] = __builtin_amdgcn_raw_buffer_load_b128(rwl, wlane, f * 1024, 0);
    }
    f32x16 acc[2], sum[2];
    for (int i = 0; i < nmine; ++i) {
      const Tile tl = decode(tile_at(i));
      [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
        ((
          [&] {
            constexpr int S = Ss, g = S / 3, ky = S % 3;
            constexpr int d = dil_of(g), SW = TW + 2 * d;
            constexpr int SN = (S + 1) % NS, gn = SN / 3, kyn = SN % 3;
            lds_barrier();                               // strip q = 9 i + S is complete in buffer q & 1
            const _Float16* Shi = ldsh + ((9 * i + S) & 1) * BUFH;
            const _Float16* Slo = Shi + NPXMAX * LDH;
            if (S == 0) { sum[0] = (f32x16)(0.f); sum[1] = (f32x16)(0.f); }
            if (ky == 0) { acc[0] = (f32x16)(0.f); acc[1] = (f32x16)(0.f); }
            const int abase = ((cw * 2) * SW + li) * LDH + lh * 8;
#pragma unroll
            for (int f = 0; f < 6; ++f) {
              const int kx = f >> 1, ks = f & 1;
              const h8 bh = __builtin_bit_cast(h8, qh[f]), bl = __builtin_bit_cast(h8, ql[f]);
              {   // the same fragment of the next strip (after the last strip: the next tile's first one)
                const int o = ((gn * 9 + kyn * 3 + kx) * 2 + ks) * 1024;
                qh[f] = __builtin_amdgcn_raw_buffer_load_b128(rwh, wlane, o, 0);
                ql[f] = __builtin_amdgcn_raw_buffer_load_b128(rwl, wlane, o, 0);
                asm volatile("" ::: "memory");           // keep the refill here (see conv_fused_1x1_3x3_f16.hip)
              }
#pragma unroll
              for (int tm = 0; tm < 2; ++tm) {
                const int o = abase + (tm * SW + kx * d) * LDH + ks * 16;
                const h8 ah = *(const h8*)&Shi[o], al = *(const h8*)&Slo[o];
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[tm], 0, 0, 0);
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[tm], 0, 0, 0);
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[tm], 0, 0, 0);
              }
            }
            if (ky == 2) {                               // this dilation is complete: bias + ReLU, join the sum
              const float bv = p.bias ? p.bias[g * p.CoutP + li] : 0.f;
#pragma unroll
              for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum[tm][r] += fmaxf(acc[tm][r] * out_scale + bv, 0.f);
            }
          }()
        ), ...);
      }(std::make_integer_sequence<int, NS>{});

      // ---- epilogue: lane holds channel li of 16 pixels x = x_lane + c_r, c_r = (r&3) + 8*(r>>2), of tile row tm ----
      const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out + (long long)tl.b * H * W * p.out_pix_stride, frame_out);
      const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.residual + (long long)tl.b * H * W * p.res_pix_stride, frame_res);
      const int xl = tl.x0 + 4 * lh;
      const int cmax = xl < W ? W - xl : 0;
      const bool nok = li < p.Cout_store;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int y = tl.y0 + cw * 2 + tm;
        const int cm = (nok && y < H) ? cmax : 0;
        const int pix = y * W + xl;
        const unsigned o0 = (unsigned)((pix * (int)p.out_pix_stride + p.out_ch_off + li) * 4);
        const unsigned r0 = (unsigned)((pix * (int)p.res_pix_stride + p.res_ch_off + li) * 4);
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, (int)(c < cm ? r0 + c * res_step : OOB), 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum[tm][r] + rv[r]), rout, (int)(c < cm ? o0 + c * out_step : OOB), 0, 0);
        }
      }
    }
    lds_barrier();
  }
}

}  // namespace

// d: the grouped dilated convolution of an MSBlock exactly as egne_conv2d_f16x3_fwd takes it (ngroups = 3, 3x3, pad 1,
// dil = {4, 8, 12}, one raw 32-channel input slice, CoutP = 32, bias [3][32], act = ReLU, residual = the input, out).
// fhi / flo: egne_pack_conv_weight_f16frag per group, 9 * 32 * 32 halfs each, consecutive.
extern "C" int egne_msblock_dil_f16_fwd(const egne_conv_desc* dp, const void* fhi, const void* flo, float a_scale, float w_scale,
                                        void* stream) {
  EGNE_REQUIRE(dp && fhi && flo, "msblock_dil: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad_mode == 0 && d.ngroups == 3 && d.pad_h == 1 && d.pad_w == 1 &&
               d.dil[0] == 4 && d.dil[1] == 8 && d.dil[2] == 12 && d.Ho == d.H && d.Wo == d.W && d.nseg == 1 && d.CoutP == 32 &&
               d.Ktot == 32 && d.act == EGNE_ACT_RELU && !d.post_scale && d.residual && d.out, "msblock_dil: descriptor");
  const egne_seg& g = d.seg[0];
  EGNE_REQUIRE(g.ptr && !g.scale && !g.shift && g.act_in == EGNE_ACT_NONE && g.Cp == 32 && g.ch_off % 4 == 0 && g.pix_stride % 4 == 0 &&
               ((uintptr_t)g.ptr & 15) == 0 && (long long)d.H * d.W * g.pix_stride * 4 < (1ll << 31), "msblock_dil: input slice");
  EGNE_REQUIRE(d.Cout_store <= 32 && d.out_ch_off + d.Cout_store <= d.out_pix_stride && (long long)d.H * d.W * d.out_pix_stride * 4 < (1ll << 31) &&
               (long long)d.H * d.W * d.res_pix_stride * 4 < (1ll << 31), "msblock_dil: output / residual");
  EGNE_REQUIRE(((uintptr_t)fhi & 15) == 0 && ((uintptr_t)flo & 15) == 0 && a_scale > 0.f && w_scale > 0.f, "msblock_dil: weights / scales");
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y * d.B;
  constexpr size_t lds = (size_t)2 * 2 * TH * (TW + 24) * LDH * sizeof(_Float16);
  static bool once = hipFuncSetAttribute((const void*)msblock_dil_kernel<4, 8, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
  if (!once) return egne::fail(EGNE_ERR_LAUNCH, "msblock_dil: cannot raise the dynamic LDS limit to %zu", lds);
  int gx = 256;
  if (gx > ntiles) gx = ntiles;
  hipLaunchKernelGGL((msblock_dil_kernel<4, 8, 12>), dim3(gx), dim3(512), lds, (hipStream_t)stream, d, (const _Float16*)fhi,
                     (const _Float16*)flo, a_scale, 1.0f / (a_scale * w_scale), tiles_x, tiles_y, ntiles);
  return egne::check_launch("egne_msblock_dil_f16_fwd");
}
