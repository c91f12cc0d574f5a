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
//             three taps arriving one strip ahead through a register ring, one accumulator pair per dilation; after the
//             ninth strip bias + ReLU per dilation, the three are summed, the exact fp32 o is added and the tile stored.
// One s_barrier per strip.  The conversion work (each element is split 9 times) sits in waves that do nothing else.
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
__device__ int g_mdbg = 0;
__device__ unsigned long long g_mstamps[256 * 8 * 4];

constexpr int TW = 32, TH = 8;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// XIN: this launch holds only tiles whose widest strip lies inside the image columns (x0 >= 12, x0 + 32 + 12 <= W): item addresses
// are then a per-lane constant + a wave-uniform term.  A map is covered by an XIN launch over the interior tile columns and a
// !XIN launch over the two border columns (cols: 0 = all tile columns, 1 = interior, 2 = the two border columns).
// PS: the input (= the residual) is held in SPLIT-PAIR storage (egne_conv_desc.out_split of its producer, written with a_scale):
// per pixel [hi x 32 | lo x 32] halves, so a producer item is a 16-byte COPY into the operand image (no conversion at all -- the
// fp32 form splits every element 13.5 times per tile) and the epilogue recovers o = (hi + lo) * inv_a.
template <int D0, int D1, int D2, int NPW, bool XIN, bool PS>       // NPW: producer waves (4 or 8); 4 consumer waves follow them
__global__ __launch_bounds__(64 * (NPW + 4))
void msblock_dil_kernel(const egne_conv_desc p, const _Float16* __restrict__ fhi, const _Float16* __restrict__ flo, float a_scale, float inv_a,
                        float out_scale, int tiles_x, int tiles_y, int ntiles, int cols, const float* __restrict__ score_w,
                        const float* __restrict__ score_c, float* __restrict__ s0, float* __restrict__ s1, int accumulate) {
  constexpr int DMAX = D2 > D1 ? (D2 > D0 ? D2 : D0) : (D1 > D0 ? D1 : D0);
  constexpr int SWMAX = TW + 2 * DMAX, NPXMAX = TH * SWMAX;
  // LDS: two strip buffers [hi | lo][NPXMAX][32 halfs] -- 64 bytes per pixel and half, NO padding: the 16-byte chunk c of
  // pixel q sits at chunk c ^ ((q >> 2) & 3), which makes the consumers' ds_read_b128 (32 consecutive pixels, one chunk)
  // and the producers' ds_write_b64 conflict-free -- then two weight buffers [12 fragments][64 lanes][8 halfs].
  constexpr int BUFH = 2 * NPXMAX * 32;                  // halfs per strip buffer
  constexpr int WBUFH = 12 * 512;                        // halfs per weight buffer: (kx, ks) x (hi, lo) fragments of one strip
  constexpr int NS = 9, NBUF = 3, DIST = 2;              // strips per tile; register buffers of the producers' prefetch
  static_assert(NS % NBUF == 0, "register buffer of a strip must not depend on the tile");
  extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
  _Float16* const lw = ldsh + 2 * BUFH;

  const int dbg = g_mdbg;
  unsigned long long t_work = 0, t_wait = 0, t_pa = 0, t_pb = 0, t_pc = 0, t_last = __builtin_amdgcn_s_memtime(), r_first = __builtin_amdgcn_s_memrealtime();
  auto stamp = [&](unsigned long long& accum) {
    if (dbg & 64) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      accum += t - t_last; t_last = t;
    }
  };
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
    const int ntx = cols == 0 ? tiles_x : (cols == 1 ? tiles_x - 2 : 2);
    const int tq = t % ntx; t /= ntx;
    const int tx = cols == 0 ? tq : (cols == 1 ? tq + 1 : tq * (tiles_x - 1));
    const int ty = t % tiles_y; t /= tiles_y;
    r.b = t; r.y0 = ty * TH; r.x0 = tx * TW;
    return r;
  };
  int nmine = 0;
  while (tile_at(nmine) < ntiles) ++nmine;
  auto dil_of = [](int g) constexpr { return g == 0 ? D0 : (g == 1 ? D1 : D2); };

  if (wave < NPW) {
    // =================================================================== producers: strips of o -> hi / lo in LDS
    const int ptid = tid;                                // 0 .. 64 NPW - 1
    constexpr int PPI = 8 * NPW;                         // pixels per item round (eight lanes per pixel)
    const int piece = ptid & 7;
    // strips are ordered ky-major (S = 3 ky + g), so register buffer S % 3 = g only ever holds strips of dilation g:
    // 10 + 12 + 14 items instead of 3 x 14 (the difference decided between spilling and not spilling)
    constexpr int NI0 = (TH * (TW + 2 * D0) + PPI - 1) / PPI, NI1 = (TH * (TW + 2 * D1) + PPI - 1) / PPI, NI2 = (TH * (TW + 2 * D2) + PPI - 1) / PPI;
    constexpr int WPW = 12 / NPW + (12 % NPW != 0);     // weight fragments per producer wave (waves past 12 / WPW repeat earlier ones)
    u32x4 st0[NI0], st1[NI1], st2[NI2], wr[NBUF][WPW];
    // byte offset of item I of a dilation-g strip relative to the tile's first pixel, ky = 1 (tile- and ky-invariant; the rest of
    // the address is wave-uniform: one v_add per item instead of a division, two multiplies and the bounds arithmetic)
    int rel0[XIN ? NI0 : 1], rel1[XIN ? NI1 : 1], rel2[XIN ? NI2 : 1];
    auto relbuf = [&](auto gc) -> int* { constexpr int Gq = decltype(gc)::value; if constexpr (Gq == 0) return rel0; else if constexpr (Gq == 1) return rel1; else return rel2; };
    if constexpr (XIN) [&]<int... Gs>(std::integer_sequence<int, Gs...>) {
      (([&] {
        constexpr int g = Gs, d = dil_of(g), SW = TW + 2 * d, NIg = (TH * SW + PPI - 1) / PPI;
#pragma unroll
        for (int I = 0; I < NIg; ++I) {
          const int px = (ptid >> 3) + PPI * I;
          const int rr = px / SW, cc = px - rr * SW;
          relbuf(std::integral_constant<int, g>{})[I] = (TH * SW % PPI != 0 && px >= TH * SW) ? (int)OOB
                                                        : ((rr * W + cc - d) * (int)sg.pix_stride + sg.ch_off + piece * 4) * 4;
        }
      }()), ...);
    }(std::make_integer_sequence<int, 3>{});
    auto stbuf = [&](auto bc) -> u32x4* { constexpr int Bq = decltype(bc)::value; if constexpr (Bq == 0) return st0; else if constexpr (Bq == 1) return st1; else return st2; };
    const unsigned wbytes = 3u * 9u * 2u * 1024u;
    const __amdgpu_buffer_rsrc_t rwh = make_rsrc(fhi, wbytes), rwl = make_rsrc(flo, wbytes);

    // item i of strip S: one 16-byte load (4 channels of one pixel; eight lanes cover a pixel).  Rows outside the image
    // fall outside the per-frame buffer resource (negative or too large an offset) and load zeros by themselves; columns
    // need a test only in tiles that touch the left / right border (XIN = false): the common path is pure address math.
    auto issue1 = [&](const Tile& tl, bool on, int pg, bool xin, auto sc, auto ic) {
      constexpr int S = decltype(sc)::value, I = decltype(ic)::value, BUF = S % NBUF, g = S % 3, ky = S / 3;
      constexpr int d = dil_of(g), SW = TW + 2 * d;
      const __amdgpu_buffer_rsrc_t r = make_rsrc(sg.ptr + (long long)tl.b * H * W * sg.pix_stride, (unsigned)H * W * (unsigned)sg.pix_stride * 4u);
      int off;
      if constexpr (XIN) {                               // interior columns: rows outside the image fall outside the resource
        const int sbase = ((tl.y0 + (ky - 1) * d) * W + tl.x0) * (int)sg.pix_stride * 4;       // wave-uniform
        const int rl = relbuf(std::integral_constant<int, g>{})[I];
        off = rl == (int)OOB ? (int)OOB : rl + sbase;
      } else {
        const int px = pg + PPI * I;                     // pixel of the strip, row-major over TH x SW
        const int rr = px / SW, cc = px - rr * SW;
        const int y = tl.y0 + (ky - 1) * d + rr, x = tl.x0 - d + cc;
        off = ((y * W + x) * (int)sg.pix_stride + sg.ch_off + piece * 4) * 4;
        off = ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) ? off : (int)OOB;
        if (TH * SW % PPI != 0 && px >= TH * SW) off = (int)OOB;
      }
      stbuf(std::integral_constant<int, BUF>{})[I] = __builtin_amdgcn_raw_buffer_load_b128(r, (on && !(dbg & 1)) ? off : (int)OOB, 0, 0);
    };
    // LDS slot of item I: pixel pg + PPI I, and (pixel >> 2) & 3 = (pg >> 2) & 3 for every I: lane constant + 64 B * PPI * I
    const int lofs = PS ? (piece >> 2) * NPXMAX * 32 + (ptid >> 3) * 32 + (((piece & 3) ^ ((ptid >> 5) & 3)) << 3)      // piece = (plane, 16-byte chunk)
                        : (ptid >> 3) * 32 + (((piece >> 1) ^ ((ptid >> 5) & 3)) << 3) + ((piece & 1) << 2);
    static_assert(PPI % 4 == 0, "the swizzle key of an item must not depend on I");
    auto convert1 = [&](_Float16* buf, auto sc, auto ic) {
      constexpr int S = decltype(sc)::value, I = decltype(ic)::value, BUF = S % NBUF, g = S % 3;
      constexpr int d = dil_of(g), SW = TW + 2 * d;
      if constexpr (PS) {
        if (TH * SW % PPI == 0 || (ptid >> 3) + PPI * I < TH * SW)
          *(u32x4*)&buf[lofs + 32 * PPI * I] = stbuf(std::integral_constant<int, BUF>{})[I];
      } else if (TH * SW % PPI == 0 || (ptid >> 3) + PPI * I < TH * SW) {
        const f32x4 v = __builtin_bit_cast(f32x4, stbuf(std::integral_constant<int, BUF>{})[I]);
        h2 h0, h1, l0, l1;                        // x * a_scale = hi + lo, plain (unpacked) VALU: split_f16.h
        egne::split2(v[0], v[1], a_scale, h0, l0);
        egne::split2(v[2], v[3], a_scale, h1, l1);
        const h4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
        const int o = lofs + 32 * PPI * I;
        *(h4*)&buf[o] = hi;
        *(h4*)&buf[NPXMAX * 32 + o] = lo;
      }
    };
    // 12 weight fragments per strip ((kx, ks) x (hi, lo)); producer wave w fetches fragments 3w .. 3w + 2
    auto issue_w = [&](auto sc) {
      constexpr int S = decltype(sc)::value, BUF = S % NBUF, g = S % 3, ky = S / 3;
#pragma unroll
      for (int t = 0; t < WPW; ++t) {
        const int j = (wave * WPW + t) % 12, f = j >> 1;
        const int o = ((g * 9 + ky * 3 + (f >> 1)) * 2 + (f & 1)) * 1024 + lane * 16;
        wr[BUF][t] = (j & 1) ? __builtin_amdgcn_raw_buffer_load_b128(rwl, o, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rwh, o, 0, 0);
      }
    };
    // One step: the loads of strip S + DIST go out INTERLEAVED with the conversion of strip S -- the 16-byte-per-lane
    // gathers are paced by the 64 B/clk vector memory path (measured: 14 back-to-back loads stall ~2000 cycles), which the
    // conversion's VALU and LDS work now hides.  9 strips per tile: the LDS buffer follows the global strip count q = 9i+S.
    auto produce_strip = [&](int i, const Tile& tl, const Tile& nx, bool nx_on, auto sc) {
      constexpr int S = decltype(sc)::value, SI = (S + DIST) % NS, gi = SI % 3, gc = S % 3;
      constexpr int NIi = (TH * (TW + 2 * dil_of(gi)) + PPI - 1) / PPI, NIc = (TH * (TW + 2 * dil_of(gc)) + PPI - 1) / PPI;
      constexpr int NIm = NIi > NIc ? NIi : NIc;
      const Tile& ti = (S + DIST < NS) ? tl : nx;
      const bool oni = (S + DIST < NS) ? true : nx_on;
      const bool xin = ti.x0 >= DMAX && ti.x0 + TW + DMAX <= W;        // wave-uniform
      _Float16* buf = ldsh + ((9 * i + S) & 1) * BUFH;
      int pg = ptid >> 3;
      asm volatile("" : "+v"(pg));                       // opaque: keeps the tile-invariant (row, column) pairs from being hoisted
                                                         // out of the tile loop into 72 live registers (= spills)
      issue_w(std::integral_constant<int, SI>{});
      stamp(t_pa);
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (([&] {
          if constexpr (Is < NIi) issue1(ti, oni, pg, xin, std::integral_constant<int, SI>{}, std::integral_constant<int, Is>{});
          if constexpr (Is < NIc) { if (!(dbg & 2)) convert1(buf, sc, std::integral_constant<int, Is>{}); }
        }()), ...);
      }(std::make_integer_sequence<int, NIm>{});
      stamp(t_pb);
      _Float16* wb = lw + ((9 * i + S) & 1) * WBUFH;
#pragma unroll
      for (int t = 0; t < WPW; ++t) *(u32x4*)&wb[(((wave * WPW + t) % 12) * 64 + lane) * 8] = wr[S % NBUF][t];
      stamp(t_pc);
    };

    auto issue_all = [&](const Tile& tl, auto sc) {      // whole strip at once: only before the first tile
      constexpr int S = decltype(sc)::value, NIs = (TH * (TW + 2 * dil_of(S % 3)) + PPI - 1) / PPI;
      issue_w(sc);
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (issue1(tl, true, ptid >> 3, false, sc, std::integral_constant<int, Is>{}), ...);
      }(std::make_integer_sequence<int, NIs>{});
    };
    if (nmine > 0) {
      const Tile t0 = decode(tile_at(0));
      issue_all(t0, std::integral_constant<int, 0>{});
      issue_all(t0, std::integral_constant<int, 1>{});
    }
    for (int i = 0; i < nmine; ++i) {
      const Tile tl = decode(tile_at(i));
      const bool nx_on = i + 1 < nmine;
      const Tile nx = decode(tile_at(nx_on ? i + 1 : i));
      [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
        ((produce_strip(i, tl, nx, nx_on, std::integral_constant<int, Ss>{}), lds_barrier(), stamp(t_wait)), ...);
      }(std::make_integer_sequence<int, NS>{});
    }
    lds_barrier();          // matches the consumers' last step
  } else {
    // =================================================================== consumers: 3 taps x 2 k-steps per strip
    if (dbg & 128) __builtin_amdgcn_s_setprio(1);
    const int cw = wave - NPW;
    const unsigned frame_out = (unsigned)H * W * (unsigned)p.out_pix_stride * 4u;
    const unsigned frame_res = (unsigned)H * W * (unsigned)p.res_pix_stride * 4u;
    const int out_step = (int)p.out_pix_stride * 4, res_step = (int)p.res_pix_stride * 4;
    f32x16 acc[3][2];                                    // one accumulator pair per dilation (strips arrive ky-major)
    for (int i = 0; i < nmine; ++i) {
      const Tile tl = decode(tile_at(i));
      [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
        ((
          [&] {
            constexpr int S = Ss, g = S % 3, ky = S / 3;
            constexpr int d = dil_of(g), SW = TW + 2 * d;
            stamp(t_work);
            lds_barrier();                               // strip q = 9 i + S and its weights are complete in buffers q & 1
            stamp(t_wait);
            const _Float16* Shi = ldsh + ((9 * i + S) & 1) * BUFH;
            const _Float16* Slo = Shi + NPXMAX * 32;
            const _Float16* wb = lw + ((9 * i + S) & 1) * WBUFH + lane * 8;
            if (ky == 0) { acc[g][0] = (f32x16)(0.f); acc[g][1] = (f32x16)(0.f); }
            // software pipeline: the fragments of step f + 1 (weights and both rows' operands) are requested from LDS before the
            // MFMAs of step f are issued -- one consumer wave per SIMD has nothing else to cover the LDS latency with
            auto offs = [&](int f, int tm) {
              const int kx = f >> 1, ks = f & 1;
              const int q = (cw * 2 + tm) * SW + li + kx * d;
              return q * 32 + (((ks * 2 + lh) ^ ((q >> 2) & 3)) << 3);
            };
            h8 bh[2], bl[2], ah[2][2], al[2][2];
            bh[0] = *(const h8*)&wb[0]; bl[0] = *(const h8*)&wb[512];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) { ah[0][tm] = *(const h8*)&Shi[offs(0, tm)]; al[0][tm] = *(const h8*)&Slo[offs(0, tm)]; }
#pragma unroll
            for (int f = 0; f < 6; ++f) {
              if (f + 1 < 6) {
                bh[(f + 1) & 1] = *(const h8*)&wb[(2 * f + 2) * 512]; bl[(f + 1) & 1] = *(const h8*)&wb[(2 * f + 3) * 512];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                  ah[(f + 1) & 1][tm] = *(const h8*)&Shi[offs(f + 1, tm)];
                  al[(f + 1) & 1][tm] = *(const h8*)&Slo[offs(f + 1, tm)];
                }
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int tm = 0; tm < 2; ++tm) {
                acc[g][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[f & 1][tm], bh[f & 1], acc[g][tm], 0, 0, 0);
                acc[g][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[f & 1][tm], bl[f & 1], acc[g][tm], 0, 0, 0);
                acc[g][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[f & 1][tm], bh[f & 1], acc[g][tm], 0, 0, 0);
              }
              __builtin_amdgcn_sched_barrier(0);
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
      bool bad = false;
      const float b0 = p.bias ? p.bias[li] : 0.f, b1 = p.bias ? p.bias[p.CoutP + li] : 0.f, b2 = p.bias ? p.bias[2 * p.CoutP + li] : 0.f;
      // optional: this block's share of the stage's two score maps (bdcn_new.py:118-166: 1x1 "down" conv 32 -> 21, summed over
      // the stage's blocks, then the two 21 -> 1 heads -- all linear, so per block and head ONE 32-vector, score_w[2][32])
      const float cw0 = score_w ? score_w[li] : 0.f, cw1 = score_w ? score_w[32 + li] : 0.f;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int y = tl.y0 + cw * 2 + tm;
        const int cm = (nok && y < H) ? cmax : 0;
        const int pix = y * W + xl;
        const unsigned o0 = (unsigned)((pix * (int)p.out_pix_stride + p.out_ch_off + li) * 4);
        // (split-pair storage keeps channel li at position 8 * ((li >> 2) & 3) + 4 * (li >> 4) + (li & 3) of either plane)
        const unsigned r0 = PS ? (unsigned)((pix * (int)p.res_pix_stride + p.res_ch_off) * 4 + (8 * ((li >> 2) & 3) + 4 * (li >> 4) + (li & 3)) * 2)
                               : (unsigned)((pix * (int)p.res_pix_stride + p.res_ch_off + li) * 4);
        float rv[16], sc[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          if constexpr (PS) {
            const unsigned short uh = __builtin_amdgcn_raw_buffer_load_b16(rres, (int)(c < cm ? r0 + c * res_step : OOB), 0, 0);
            const unsigned short ul = __builtin_amdgcn_raw_buffer_load_b16(rres, (int)(c < cm ? r0 + c * res_step : OOB), 64, 0);
            rv[r] = ((float)__builtin_bit_cast(_Float16, uh) + (float)__builtin_bit_cast(_Float16, ul)) * inv_a;
          } else {
            rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, (int)(c < cm ? r0 + c * res_step : OOB), 0, 0));
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const float v = fmaxf(acc[0][tm][r] * out_scale + b0, 0.f) + fmaxf(acc[1][tm][r] * out_scale + b1, 0.f) +
                          fmaxf(acc[2][tm][r] * out_scale + b2, 0.f) + rv[r];      // o + o1 + o2 + o3 (bdcn_new.py:54)
          // (fmaxf(x, 0) swallows NaN and -inf: the overflow test looks at the accumulators themselves, egne_conv_desc.ovf_flag)
          bad |= egne_nonfinite(acc[0][tm][r] + acc[1][tm][r] + acc[2][tm][r]) | egne_nonfinite(rv[r]);
          if (p.out) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, (int)(c < cm ? o0 + c * out_step : OOB), 0, 0);
          sc[r] = v * cw0; sc[16 + r] = v * cw1;
        }
        if (score_w) {
          // sum over the 32 channels (= lanes of one half): transpose-reduce, 31 exchanges for the 32 (head, pixel) values.  The
          // partner of a lane is chosen per stage so that 30 of the 31 exchanges are DPP moves (4 cycles of VALU, no LDS
          // crossbar and no wait): lane ^ 1 and lane ^ 2 (quad_perm), lane ^ 7 (row_half_mirror), lane ^ 15 (row_mirror);
          // only the last, single exchange crosses the 16-lane row (lane ^ 16).  A chain of 31 ds_bpermute round trips took
          // ~7 k of the consumer's 21 k cycles per tile.  Stage k halves the value index (bit VB) by lane bit LB, so
          // lane li ends up with value index v = ((li & 15) << 1) | (li >> 4): (head v >> 4, pixel register v & 15).
          auto stage = [&](auto vbc, auto lbc, auto ctlc) {
            constexpr int VB = decltype(vbc)::value, LB = decltype(lbc)::value, CTL = decltype(ctlc)::value;
            const bool up = (li & LB) != 0;
#pragma unroll
            for (int j = 0; j < VB; ++j) {
              const float send = up ? sc[j] : sc[j + VB], keep = up ? sc[j + VB] : sc[j];
              float got;
              if constexpr (CTL >= 0) got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), CTL, 0xf, 0xf, false));
              else got = __shfl_xor(send, 16);
              sc[j] = keep + got;
            }
          };
          using IC = std::integral_constant<int, 0>;
          // a partner must hold the SAME value subset (agree in the lane bits of earlier stages) and a disjoint set of summed lanes:
          // the mirrors (which flip all lower bits) therefore come first
          stage(std::integral_constant<int, 16>{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 0x140>{});     // lane ^ 15: row_mirror
          stage(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 0x141>{});      // lane ^ 7: row_half_mirror
          stage(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 0x4E>{});       // lane ^ 2: quad_perm [2,3,0,1]
          stage(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 0xB1>{});       // lane ^ 1: quad_perm [1,0,3,2]
          stage(std::integral_constant<int, 1>{}, std::integral_constant<int, 16>{}, std::integral_constant<int, -1>{});        // lane ^ 16: across the rows
          (void)sizeof(IC);
          const int vi = ((li & 15) << 1) | (li >> 4);      // value index of this lane: bits (8, 4, 2, 1, 16) of li select value bits (16, 8, 4, 2, 1)
          const int r = vi & 15, c = (r & 3) + 8 * (r >> 2), x = xl + c;
          if (y < H && x < W) {
            float* dstp = ((vi >> 4) ? s1 : s0) + ((long long)tl.b * H + y) * W + x;
            const float t = sc[0] + (accumulate ? *dstp : score_c[vi >> 4]);
            *dstp = t;
          }
        }
      }
      egne_ovf_commit(bad, p.ovf_flag);
    }
    stamp(t_work);
    lds_barrier();
  }
  if ((dbg & 64) && lane == 0) {
    unsigned long long* o = g_mstamps + ((long long)blockIdx.x * 8 + wave) * 4;
    o[0] = wave < NPW ? t_pa : t_work; o[1] = wave < NPW ? t_pb : t_wait; o[2] = nmine; o[3] = wave < NPW ? t_pc : 0;
    if (wave < NPW) o[2] |= (unsigned long long)(t_wait / (nmine ? nmine : 1)) << 32;
  }
}

}  // namespace

// d: the grouped dilated convolution of an MSBlock exactly as egne_conv2d_f16x3_fwd takes it (ngroups = 3, 3x3, pad 1,
// dil = {4, 8, 12}, one raw 32-channel input slice, CoutP = 32, bias [3][32], act = ReLU, residual = the input, out).
// fhi / flo: egne_pack_conv_weight_f16frag per group, 9 * 32 * 32 halfs each, consecutive.
extern "C" int egne_msblock_dil_scores_f16_fwd(const egne_conv_desc* dp, const void* fhi, const void* flo, float a_scale, float w_scale,
                                               const float* score_w, const float* score_c, float* s0, float* s1, int accumulate,
                                               void* stream) {
  EGNE_REQUIRE(dp && fhi && flo, "msblock_dil: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad_mode == 0 && d.ngroups == 3 && d.pad_h == 1 && d.pad_w == 1 &&
               d.dil[0] == 4 && d.dil[1] == 8 && d.dil[2] == 12 && d.Ho == d.H && d.Wo == d.W && d.nseg == 1 && d.CoutP == 32 &&
               d.Ktot == 32 && d.act == EGNE_ACT_RELU && !d.post_scale && d.residual && (d.out || score_w), "msblock_dil: descriptor");
  const egne_seg& g = d.seg[0];
  EGNE_REQUIRE(g.ptr && !g.scale && !g.shift && g.act_in == EGNE_ACT_NONE && g.Cp == 32 && g.ch_off % 4 == 0 && g.pix_stride % 4 == 0 &&
               ((uintptr_t)g.ptr & 15) == 0 && (long long)d.H * d.W * g.pix_stride * 4 < (1ll << 31), "msblock_dil: input slice");
  EGNE_REQUIRE(d.Cout_store <= 32 && (!d.out || (d.out_ch_off + d.Cout_store <= d.out_pix_stride && (long long)d.H * d.W * d.out_pix_stride * 4 < (1ll << 31))) &&
               (long long)d.H * d.W * d.res_pix_stride * 4 < (1ll << 31), "msblock_dil: output / residual");
  EGNE_REQUIRE(((uintptr_t)fhi & 15) == 0 && ((uintptr_t)flo & 15) == 0 && a_scale > 0.f && w_scale > 0.f, "msblock_dil: weights / scales");
  EGNE_REQUIRE(!score_w || (score_c && s0 && s1 && d.Cout_store == 32), "msblock_dil: score maps need score_c, s0, s1 and all 32 channels");
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y * d.B;
  constexpr size_t lds = ((size_t)2 * 2 * TH * (TW + 24) * 32 + 2 * 12 * 512) * sizeof(_Float16);
  static bool once = hipFuncSetAttribute((const void*)msblock_dil_kernel<4, 8, 12, 4, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                     hipFuncSetAttribute((const void*)msblock_dil_kernel<4, 8, 12, 4, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                     hipFuncSetAttribute((const void*)msblock_dil_kernel<4, 8, 12, 4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                     hipFuncSetAttribute((const void*)msblock_dil_kernel<4, 8, 12, 4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
  if (!once) return egne::fail(EGNE_ERR_LAUNCH, "msblock_dil: cannot raise the dynamic LDS limit to %zu", lds);
  const float os = 1.0f / (a_scale * w_scale), inv_a = 1.0f / a_scale;
  const bool ps = g.presplit != 0;
  // split-pair input: the residual must be that same slice (bdcn_new.py:54 adds `o` itself), whole 32-channel blocks
  EGNE_REQUIRE(!ps || (d.residual == g.ptr && d.res_pix_stride == g.pix_stride && d.res_ch_off == g.ch_off && g.ch_off % 32 == 0),
               "msblock_dil: a split-pair input must also be the residual");
  static const bool ps_sym = getenv("EGNE_MSDIL_PS_OLD") == nullptr;      // (diagnostics: the producer / consumer kernel with copying producers)
  if (ps && ps_sym) return egne::msdil_ps_launch(d, fhi, flo, a_scale, w_scale, score_w, score_c, s0, s1, accumulate, (hipStream_t)stream);
  auto launch = [&](auto xin, int grid, int nt, int cols) {
    constexpr bool XIN = decltype(xin)::value;
    if (ps) hipLaunchKernelGGL((msblock_dil_kernel<4, 8, 12, 4, XIN, true>), dim3(grid), dim3(512), lds, (hipStream_t)stream, d, (const _Float16*)fhi,
                               (const _Float16*)flo, a_scale, inv_a, os, tiles_x, tiles_y, nt, cols, score_w, score_c, s0, s1, accumulate);
    else hipLaunchKernelGGL((msblock_dil_kernel<4, 8, 12, 4, XIN, false>), dim3(grid), dim3(512), lds, (hipStream_t)stream, d, (const _Float16*)fhi,
                            (const _Float16*)flo, a_scale, inv_a, os, tiles_x, tiles_y, nt, cols, score_w, score_c, s0, s1, accumulate);
  };
  // interior tile columns (x0 >= 12 and x0 + 44 <= W <=> tile column 1 .. tiles_x - 2 when W >= 32 * (tiles_x - 1) + 12) on the
  // fast-address kernel, the two border columns (or everything on narrow maps) on the checked one
  const bool split = tiles_x > 2 && d.W >= TW * (tiles_x - 1) + 12;
  if (split) {
    const int nt_in = (tiles_x - 2) * tiles_y * d.B, nt_b = 2 * tiles_y * d.B;
    launch(std::true_type{}, nt_in < 256 ? nt_in : 256, nt_in, 1);
    static const bool only_interior = getenv("EGNE_MSDIL_ONLY_INTERIOR") != nullptr;      // diagnostics: stamps of the interior launch
    if (!only_interior) launch(std::false_type{}, nt_b < 256 ? nt_b : 256, nt_b, 2);
  } else {
    int gx = 256;
    if (gx > ntiles) gx = ntiles;
    launch(std::false_type{}, gx, ntiles, 0);
  }
  return egne::check_launch("egne_msblock_dil_f16_fwd");
}

extern "C" int egne_msblock_dil_f16_fwd(const egne_conv_desc* dp, const void* fhi, const void* flo, float a_scale, float w_scale,
                                        void* stream) {
  return egne_msblock_dil_scores_f16_fwd(dp, fhi, flo, a_scale, w_scale, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int egne_msdil_debug(int dbg, void* out_stamps) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_mdbg), &dbg, sizeof(int)) != hipSuccess) return -2;
  if (out_stamps && hipMemcpyFromSymbol(out_stamps, HIP_SYMBOL(g_mstamps), sizeof(unsigned long long) * 256 * 8 * 4) != hipSuccess) return -2;
  return 0;
}
