// The dilated branch of a BDCN MSBlock as ONE launch, for an input held in SPLIT-PAIR storage (egne_conv_desc.out_split of its
// producer: per pixel [hi x 32 | lo x 32] f16 halves of o * a_scale, channel order engine.SPLIT_PAIR_PERM):
//   out = o + sum_g relu(conv3x3_{dil g}(o) + b_g), g = 0..2, dilations 4 / 8 / 12 (bdcn_new.py:51-54),
// on the split-f16 MFMA path (three v_mfma_f32_16x16x32_f16 per product, fp32 accumulate; numerics: conv_f16x3.hip).
//
// msblock_dil_f16.hip splits its eight waves into four producers (gather + fp32 -> hi / lo conversion) and four consumers (MFMA).
// With the conversion gone a producer item is a 16-byte copy and the four consumer waves are what the tile waits for (stamps,
// scratch/msdil_ps_stamps.py: 24.7 k cycles of consumer work per tile for 10.4 k cycles of MFMA issue, the producers idle at the
// barriers 40 % of the time): ONE MFMA-issuing wave per SIMD has nothing to cover its LDS latencies, its dependent accumulator chains
// and its epilogue with.  Here all EIGHT waves are symmetric: every wave copies its share of the next strips (global -> registers ->
// LDS, two strips ahead) AND owns one of the tile's eight rows for all three dilations, so that each SIMD has two MFMA-issuing waves
// that fill each other's gaps.  The product is transposed as in conv3x3_rw_f16.hip (weights as the A operand of
// v_mfma_f32_16x16x32_f16: four independent accumulators per dilation, a lane ends with 4 consecutive channels of one pixel), which
// turns the epilogue into 16-byte accesses: the residual o = (hi + lo) / a_scale is two 16-byte loads per 16 pixels, the score
// heads reduce 8 channels in the lane and the rest over the lane's three partners.
//
// Per tile (8 x 32 pixels) nine strips, ky-major (S = 3 ky + g): the 8 x (32 + 2 d_g) pixels that hold the three kx taps of kernel row
// ky of dilation g, in one of two LDS strip buffers [hi | lo][448 pixels][32 halfs] -- 16-byte chunk c of pixel q at c ^ ((q >> 1) & 3),
// conflict-free for the 16-pixel x 4-chunk ds_read_b128 pattern of the 16x16x32 operand; the lo plane starts 64 bytes past a
// multiple of 128 so that the eight lanes of a pixel (4 hi + 4 lo chunks) write 128 distinct bank bytes -- and the strip's 12 weight
// fragments in one of two weight buffers.  One s_barrier per strip.
#include "common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

namespace {
__device__ int g_pdbg = 0;
__device__ unsigned long long g_pstamps[256 * 8 * 4];

constexpr int TW = 32, TH = 8;
constexpr int D0 = 4, D1 = 8, D2 = 12, DMAX = 12;
constexpr int SWMAX = TW + 2 * DMAX, NPXMAX = TH * SWMAX;      // 56, 448
constexpr int PLANE = NPXMAX * 32 + 32;                        // halfs per plane: the lo plane sits 64 bytes past a multiple of 128
constexpr int BUFH = 2 * PLANE;                                // halfs per strip buffer
constexpr int WBUFH = 12 * 512;                                // halfs per weight buffer: [kx 3][ks 2][hi | lo][64 lanes][8]
constexpr int NS = 9;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

constexpr int dil_of(int g) { return g == 0 ? D0 : (g == 1 ? D1 : D2); }
constexpr int ni_of(int g) { return TH * (TW + 2 * dil_of(g)) * 8 / 512; }     // 16-byte items per lane and strip: 5 / 6 / 7 (exact)

// XIN: this launch holds only tiles whose widest strip lies inside the image columns (x0 >= 12, x0 + 32 + 12 <= W): item addresses are
// then a per-lane constant + a wave-uniform term (rows outside the image fall outside the per-frame buffer resource and read zeros).
// cols: 0 = all tile columns, 1 = interior, 2 = the two border columns (as msblock_dil_f16.hip).
// NP: products per multiply (egne_conv_desc.f16_products): 3 = hi hi + hi lo + lo hi; 1 = hi hi only -- the lo plane of the input is
// then not even copied (its pieces read zeros through the out-of-range offset), which halves what the copy side pulls from L2
template <bool XIN, int NP = 3>
__global__ __launch_bounds__(512)
void msdil_ps_kernel(const egne_conv_desc p, const _Float16* __restrict__ fhi, const _Float16* __restrict__ flo, float inv_a, float out_scale,
                     int tiles_x, int tiles_y, int ntiles, int cols, const float* __restrict__ score_w, const float* __restrict__ score_c,
                     float* __restrict__ s0, float* __restrict__ s1, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
  _Float16* const lw = ldsh + 2 * BUFH;
  float* const lconst = (float*)(lw + 3 * WBUFH);             // [3][32] biases, [2][32] score vectors, [2] constants: the epilogue issues no vector memory loads
  int* const lrel = (int*)(lconst + 162);                     // XIN: [18 items][64 pixel groups] item offsets relative to the strip's first pixel

  const int dbg = g_pdbg;
  unsigned long long t_work = 0, t_wait = 0, t_last = __builtin_amdgcn_s_memtime();
  auto stamp = [&](unsigned long long& accum) {
    if (dbg & 64) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      accum += t - t_last; t_last = t;
    }
  };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = the tile row this wave owns
  const int l15 = lane & 15, kg = lane >> 4;
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
  if (tid < 160) {
    const int r = tid >> 5, c = tid & 31;
    lconst[tid] = r < 3 ? (p.bias ? p.bias[r * p.CoutP + c] : 0.f) : (score_w ? score_w[(r - 3) * 32 + c] : 0.f);
  } else if (tid < 162) {
    lconst[tid] = score_c ? score_c[tid - 160] : 0.f;
  }

  // ---- copy side: strip items.  Item I of a lane: pixel (tid >> 3) + 64 I of the strip (row-major over 8 x SW), 16-byte piece tid & 7
  // of its 128 bytes: pieces 0-3 = chunks of the hi plane, 4-7 = of the lo plane.
  const int piece = tid & 7, pg = tid >> 3;
  constexpr int NI0 = ni_of(0), NI1 = ni_of(1), NI2 = ni_of(2);
  u32x4 st0[NI0], st1[NI1], st2[NI2];                          // register buffer S % 3 = g holds strips of dilation g only
  auto stbuf = [&](auto gc) -> u32x4* { constexpr int Gq = decltype(gc)::value; if constexpr (Gq == 0) return st0; else if constexpr (Gq == 1) return st1; else return st2; };
  constexpr int RB1 = NI0, RB2 = NI0 + NI1;                    // first table row of dilation 1 / 2
  if constexpr (XIN) {
    // byte offset of item I of a dilation-g strip relative to the tile's first pixel, ky = 1, piece 0 (tile- and ky-invariant; the rest
    // of the address is wave-uniform): a table in LDS, one ds_read per item instead of a division and 18 live registers
    for (int e = tid; e < (NI0 + NI1 + NI2) * 64; e += 512) {
      const int row = e >> 6, pgq = e & 63;
      const int g = row < RB1 ? 0 : (row < RB2 ? 1 : 2), I = row - (g == 0 ? 0 : (g == 1 ? RB1 : RB2));
      const int d = g == 0 ? D0 : (g == 1 ? D1 : D2), SW = TW + 2 * d;
      const int px = pgq + 64 * I;
      const int rr = px / SW, cc = px - rr * SW;
      lrel[e] = ((rr * W + cc - d) * (int)sg.pix_stride + sg.ch_off) * 4;
    }
  }
  // LDS slot of an item: plane (piece >> 2), pixel, chunk (piece & 3) ^ ((pixel >> 1) & 3); 64 I pixels further for item I (key unchanged)
  const int lofs = (piece >> 2) * PLANE + pg * 32 + (((piece & 3) ^ ((pg >> 1) & 3)) << 3);
  const unsigned wbytes = 3u * 9u * 2u * 1024u;
  const __amdgpu_buffer_rsrc_t rwh = make_rsrc(fhi, wbytes), rwl = make_rsrc(flo, wbytes);

  // the 12 weight fragments of strip S ((kx, ks) x (hi, lo), 1 KB each) go straight into weight buffer S % 3 (one per dilation) by
  // LDS-DMA, TWO steps ahead of their use: wave w moves fragments w and 8 + (w & 3) (waves 4-7 repeat what waves 0-3 moved: every wave
  // issues exactly two, so that the in-order load counter can be waited on with a constant).  No registers.
  auto dma_weights = [&](auto sc) {
    constexpr int S = decltype(sc)::value, g = S % 3, ky = S / 3;
    char* wb = (char*)(lw + g * WBUFH);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = t == 0 ? wave : 8 + (wave & 3), f = j >> 1;
      const int o = ((g * 9 + ky * 3 + (f >> 1)) * 2 + (f & 1)) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds((j & 1) ? rwl : rwh, (lds_ptr)(wb + j * 1024), 16, lane * 16, o, 0, 0);
    }
  };
  auto issue_strip = [&](const Tile& tl, bool on, auto sc) {
    constexpr int S = decltype(sc)::value, g = S % 3, ky = S / 3, d = dil_of(g), SW = TW + 2 * d;
    const __amdgpu_buffer_rsrc_t r = make_rsrc(sg.ptr + (long long)tl.b * H * W * sg.pix_stride, (unsigned)H * W * (unsigned)sg.pix_stride * 4u);
    int pq = pg;
    asm volatile("" : "+v"(pq));            // opaque: no hoisting of the per-item coordinates out of the tile loop
#pragma unroll
    for (int I = 0; I < ni_of(g); ++I) {
      int off;
      if constexpr (XIN) {
        const int sbase = ((tl.y0 + (ky - 1) * d) * W + tl.x0) * (int)sg.pix_stride * 4 + piece * 16;       // wave-uniform + lane constant
        off = lrel[((g == 0 ? 0 : (g == 1 ? RB1 : RB2)) + I) * 64 + pq] + sbase;
      } else {
        const int px = pq + 64 * I;
        const int rr = px / SW, cc = px - rr * SW;
        const int y = tl.y0 + (ky - 1) * d + rr, x = tl.x0 - d + cc;
        off = ((y * W + x) * (int)sg.pix_stride + sg.ch_off + piece * 4) * 4;
        off = ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) ? off : (int)OOB;
      }
      stbuf(std::integral_constant<int, g>{})[I] = __builtin_amdgcn_raw_buffer_load_b128(r, (on && !(dbg & 1) && (NP == 3 || piece < 4)) ? off : (int)OOB, 0, 0);
    }
  };
  auto write_strip = [&](int q, auto sc) {          // strip S (global count q) out of its registers into strip / weight buffer q & 1
    constexpr int S = decltype(sc)::value, g = S % 3;
    _Float16* buf = ldsh + (q & 1) * BUFH;
    if (!(dbg & 2)) {
#pragma unroll
      for (int I = 0; I < ni_of(g); ++I) *(u32x4*)&buf[lofs + 32 * 64 * I] = stbuf(std::integral_constant<int, g>{})[I];
    }
  };

  // ---- MFMA side: row `wave` of the tile, 32 pixels x 32 channels per dilation = four 16 x 16 accumulators (ph, nh) each
  const int wl = (kg >> 1) * 1024 + ((kg & 1) * 32 + l15) * 8;       // + kx * 2048 + hl * 512 + nh * 128 (halfs), as conv3x3_rw_f16.hip
  f32x4 acc[3][2][2];
  h8 resh[2], resl[2];          // o itself: the operands of the centre tap (strip 3 = (dilation 4, ky 1), kx 1) ARE the lane's 8 channels of its two pixels
  auto compute_strip = [&](int q, auto sc) {
    constexpr int S = decltype(sc)::value, g = S % 3, ky = S / 3, d = dil_of(g), SW = TW + 2 * d;
    const _Float16* Shi = ldsh + (q & 1) * BUFH;
    const _Float16* Slo = Shi + PLANE;
    const _Float16* wb = lw + g * WBUFH + wl;
    if constexpr (ky == 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a) (&acc[g][0][0])[a] = (f32x4)(0.f);
    }
    h8 wh[2], wo[2], ah[2][2], al[2][2];          // activations of tap kx + 1 requested before the MFMAs of tap kx; weights when needed
    auto fetch_w = [&](auto kc) {
      constexpr int KX = decltype(kc)::value;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        wh[nh] = *(const h8*)&wb[KX * 2048 + nh * 128];
        if constexpr (NP == 3) wo[nh] = *(const h8*)&wb[KX * 2048 + 512 + nh * 128];
      }
    };
    auto fetch = [&](auto kc) {
      constexpr int KX = decltype(kc)::value, Bq = KX & 1;
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const int qq = wave * SW + ph * 16 + l15 + KX * d;
        const int o = qq * 32 + ((kg ^ ((qq >> 1) & 3)) << 3);
        ah[Bq][ph] = *(const h8*)&Shi[o];
        if constexpr (NP == 3 || (S == 3 && KX == 1)) al[Bq][ph] = *(const h8*)&Slo[o];       // (NP == 1: zeros, read once for the residual o)
      }
    };
    fetch(std::integral_constant<int, 0>{});
    [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
      (([&] {
        constexpr int kx = Ks, Bq = kx & 1;
        fetch_w(std::integral_constant<int, kx>{});
        if constexpr (kx + 1 < 3) fetch(std::integral_constant<int, kx + 1>{});
        if constexpr (S == 3 && kx == 1) { resh[0] = ah[1][0]; resh[1] = ah[1][1]; resl[0] = al[1][0]; resl[1] = al[1][1]; }
        __builtin_amdgcn_sched_barrier(0);
        // three products per accumulator, the four accumulators interleaved: no MFMA reads the result of the one before it
        if constexpr (NP == 3) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) acc[g][ph][nh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nh], al[Bq][ph], acc[g][ph][nh], 0, 0, 0);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) acc[g][ph][nh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wo[nh], ah[Bq][ph], acc[g][ph][nh], 0, 0, 0);
        }
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) acc[g][ph][nh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nh], ah[Bq][ph], acc[g][ph][nh], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }()), ...);
    }(std::make_integer_sequence<int, 3>{});
  };

  // ---- epilogue of a tile row: lane holds channels n = 16 nh + 4 kg + r of pixel x0 + 16 ph + l15.  Nothing in it waits for a vector
  // memory load (loads retire in order: a load issued here would wait for the strips requested for the next tile): o comes from the
  // centre tap's operands, the constants from LDS, the running score sums were requested at the start of the tile's last step.
  const unsigned frame_out = (unsigned)H * W * (unsigned)p.out_pix_stride * 4u;
  float* sdst = nullptr;
  float sprev = 0.f;
  bool ovf_bad = false;
  auto score_prefetch = [&](const Tile& tl) {      // lane (l15, kg) finishes head kg >> 1 of pixel block kg & 1
    const int h = kg >> 1, y = tl.y0 + wave, x = tl.x0 + (kg & 1) * 16 + l15;
    sdst = (score_w && y < H && x < W) ? (h ? s1 : s0) + ((long long)tl.b * H + y) * W + x : nullptr;
    sprev = *(sdst ? (const volatile float*)sdst : (const volatile float*)p.residual);     // unconditional: the load counter below counts instructions
  };
  auto epilogue = [&](const Tile& tl) {
    const int y = tl.y0 + wave;
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out ? p.out + (long long)tl.b * H * W * p.out_pix_stride : nullptr, p.out ? frame_out : 0u);
    f32x4 bq[3][2], cwq[2][2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const int n = nh * 16 + 4 * kg;
#pragma unroll
      for (int g = 0; g < 3; ++g) bq[g][nh] = *(const f32x4*)&lconst[g * 32 + n];
#pragma unroll
      for (int h = 0; h < 2; ++h) cwq[h][nh] = *(const f32x4*)&lconst[(3 + h) * 32 + n];
    }
    float sc[2][2];                                      // [head][ph]
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int x = tl.x0 + ph * 16 + l15;
      const bool okp = y < H && x < W;
      const int pix = y * W + x;
      sc[0][ph] = sc[1][ph] = 0.f;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // positions 8 kg .. 8 kg + 7 of either plane = channels {4 kg ..} and {16 + 4 kg ..} (engine.SPLIT_PAIR_PERM)
          const float o = ((float)resh[ph][nh * 4 + e] + (float)resl[ph][nh * 4 + e]) * inv_a;
          v[e] = fmaxf(acc[0][ph][nh][e] * out_scale + bq[0][nh][e], 0.f) + fmaxf(acc[1][ph][nh][e] * out_scale + bq[1][nh][e], 0.f) +
                 fmaxf(acc[2][ph][nh][e] * out_scale + bq[2][nh][e], 0.f) + o;        // o + o1 + o2 + o3 (bdcn_new.py:54)
          sc[0][ph] += v[e] * cwq[0][nh][e];
          sc[1][ph] += v[e] * cwq[1][nh][e];
          // (fmaxf(x, 0) swallows NaN and -inf: the overflow test takes the accumulators themselves; lane = pixel: one channel per
          //  pixel, common.h)
          if (nh == 0 && e == 0) ovf_bad |= egne_nonfinite(acc[0][ph][nh][e] + acc[1][ph][nh][e] + acc[2][ph][nh][e] + o);
        }
        if (p.out) {
          const int n = nh * 16 + 4 * kg;
          const int oo = (okp && n < p.Cout_store) ? (pix * (int)p.out_pix_stride + p.out_ch_off + n) * 4 : (int)OOB;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout, oo, 0, 0);
        }
      }
    }
    if (score_w) {
      // this block's share of the stage's two score maps (bdcn_new.py:118-166 is linear behind the block: per block and head ONE
      // 32-vector): 8 channels were summed in the lane, the other 24 sit in lanes l15 + 16, + 32, + 48 -- two exchanges, fixed order
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          float t = sc[h][ph];
          t += __shfl_xor(t, 16);
          t += __shfl_xor(t, 32);
          sc[h][ph] = t;
        }
      const int h = kg >> 1, ph = kg & 1;
      const float v = h ? (ph ? sc[1][1] : sc[1][0]) : (ph ? sc[0][1] : sc[0][0]);
      if (sdst) *sdst = v + (accumulate ? sprev : lconst[160 + h]);
    }
  };

  __syncthreads();                  // lconst / lrel
  // ---- the loop.  Step S of a tile: [weights of strip S + 2 -> LDS by DMA] [registers of strip S + 1 -> LDS] [request strip S + 4]
  // [MFMAs of strip S] [epilogue behind the last strip] [wait for the weights of strip S + 1] [barrier].  An activation strip is requested
  // four steps and written one step before its use into the buffer that was last read two steps before it.  Loads retire IN ORDER: the
  // end-of-step wait for the DMA issued in the previous step allows exactly what was issued after it -- the previous step's strip and
  // everything of this step -- to stay in flight (the strip it does wait for is due at the start of the next step anyway).
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  if (nmine > 0) {
    const Tile t0 = decode(tile_at(0));
    dma_weights(I0{});
    dma_weights(I1{});
    issue_strip(t0, true, I0{});
    issue_strip(t0, true, I1{});
    issue_strip(t0, true, I2{});
    write_strip(0, I0{});
    issue_strip(t0, true, std::integral_constant<int, 3>{});       // (register buffer 0 again)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NI1 + NI2 + NI0) : "memory");      // both weight DMAs have landed
  }
  lds_barrier();
  for (int i = 0; i < nmine; ++i) {
    const Tile tl = decode(tile_at(i));
    const bool nx_on = i + 1 < nmine;
    const Tile nx = decode(tile_at(nx_on ? i + 1 : i));
    [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
      (([&] {
        constexpr int S = Ss, S2 = (S + 2) % NS, S3 = (S + 3) % NS, S4 = (S + 4) % NS;
        const int q = 9 * i + S;
        dma_weights(std::integral_constant<int, S2>{});          // into the weight buffer strip S - 1 has left (9 % 3 = 0: same dilation)
        if constexpr (S == NS - 1) score_prefetch(tl);
        // strip S + 1 (requested three steps ago) into the strip buffer that strip S - 1 has left ...
        if constexpr (S + 1 < NS) write_strip(q + 1, std::integral_constant<int, S + 1>{});
        else if (nx_on) write_strip(q + 1, I0{});
        // ... and its registers (buffer (S + 1) % 3) take strip S + 4, which has the same dilation
        if constexpr (S + 4 < NS) issue_strip(tl, true, std::integral_constant<int, S4>{});
        else issue_strip(nx, nx_on, std::integral_constant<int, S4>{});
        compute_strip(q, std::integral_constant<int, S>{});
        if constexpr (S == NS - 1) epilogue(tl);
        stamp(t_work);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ni_of(S3 % 3) + 2 + ni_of(S4 % 3)) : "memory");     // the weights of strip S + 1 have landed
        lds_barrier();
        stamp(t_wait);
      }()), ...);
    }(std::make_integer_sequence<int, NS>{});
  }
  egne_ovf_commit(ovf_bad, p.ovf_flag);
  if ((dbg & 64) && lane == 0) {
    unsigned long long* o = g_pstamps + ((long long)blockIdx.x * 8 + wave) * 4;
    o[0] = t_work; o[1] = t_wait; o[2] = nmine; o[3] = 0;
  }
}

}  // namespace

namespace egne {
// Called by egne_msblock_dil_scores_f16_fwd (msblock_dil_f16.hip) for a descriptor whose input slice is in split-pair storage; the
// descriptor has been validated there.
int msdil_ps_launch(const egne_conv_desc& d, const void* fhi, const void* flo, float a_scale, float w_scale, const float* score_w,
                    const float* score_c, float* s0, float* s1, int accumulate, hipStream_t st) {
  if (msdil1_wanted(d)) return msdil1_launch(d, fhi, a_scale, w_scale, score_w, score_c, s0, s1, accumulate, st);
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y * d.B;
  constexpr size_t lds = ((size_t)2 * BUFH + 2 * WBUFH) * sizeof(_Float16) + 162 * sizeof(float) + 18 * 64 * sizeof(int) + WBUFH * sizeof(_Float16);
  static_assert(lds <= 163840, "LDS budget");
  const float os = 1.0f / (a_scale * w_scale), inv_a = 1.0f / a_scale;
  const bool split = tiles_x > 2 && d.W >= TW * (tiles_x - 1) + 12;
  auto go = [&](auto npc) -> int {
    constexpr int NP = decltype(npc)::value;
    static bool once = hipFuncSetAttribute((const void*)msdil_ps_kernel<false, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                       hipFuncSetAttribute((const void*)msdil_ps_kernel<true, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (!once) return egne::fail(EGNE_ERR_LAUNCH, "msblock_dil (split-pair input): cannot raise the dynamic LDS limit to %zu", lds);
    if (split) {
      const int nt_in = (tiles_x - 2) * tiles_y * d.B, nt_b = 2 * tiles_y * d.B;
      hipLaunchKernelGGL((msdil_ps_kernel<true, NP>), dim3(nt_in < 256 ? nt_in : 256), dim3(512), lds, st, d, (const _Float16*)fhi, (const _Float16*)flo,
                         inv_a, os, tiles_x, tiles_y, nt_in, 1, score_w, score_c, s0, s1, accumulate);
      static const bool only_interior = getenv("EGNE_MSDIL_ONLY_INTERIOR") != nullptr;      // diagnostics: stamps of the interior launch
      if (!only_interior) hipLaunchKernelGGL((msdil_ps_kernel<false, NP>), dim3(nt_b < 256 ? nt_b : 256), dim3(512), lds, st, d, (const _Float16*)fhi,
                                             (const _Float16*)flo, inv_a, os, tiles_x, tiles_y, nt_b, 2, score_w, score_c, s0, s1, accumulate);
    } else {
      hipLaunchKernelGGL((msdil_ps_kernel<false, NP>), dim3(ntiles < 256 ? ntiles : 256), dim3(512), lds, st, d, (const _Float16*)fhi, (const _Float16*)flo,
                         inv_a, os, tiles_x, tiles_y, ntiles, 0, score_w, score_c, s0, s1, accumulate);
    }
    return EGNE_OK;
  };
  // egne_conv_desc.f16_products == 1: plain f16 operands (the edge network next to a bf16-storage training plan)
  const int rc = d.f16_products == 1 ? go(std::integral_constant<int, 1>{}) : go(std::integral_constant<int, 3>{});
  if (rc != EGNE_OK) return rc;
  return egne::check_launch("egne_msblock_dil_f16_fwd (split-pair input)");
}
}  // namespace egne

extern "C" int egne_msdil_ps_debug(int dbg, void* out_stamps) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_pdbg), &dbg, sizeof(int)) != hipSuccess) return -2;
  if (out_stamps && hipMemcpyFromSymbol(out_stamps, HIP_SYMBOL(g_pstamps), sizeof(unsigned long long) * 256 * 8 * 4) != hipSuccess) return -2;
  return 0;
}
