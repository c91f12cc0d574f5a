// The dilated branch of a BDCN MSBlock on PLAIN f16 operands (egne_conv_desc.f16_products = 1: the frozen edge network next to a
// training plan with bf16 activation storage), input in split-pair storage (only its hi plane is read):
//   out = o + sum_g relu(conv3x3_{dil g}(o) + b_g), g = 0..2, dilations 4 / 8 / 12 (bdcn_new.py:51-54), score heads fused,
// one v_mfma_f32_16x16x32_f16 per product, fp32 accumulate; the same operands and per-accumulator order as msblock_dil_ps_f16.hip
// with NP = 1, so the two agree bit for bit.
//
// Why a kernel of its own.  msblock_dil_ps_f16.hip stages, per 8 x 32 tile, nine strips (one per dilation and kernel row) through two
// LDS buffers: 13.7x the tile's own bytes from L2 and a barrier every 12 MFMAs per wave once the three-product split is gone
// (277-318 TFLOP/s).  With hi halves only a pixel is 64 bytes, and the WHOLE reach of the three dilations fits the LDS at once:
//   * a workgroup walks a column of 32-pixel-wide tiles DOWN the frame and keeps a ring of 32 rows x 56 pixels (8 + 2 x 12 rows,
//     32 + 2 x 12 columns: 112 KB) -- every 8-row step brings in 8 new rows (28 LDS-DMA instructions of 1 KB, no registers, no
//     VALU) and all 27 taps read from the ring: 1.75x the tile's bytes instead of 13.7x;
//   * the rows a step's DMA replaces are read by six of its taps only (kernel row 0 of dilations 12 and 8): those go first, ONE
//     barrier, the DMA is issued and has the other 21 taps to land -- one barrier per 108 MFMAs of a wave;
//   * the weights of dilations 4 and 8 stay in LDS for the whole launch (36 KB), those of dilation 12 in registers (72 per lane):
//     ring + weights = 148 KB;
//   * 16-byte chunk c of ring pixel q sits at c ^ (2 * ((q >> 2) & 1)): conflict free for the 16-pixel x 4-chunk ds_read_b128
//     pattern at every column offset that is a multiple of four (all tap offsets are).
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

namespace {

constexpr int TW = 32, TH = 8, HALO = 12;
constexpr int RPX = TW + 2 * HALO, RROWS = TH + 2 * HALO;            // 56 pixels, 32 rows
constexpr int ROWB = RPX * 64, RINGB = RROWS * ROWB;                  // 3 584, 114 688 bytes
constexpr int WLDS = 2 * 9 * 2 * 1024;                                // dilations 4 and 8: [g][tap][ks][64 lanes][16 B]
constexpr int OFF_W = RINGB, OFF_DUMMY = OFF_W + WLDS, OFF_CONST = OFF_DUMMY + 1024;
constexpr int LDS_BYTES = OFF_CONST + 164 * 4;
constexpr int GI = RPX * TH / 16;                                     // 28 LDS-DMA instructions per 8-row group
constexpr unsigned OOB = 0x80000000u;
static_assert(LDS_BYTES <= 163840, "LDS budget");

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

struct Tap { int g, ky, kx; };
// a step's taps: the two tap rows that read the ring rows the step's DMA replaces go first (dilation 12 / kernel row 0: frame rows
// y0 - 12 .. y0 - 5; dilation 8 / kernel row 0: y0 - 8 .. y0 - 1), then dilation 4 and the rest of dilations 8 and 12 -- per dilation in
// (ky, kx) order, as msblock_dil_ps_f16.hip accumulates them
constexpr int NFIRST = 6;
constexpr Tap tap_at(int i) {
  if (i < 3) return Tap{2, 0, i};
  if (i < 6) return Tap{1, 0, i - 3};
  i -= 6;
  if (i < 9) return Tap{0, i / 3, i % 3};
  i -= 9;
  if (i < 6) return Tap{1, 1 + i / 3, i % 3};
  i -= 6;
  return Tap{2, 1 + i / 3, i % 3};
}

__global__ __launch_bounds__(512)
void msdil1_kernel(const egne_conv_desc p, const _Float16* __restrict__ fhi, float inv_a, float out_scale, int ncols, int nseg, int seg_rows,
                   int nitems, const float* __restrict__ score_w, const float* __restrict__ score_c, float* __restrict__ s0,
                   float* __restrict__ s1, int accumulate) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  float* const lconst = (float*)(lds + OFF_CONST);       // [3][32] biases, [2][32] score vectors, [2] constants
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = the tile row this wave owns
  const int l15 = lane & 15, kg = lane >> 4;
  const int H = p.H, W = p.W;
  const egne_seg sg = p.seg[0];

  if (tid < 160) {
    const int r = tid >> 5, c = tid & 31;
    lconst[tid] = r < 3 ? (p.bias ? p.bias[r * p.CoutP + c] : 0.f) : (score_w ? score_w[(r - 3) * 32 + c] : 0.f);
  } else if (tid < 162) {
    lconst[tid] = score_c ? score_c[tid - 160] : 0.f;
  }
  // weights: dilations 4 and 8 -> LDS (36 blocks of 1 KB, linear), dilation 12 -> registers (the A operand of tap t, channel block nh)
  const __amdgpu_buffer_rsrc_t rwh = make_rsrc(fhi, 3u * 9u * 2u * 1024u);
  for (int k = wave; k < 36; k += 8)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rwh, (lds_ptr)(lds + OFF_W + k * 1024), 16, lane * 16, k * 1024, 0, 0);
  const int wl = (kg >> 1) * 1024 + ((kg & 1) * 32 + l15) * 16;      // + tap * 2048 + nh * 256 (bytes)
  h8 w2[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) w2[t][nh] = *(const h8*)((const char*)fhi + (18 + t) * 2048 + wl + nh * 256);

  // ---- copy side.  An 8-row group of the ring is 448 pixel slots (row-major over 8 x 56) = 28 instructions of 16 slots; wave w issues
  // instructions 4 w .. 4 w + 3 (28-31: into a scratch KB, so that every wave issues the same number).  Lane l of instruction k: slot
  // 16 k + (l >> 2), LDS chunk l & 3, which holds the pixel's chunk (l & 3) ^ key(column).
  int rel[4], rr[4], cc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = wave * 4 + j, slot = 16 * k + (lane >> 2);
    const int r = slot / RPX, c = slot - r * RPX;
    rr[j] = k < GI ? r : -100000;
    cc[j] = c;
    rel[j] = ((r * W + c - HALO) * (int)sg.pix_stride + sg.ch_off) * 4 + (((lane & 3) ^ (((c >> 2) & 1) << 1)) << 4);
  }
  // read side: lane (l15, kg) reads chunk kg of pixel `column offset + l15`; the key's parity follows the offset (a multiple of 4)
  int lanebase[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) lanebase[par] = l15 * 64 + ((kg ^ ((((l15 >> 2) + par) & 1) << 1)) << 4);

  const unsigned frame_in = (unsigned)H * W * (unsigned)sg.pix_stride * 4u;
  const unsigned frame_out = (unsigned)H * W * (unsigned)p.out_pix_stride * 4u;
  const int per = (int)gridDim.x >> 3;
  auto item_at = [&](int i) { return ((int)gridDim.x & 7) ? (int)blockIdx.x + i * (int)gridDim.x : ((i * 8 + ((int)blockIdx.x & 7)) * per + ((int)blockIdx.x >> 3)); };
  bool ovf_bad = false;

  for (int ii = 0;; ++ii) {
    const int it = item_at(ii);
    if (it >= nitems) break;
    const int sgi = it % nseg, cx = (it / nseg) % ncols, b = it / (nseg * ncols);
    const int x0 = cx * TW, ys = sgi * seg_rows;
    const int ye = ys + seg_rows < H ? ys + seg_rows : H;
    const int nst = (ye - ys + TH - 1) / TH;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(sg.ptr + (long long)b * H * W * sg.pix_stride, frame_in);
    // rows yg .. yg + 7 of the frame (zeros outside it) into ring rows ringrow .. ringrow + 7
    auto issue_group = [&](int yg, int ringrow, bool on) {
      const int sbase = (yg * W + x0) * (int)sg.pix_stride * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int y = yg + rr[j], x = x0 - HALO + cc[j];
        const bool ok = on && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const int k = wave * 4 + j;
        char* dst = k < GI ? lds + ringrow * ROWB + k * 1024 : lds + OFF_DUMMY;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_ptr)dst, 16, ok ? rel[j] + sbase : (int)OOB, 0, 0, 0);
      }
    };
    lds_barrier();                                        // the previous item's reads are done
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_group(ys - HALO + 8 * q, (ys + 8 * q) & 31, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    for (int t = 0; t < nst; ++t) {
      const int y0 = ys + TH * t;
      // ring row of frame row y: (y + 12) & 31; a tap of kernel row ky and dilation d reads row y0 + wave + (ky - 1) d: seven classes j
      int rb[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) rb[j] = ((y0 + 4 * j + wave) & 31) * ROWB;

      f32x4 acc[3][2][2];
#pragma unroll
      for (int a = 0; a < 12; ++a) (&acc[0][0][0])[a] = (f32x4)(0.f);
      h8 ah[2][2], wh[2][2], resh[2];
      float* sdst = nullptr;
      float sprev = 0.f;

      auto fetch = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr Tap tp = tap_at(I);
        constexpr int d = 4 * (tp.g + 1), j = 3 + (tp.ky - 1) * (tp.g + 1);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          constexpr int cconst = HALO + (tp.kx - 1) * d;    // + 16 ph: same key parity
          ah[I & 1][ph] = *(const h8*)(lds + rb[j] + lanebase[(cconst >> 2) & 1] + (cconst + 16 * ph) * 64);
        }
        if constexpr (tp.g < 2) {
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) wh[I & 1][nh] = *(const h8*)(lds + OFF_W + wl + ((tp.g * 9 + tp.ky * 3 + tp.kx) * 2048 + nh * 256));
        }
      };
      auto multiply = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr Tap tp = tap_at(I);
        if constexpr (tp.g == 0 && tp.ky == 1 && tp.kx == 1) { resh[0] = ah[I & 1][0]; resh[1] = ah[I & 1][1]; }    // o itself
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) {
            if constexpr (tp.g < 2)
              acc[tp.g][ph][nh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[I & 1][nh], ah[I & 1][ph], acc[tp.g][ph][nh], 0, 0, 0);
            else
              acc[2][ph][nh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[tp.ky * 3 + tp.kx][nh], ah[I & 1][ph], acc[2][ph][nh], 0, 0, 0);
          }
      };

      fetch(std::integral_constant<int, 0>{});
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (([&] {
          constexpr int I = Is;
          if constexpr (I + 1 < 27) fetch(std::integral_constant<int, I + 1>{});
          __builtin_amdgcn_sched_barrier(0);
          multiply(std::integral_constant<int, I>{});
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (I == NFIRST - 1) {
            // every wave has read what it needs of ring rows y0 & 31 .. + 7 (frame rows y0 - 12 .. y0 - 5): they take frame rows y0 + 20 .. y0 + 27
            lds_barrier();
            issue_group(y0 + TH + HALO, y0 & 31, t + 1 < nst);
            // the running score sums of this tile (lane (l15, kg) finishes head kg >> 1 of pixel block kg & 1); unconditional load
            const int h = kg >> 1, y = y0 + wave, x = x0 + (kg & 1) * 16 + l15;
            sdst = (score_w && y < H && x < W) ? (h ? s1 : s0) + ((long long)b * H + y) * W + x : nullptr;
            sprev = *(sdst ? (const volatile float*)sdst : (const volatile float*)p.residual);
          }
        }()), ...);
      }(std::make_integer_sequence<int, 27>{});

      // ---- epilogue of the tile row: lane holds channels n = 16 nh + 4 kg + e of pixel x0 + 16 ph + l15 (msblock_dil_ps_f16.hip)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the step's rows have landed (and sprev): nothing is waited for at the barrier
      {
        const int y = y0 + wave;
        const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out ? p.out + (long long)b * H * W * p.out_pix_stride : nullptr, p.out ? frame_out : 0u);
        f32x4 bq[3][2], cwq[2][2];
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          const int n = nh * 16 + 4 * kg;
#pragma unroll
          for (int g = 0; g < 3; ++g) bq[g][nh] = *(const f32x4*)&lconst[g * 32 + n];
#pragma unroll
          for (int h = 0; h < 2; ++h) cwq[h][nh] = *(const f32x4*)&lconst[(3 + h) * 32 + n];
        }
        float sc[2][2];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          const int x = x0 + ph * 16 + l15;
          const bool okp = y < H && x < W;
          const int pix = y * W + x;
          sc[0][ph] = sc[1][ph] = 0.f;
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // positions 8 kg .. 8 kg + 7 of the hi plane = channels {4 kg ..} and {16 + 4 kg ..} (engine.SPLIT_PAIR_PERM); with plain
              // f16 operands the residual is the hi half alone, as in msblock_dil_ps_f16.hip (NP = 1)
              const float o = ((float)resh[ph][nh * 4 + e] + 0.f) * inv_a;
              v[e] = fmaxf(acc[0][ph][nh][e] * out_scale + bq[0][nh][e], 0.f) + fmaxf(acc[1][ph][nh][e] * out_scale + bq[1][nh][e], 0.f) +
                     fmaxf(acc[2][ph][nh][e] * out_scale + bq[2][nh][e], 0.f) + o;        // o + o1 + o2 + o3 (bdcn_new.py:54)
              sc[0][ph] += v[e] * cwq[0][nh][e];
              sc[1][ph] += v[e] * cwq[1][nh][e];
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
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
              float tt = sc[h][ph];
              tt += __shfl_xor(tt, 16);
              tt += __shfl_xor(tt, 32);
              sc[h][ph] = tt;
            }
          const int h = kg >> 1, ph = kg & 1;
          const float v = h ? (ph ? sc[1][1] : sc[1][0]) : (ph ? sc[0][1] : sc[0][0]);
          if (sdst) *sdst = v + (accumulate ? sprev : lconst[160 + h]);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  egne_ovf_commit(ovf_bad, p.ovf_flag);
}

}  // namespace

namespace egne {

bool msdil1_wanted(const egne_conv_desc& d) {
  const char* e = getenv("EGNE_MSDIL1");                   // (read per launch: the tests compare both forms inside one process)
  return !(e && atoi(e) == 0) && d.f16_products == 1 && d.W >= 64 && d.H >= 48;
}

// Called by msdil_ps_launch for plain f16 operands on maps of 48 x 64 and larger; the descriptor has been validated by
// egne_msblock_dil_scores_f16_fwd.  Work items: (frame, 32-pixel column, vertical segment); a column is cut into segments where whole
// columns would leave compute units without work (each segment pays a 32-row warm-up).
int msdil1_launch(const egne_conv_desc& d, const void* fhi, float a_scale, float w_scale, const float* score_w, const float* score_c,
                  float* s0, float* s1, int accumulate, hipStream_t st) {
  const int ncols = (d.W + TW - 1) / TW;
  int best = 1;
  double best_t = 1e30;
  for (int ns = 1; ns <= 4; ++ns) {
    const int rows = ((d.H + ns - 1) / ns + TH - 1) / TH * TH;
    if (ns > 1 && rows * (ns - 1) >= d.H) continue;           // an empty last segment
    const long long items = (long long)d.B * ncols * ns;
    const double t = (double)((items + 255) / 256) * (rows / TH + 3.5);
    if (t < best_t) { best_t = t; best = ns; }
  }
  const int seg_rows = ((d.H + best - 1) / best + TH - 1) / TH * TH;
  const long long nitems = (long long)d.B * ncols * best;
  if (!egne::raise_lds((const void*)msdil1_kernel, LDS_BYTES))
    return egne::fail(EGNE_ERR_LAUNCH, "msblock_dil (plain f16, ring form): cannot raise the dynamic LDS limit to %d", LDS_BYTES);
  const float os = 1.0f / (a_scale * w_scale), inv_a = 1.0f / a_scale;
  hipLaunchKernelGGL(msdil1_kernel, dim3((unsigned)(nitems < 256 ? nitems : 256)), dim3(512), LDS_BYTES, st, d, (const _Float16*)fhi, inv_a, os,
                     ncols, best, seg_rows, (int)nitems, score_w, score_c, s0, s1, accumulate);
  return egne::check_launch("egne_msblock_dil_f16_fwd (plain f16, ring form)");
}

}  // namespace egne
