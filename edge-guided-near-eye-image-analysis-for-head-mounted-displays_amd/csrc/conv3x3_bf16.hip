// 3x3 "same" convolution over ONE bf16 input slice on v_mfma_f32_16x16x32_bf16 (fp32 accumulate, bf16 output): the 3x3
// convolutions of ESF-Net (models/RITnet_v2.py:57-62,85-87; utils.py:1047-1048) and their data gradients in training plans that
// keep activations and activation gradients in HBM as bf16 (BASELINE.json configs[2..4]; reference loop train.py:262-287).
//
// Same structure as conv3x3_rw_f16.hip (fixed wave roles, weights of a 32-channel output block resident in LDS, one s_barrier
// per job) minus everything the split-f16 arithmetic needed: the tensor IS bf16, so an element is one MFMA operand -- no hi / lo
// split, no pre-scale, one MFMA per product instead of three, and a raw input goes HBM -> register -> LDS without touching the
// vector ALU.  Per 240x320x32 layer the kernel moves half the bytes of the fp32-storage form and is HBM-bound.
//
// LDS: weights [chunk][tap][k16][64 lanes][8 bf16] (18 KB per 32 input channels; up to four chunks resident, streamed per chunk
// above that), two halo images of ONE 32-channel chunk of a 32 x 8 tile, [340 pixels][32 bf16] without padding: the 16-byte
// group c of pixel q sits at c ^ ((q >> 1) & 3) (conflict-free for the 16-pixel x 4-group ds_read_b128 pattern of the 16x16x32
// MFMA at every alignment).  A tile is `nk` jobs (one per chunk), accumulators persist.
//   producers (waves 0-3)  halo gather, 16 bytes = 8 channels per item, two jobs of loads in flight; optional fused
//             InstanceNorm affine + activation (fp32) with the zero padding applied after it; image (job + 1) & 1;
//   consumers (waves 4-7)  two rows x 32 channels each: 9 taps x 8 MFMAs per job, operands of tap t + 1 requested before the
//             MFMAs of tap t; transposed product, so a lane ends with 4 consecutive channels of a pixel per accumulator:
//             8-byte stores, issued between the MFMAs of the next tile's first job.
#include "common.h"
#include <cstdlib>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef egne_bf16x8 b8;

#ifdef EGNE_B3_STAMPS      // diagnostic build only (scratch/b3_stamps.py): cycles per phase and wave, [block][wave][4]
__device__ unsigned long long egne_b3_stamps_buf[256 * 8 * 4];
extern "C" int egne_b3_read_stamps(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(egne_b3_stamps_buf), sizeof(egne_b3_stamps_buf)) == hipSuccess ? 0 : 1;
}
#define B3_T0() unsigned long long st_last = __builtin_amdgcn_s_memtime(), st_acc[4] = {0, 0, 0, 0}; const unsigned long long st_begin = st_last
#define B3_ADD(i) do { const unsigned long long st_now = __builtin_amdgcn_s_memtime(); st_acc[i] += st_now - st_last; st_last = st_now; } while (0)
#define B3_OUT() do { if (lane == 0) { unsigned long long* o = egne_b3_stamps_buf + (blockIdx.x * 8 + wave) * 4; o[0] = st_acc[0]; o[1] = st_acc[1]; o[2] = st_acc[2]; o[3] = __builtin_amdgcn_s_memtime() - st_begin; } } while (0)
#else
#define B3_T0() do {} while (0)
#define B3_ADD(i) do {} while (0)
#define B3_OUT() do {} while (0)
#endif

namespace {

constexpr int TW = 32, TH = 8, HWd = TW + 2, HHd = TH + 2, NPX = HHd * HWd;       // 340 halo pixels
constexpr int IMG = NPX * 32;                            // bf16 elements per image
constexpr int WCH = 9 * 2 * 512;                         // bf16 elements of weights per 32-channel chunk
constexpr int NI = (NPX * 4 + 255) / 256;                // 16-byte items per producer lane and job (6)
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Output-channel order of a 32-channel block inside the LDS weight images (round 6).  The consumer's two fragments nh = 0 / 1 of a block
// are the LDS lane positions 16 nh + i (i = MFMA row), and the transposed product leaves rows 4 kg .. 4 kg + 3 of a fragment in lane
// group kg.  With channel 8 (i >> 2) + 4 nh + (i & 3) at position 16 nh + i a lane ends with the EIGHT CONSECUTIVE channels 8 kg .. 8 kg + 7
// of its pixel in its two accumulators: one 16-byte store per pixel, block and lane instead of two of 8 bytes (a consumer wave's 16
// stores per tile were 800-1000 of its 8 300 cycles, scratch/b3_stamps.py; the reads keep their conflict-free lane order).
// pack lane (k-half h, channel c) = 32 h + c  <-  LDS lane position l = 32 h + m:
__device__ __forceinline__ int chan_lane(int l) {
  const int m = l & 31;
  return (l & 32) + 8 * ((m & 15) >> 2) + 4 * (m >> 4) + (m & 3);
}

__device__ __forceinline__ f32x4 unpack_lo(u32x2 v) {     // 4 bf16 -> 4 floats
  const u32x4 w = {v[0] << 16, v[0] & 0xffff0000u, v[1] << 16, v[1] & 0xffff0000u};
  return __builtin_bit_cast(f32x4, w);
}

// KCH: 32-channel chunks of the input slice (1..4: all weights resident) or 0: any number of chunks (p.Ktot / 32), the weights of
// a job's chunk (18 KB per 32 output channels) STREAMED into one of two LDS weight buffers by the producers one job ahead.
// MB: 32-channel output blocks per WORKGROUP (round 6).  MB = 1: a consumer wave owns two rows x 32 pixels x 32 channels (8 accumulators;
// 2 weight + 4 activation fragments read from LDS per 8 MFMAs).  MB = 2: two rows x 32 pixels x 64 channels (16 accumulators; 4 + 4
// fragments per 16 MFMAs: a third less LDS traffic per MFMA and twice the MFMA time per tap to cover a read's latency with), and the
// halo of a tile is staged ONCE per 64 output channels -- with one 32-channel block per workgroup the 64+-channel layers ran at
// 27-29 % MFMA-busy, every tap waiting for its operands (a tap's 128 MFMA cycles do not cover an LDS round trip with 24 reads of four
// waves queued) and every tile's halo gathered by two workgroups.  A workgroup whose second block lies past the pack (CoutP = 96:
// blocks of 64 + 32) skips that block's MFMAs (mbn).
// ncb = 32-channel blocks in the pack, nrun = workgroup blocks (of 32 MB channels) that hold stored channels.
// RM: 0 = neither residual nor mask; 1 = a residual (accumulating data gradients); 2 = a mask, with or without a residual (the last writer of
// a gradient slice).  Their vectors are prefetched into 8 MB register pairs each -- compiled out of the launches that have none: the
// 64-channel form has registers for ONE of the two sets (masked launches of 64+ channels take the MB = 1 form).
// RW: tile rows per consumer wave.  2: four consumer waves (one MFMA-issuing wave per SIMD).  1 (round 6, MB = 2 without statistics or mask):
// EIGHT consumer waves of one row each next to the four producers -- two MFMA-issuing waves per SIMD, so that one wave's operand waits,
// hand-over and stores run under the other's MFMAs (stamps: a consumer wave of the four-wave form spends 4 600 of 10 300 cycles per tile
// issuing MFMAs and nobody else on its SIMD issues any).
template <int KCH, int MB, int RM, int RW>
__global__ __launch_bounds__(64 * (4 + 8 / RW))
void conv3x3_bf16_kernel(const egne_conv_desc p, const egne_bf16* __restrict__ wfrag, int tiles_x, int tiles_y, int ntiles, int ncb,
                         int nrun) {
  extern __shared__ __attribute__((aligned(16))) egne_bf16 ldsb[];
  constexpr int WCHM = MB * WCH;                         // bf16 elements of weights per 32-channel input chunk: [tap][mb][k16][64 lanes][8]
  egne_bf16* const lw = ldsb + 2 * IMG;                  // weights behind the two images
  // behind the weights: per consumer wave [16 values][64 lanes] DOUBLES -- the per-tile channel sums of egne_conv_desc.stats_ws on their
  // way from "4 channels x 4 pixels per lane" to "one (channel, statistic) per lane" (fp64 throughout: E[x^2] - mean^2 of a nearly constant
  // channel cancels seven digits, and float partial sums moved the decoder's gradients by 30 %), or -- a launch has one or the other --
  // the running bias sums of a masked data gradient (mask_sums: [8 MB][64] doubles per wave)
  double* const lstat_all = (double*)(lw + (size_t)(KCH == 0 ? 2 : KCH) * WCHM);
  // ... and the block's bias [32 MB]: the accumulators' initial value
  float* const lepi = (float*)(lstat_all + 4 * 16 * 64);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  const egne_seg sg = p.seg[0];
  const egne_bf16* const xin = (const egne_bf16*)sg.ptr;

  // workgroup -> (output block, worker): the nrun workgroups of one worker walk the SAME tiles on the same XCD (round-robin
  // dispatch: a speed assumption only), so the halo of a tile is read from HBM once
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int wpx = ((int)gridDim.x >> 3) / nrun;          // workers per XCD
  if (q >= wpx * nrun) return;                           // spare workgroups of an XCD stay idle (uniform per workgroup)
  const int cb = q % nrun, worker = (q / nrun) * 8 + xcd, nworkers = wpx * 8;
  const int mbn = (ncb - cb * MB) < MB ? (ncb - cb * MB) : MB;       // 32-channel blocks of this workgroup that exist in the pack
  auto tile_at = [&](int i) { return worker + i * nworkers; };
  struct Tile { int b, y0, x0; };
  auto decode = [&](int t) {
    Tile r;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    r.b = t; r.y0 = ty * TH; r.x0 = tx * TW;
    return r;
  };
  const int ntl = worker < ntiles ? (ntiles - 1 - worker) / nworkers + 1 : 0;
  constexpr bool STREAM = KCH == 0;
  const int nk = STREAM ? p.Ktot / 32 : KCH;             // chunks per tile
  const int nmine = ntl * nk;                            // jobs: (tile, chunk), chunk fastest
  const int nloop = (nmine + 1) & ~1;                    // both roles run an even number of steps (register buffer = step parity)

  // the block's weights: fragments (tap, k16, nt = cb MB + mb) of the pack [tap][Ktot/16][CoutP/32][lane][8]
  const int KT16 = nk * 2;
  if constexpr (!STREAM) {
    constexpr int NTHR = 64 * (4 + 8 / RW);
    for (int it = tid; it < KCH * 9 * MB * 2 * 64; it += NTHR) {       // 16-byte items, LDS order [chunk][tap][mb][ks][lane]
      const int l = it & 63, ks = (it >> 6) & 1, r0 = it >> 7, mb = r0 % MB, r = r0 / MB, tap = r % 9, ch = r / 9;
      const long long src = (((long long)tap * KT16 + ch * 2 + ks) * ncb + cb * MB + mb) * 512 + chan_lane(l) * 8;
      *(u32x4*)&lw[(long long)it * 8] = mb < mbn ? *(const u32x4*)(wfrag + src) : u32x4{0u, 0u, 0u, 0u};
    }
  }
  if (tid < 32 * MB) {
    const int n = cb * MB * 32 + tid;
    const bool okn = n < ncb * 32;
    lepi[tid] = (p.bias && okn) ? p.bias[n] : 0.f;
  }
  __syncthreads();

  if (wave < 4) {
    // =================================================================== producers: halo chunk -> LDS image
    const int piece = tid & 3, pg = tid >> 2;            // 16-byte group of the pixel's 32-channel chunk, pixel (64 per round)
    const float slope_in = sg.act_in == EGNE_ACT_RELU ? 0.f : (sg.act_in == EGNE_ACT_LEAKY ? 0.01f : 1.f);
    // LDS slot of item I: pixel pg + 64 I; (pixel >> 1) & 3 = (pg >> 1) & 3 for every I: lane constant + 64 B * 64 * I
    const int lofs = pg * 32 + ((piece ^ ((pg >> 1) & 3)) << 3);
    u32x4 st[2][NI];
    int rel[NI];          // byte offset of item I relative to the tile's first pixel (tile- and chunk-invariant)
#pragma unroll
    for (int I = 0; I < NI; ++I) {
      const int px = pg + 64 * I, hy = px / HWd, hx = px - hy * HWd;
      rel[I] = px < NPX ? (((hy - 1) * W + hx - 1) * (int)sg.pix_stride + sg.ch_off + piece * 8) * 2 : (int)OOB;
    }
    // jobs s + 1 (converted), s + 2 (coefficients, streamed weights) and s + 3 (requested) of step s as a ring of (tile, chunk, index):
    // advanced once per step, the tile decoded (two divisions) only where the chunk wraps -- three job_at() calls per step had cost
    // a dozen scalar divisions
    struct Job { Tile t; int ch, ti, idx; };
    auto first_job = [&]() { Job r; r.t = decode(tile_at(0)); r.ch = 0; r.ti = 0; r.idx = 0; return r; };
    auto next_job = [&](const Job& j) {
      Job r = j;
      ++r.idx;
      if (++r.ch == nk) { r.ch = 0; ++r.ti; r.t = decode(tile_at(r.ti)); }
      return r;
    };
    auto issue1 = [&](const Job& jb, bool on, auto bc, auto ic) {
      constexpr int BUF = decltype(bc)::value, I = decltype(ic)::value;
      const Tile& tl = jb.t;
      const __amdgpu_buffer_rsrc_t r = make_rsrc(xin + (long long)tl.b * H * W * sg.pix_stride, (unsigned)H * W * (unsigned)sg.pix_stride * 2u);
      const int c0 = jb.ch * 32 + piece * 8;            // channels past the slice (padding up to 32 * nk) read zeros
      int off;
      if (tl.x0 >= 1 && tl.x0 + TW + 1 <= W) {          // wave-uniform: interior columns (rows outside the frame fall outside the resource)
        const int sbase = ((tl.y0 * W + tl.x0) * (int)sg.pix_stride + jb.ch * 32) * 2;
        off = (on && c0 < sg.Cp && rel[I] != (int)OOB) ? rel[I] + sbase : (int)OOB;
      } else {
        int pq = pg;
        asm volatile("" : "+v"(pq));                    // opaque: no hoisting of the per-item coordinates out of the job loop
        const int px = pq + 64 * I;
        const int hy = px / HWd, hx = px - hy * HWd;
        const int y = tl.y0 - 1 + hy, x = tl.x0 - 1 + hx;
        const bool ok = on && c0 < sg.Cp && px < NPX && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        off = ok ? ((y * W + x) * (int)sg.pix_stride + sg.ch_off + c0) * 2 : (int)OOB;
      }
      st[BUF][I] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    };
    // fused affine: the coefficients of a job are requested one step EARLIER than its conversion
    f32x4 asc[2][2], ash[2][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) { (&asc[0][0])[a] = (f32x4)(1.f); (&ash[0][0])[a] = (f32x4)(0.f); }
    auto load_aff = [&](const Job& jb, bool on, auto bc) {
      constexpr int BUF = decltype(bc)::value;
      if (sg.scale && on) {       // (a job past the workgroup's last decodes a frame past the batch: no table row to read)
        const int c0 = jb.ch * 32 + piece * 8;
        const float* zs = c0 < sg.Cp ? sg.scale + (long long)jb.t.b * sg.Cp + c0 : egne_zero_page;
        const float* zh = c0 < sg.Cp ? sg.shift + (long long)jb.t.b * sg.Cp + c0 : egne_zero_page;
        asc[BUF][0] = *(const f32x4*)zs; asc[BUF][1] = *(const f32x4*)(zs + 4);
        ash[BUF][0] = *(const f32x4*)zh; ash[BUF][1] = *(const f32x4*)(zh + 4);
      }
    };
    auto convert1 = [&](const Job& jb, egne_bf16* img, auto bc, auto ic) {
      constexpr int BUF = decltype(bc)::value, I = decltype(ic)::value;
      const int px = pg + 64 * I;
      if (I < NI - 1 || px < NPX) {
        u32x4 raw = st[BUF][I];
        if (sg.scale) {      // fused InstanceNorm affine (+ activation) of the consumer; zero padding applied after it
          const int hy = px / HWd, hx = px - hy * HWd;
          const int y = jb.t.y0 - 1 + hy, x = jb.t.x0 - 1 + hx;
          f32x4 v0 = unpack_lo(u32x2{raw[0], raw[1]}), v1 = unpack_lo(u32x2{raw[2], raw[3]});
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t0 = v0[e] * asc[BUF][0][e] + ash[BUF][0][e], t1 = v1[e] * asc[BUF][1][e] + ash[BUF][1][e];
            v0[e] = fmaxf(t0, t0 * slope_in); v1[e] = fmaxf(t1, t1 * slope_in);
          }
          if (!((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)) { v0 = (f32x4)(0.f); v1 = (f32x4)(0.f); }
          const u32x2 p0 = __builtin_bit_cast(u32x2, __builtin_convertvector(v0, egne_bf16x4));
          const u32x2 p1 = __builtin_bit_cast(u32x2, __builtin_convertvector(v1, egne_bf16x4));
          raw = u32x4{p0[0], p0[1], p1[0], p1[1]};
        }
        *(u32x4*)&img[lofs + 32 * 64 * I] = raw;
      }
    };
    // STREAM: the 1152 MB 16-byte pieces of a chunk's weights (LDS order [tap][mb][ks][lane]), five (nine) per producer lane (the last
    // round is partial), requested one job ahead and written at the start of the next step
    constexpr int NWP = 1152 * MB, NWI = (NWP + 255) / 256;
    u32x4 wreg[STREAM ? NWI : 1];
    const unsigned wbytes = 9u * (unsigned)KT16 * (unsigned)ncb * 1024u;
    const __amdgpu_buffer_rsrc_t rwf = make_rsrc(wfrag, wbytes);
    auto w_issue = [&](int ch, bool on) {
#pragma unroll
      for (int i = 0; i < (STREAM ? NWI : 0); ++i) {
        const int it = tid + 256 * i, l = it & 63, ks = (it >> 6) & 1, r0 = it >> 7, mb = r0 % MB, tap = r0 / MB;
        const int off = (on && it < NWP && mb < mbn) ? ((((tap * KT16 + ch * 2 + ks) * ncb + cb * MB + mb) * 512 + chan_lane(l) * 8) * 2) : (int)OOB;
        wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rwf, off, 0, 0);
      }
    };
    auto w_store = [&](int parity) {
#pragma unroll
      for (int i = 0; i < (STREAM ? NWI : 0); ++i)
        if (tid + 256 * i < NWP) *(u32x4*)&lw[parity * WCHM + (tid + 256 * i) * 8] = wreg[i];
    };
    Job j1, j2, j3;       // jobs s + 1, s + 2, s + 3 of the coming step
    // step s (job s): convert job s+1 out of register buffer (s+1)&1 and refill every freed register with job s+3
    auto step = [&](int s, auto bc) {
      constexpr int BUF = decltype(bc)::value;          // = (s + 1) & 1
      const bool c_on = s + 1 < nmine, a_on = s + 2 < nmine, i_on = s + 3 < nmine;
      if constexpr (STREAM) {
        w_store((s + 1) & 1);                             // weights of job s+1 (requested in step s-1)
        w_issue(j2.ch, a_on);
      }
      load_aff(j2, a_on, std::integral_constant<int, BUF ^ 1>{});
      egne_bf16* img = ldsb + ((s + 1) & 1) * IMG;
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (([&] {
          if (c_on) convert1(j1, img, bc, std::integral_constant<int, Is>{});
          issue1(j3, i_on, bc, std::integral_constant<int, Is>{});
        }()), ...);
      }(std::make_integer_sequence<int, NI>{});
      j1 = j2; j2 = j3;
      if (i_on) j3 = next_job(j3);
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    {   // prologue: jobs 0 and 1 requested, job 0 converted (its registers refilled with job 2)
      const Job j0 = first_job();
      j1 = nmine > 1 ? next_job(j0) : j0;
      j2 = nmine > 2 ? next_job(j1) : j1;
      j3 = nmine > 3 ? next_job(j2) : j2;
      if constexpr (STREAM) {
        w_issue(j0.ch, nmine > 0);
        w_store(0);
        w_issue(j1.ch, nmine > 1);
      }
      load_aff(j0, nmine > 0, B0{});
      load_aff(j1, nmine > 1, B1{});
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (issue1(j0, nmine > 0, B0{}, std::integral_constant<int, Is>{}), ...);
        (issue1(j1, nmine > 1, B1{}, std::integral_constant<int, Is>{}), ...);
      }(std::make_integer_sequence<int, NI>{});
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (([&] {
          if (nmine > 0) convert1(j0, ldsb, B0{}, std::integral_constant<int, Is>{});
          issue1(j2, nmine > 2, B0{}, std::integral_constant<int, Is>{});
        }()), ...);
      }(std::make_integer_sequence<int, NI>{});
    }
    lds_barrier();
    B3_T0();
    for (int s = 0; s < nloop; s += 2) {
      step(s, B1{});
      B3_ADD(0);
      lds_barrier();
      B3_ADD(1);
      step(s + 1, B0{});
      B3_ADD(0);
      lds_barrier();
      B3_ADD(1);
    }
    B3_OUT();
  } else {
    // =================================================================== consumers: 9 taps per job from LDS only
    // v_mfma_f32_16x16x32_bf16: one instruction per (16 channels, 16 pixels, 32 input channels of a tap); a wave's two rows x 32
    // pixels x 32 MB channels are 8 MB independent accumulators
    const int cw = wave - 4, row0 = cw * RW;
    const int l15 = lane & 15, kg = lane >> 4;
    egne_bf16* const outp = (egne_bf16*)p.out;
    const egne_bf16* const resp = RM ? (const egne_bf16*)p.residual : nullptr;       // (RM = 2: may be null)
    const unsigned frame_out = (unsigned)H * W * (unsigned)p.out_pix_stride * 2u;
    const unsigned frame_res = (unsigned)H * W * (unsigned)p.res_pix_stride * 2u;
    const float slope_out = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
    // transposed product (weights as the A operand): the lane holds channels n = 32 (cb MB + mb) + 8 kg + 4 nh + r of pixel 16 ph + l15
    const int nbase = cb * MB * 32 + 8 * kg;             // + mb * 32 + nh * 4 (chan_lane: eight consecutive channels per lane and block)
    // Activation operands: pixel q = (row0 + r) * 34 + 16 ph + kx + l15 (r = tm + ky), 8-channel group kg at group kg ^ ((q >> 1) & 3).
    // With c = r * 34 + 16 ph + kx (compile time) and row0 * 34 = 68 cw even: (q >> 1) = ((l15 + (c & 1)) >> 1) + (c >> 1) + 34 cw, so the
    // lane-dependent part of the address takes EIGHT values (parity of c, (c >> 1) & 3) and the rest is the instruction's offset
    // (36 address registers before)
    int atab[2][4];
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int mm = 0; mm < 4; ++mm)
        atab[par][mm] = (l15 + cw * RW * 34) * 64 + ((kg ^ ((((l15 + par) >> 1) + mm + cw * RW * 17) & 3)) << 4);
    // weights: k-group kg of output channel m = fragment k16 = kg >> 1, lane position (kg & 1) * 32 + m
    const int wl = (kg >> 1) * 512 + ((kg & 1) * 32 + l15) * 8;       // + (tap * MB + mb) * 1024 + nh * 128 (elements)
    f32x4 acc[RW][2][MB][2];
    // the finished tile as it will be stored (bf16 x 4 per group).  MB = 1: TWO sets, by tile parity -- a set is rewritten two tiles after
    // its stores went out; with one set the rounding at the tile's last tap waited for the acknowledgements of the previous tile's stores
    // (the compiler guards the data registers of a store in flight with s_waitcnt vmcnt(0))
    constexpr int PP = 1;       // (PP = 2 for MB = 1 measured: the parity branches per group cost more than the wait they avoid)
    u32x2 prevp[PP][RW][2][MB][2];
    int tpar = 0;                                         // parity of the tile being computed
    constexpr int NG = 4 * RW * MB;
#pragma unroll
    for (int a = 0; a < NG; ++a) (&acc[0][0][0][0])[a] = (f32x4)(0.f);
#pragma unroll
    for (int a = 0; a < NG * PP; ++a) (&prevp[0][0][0][0][0])[a] = u32x2{0u, 0u};
    __amdgpu_buffer_rsrc_t rout = make_rsrc(outp, 0u);
    int tvo[RW][2];
#pragma unroll
    for (int a = 0; a < 2 * RW; ++a) (&tvo[0][0])[a] = (int)OOB;
    // mask-on-write (egne_conv_desc.mask_y): the launch is the last writer of a gradient slice -- the stored value is v * act'(y), and the
    // wave keeps the sums of what it stores (fp64, all its tiles) for the producing layer's bias gradient
    const egne_bf16* const mskp = RM == 2 ? (const egne_bf16*)p.mask_y : nullptr;
    const unsigned frame_msk = (unsigned)H * W * (unsigned)p.mask_pix_stride * 2u;
    const float slope_m = p.mask_act == EGNE_ACT_RELU ? 0.f : (p.mask_act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
    // (the wave's running sums live in LDS, [8 MB values][64 lanes] doubles per consumer wave behind the statistics staging area)
    double* const lms = lstat_all + cw * 16 * 64;
    if (mskp && p.mask_sums) {
#pragma unroll
      for (int a = 0; a < NG; ++a) lms[a * 64 + lane] = 0.;
    }
    // residual and mask vectors of a tile are requested in the second half of its last job (round 5: loaded at hand-over, every tile
    // waited a memory round trip for them -- a masked data gradient took twice the time of a plain one)
    // (declared per job below: values that live from tap 5 of a tile's last job to its hand-over only -- as loop-carried variables they
    //  held 32 MB registers through every tap)
    auto jok = [&](int mb, int nh) { return nbase + mb * 32 + nh * 4 < p.Cout_store; };       // Cout_store is a multiple of 4
    // values are finished (activation [, residual, mask]) and rounded at hand-over, behind the barrier that releases the tile's last image
    // to the producers; the deferred part is the bare store.  (Round 6 measured the 16 groups at 2 450 cycles per tile next to 4 600 of MFMA
    // issue, scratch/b3_stamps.py, and tried them BETWEEN the MFMAs of the tile's last tap: the same total -- the wave's vector work does
    // not hide behind its own matrix work -- and 45 spilled registers in the residual form; the bias now enters as the accumulators'
    // initial value and the activation is branch-free.)
    auto finish_group = [&](auto gc, const u32x2 (&pre_r)[RW][2][MB][2], const u32x2 (&pre_m)[RW][2][MB][2]) {
      constexpr int Gi = decltype(gc)::value, nh = Gi & 1, mb = (Gi >> 1) % MB, ph = ((Gi >> 1) / MB) & 1, tm = (Gi >> 1) / MB / 2;
      f32x4 v = acc[tm][ph][mb][nh];                       // (the bias went in with the accumulator's initial value)
      // branch-free: none / ReLU / LeakyReLU are max(v, v * slope) with slope 1 / 0 / 0.01 -- a taken scalar branch between two MFMAs
      // costs more than the two instructions it saves (the bf16 plans are training plans: no folded BatchNorm, so no post affine here)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope_out);
      if (RM == 1 || (RM == 2 && resp)) {
        const f32x4 rv = unpack_lo(pre_r[tm][ph][mb][nh]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rv[e];
      }
      if constexpr (RM == 2) {
        const bool on = jok(mb, nh) && tvo[tm][ph] != (int)OOB;
        const f32x4 yv = unpack_lo(pre_m[tm][ph][mb][nh]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = yv[e] > 0.f ? v[e] : slope_m * v[e];
        if (p.mask_sums) {       // sums of what is STORED (bf16-rounded), as a pass over the stored tensor would see it
          const egne_bf16x4 r4 = __builtin_convertvector(v, egne_bf16x4);
#pragma unroll
          for (int e = 0; e < 4; ++e) lms[((mb * 2 + nh) * 4 + e) * 64 + lane] += on ? (double)(float)r4[e] : 0.;
        }
      }
      const u32x2 pk = __builtin_bit_cast(u32x2, __builtin_convertvector(v, egne_bf16x4));
      if (PP == 1 || tpar == 0) prevp[0][tm][ph][mb][nh] = pk; else prevp[PP - 1][tm][ph][mb][nh] = pk;
    };
    // store group Gs = (tm, ph, mb): both halves nh of a block as ONE 16-byte store (8-byte stores where the slice ends or starts on a multiple of 4
    // that is not one of 8 -- uniform per launch)
    const bool narrow = ((p.Cout_store | p.out_ch_off | (int)p.out_pix_stride) & 4) != 0;
    auto store_group = [&](auto gc) {
      constexpr int Gi = decltype(gc)::value, mb = Gi % MB, ph = (Gi / MB) & 1, tm = Gi / MB / 2;
      // (the tile being stored is the PREVIOUS one: the other parity; after the last tile tpar has flipped once more)
      const u32x2 p0 = (PP == 1 || tpar == 1) ? prevp[0][tm][ph][mb][0] : prevp[PP - 1][tm][ph][mb][0];
      const u32x2 p1 = (PP == 1 || tpar == 1) ? prevp[0][tm][ph][mb][1] : prevp[PP - 1][tm][ph][mb][1];
      if (narrow) {
        __builtin_amdgcn_raw_buffer_store_b64(p0, rout, jok(mb, 0) ? tvo[tm][ph] : (int)OOB, (mb * 32) * 2, 0);
        __builtin_amdgcn_raw_buffer_store_b64(p1, rout, jok(mb, 1) ? tvo[tm][ph] : (int)OOB, (mb * 32 + 4) * 2, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{p0[0], p0[1], p1[0], p1[1]}, rout, jok(mb, 1) ? tvo[tm][ph] : (int)OOB, (mb * 32) * 2, 0);
      }
    };
    bool have_prev = false;
    // The jobs of a tile as straight-line code per position -- FIRST job (the previous tile's stores ride on its first four taps), middle
    // jobs, LAST job (residual / mask vectors requested at tap 5, hand-over behind it) -- so that the compiler sees where the packed
    // results of the previous tile and the prefetched vectors are live: as conditions inside one loop body both sets stayed allocated
    // through every tap and the 64-channel form spilled 250 registers.
    auto run_job = [&](auto firstc, auto lastc, const Tile& tl, int ch, int s, u32x2 (&pre_r)[RW][2][MB][2], u32x2 (&pre_m)[RW][2][MB][2]) {
      constexpr bool FIRST = decltype(firstc)::value, LAST = decltype(lastc)::value;
      const char* Timg = (const char*)(ldsb + (s & 1) * IMG);
      const egne_bf16* wb = lw + (STREAM ? (s & 1) : ch) * WCHM + wl;
      auto prefetch_rm = [&]() {
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(resp ? resp + (long long)tl.b * H * W * p.res_pix_stride : nullptr, resp ? frame_res : 0u);
        const __amdgpu_buffer_rsrc_t rm = make_rsrc(mskp ? mskp + (long long)tl.b * H * W * p.mask_pix_stride : nullptr, mskp ? frame_msk : 0u);
#pragma unroll
        for (int tm = 0; tm < RW; ++tm)
#pragma unroll
          for (int ph = 0; ph < 2; ++ph) {
            const int yy = tl.y0 + row0 + tm, x = tl.x0 + ph * 16 + l15;
            const bool okp = yy < H && x < W;
            const int orr = okp ? ((yy * W + x) * (int)p.res_pix_stride + p.res_ch_off + nbase) * 2 : (int)OOB;
            const int om = okp ? ((yy * W + x) * (int)p.mask_pix_stride + p.mask_ch_off + nbase) * 2 : (int)OOB;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
              for (int nh = 0; nh < 2; ++nh) {
                if (resp) pre_r[tm][ph][mb][nh] = __builtin_amdgcn_raw_buffer_load_b64(rr, jok(mb, nh) ? orr : (int)OOB, (mb * 32 + nh * 4) * 2, 0);
                if (mskp) pre_m[tm][ph][mb][nh] = __builtin_amdgcn_raw_buffer_load_b64(rm, jok(mb, nh) ? om : (int)OOB, (mb * 32 + nh * 4) * 2, 0);
              }
          }
      };
      // operands of tap t + 1 are requested before the MFMAs of tap t (two register sets).  NB = blocks computed (mbn).
      // (Round 6 also ran the taps column-major with ONE set of activation fragments, each halo row's fragment serving the taps (ky, tm)
      //  with tm + ky = row: 24 + 36 fragment reads per job instead of 36 + 36.  Same time on every shape -- the loop waits on the LATENCY
      //  of a tap's reads, not on the LDS array's throughput -- and 20 registers more; this is the plain row-major form.)
      auto job = [&](auto nbc) {
        constexpr int NB = decltype(nbc)::value;
        b8 wh[2][NB][2], ah[2][2][2];
        auto fetch = [&](auto tc) {
          constexpr int T = decltype(tc)::value, Bq = T & 1;
#pragma unroll
          for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) wh[Bq][mb][nh] = *(const b8*)&wb[(T * MB + mb) * 1024 + nh * 128];
#pragma unroll
          for (int tm = 0; tm < RW; ++tm)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
              const int c = (tm + T / 3) * HWd + ph * 16 + T % 3;
              ah[Bq][tm][ph] = *(const b8*)(Timg + atab[c & 1][(c >> 1) & 3] + c * 64);
            }
        };
        fetch(std::integral_constant<int, 0>{});
        [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
          (([&] {
            constexpr int S = Ss, Bq = S & 1;
            if constexpr (S + 1 < 9) fetch(std::integral_constant<int, S + 1>{});
            // the previous tile's results leave between the MFMAs of the first four taps of this tile's first job (MB stores of 16 bytes
            // per tap); the residual / mask vectors of the tile are requested at tap 5 of its last job
            if constexpr (FIRST && S < 4) {
              if (have_prev) {
                [&]<int... Gs>(std::integer_sequence<int, Gs...>) {
                  (store_group(std::integral_constant<int, (RW * MB / 2) * S + Gs>{}), ...);
                }(std::make_integer_sequence<int, RW * MB / 2>{});
              }
            }
            if constexpr (RM && LAST && S == 5) prefetch_rm();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tm = 0; tm < RW; ++tm)
#pragma unroll
              for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int mb = 0; mb < NB; ++mb)
#pragma unroll
                  for (int nh = 0; nh < 2; ++nh) {
                    f32x4& c = acc[tm][ph][mb][nh];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[Bq][mb][nh], ah[Bq][tm][ph], c, 0, 0, 0);
                  }
            __builtin_amdgcn_sched_barrier(0);
          }()), ...);
        }(std::make_integer_sequence<int, 9>{});
      };
      if constexpr (MB == 1) job(std::integral_constant<int, 1>{});
      else { if (mbn == 2) job(std::integral_constant<int, 2>{}); else job(std::integral_constant<int, 1>{}); }
    };
    auto hand_over = [&](const Tile& tl, const u32x2 (&pre_r)[RW][2][MB][2], const u32x2 (&pre_m)[RW][2][MB][2]) {
      rout = make_rsrc(outp + (long long)tl.b * H * W * p.out_pix_stride, frame_out);
#pragma unroll
      for (int tm = 0; tm < RW; ++tm)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          const int yy = tl.y0 + row0 + tm, x = tl.x0 + ph * 16 + l15;
          const bool okp = yy < H && x < W;
          tvo[tm][ph] = okp ? ((yy * W + x) * (int)p.out_pix_stride + p.out_ch_off + nbase) * 2 : (int)OOB;
        }
      [&]<int... Gs>(std::integer_sequence<int, Gs...>) { (finish_group(std::integral_constant<int, Gs>{}, pre_r, pre_m), ...); }(std::make_integer_sequence<int, NG>{});
      have_prev = true;
      if (p.stats_ws) {
        // InstanceNorm / BatchNorm statistics of the consumer without a pass over the tensor (round 5, training plans): sums and sums of
        // squares of what this wave STORES of the tile (bf16-rounded), one chunk per (tile, consumer wave):
        // stats_ws [b][tile * 4 + cw][Cout_store][2] doubles, finished by egne_norm_stats_finish in a fixed order
        double* lst = lstat_all + cw * 16 * 64;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {              // (one 32-channel block at a time through the wave's [16][64] staging area)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) {            // (one half of the channels at a time: eight live sums instead of sixteen)
            double ss[4], qq[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { ss[e] = 0.; qq[e] = 0.; }
#pragma unroll
            for (int tm = 0; tm < RW; ++tm)
#pragma unroll
              for (int ph = 0; ph < 2; ++ph) {
                const bool okp = tvo[tm][ph] != (int)OOB;
                const f32x4 r4 = unpack_lo((PP == 1 || tpar == 0) ? prevp[0][tm][ph][mb][nh] : prevp[PP - 1][tm][ph][mb][nh]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const double f = okp ? (double)r4[e] : 0.;
                  ss[e] += f; qq[e] += f * f;
                }
              }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              lst[((nh * 4 + e) * 2 + 0) * 64 + lane] = ss[e];
              lst[((nh * 4 + e) * 2 + 1) * 64 + lane] = qq[e];
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (one wave: its own LDS writes are visible to its own reads after this)
          // lane (kg' = lane >> 4, v = lane & 15): value v = (nh, e, statistic) of k-group kg' over its 16 pixel lanes
          const int kgq = lane >> 4, vq = lane & 15;
          double t = 0.;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const double2 q2 = *(const double2*)&lst[vq * 64 + kgq * 16 + 2 * j];
            t += q2.x + q2.y;
          }
          const int nhq = vq >> 3, eq = (vq >> 1) & 3, stq = vq & 1;
          const int n = (cb * MB + mb) * 32 + 8 * kgq + 4 * nhq + eq;
          const long long chunk = (long long)(tl.y0 / TH) * tiles_x + tl.x0 / TW;
          if (n < p.Cout_store)
            p.stats_ws[((((long long)tl.b * p.stats_nchunk + chunk * 4 + cw) * p.Cout_store) + n) * 2 + stq] = t;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    };
    using TT = std::true_type;
    using FF = std::false_type;
    int s = 0;                                           // job counter: its parity selects the image (and the streamed weights' buffer)
    lds_barrier();
    B3_T0();
    for (int ti = 0; ti < ntl; ++ti) {
      const Tile tl = decode(tile_at(ti));
      // accumulators start from the bias: sixteen LDS reads straight into their registers (zero fill + bias add at the end were 130 vector
      // instructions per tile and 32-channel block); the opaque offset keeps the compiler from turning them into four reads + copies
#pragma unroll
      for (int tm = 0; tm < RW; ++tm)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          int lo = 8 * kg;
          asm volatile("" : "+v"(lo));                    // (opaque OFFSET: an opaque pointer would lose its LDS address space -- flat loads)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) acc[tm][ph][mb][nh] = *(const f32x4*)&lepi[lo + mb * 32 + nh * 4];
        }
      u32x2 pre_r[RW][2][MB][2], pre_m[RW][2][MB][2];      // (assigned at tap 5 of the tile's last job, read at its hand-over)
      if constexpr (KCH == 1) {
        run_job(TT{}, TT{}, tl, 0, s, pre_r, pre_m);
        B3_ADD(0);
        lds_barrier(); ++s;
        B3_ADD(1);
      } else {                                           // nk >= 2 (the launcher sends streamed weights only at three chunks and more)
        run_job(TT{}, FF{}, tl, 0, s, pre_r, pre_m);
        B3_ADD(0);
        lds_barrier(); ++s;
        B3_ADD(1);
        for (int ch = 1; ch < nk - 1; ++ch) {
          run_job(FF{}, FF{}, tl, ch, s, pre_r, pre_m);
          B3_ADD(0);
          lds_barrier(); ++s;
          B3_ADD(1);
        }
        run_job(FF{}, TT{}, tl, nk - 1, s, pre_r, pre_m);
        B3_ADD(0);
        lds_barrier(); ++s;
        B3_ADD(1);
      }
      hand_over(tl, pre_r, pre_m);
      tpar ^= 1;
      B3_ADD(2);
    }
    B3_OUT();
    for (; s < nloop; ++s) lds_barrier();                // (the producers run an even number of steps)
    if (have_prev)
      [&]<int... Gs>(std::integer_sequence<int, Gs...>) { (store_group(std::integral_constant<int, Gs>{}), ...); }(std::make_integer_sequence<int, 2 * RW * MB>{});
    if (mskp && p.mask_sums) {
      // over the 16 pixel lanes that share the channel vector (fixed order), then row (workgroup, consumer wave) of mask_sums [rows][Cout_store]
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double msum[MB * RW][4];
#pragma unroll
      for (int a = 0; a < NG; ++a) (&msum[0][0])[a] = lms[a * 64 + lane];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
        for (int a = 0; a < NG; ++a) (&msum[0][0])[a] += __shfl_xor((&msum[0][0])[a], o);
      }
      if (l15 == 0) {
        float* row = p.mask_sums + ((long long)blockIdx.x * 4 + cw) * p.Cout_store;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh)
            if (jok(mb, nh)) {
#pragma unroll
              for (int e = 0; e < 4; ++e) row[nbase + mb * 32 + nh * 4 + e] = (float)msum[mb * 2 + nh][e];
            }
      }
    }
  }
}

}  // namespace
extern "C" int egne_conv3x3_bf16_sum_rows(void) { return 256 * 4; }
namespace {

template <int KCH, int MB, int RM, int RW = 2>
int launch_b3(const egne_conv_desc& d, const egne_bf16* wf, hipStream_t st) {
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y * d.B, ncb = d.CoutP / 32, nrun = (d.Cout_store + 32 * MB - 1) / (32 * MB);
  constexpr size_t lds = ((size_t)2 * IMG + (size_t)(KCH == 0 ? 2 : KCH) * MB * WCH) * sizeof(egne_bf16) + 4 * 16 * 64 * sizeof(double) +
                         32 * MB * sizeof(float);
  if (d.stats_ws && d.mask_sums) return egne::fail(EGNE_ERR_ARG, "conv3x3_bf16: stats_ws and mask_sums share the waves' LDS scratch: one of them per launch");
  if (d.stats_ws && d.stats_nchunk != tiles_x * tiles_y * 4)
    return egne::fail(EGNE_ERR_ARG, "conv3x3_bf16: stats_nchunk %d, the launch has %d chunks per frame (tiles of 32 x 8 pixels x 4 waves)", d.stats_nchunk, tiles_x * tiles_y * 4);
  static_assert(lds <= 163840, "LDS budget");
  const bool once = egne::raise_lds((const void*)conv3x3_bf16_kernel<KCH, MB, RM, RW>, lds);
  if (!once) return egne::fail(EGNE_ERR_LAUNCH, "conv3x3_bf16: cannot raise the dynamic LDS limit to %zu", lds);
  // 256 workgroups = 8 XCDs x 32; the nrun blocks of a worker sit on one XCD: 32 / nrun workers per XCD
  hipLaunchKernelGGL((conv3x3_bf16_kernel<KCH, MB, RM, RW>), dim3(256), dim3(64 * (4 + 8 / RW)), lds, st, d, wf, tiles_x, tiles_y, ntiles, ncb, nrun);
  return egne::check_launch("egne_conv3x3_bf16_fwd");
}

// OIHW fp32 -> bf16 (round to nearest even) in fragment order [tap][Ktot/16][CoutP/32][lane = h*32 + n%32][8]: k = 16*k16 + 8*h + j
__global__ void pack_weight_bf16frag_k(const float* __restrict__ w, int Cout, int Cin, int T, int CoutP, int Ktot,
                                       egne_bf16* __restrict__ out) {
  const long long total = (long long)T * CoutP * Ktot;
  const int NT = CoutP >> 5;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), nn = (int)((i >> 3) & 31), h = (int)((i >> 8) & 1);
    long long q = i >> 9;
    const int nt = (int)(q % NT); q /= NT;
    const int k16 = (int)(q % (Ktot >> 4));
    const int t = (int)(q / (Ktot >> 4));
    const int n = nt * 32 + nn, k = k16 * 16 + h * 8 + j;
    out[i] = (egne_bf16)((n < Cout && k < Cin) ? w[((long long)n * Cin + k) * T + t] : 0.f);
  }
}

// the same fragments for the DATA GRADIENT w.r.t. input channels [c0, c0 + cn) of the forward convolution w [Co][Ci][kh][kw]: the
// ordinary convolution over gz with W'[ci][co][j][i] = w[co][c0 + ci][kh-1-j][kw-1-i], read straight out of the forward weight
// (the host used to build W' with two flips and a strided copy per layer and step)
__global__ void pack_weight_bf16frag_dgrad_k(const float* __restrict__ w, int Co, int Ci, int c0, int cn, int T, int CoutP, int Ktot,
                                             egne_bf16* __restrict__ out) {
  const long long total = (long long)T * CoutP * Ktot;
  const int NT = CoutP >> 5;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), nn = (int)((i >> 3) & 31), h = (int)((i >> 8) & 1);
    long long q = i >> 9;
    const int nt = (int)(q % NT); q /= NT;
    const int k16 = (int)(q % (Ktot >> 4));
    const int t = (int)(q / (Ktot >> 4));
    const int n = nt * 32 + nn, k = k16 * 16 + h * 8 + j;          // n: input channel of the forward layer, k: its output channel
    out[i] = (egne_bf16)((n < cn && k < Co) ? w[((long long)k * Ci + c0 + n) * T + (T - 1 - t)] : 0.f);
  }
}

}  // namespace

extern "C" int egne_pack_conv_weight_bf16frag_dgrad(const float* w_oihw, int Cout, int Cin, int kh, int kw, int c0, int cn, int CoutP, int Ktot,
                                                    void* wfrag, void* stream) {
  EGNE_REQUIRE(w_oihw && wfrag && Cout > 0 && Cin > 0 && c0 >= 0 && cn > 0 && c0 + cn <= Cin && CoutP >= cn && CoutP % 32 == 0 && Ktot >= Cout &&
               Ktot % 32 == 0, "pack_bf16frag_dgrad: bad sizes Cout %d Cin %d c0 %d cn %d CoutP %d Ktot %d", Cout, Cin, c0, cn, CoutP, Ktot);
  long long total = (long long)kh * kw * CoutP * Ktot, g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pack_weight_bf16frag_dgrad_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, c0, cn, kh * kw,
                     CoutP, Ktot, (egne_bf16*)wfrag);
  return egne::check_launch("egne_pack_conv_weight_bf16frag_dgrad");
}

extern "C" int egne_pack_conv_weight_bf16frag(const float* w_oihw, int Cout, int Cin, int kh, int kw, int CoutP, int Ktot, void* wfrag,
                                              void* stream) {
  EGNE_REQUIRE(w_oihw && wfrag && Cout > 0 && Cin > 0 && CoutP >= Cout && CoutP % 32 == 0 && Ktot >= Cin && Ktot % 32 == 0,
               "pack_bf16frag: bad sizes Cout %d Cin %d CoutP %d Ktot %d", Cout, Cin, CoutP, Ktot);
  long long total = (long long)kh * kw * CoutP * Ktot, g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pack_weight_bf16frag_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, kh * kw, CoutP,
                     Ktot, (egne_bf16*)wfrag);
  return egne::check_launch("egne_pack_conv_weight_bf16frag");
}

extern "C" int egne_conv3x3_bf16_fwd(const egne_conv_desc* dp, const void* wfrag, void* stream) {
  EGNE_REQUIRE(dp && wfrag, "conv3x3_bf16: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.dtype == 1, "conv3x3_bf16: the descriptor must say bf16 tensors (dtype 1)");
  EGNE_REQUIRE(d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad_mode == 0 && d.ngroups == 1 && d.nseg == 1 && d.pad_h == 1 &&
               d.pad_w == 1 && d.dil[0] == 1 && d.Ho == d.H && d.Wo == d.W && !d.pool_out && !d.dyn_scale && !d.absmax_out &&
               (!d.stats_ws || ((uintptr_t)d.stats_ws & 15) == 0), "conv3x3_bf16: geometry / options not supported");
  const egne_seg& g = d.seg[0];
  EGNE_REQUIRE(g.ptr && g.Cp % 8 == 0 && (g.Cp + 31) / 32 * 32 == d.Ktot && d.Ktot >= 32 && d.Ktot <= 1024 && g.ch_off % 8 == 0 &&
               g.pix_stride % 8 == 0 && ((uintptr_t)g.ptr & 15) == 0 && (g.scale == nullptr) == (g.shift == nullptr) &&
               g.ch_off + g.Cp <= g.pix_stride, "conv3x3_bf16: input slice (16-byte groups of 8 channels)");
  EGNE_REQUIRE(d.CoutP % 32 == 0 && d.CoutP <= 256 && d.Cout_store >= 8 && d.Cout_store <= d.CoutP && d.Cout_store % 4 == 0 && d.out &&
               ((uintptr_t)d.out & 15) == 0 && d.out_pix_stride % 4 == 0 && d.out_ch_off % 4 == 0 &&
               d.out_ch_off + d.Cout_store <= d.out_pix_stride && (!d.bias || ((uintptr_t)d.bias & 15) == 0), "conv3x3_bf16: output");
  EGNE_REQUIRE(!d.residual || (((uintptr_t)d.residual & 15) == 0 && d.res_pix_stride % 4 == 0 && d.res_ch_off % 4 == 0), "conv3x3_bf16: residual alignment");
  EGNE_REQUIRE(!d.post_scale && !d.post_shift, "conv3x3_bf16: no post affine (bf16 plans are training plans: BatchNorm is never folded)");
  EGNE_REQUIRE(!d.mask_y || (((uintptr_t)d.mask_y & 7) == 0 && d.mask_pix_stride % 4 == 0 && d.mask_ch_off % 4 == 0 &&
                              (long long)d.H * d.W * d.mask_pix_stride * 2 < (1ll << 31) && !d.post_scale &&
                              (d.mask_act == EGNE_ACT_NONE || d.mask_act == EGNE_ACT_RELU || d.mask_act == EGNE_ACT_LEAKY)),
               "conv3x3_bf16: mask tensor (8-byte groups of 4 channels, no post affine)");
  EGNE_REQUIRE(!d.mask_sums || (d.mask_y && ((uintptr_t)d.mask_sums & 3) == 0), "conv3x3_bf16: mask_sums needs mask_y");
  EGNE_REQUIRE(((uintptr_t)wfrag & 15) == 0, "conv3x3_bf16: weight alignment");
  EGNE_REQUIRE((long long)d.H * d.W * g.pix_stride * 2 < (1ll << 31) && (long long)d.H * d.W * d.out_pix_stride * 2 < (1ll << 31) &&
               (!d.residual || (long long)d.H * d.W * d.res_pix_stride * 2 < (1ll << 31)), "conv3x3_bf16: frame too large for 32-bit byte offsets");
  hipStream_t st = (hipStream_t)stream;
  const egne_bf16* wf = (const egne_bf16*)wfrag;
  // two 32-channel output blocks per workgroup wherever the pack has them (EGNE_B3_MB=1: the round-5 form, one block per workgroup)
  static const int mb_env = [] { const char* e = getenv("EGNE_B3_MB"); return e ? atoi(e) : 2; }();
  const int rm = d.mask_y ? 2 : (d.residual ? 1 : 0);
  // EGNE_B3_RW=1: eight consumer waves of one row each where the launch writes neither statistics nor mask sums.  Measured (round 6,
  // scratch/b3_bench.py, 128 frames): 64 -> 64 at 120x160 0.193 -> 0.177 ms, at 240x320 0.841 -> 0.819, with a residual 1.047 -> 0.987,
  // streamed-weights layers (128+ channels) level; on the training step level within the boxes' noise (1607 / 1613 vs 1608 / 1604 frames/s,
  // chz 64: 688 / 694 vs 693 / 692) -- the launches that carry statistics keep the four-wave form anyway.  OFF by default.
  static const int rw_env = [] { const char* e = getenv("EGNE_B3_RW"); return e ? atoi(e) : 2; }();
  if (d.CoutP >= 64 && mb_env >= 2 && rm < 2 && rw_env == 1 && !d.stats_ws) {
    switch (d.Ktot) {
      case 32: return rm ? launch_b3<1, 2, 1, 1>(d, wf, st) : launch_b3<1, 2, 0, 1>(d, wf, st);
      case 64: return rm ? launch_b3<2, 2, 1, 1>(d, wf, st) : launch_b3<2, 2, 0, 1>(d, wf, st);
      default: return rm ? launch_b3<0, 2, 1, 1>(d, wf, st) : launch_b3<0, 2, 0, 1>(d, wf, st);
    }
  }
  if (d.CoutP >= 64 && mb_env >= 2 && rm < 2) {
    switch (d.Ktot) {
      case 32: return rm ? launch_b3<1, 2, 1>(d, wf, st) : launch_b3<1, 2, 0>(d, wf, st);
      case 64: return rm ? launch_b3<2, 2, 1>(d, wf, st) : launch_b3<2, 2, 0>(d, wf, st);
      default: return rm ? launch_b3<0, 2, 1>(d, wf, st) : launch_b3<0, 2, 0>(d, wf, st);
    }
  }
  switch (d.Ktot) {
    case 32: return rm == 2 ? launch_b3<1, 1, 2>(d, wf, st) : (rm ? launch_b3<1, 1, 1>(d, wf, st) : launch_b3<1, 1, 0>(d, wf, st));
    case 64: return rm == 2 ? launch_b3<2, 1, 2>(d, wf, st) : (rm ? launch_b3<2, 1, 1>(d, wf, st) : launch_b3<2, 1, 0>(d, wf, st));
    case 96: return rm == 2 ? launch_b3<3, 1, 2>(d, wf, st) : (rm ? launch_b3<3, 1, 1>(d, wf, st) : launch_b3<3, 1, 0>(d, wf, st));
    case 128: return rm == 2 ? launch_b3<4, 1, 2>(d, wf, st) : (rm ? launch_b3<4, 1, 1>(d, wf, st) : launch_b3<4, 1, 0>(d, wf, st));
    default: return rm == 2 ? launch_b3<0, 1, 2>(d, wf, st) : (rm ? launch_b3<0, 1, 1>(d, wf, st) : launch_b3<0, 1, 0>(d, wf, st));
  }
}
