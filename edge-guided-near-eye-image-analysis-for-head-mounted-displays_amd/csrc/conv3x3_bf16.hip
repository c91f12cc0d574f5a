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
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef egne_bf16x8 b8;

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

__device__ __forceinline__ f32x4 unpack_lo(u32x2 v) {     // 4 bf16 -> 4 floats
  const u32x4 w = {v[0] << 16, v[0] & 0xffff0000u, v[1] << 16, v[1] & 0xffff0000u};
  return __builtin_bit_cast(f32x4, w);
}

// KCH: 32-channel chunks of the input slice (1..4: all weights resident) or 0: any number of chunks (p.Ktot / 32), the weights of
// a job's chunk (18 KB) STREAMED into one of two LDS weight buffers by the producers one job ahead.
// ncb = output blocks of 32 channels in the pack, nrun = blocks that hold stored channels.
template <int KCH>
__global__ __launch_bounds__(512)
void conv3x3_bf16_kernel(const egne_conv_desc p, const egne_bf16* __restrict__ wfrag, int tiles_x, int tiles_y, int ntiles, int ncb,
                         int nrun) {
  extern __shared__ __attribute__((aligned(16))) egne_bf16 ldsb[];
  egne_bf16* const lw = ldsb + 2 * IMG;                  // weights behind the two images
  // behind the weights: per consumer wave [16 values][64 lanes] DOUBLES -- the per-tile channel sums of egne_conv_desc.stats_ws on their
  // way from "4 channels x 4 pixels per lane" to "one (channel, statistic) per lane" (fp64 throughout: E[x^2] - mean^2 of a nearly constant
  // channel cancels seven digits, and float partial sums moved the decoder's gradients by 30 %), or -- a launch has one or the other --
  // the running bias sums of a masked data gradient (mask_sums: [8][64] doubles per wave)
  double* const lstat_all = (double*)(lw + (size_t)(KCH == 0 ? 2 : KCH) * WCH);

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
  auto tile_at = [&](int i) { return worker + i * nworkers; };
  struct Tile { int b, y0, x0; };
  auto decode = [&](int t) {
    Tile r;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    r.b = t; r.y0 = ty * TH; r.x0 = tx * TW;
    return r;
  };
  int ntl = 0;
  while (tile_at(ntl) < ntiles) ++ntl;
  constexpr bool STREAM = KCH == 0;
  const int nk = STREAM ? p.Ktot / 32 : KCH;             // chunks per tile
  const int nmine = ntl * nk;                            // jobs: (tile, chunk), chunk fastest
  const int nloop = (nmine + 1) & ~1;                    // both roles run an even number of steps (register buffer = step parity)

  // the block's weights: fragments (tap, k16, nt = cb) of the pack [tap][Ktot/16][CoutP/32][lane][8]
  const int KT16 = nk * 2;
  if constexpr (!STREAM) {
    for (int it = tid; it < KCH * 9 * 2 * 64; it += 512) {            // 16-byte items, LDS order [chunk][tap][ks][lane]
      const int l = it & 63, ks = (it >> 6) & 1, r = it >> 7, tap = r % 9, ch = r / 9;
      const long long src = (((long long)tap * KT16 + ch * 2 + ks) * ncb + cb) * 512 + l * 8;
      *(u32x4*)&lw[(long long)it * 8] = *(const u32x4*)(wfrag + src);
    }
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
    struct Job { Tile t; int ch; };
    auto job_at = [&](int j) { Job r; r.t = decode(tile_at(j / nk)); r.ch = j % nk; return r; };
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
    auto load_aff = [&](const Job& jb, auto bc) {
      constexpr int BUF = decltype(bc)::value;
      if (sg.scale && nmine > 0) {       // (a workgroup without tiles decodes a frame past the batch: no table row to read)
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
    // STREAM: the 1152 16-byte pieces of a chunk's weights (LDS order [tap][ks][lane]), five per producer lane (the last
    // round is partial), requested one job ahead and written at the start of the next step
    constexpr int NWI = 5;
    u32x4 wreg[STREAM ? NWI : 1];
    const unsigned wbytes = 9u * (unsigned)KT16 * (unsigned)ncb * 1024u;
    const __amdgpu_buffer_rsrc_t rwf = make_rsrc(wfrag, wbytes);
    auto w_issue = [&](int ch, bool on) {
#pragma unroll
      for (int i = 0; i < (STREAM ? NWI : 0); ++i) {
        const int it = tid + 256 * i, l = it & 63, ks = (it >> 6) & 1, tap = it >> 7;
        const int off = (on && it < 1152) ? ((((tap * KT16 + ch * 2 + ks) * ncb + cb) * 512 + l * 8) * 2) : (int)OOB;
        wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rwf, off, 0, 0);
      }
    };
    auto w_store = [&](int parity) {
#pragma unroll
      for (int i = 0; i < (STREAM ? NWI : 0); ++i)
        if (tid + 256 * i < 1152) *(u32x4*)&lw[parity * WCH + (tid + 256 * i) * 8] = wreg[i];
    };
    // step s (job s): convert job s+1 out of register buffer (s+1)&1 and refill every freed register with job s+3
    auto step = [&](int s, auto bc) {
      constexpr int BUF = decltype(bc)::value;          // = (s + 1) & 1
      const bool c_on = s + 1 < nmine, i_on = s + 3 < nmine;
      const Job jc = job_at(c_on ? s + 1 : 0), ji = job_at(i_on ? s + 3 : 0);
      if constexpr (STREAM) {
        w_store((s + 1) & 1);                             // weights of job s+1 (requested in step s-1)
        w_issue(job_at(s + 2 < nmine ? s + 2 : 0).ch, s + 2 < nmine);
      }
      load_aff(job_at(s + 2 < nmine ? s + 2 : 0), std::integral_constant<int, BUF ^ 1>{});
      egne_bf16* img = ldsb + ((s + 1) & 1) * IMG;
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (([&] {
          if (c_on) convert1(jc, img, bc, std::integral_constant<int, Is>{});
          issue1(ji, i_on, bc, std::integral_constant<int, Is>{});
        }()), ...);
      }(std::make_integer_sequence<int, NI>{});
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    {   // prologue: jobs 0 and 1 requested, job 0 converted (its registers refilled with job 2)
      const Job j0 = job_at(0), j1 = job_at(nmine > 1 ? 1 : 0), j2 = job_at(nmine > 2 ? 2 : 0);
      if constexpr (STREAM) {
        w_issue(j0.ch, nmine > 0);
        w_store(0);
        w_issue(j1.ch, nmine > 1);
      }
      load_aff(j0, B0{});
      load_aff(j1, B1{});
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
    for (int s = 0; s < nloop; s += 2) {
      step(s, B1{});
      lds_barrier();
      step(s + 1, B0{});
      lds_barrier();
    }
  } else {
    // =================================================================== consumers: 9 taps per job from LDS only
    // v_mfma_f32_16x16x32_bf16: one instruction per (16 channels, 16 pixels, 32 input channels of a tap); a wave's two rows x 32
    // pixels x 32 channels are EIGHT independent accumulators
    const int cw = wave - 4, row0 = cw * 2;
    const int l15 = lane & 15, kg = lane >> 4;
    egne_bf16* const outp = (egne_bf16*)p.out;
    const egne_bf16* const resp = (const egne_bf16*)p.residual;
    const unsigned frame_out = (unsigned)H * W * (unsigned)p.out_pix_stride * 2u;
    const unsigned frame_res = (unsigned)H * W * (unsigned)p.res_pix_stride * 2u;
    const float slope_out = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
    // transposed product (weights as the A operand): the lane holds channels n = 32 cb + 16 nh + 4 kg + r of pixel 16 ph + l15
    f32x4 b4[2], ps4[2], pt4[2];
    bool jok[2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const int n = cb * 32 + nh * 16 + 4 * kg;
      b4[nh] = p.bias ? *(const f32x4*)(p.bias + n) : (f32x4)(0.f);
      ps4[nh] = p.post_scale ? *(const f32x4*)(p.post_scale + n) : (f32x4)(1.f);
      pt4[nh] = p.post_scale ? *(const f32x4*)(p.post_shift + n) : (f32x4)(0.f);
      jok[nh] = n < p.Cout_store;                        // Cout_store is a multiple of 4
    }
    // operand addresses (independent of the tile).  Activations: pixel q = (row + ky) * 34 + 16 ph + l15 + kx, 8-channel group kg at
    // group kg ^ ((q >> 1) & 3); weights: k-group kg of output channel m = fragment k16 = kg >> 1, lane position (kg & 1) * 32 + m
    int aofs[9][2][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          const int qq = (row0 + tm + tap / 3) * HWd + ph * 16 + l15 + tap % 3;
          aofs[tap][tm][ph] = qq * 32 + ((kg ^ ((qq >> 1) & 3)) << 3);
        }
    const int wl = (kg >> 1) * 512 + ((kg & 1) * 32 + l15) * 8;       // + tap * 1024 + nh * 128 (elements)
    f32x4 acc[2][2][2], prev[2][2][2];
#pragma unroll
    for (int a = 0; a < 8; ++a) { (&acc[0][0][0])[a] = (f32x4)(0.f); (&prev[0][0][0])[a] = (f32x4)(0.f); }
    __amdgpu_buffer_rsrc_t rout = make_rsrc(outp, 0u);
    int tvo[2][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) (&tvo[0][0])[a] = (int)OOB;
    // mask-on-write (egne_conv_desc.mask_y): the launch is the last writer of a gradient slice -- the stored value is v * act'(y), and the
    // wave keeps the sums of what it stores (fp64, all its tiles) for the producing layer's bias gradient
    const egne_bf16* const mskp = (const egne_bf16*)p.mask_y;
    const unsigned frame_msk = (unsigned)H * W * (unsigned)p.mask_pix_stride * 2u;
    const float slope_m = p.mask_act == EGNE_ACT_RELU ? 0.f : (p.mask_act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
    // (the wave's running sums live in LDS, [8 values][64 lanes] doubles per consumer wave behind the statistics staging area: sixteen
    //  registers the MFMA loop of the wider variants does not have)
    double* const lms = lstat_all + cw * 16 * 64;
    if (mskp && p.mask_sums) {
#pragma unroll
      for (int a = 0; a < 8; ++a) lms[a * 64 + lane] = 0.;
    }
    // residual and mask vectors of a tile are requested at the START of its last job (round 5: loaded at hand-over, every tile waited a
    // memory round trip for them -- a masked data gradient took twice the time of a plain one)
    u32x2 pre_r[2][2][2], pre_m[2][2][2];
#pragma unroll
    for (int a = 0; a < 8; ++a) { (&pre_r[0][0][0])[a] = u32x2{0u, 0u}; (&pre_m[0][0][0])[a] = u32x2{0u, 0u}; }
    // values are finished (bias, activation [, post affine, residual]) in place at hand-over; the deferred part is the bare store
    auto finish_group = [&](auto gc) {
      constexpr int Gi = decltype(gc)::value, nh = Gi & 1, ph = (Gi >> 1) & 1, tm = Gi >> 2;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = prev[tm][ph][nh][e] + b4[nh][e];
        v[e] = fmaxf(t, t * slope_out) * ps4[nh][e] + pt4[nh][e];
      }
      if (resp) {
        const f32x4 rv = unpack_lo(pre_r[tm][ph][nh]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rv[e];
      }
      if (mskp) {
        const bool on = jok[nh] && tvo[tm][ph] != (int)OOB;
        const f32x4 yv = unpack_lo(pre_m[tm][ph][nh]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = yv[e] > 0.f ? v[e] : slope_m * v[e];
        if (p.mask_sums) {       // sums of what is STORED (bf16-rounded), as a pass over the stored tensor would see it
          const egne_bf16x4 r4 = __builtin_convertvector(v, egne_bf16x4);
#pragma unroll
          for (int e = 0; e < 4; ++e) lms[(nh * 4 + e) * 64 + lane] += on ? (double)(float)r4[e] : 0.;
        }
      }
      prev[tm][ph][nh] = v;
    };
    auto store_group = [&](auto gc) {
      constexpr int Gi = decltype(gc)::value, nh = Gi & 1, ph = (Gi >> 1) & 1, tm = Gi >> 2;
      const u32x2 pk = __builtin_bit_cast(u32x2, __builtin_convertvector(prev[tm][ph][nh], egne_bf16x4));
      __builtin_amdgcn_raw_buffer_store_b64(pk, rout, jok[nh] ? tvo[tm][ph] : (int)OOB, nh * 32, 0);
    };
    bool have_prev = false;
    lds_barrier();
    for (int s = 0; s < nloop; ++s) {
      if (s < nmine) {
        const int ch = s % nk;
        const Tile tl = decode(tile_at(s / nk));
        const egne_bf16* Timg = ldsb + (s & 1) * IMG;
        const egne_bf16* wb = lw + (STREAM ? (s & 1) : ch) * WCH + wl;
        if (ch == 0) {
#pragma unroll
          for (int a = 0; a < 8; ++a) (&acc[0][0][0])[a] = (f32x4)(0.f);
        }
        if (ch == nk - 1 && (resp || mskp)) {
          const __amdgpu_buffer_rsrc_t rr = make_rsrc(resp ? resp + (long long)tl.b * H * W * p.res_pix_stride : nullptr, resp ? frame_res : 0u);
          const __amdgpu_buffer_rsrc_t rm = make_rsrc(mskp ? mskp + (long long)tl.b * H * W * p.mask_pix_stride : nullptr, mskp ? frame_msk : 0u);
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
              const int yy = tl.y0 + row0 + tm, x = tl.x0 + ph * 16 + l15;
              const bool okp = yy < H && x < W;
              const int orr = okp ? ((yy * W + x) * (int)p.res_pix_stride + p.res_ch_off + cb * 32 + 4 * kg) * 2 : (int)OOB;
              const int om = okp ? ((yy * W + x) * (int)p.mask_pix_stride + p.mask_ch_off + cb * 32 + 4 * kg) * 2 : (int)OOB;
#pragma unroll
              for (int nh = 0; nh < 2; ++nh) {
                if (resp) pre_r[tm][ph][nh] = __builtin_amdgcn_raw_buffer_load_b64(rr, jok[nh] ? orr : (int)OOB, nh * 32, 0);
                if (mskp) pre_m[tm][ph][nh] = __builtin_amdgcn_raw_buffer_load_b64(rm, jok[nh] ? om : (int)OOB, nh * 32, 0);
              }
            }
        }
        // operands of tap t + 1 are requested before the MFMAs of tap t (two register sets)
        b8 wh[2][2], ah[2][2][2];
        auto fetch = [&](auto tc) {
          constexpr int T = decltype(tc)::value, Bq = T & 1;
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) wh[Bq][nh] = *(const b8*)&wb[T * 1024 + nh * 128];
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) ah[Bq][tm][ph] = *(const b8*)&Timg[aofs[T][tm][ph]];
        };
        fetch(std::integral_constant<int, 0>{});
        [&]<int... Ts>(std::integer_sequence<int, Ts...>) {
          (([&] {
            constexpr int t = Ts, Bq = t & 1;
            if constexpr (t + 1 < 9) fetch(std::integral_constant<int, t + 1>{});
            // the previous tile's results leave between the MFMAs of this tile's first job (8 stores of 8 bytes)
            if constexpr (t < 8) { if (ch == 0 && have_prev) store_group(std::integral_constant<int, t>{}); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
              for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                  f32x4& c = acc[tm][ph][nh];
                  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[Bq][nh], ah[Bq][tm][ph], c, 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
          }()), ...);
        }(std::make_integer_sequence<int, 9>{});
        if (ch == nk - 1) {                                // tile complete: hand it to the deferred stores
          const int y = tl.y0 + row0;
#pragma unroll
          for (int a = 0; a < 8; ++a) (&prev[0][0][0])[a] = (&acc[0][0][0])[a];
          rout = make_rsrc(outp + (long long)tl.b * H * W * p.out_pix_stride, frame_out);

#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
              const int yy = y + tm, x = tl.x0 + ph * 16 + l15;
              const bool okp = yy < H && x < W;
              tvo[tm][ph] = okp ? ((yy * W + x) * (int)p.out_pix_stride + p.out_ch_off + cb * 32 + 4 * kg) * 2 : (int)OOB;

            }
          [&]<int... Gs>(std::integer_sequence<int, Gs...>) { (finish_group(std::integral_constant<int, Gs>{}), ...); }(std::make_integer_sequence<int, 8>{});
          have_prev = true;
          if (p.stats_ws) {
            // InstanceNorm / BatchNorm statistics of the consumer without a pass over the tensor (round 5, training plans): sums and sums of
            // squares of what this wave STORES of the tile (bf16-rounded), one chunk per (tile, consumer wave):
            // stats_ws [b][tile * 4 + cw][Cout_store][2] doubles, finished by egne_norm_stats_finish in a fixed order
            double* lst = lstat_all + cw * 16 * 64;
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {            // (one half of the channels at a time: eight live sums instead of sixteen)
              double ss[4], qq[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) { ss[e] = 0.; qq[e] = 0.; }
#pragma unroll
              for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                  const bool okp = tvo[tm][ph] != (int)OOB;
                  const egne_bf16x4 r4 = __builtin_convertvector(prev[tm][ph][nh], egne_bf16x4);
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    const double f = okp ? (double)(float)r4[e] : 0.;
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
            const int n = cb * 32 + nhq * 16 + 4 * kgq + eq;
            const long long chunk = (long long)(tl.y0 / TH) * tiles_x + tl.x0 / TW;
            if (n < p.Cout_store)
              p.stats_ws[((((long long)tl.b * p.stats_nchunk + chunk * 4 + cw) * p.Cout_store) + n) * 2 + stq] = t;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
        }
      }
      lds_barrier();
    }
    if (have_prev)
      [&]<int... Gs>(std::integer_sequence<int, Gs...>) { (store_group(std::integral_constant<int, Gs>{}), ...); }(std::make_integer_sequence<int, 8>{});
    if (mskp && p.mask_sums) {
      // over the 16 pixel lanes that share the channel vector (fixed order), then row (workgroup, consumer wave) of mask_sums [rows][Cout_store]
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double msum[2][4];
#pragma unroll
      for (int a = 0; a < 8; ++a) (&msum[0][0])[a] = lms[a * 64 + lane];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
        for (int a = 0; a < 8; ++a) (&msum[0][0])[a] += __shfl_xor((&msum[0][0])[a], o);
      }
      if (l15 == 0) {
        float* row = p.mask_sums + ((long long)blockIdx.x * 4 + cw) * p.Cout_store;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
          if (jok[nh]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) row[cb * 32 + nh * 16 + 4 * kg + e] = (float)msum[nh][e];
          }
      }
    }
  }
}

}  // namespace
extern "C" int egne_conv3x3_bf16_sum_rows(void) { return 256 * 4; }
namespace {

template <int KCH>
int launch_b3(const egne_conv_desc& d, const egne_bf16* wf, hipStream_t st) {
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y * d.B, ncb = d.CoutP / 32, nrun = (d.Cout_store + 31) / 32;
  constexpr size_t lds = ((size_t)2 * IMG + (size_t)(KCH == 0 ? 2 : KCH) * WCH) * sizeof(egne_bf16) + 4 * 16 * 64 * sizeof(double);
  if (d.stats_ws && d.mask_sums) return egne::fail(EGNE_ERR_ARG, "conv3x3_bf16: stats_ws and mask_sums share the waves' LDS scratch: one of them per launch");
  if (d.stats_ws && d.stats_nchunk != tiles_x * tiles_y * 4)
    return egne::fail(EGNE_ERR_ARG, "conv3x3_bf16: stats_nchunk %d, the launch has %d chunks per frame (tiles of 32 x 8 pixels x 4 waves)", d.stats_nchunk, tiles_x * tiles_y * 4);
  static_assert(lds <= 163840, "LDS budget");
  static bool once = hipFuncSetAttribute((const void*)conv3x3_bf16_kernel<KCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
  if (!once) return egne::fail(EGNE_ERR_LAUNCH, "conv3x3_bf16: cannot raise the dynamic LDS limit to %zu", lds);
  // 256 workgroups = 8 XCDs x 32; the nrun blocks of a worker sit on one XCD: 32 / nrun workers per XCD
  hipLaunchKernelGGL((conv3x3_bf16_kernel<KCH>), dim3(256), dim3(512), lds, st, d, wf, tiles_x, tiles_y, ntiles, ncb, nrun);
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
  EGNE_REQUIRE((d.post_scale == nullptr) == (d.post_shift == nullptr) &&
               (!d.post_scale || (((uintptr_t)d.post_scale & 15) == 0 && ((uintptr_t)d.post_shift & 15) == 0)), "conv3x3_bf16: post affine");
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
  switch (d.Ktot) {
    case 32: return launch_b3<1>(d, wf, st);
    case 64: return launch_b3<2>(d, wf, st);
    case 96: return launch_b3<3>(d, wf, st);
    case 128: return launch_b3<4>(d, wf, st);
    default: return launch_b3<0>(d, wf, st);
  }
}
