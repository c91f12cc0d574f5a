// 3x3 "same" convolution on the split-f16 MFMA path with fixed wave roles and RESIDENT WEIGHTS (fp32 tensors, three
// v_mfma_f32_32x32x16_f16 per product, fp32 accumulate; numerics: conv_f16x3.hip): one input slice of <= 64 channels, any number of
// 32-channel output blocks -- vgg16_c.py:66-69 (conv1_2, conv2_1), bdcn_new.py:50 (stage-1 MSBlock convs), models/RITnet_v2.py:57
// (down-block conv1 behind its InstanceNorm).
//
// Why another form of conv3x3_rs_f16.hip: there the consumer waves pull their weight fragments from L2 through a register ring and
// ALSO store the results, and on this hardware loads and stores retire through ONE in-order counter (vmcnt): every wait for a
// weight fragment also waits for the acknowledgement of all older stores (~3 k cycles under load).  Measured on 64 -> 64 at
// 240x320x64 (s_memtime stamps, scratch/rs_dbg.py): 11.7 k cycles per tile with both, 8.5 k with loads only, 8.6 k with stores only,
// 7.8 k with neither, 6.9 k of MFMA issue.  Here a workgroup owns ONE 32-channel output block and keeps all of its weights
// (<= 64 x 9 x 32 as hi | lo = 72 KB) in LDS for the whole launch, so the consumers issue no vector loads at all: their stores are
// never waited for.  The price: the input is staged once per output block (the blocks of one tile run on the same XCD at the same
// time, so the second read comes from L2) -- the producers have the slack for it.
//
// LDS: weights [chunk][tap][k16][hi | lo][64 lanes][8 halfs] (36 KB per 32 input channels), two halo images of ONE 32-channel chunk
// of a 32 x 8 tile, [hi | lo][340 pixels][32 halfs] without padding: the 16-byte chunk c of pixel q sits at c ^ ((q >> 1) & 3)
// (conflict-free for the 16-pixel x 4-chunk ds_read_b128 pattern of the 16x16x32 MFMA at every alignment, and for ds_write_b64).  A tile is KCH jobs (one per chunk), accumulators persist.
//   producers (waves 0-3)  halo gather (two jobs of loads in flight), optional fused InstanceNorm affine + activation, fp32 ->
//             hi / lo with plain VALU (split_f16.h), image (job + 1) & 1;
//   consumers (waves 4-7)  two rows x 32 channels each on v_mfma_f32_16x16x32_f16 (eight independent accumulators): 9 taps x 24
//             MFMAs per job, operands of tap t + 1 requested before the MFMAs of tap t; transposed product, so a lane ends with 16 channels of one pixel: 8 stores of 16 bytes per tile,
//             issued between the MFMAs of the next job; optional second output = 2x2 ceil-mode max pooling (pool1 of conv1_2).
// One s_barrier per job.
#include "common.h"
#include "split_f16.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {
__device__ int g_wdbg = 0;
__device__ unsigned long long g_wstamps[256 * 8 * 4];

constexpr int TW = 32, TH = 8, HWd = TW + 2, HHd = TH + 2, NPX = HHd * HWd;       // 340 halo pixels
constexpr int IMGH = 2 * NPX * 32;                       // halfs per image: [hi | lo][NPX][32]
constexpr int WCH = 9 * 2 * 2 * 512;                     // halfs of weights per 32-channel chunk
constexpr int NI = (NPX * 8 + 255) / 256;                // 16-byte items per producer lane and job
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// KCH: 32-channel chunks of the input slice (1 or 2: all weights resident) or 0: any number of chunks (p.Ktot / 32), the weights of
// a job's chunk (36 KB) STREAMED into one of two LDS weight buffers by the producers one job ahead -- the wider layers
// (conv2_2, the MSBlock convs of stages 2-5, the dense blocks at 60x80) on the same consumer loop.
// ncb = output blocks of 32 channels; tiles_x / tiles_y / ntiles as usual.
// NP: products per multiply (egne_conv_desc.f16_products): 3 = hi hi + hi lo + lo hi, 1 = hi hi only (plain f16 operands: the lo
// halves are neither derived, stored nor read; the frozen edge network next to a bf16-storage training plan)
// F16IN: the input slice is held as f16 (egne_seg.presplit = 2: plain halves of x * a_scale in channel order, written by a producer with
// egne_conv_desc.out_split = 2): a producer item is a 16-byte copy of eight channels, no conversion (NP = 1 only)
template <int KCH, int NP = 3, bool F16IN = false>
__global__ __launch_bounds__(512)
void conv3x3_rw_kernel(const egne_conv_desc p, const _Float16* __restrict__ fhi, const _Float16* __restrict__ flo, float a_scale,
                       float out_scale, int tiles_x, int tiles_y, int ntiles, int ncb, int nrun) {
  egne::dyn_scales(p.dyn_scale, a_scale, out_scale);
  extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
  _Float16* const lw = ldsh + 2 * IMGH;                  // weights behind the two images

  const int dbg = g_wdbg;
  unsigned long long t_work = 0, t_wait = 0, t_last = 0;
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

  // workgroup -> (output block, worker): blocks b and b + 8 share an XCD (round-robin dispatch: a speed assumption only); the ncb
  // workgroups of one worker walk the SAME tiles on the same XCD, so the halo of a tile is read from HBM once
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  // ncb: output blocks in the weight pack, nrun <= ncb: blocks that hold stored channels (the others are never computed)
  const int wpx = ((int)gridDim.x >> 3) / nrun;          // workers per XCD
  if (q >= wpx * nrun) return;                           // 32 is not a multiple of nrun: the spare workgroups of each XCD stay idle
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

  // the block's weights: fragments (tap, k16, nt = cb) of the 32x32x16 pack [tap][Ktot/16][CoutP/32][lane][8]
  const int KT16 = nk * 2;
  if constexpr (!STREAM) {
    for (int it = tid; it < KCH * 9 * 2 * 2 * 64; it += 512) {        // 16-byte items, LDS order [chunk][tap][ks][hl][lane]
      const int l = it & 63, hl = (it >> 6) & 1, ks = (it >> 7) & 1, r = it >> 8, tap = r % 9, ch = r / 9;
      const long long src = (((long long)tap * KT16 + ch * 2 + ks) * ncb + cb) * 512 + l * 8;
      *(u32x4*)&lw[(long long)it * 8] = *(const u32x4*)((hl ? flo : fhi) + src);
    }
  }
  __syncthreads();

  if (wave < 4) {
    // =================================================================== producers: halo chunk -> hi / lo image
    // 16-byte piece of the pixel's 32-channel chunk (4 fp32 or 8 f16 channels), pixel group (32 or 64 per round of 256 lanes)
    constexpr int NPC = F16IN ? 4 : 8, PGS = 256 / NPC, CPP = 32 / NPC, ESZ = F16IN ? 2 : 4, NQ = (NPX * NPC + 255) / 256;
    const int piece = tid & (NPC - 1), pg = tid / NPC;
    const float slope_in = sg.act_in == EGNE_ACT_RELU ? 0.f : (sg.act_in == EGNE_ACT_LEAKY ? 0.01f : 1.f);
    // LDS slot of item I: pixel pg + PGS I; (pixel >> 1) & 3 = (pg >> 1) & 3 for every I: lane constant + 64 B * PGS * I
    const int lofs = F16IN ? pg * 32 + ((piece ^ ((pg >> 1) & 3)) << 3) : pg * 32 + (((piece >> 1) ^ ((pg >> 1) & 3)) << 3) + ((piece & 1) << 2);
    u32x4 st[2][NQ];
    // byte offset of item I relative to the tile's first pixel (tile- and chunk-invariant): with it an item's address is one add
    // for tiles whose halo columns lie inside the image (rows outside it fall outside the per-frame resource and read zeros)
    int rel[NQ];
#pragma unroll
    for (int I = 0; I < NQ; ++I) {
      const int px = pg + PGS * I, hy = px / HWd, hx = px - hy * HWd;
      rel[I] = px < NPX ? (((hy - 1) * W + hx - 1) * (int)sg.pix_stride + sg.ch_off + piece * CPP) * ESZ : (int)OOB;
    }
    struct Job { Tile t; int ch; };
    auto job_at = [&](int j) { Job r; r.t = decode(tile_at(j / nk)); r.ch = j % nk; return r; };
    auto issue1 = [&](const Job& jb, bool on, auto bc, auto ic) {
      constexpr int BUF = decltype(bc)::value, I = decltype(ic)::value;
      const Tile& tl = jb.t;
      const __amdgpu_buffer_rsrc_t r = make_rsrc((const char*)sg.ptr + (long long)tl.b * H * W * sg.pix_stride * ESZ, (unsigned)H * W * (unsigned)sg.pix_stride * (unsigned)ESZ);
      const int c0 = jb.ch * 32 + piece * CPP;          // channels past the slice (padding up to 32 * KCH) read zeros
      int off;
      if (tl.x0 >= 1 && tl.x0 + TW + 1 <= W) {          // wave-uniform: interior columns
        const int sbase = ((tl.y0 * W + tl.x0) * (int)sg.pix_stride + jb.ch * 32) * ESZ;
        off = (on && c0 < sg.Cp && rel[I] != (int)OOB) ? rel[I] + sbase : (int)OOB;
      } else {
        int pq = pg;
        asm volatile("" : "+v"(pq));                    // opaque: no hoisting of the per-item coordinates out of the job loop
        const int px = pq + PGS * I;
        const int hy = px / HWd, hx = px - hy * HWd;
        const int y = tl.y0 - 1 + hy, x = tl.x0 - 1 + hx;
        const bool ok = on && c0 < sg.Cp && px < NPX && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        off = ok ? ((y * W + x) * (int)sg.pix_stride + sg.ch_off + c0) * ESZ : (int)OOB;
      }
      st[BUF][I] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    };
    // fused affine: the coefficients of a job are requested one step EARLIER than its conversion, ahead of that step's halo loads
    f32x4 asc[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}}, ash[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto load_aff = [&](const Job& jb, auto bc) {
      constexpr int BUF = decltype(bc)::value;
      if (sg.scale && nmine > 0) {       // (a workgroup without tiles decodes a frame past the batch: no table row to read)
        const int c0 = jb.ch * 32 + piece * 4;
        const float* zs = c0 < sg.Cp ? sg.scale + (long long)jb.t.b * sg.Cp + c0 : egne_zero_page;
        const float* zh = c0 < sg.Cp ? sg.shift + (long long)jb.t.b * sg.Cp + c0 : egne_zero_page;
        asc[BUF] = *(const f32x4*)zs;
        ash[BUF] = *(const f32x4*)zh;
      }
    };
    auto convert1 = [&](const Job& jb, _Float16* img, auto bc, auto ic) {
      constexpr int BUF = decltype(bc)::value, I = decltype(ic)::value;
      const int px = pg + PGS * I;
      if constexpr (F16IN) {
        if (I < NQ - 1 || px < NPX) *(u32x4*)&img[lofs + 32 * PGS * I] = st[BUF][I];     // already f16(x * a_scale): a copy
        return;
      }
      if (I < NQ - 1 || px < NPX) {
        f32x4 v = __builtin_bit_cast(f32x4, st[BUF][I]);
        if (sg.scale) {      // fused InstanceNorm affine (+ activation) of the consumer; zero padding applied after it
          const int hy = px / HWd, hx = px - hy * HWd;
          const int y = jb.t.y0 - 1 + hy, x = jb.t.x0 - 1 + hx;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = v[e] * asc[BUF][e] + ash[BUF][e];
            v[e] = fmaxf(t, t * slope_in);
          }
          if (!((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)) v = (f32x4)(0.f);
        }
        const int o = lofs + 32 * PGS * I;
        if constexpr (NP == 1) {
          float t0, t1, t2, t3;                   // (one-lane-value multiplies, as split_f16.h: no packed f32 next to the MFMAs)
          asm("v_mul_f32_e32 %0, %1, %2" : "=v"(t0) : "s"(a_scale), "v"(v[0]));
          asm("v_mul_f32_e32 %0, %1, %2" : "=v"(t1) : "s"(a_scale), "v"(v[1]));
          asm("v_mul_f32_e32 %0, %1, %2" : "=v"(t2) : "s"(a_scale), "v"(v[2]));
          asm("v_mul_f32_e32 %0, %1, %2" : "=v"(t3) : "s"(a_scale), "v"(v[3]));
          const egne::sp_f32x2 u0 = {t0, t1}, u1 = {t2, t3};
          const h2 h0 = __builtin_convertvector(u0, h2), h1 = __builtin_convertvector(u1, h2);
          const h4 hi = {h0[0], h0[1], h1[0], h1[1]};
          *(h4*)&img[o] = hi;
        } else {
        h2 h0, h1, l0, l1;                        // x * a_scale = hi + lo, plain (unpacked) VALU: split_f16.h
        egne::split2(v[0], v[1], a_scale, h0, l0);
        egne::split2(v[2], v[3], a_scale, h1, l1);
        const h4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
        *(h4*)&img[o] = hi;
        *(h4*)&img[NPX * 32 + o] = lo;
        }
      }
    };
    // STREAM: the 2304 16-byte pieces of a chunk's weights (LDS order [tap][ks][hl][lane]), nine per producer lane, requested one
    // job ahead at the START of a step (in front of that step's halo loads in the in-order queue) and written at the start of the next
    u32x4 wreg[STREAM ? 9 : 1];
    const unsigned wbytes = 9u * (unsigned)KT16 * (unsigned)ncb * 1024u;
    const __amdgpu_buffer_rsrc_t rwh = make_rsrc(fhi, wbytes), rwl = make_rsrc(flo, wbytes);
    auto w_issue = [&](int ch, bool on) {
#pragma unroll
      for (int i = 0; i < (STREAM ? 9 : 0); ++i) {
        const int it = tid + 256 * i, l = it & 63, hl = (it >> 6) & 1, ks = (it >> 7) & 1, tap = it >> 8;
        const int off = on ? ((((tap * KT16 + ch * 2 + ks) * ncb + cb) * 512 + l * 8) * 2) : (int)OOB;
        wreg[i] = hl ? __builtin_amdgcn_raw_buffer_load_b128(rwl, off, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rwh, off, 0, 0);
      }
    };
    auto w_store = [&](int parity) {
#pragma unroll
      for (int i = 0; i < (STREAM ? 9 : 0); ++i) *(u32x4*)&lw[parity * WCH + (tid + 256 * i) * 8] = wreg[i];
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
      _Float16* img = ldsh + ((s + 1) & 1) * IMGH;
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (([&] {
          if (c_on) convert1(jc, img, bc, std::integral_constant<int, Is>{});
          issue1(ji, i_on, bc, std::integral_constant<int, Is>{});
        }()), ...);
      }(std::make_integer_sequence<int, NQ>{});
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
      }(std::make_integer_sequence<int, NQ>{});
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (([&] {
          if (nmine > 0) convert1(j0, ldsh, B0{}, std::integral_constant<int, Is>{});
          issue1(j2, nmine > 2, B0{}, std::integral_constant<int, Is>{});
        }()), ...);
      }(std::make_integer_sequence<int, NQ>{});
    }
    lds_barrier();
    t_last = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < nloop; s += 2) {
      step(s, B1{});
      stamp(t_work); lds_barrier(); stamp(t_wait);
      step(s + 1, B0{});
      stamp(t_work); lds_barrier(); stamp(t_wait);
    }
  } else {
    // =================================================================== consumers: 9 taps per job from LDS only
    // v_mfma_f32_16x16x32_f16: one instruction per (16 channels, 16 pixels, 32 input channels of a tap); a wave's two rows x 32
    // pixels x 32 channels are EIGHT independent accumulators -- with the 32x32x16 form the same work is two dependent chains,
    // which issued at ~42 cycles per MFMA instead of 32 (and the chip holds a higher clock on this shape).
    const int cw = wave - 4, row0 = cw * 2;
    const int l15 = lane & 15, kg = lane >> 4;
    // out_split = 2: the output (and the pooled second output) is stored as f16 halves of v * out_split_scale, channel order kept -- the
    // input format of this kernel's F16IN form (the frozen edge network's stage-1 tensors next to a bf16-storage training plan)
    const bool o16 = NP == 1 && p.out_split == 2;          // (plain-f16 plans only: nothing of it in the three-product kernels)
    const int oesz = o16 ? 2 : 4;
    const unsigned frame_out = (unsigned)H * W * (unsigned)p.out_pix_stride * (unsigned)oesz;
    const unsigned frame_res = (unsigned)H * W * (unsigned)p.res_pix_stride * 4u;
    const float slope_out = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
    const bool full_epi = p.post_scale != nullptr || p.residual != nullptr;
    // transposed product (weights as the A operand): the lane holds channels n = 32 cb + 16 nh + 4 kg + r of pixel 16 ph + l15
    f32x4 b4[2];
    bool jok[2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const int n = cb * 32 + nh * 16 + 4 * kg;
      b4[nh] = p.bias ? *(const f32x4*)(p.bias + n) : (f32x4)(0.f);
      jok[nh] = n < p.Cout_store;                        // Cout_store is a multiple of 4
    }
    // operand addresses (independent of the tile).  Activations: pixel q = (row + ky) * 34 + 16 ph + l15 + kx, 8-channel group kg at
    // chunk kg ^ ((q >> 1) & 3); weights: the 32x32x16 fragments hold k-group kg of output channel m at fragment k16 = kg >> 1, lane
    // position (kg & 1) * 32 + m
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
    const int wl = (kg >> 1) * 1024 + ((kg & 1) * 32 + l15) * 8;       // + tap * 2048 + hl * 512 + nh * 128 (halfs)
    f32x4 acc[2][2][2], prev[2][2][2];
#pragma unroll
    for (int a = 0; a < 8; ++a) { (&acc[0][0][0])[a] = (f32x4)(0.f); (&prev[0][0][0])[a] = (f32x4)(0.f); }
    __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out, 0u), rres = make_rsrc(nullptr, 0u);
    int tvo[2][2], tvr[2][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) { (&tvo[0][0])[a] = (int)OOB; (&tvr[0][0])[a] = (int)OOB; }
    // The values are finished (scale, bias, activation [, post affine, residual]) IN PLACE at hand-over; the deferred part is the bare
    // store.  Computing them next to the store would reuse the store's data registers group after group, and overwriting the source
    // of a store in flight costs a wait for its completion (vmcnt): eight write round trips per tile.
    bool ovf_bad = false;                                // a non-finite value was stored (egne_conv_desc.ovf_flag)
    auto finish_group = [&](auto gc) {
      constexpr int Gi = decltype(gc)::value, nh = Gi & 1, ph = (Gi >> 1) & 1, tm = Gi >> 2;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = prev[tm][ph][nh][e] * out_scale + b4[nh][e];
        v[e] = fmaxf(t, t * slope_out);
      }
      if (full_epi) {                                    // rare in these layers: post affine / residual straight from memory
        const int n = cb * 32 + nh * 16 + 4 * kg;
        if (p.post_scale) {
          const f32x4 ps = *(const f32x4*)(p.post_scale + n), pt = *(const f32x4*)(p.post_shift + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] * ps[e] + pt[e];
        }
        if (p.residual) {
          const f32x4 rv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, jok[nh] ? tvr[tm][ph] : (int)OOB, nh * 64, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += rv[e];
        }
      }
      if constexpr (nh == 0) ovf_bad |= egne_nonfinite(v[0]);      // lane = pixel: one channel per pixel (common.h)
      if (o16) {
        const egne::sp_f32x2 u0 = {v[0] * p.out_split_scale, v[1] * p.out_split_scale}, u1 = {v[2] * p.out_split_scale, v[3] * p.out_split_scale};
        const h2 h0 = __builtin_convertvector(u0, h2), h1 = __builtin_convertvector(u1, h2);
        if constexpr (nh == 0) ovf_bad |= egne_nonfinite((float)h0[0]);      // (a value beyond the f16 range under the calibrated scale)
        v[0] = __builtin_bit_cast(float, h0);
        v[1] = __builtin_bit_cast(float, h1);
      } else if (p.out_split) {
        // split-pair storage (egne_conv_desc.out_split): the consumer's hi / lo f16 halves of v * out_split_scale, written here ONCE
        // instead of being derived by every consumer workgroup that stages the element (the dilated group stages it 13.5 times)
        h2 h0, h1, l0, l1;
        egne::split2(v[0], v[1], p.out_split_scale, h0, l0);
        egne::split2(v[2], v[3], p.out_split_scale, h1, l1);
        const u32x4 pk = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1), __builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
        v = __builtin_bit_cast(f32x4, pk);
      }
      prev[tm][ph][nh] = v;
    };
    auto store_group = [&](auto gc) {
      constexpr int Gi = decltype(gc)::value, nh = Gi & 1, ph = (Gi >> 1) & 1, tm = Gi >> 2;
      if (o16) {
        const u32x4 pk = __builtin_bit_cast(u32x4, prev[tm][ph][nh]);
        const u32x2 two = {pk[0], pk[1]};
        __builtin_amdgcn_raw_buffer_store_b64(two, rout, jok[nh] ? tvo[tm][ph] : (int)OOB, nh * 32, 0);
      } else if (p.out_split) {
        // split-pair storage: per pixel and 32-channel block [hi x 32 | lo x 32] halves, the lane's channels {4 kg ..} and {16 + 4 kg ..}
        // side by side at positions 8 kg .. 8 kg + 7 of either plane (the consumer's weights are packed in that channel order): one
        // 16-byte store per plane, 64 contiguous bytes per pixel and store instruction as with plain fp32
        if constexpr (nh == 1) {
          const u32x4 p0 = __builtin_bit_cast(u32x4, prev[tm][ph][0]), p1 = __builtin_bit_cast(u32x4, prev[tm][ph][1]);
          const u32x4 hi = {p0[0], p0[1], p1[0], p1[1]}, lo = {p0[2], p0[3], p1[2], p1[3]};
          __builtin_amdgcn_raw_buffer_store_b128(hi, rout, tvo[tm][ph], 0, 0);
          // (a plain-f16 plan never reads the lo plane: its one consumer takes the hi halves as operands AND as the residual)
          if constexpr (NP == 3) __builtin_amdgcn_raw_buffer_store_b128(lo, rout, tvo[tm][ph], 64, 0);
        }
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, prev[tm][ph][nh]), rout, jok[nh] ? tvo[tm][ph] : (int)OOB, nh * 64, 0);
      }
    };
    bool have_prev = false;
    lds_barrier();
    t_last = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < nloop; ++s) {
      if (s < nmine) {
        const int ch = s % nk;
        const Tile tl = decode(tile_at(s / nk));
        const _Float16* Thi = ldsh + (s & 1) * IMGH;
        const _Float16* Tlo = Thi + NPX * 32;
        const _Float16* wb = lw + (STREAM ? (s & 1) : ch) * WCH + wl;
        if (ch == 0) {
#pragma unroll
          for (int a = 0; a < 8; ++a) (&acc[0][0][0])[a] = (f32x4)(0.f);
        }
        // operands of tap t + 1 are requested before the MFMAs of tap t (two register sets)
        h8 wh[2][2], wo[2][2], ah[2][2][2], al[2][2][2];
        auto fetch = [&](auto tc) {
          constexpr int T = decltype(tc)::value, Bq = T & 1;
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) {
            wh[Bq][nh] = *(const h8*)&wb[T * 2048 + nh * 128];
            if constexpr (NP == 3) wo[Bq][nh] = *(const h8*)&wb[T * 2048 + 512 + nh * 128];
          }
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
              ah[Bq][tm][ph] = *(const h8*)&Thi[aofs[T][tm][ph]];
              if constexpr (NP == 3) al[Bq][tm][ph] = *(const h8*)&Tlo[aofs[T][tm][ph]];
            }
        };
        fetch(std::integral_constant<int, 0>{});
        [&]<int... Ts>(std::integer_sequence<int, Ts...>) {
          (([&] {
            constexpr int t = Ts, Bq = t & 1;
            if constexpr (t + 1 < 9) fetch(std::integral_constant<int, t + 1>{});
            // the previous tile's results leave between the MFMAs of this tile's first job (8 stores of 16 bytes)
            if constexpr (t < 8) { if (ch == 0 && have_prev) store_group(std::integral_constant<int, t>{}); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
              for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                  f32x4& c = acc[tm][ph][nh];
                  if constexpr (NP == 3) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[Bq][nh], al[Bq][tm][ph], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wo[Bq][nh], ah[Bq][tm][ph], c, 0, 0, 0);
                  }
                  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[Bq][nh], ah[Bq][tm][ph], c, 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
          }()), ...);
        }(std::make_integer_sequence<int, 9>{});
        if (ch == nk - 1) {                                // tile complete: hand it to the deferred stores
          const int y = tl.y0 + row0;
          if (p.pool_out) {      // second output: 2x2 / stride 2 / ceil-mode max pooling (act(max) = max(act): monotonic activation)
            const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
            const __amdgpu_buffer_rsrc_t rpo = make_rsrc((char*)p.pool_out + (long long)tl.b * Hp * Wp * p.pool_pix_stride * oesz,
                                                         (unsigned)Hp * Wp * (unsigned)p.pool_pix_stride * (unsigned)oesz);
            const bool y1 = y + 1 < H;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
              const int x = tl.x0 + ph * 16 + l15;
              const bool x1 = x + 1 < W;
              const int poff = (!(l15 & 1) && x < W && y < H) ? (((y >> 1) * Wp + (x >> 1)) * (int)p.pool_pix_stride + p.pool_ch_off + cb * 32 + 4 * kg) * oesz : (int)OOB;
#pragma unroll
              for (int nh = 0; nh < 2; ++nh) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float m = acc[0][ph][nh][e];
                  if (y1) m = fmaxf(m, acc[1][ph][nh][e]);
                  const float qn = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xf, 0xf, false));   // lane ^ 1
                  if (x1) m = fmaxf(m, qn);
                  const float t = m * out_scale + b4[nh][e];
                  v[e] = fmaxf(t, t * slope_out);
                }
                if (o16) {
                  const egne::sp_f32x2 u0 = {v[0] * p.out_split_scale, v[1] * p.out_split_scale}, u1 = {v[2] * p.out_split_scale, v[3] * p.out_split_scale};
                  const h2 h0 = __builtin_convertvector(u0, h2), h1 = __builtin_convertvector(u1, h2);
                  const u32x2 two = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
                  __builtin_amdgcn_raw_buffer_store_b64(two, rpo, jok[nh] ? poff : (int)OOB, nh * 32, 0);
                } else {
                  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rpo, jok[nh] ? poff : (int)OOB, nh * 64, 0);
                }
              }
            }
          }
#pragma unroll
          for (int a = 0; a < 8; ++a) (&prev[0][0][0])[a] = (&acc[0][0][0])[a];
          rout = make_rsrc((char*)p.out + (long long)tl.b * H * W * p.out_pix_stride * oesz, frame_out);
          rres = make_rsrc(p.residual ? p.residual + (long long)tl.b * H * W * p.res_pix_stride : nullptr, p.residual ? frame_res : 0u);
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
              const int yy = y + tm, x = tl.x0 + ph * 16 + l15;
              const bool okp = yy < H && x < W;
              tvo[tm][ph] = okp ? ((yy * W + x) * (int)p.out_pix_stride + p.out_ch_off + cb * 32 + 4 * kg) * oesz : (int)OOB;
              tvr[tm][ph] = okp ? ((yy * W + x) * (int)p.res_pix_stride + p.res_ch_off + cb * 32 + 4 * kg) * 4 : (int)OOB;
            }
          [&]<int... Gs>(std::integer_sequence<int, Gs...>) { (finish_group(std::integral_constant<int, Gs>{}), ...); }(std::make_integer_sequence<int, 8>{});
          have_prev = true;
        }
      }
      stamp(t_work); lds_barrier(); stamp(t_wait);
    }
    if (have_prev)
      [&]<int... Gs>(std::integer_sequence<int, Gs...>) { (store_group(std::integral_constant<int, Gs>{}), ...); }(std::make_integer_sequence<int, 8>{});
    egne_ovf_commit(ovf_bad, p.ovf_flag);
  }
  if ((dbg & 64) && lane == 0) {
    unsigned long long* o = g_wstamps + ((long long)blockIdx.x * 8 + wave) * 4;
    o[0] = t_work; o[1] = t_wait; o[2] = nmine; o[3] = 0;
  }
}

template <int KCH, int NP = 3, bool F16IN = false>
int launch_rw(const egne_conv_desc& d, const _Float16* fhi, const _Float16* flo, float a_scale, float os, hipStream_t st) {
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y * d.B, ncb = d.CoutP / 32, nrun = (d.Cout_store + 31) / 32;
  constexpr size_t lds = ((size_t)2 * IMGH + (size_t)(KCH == 0 ? 2 : KCH) * WCH) * sizeof(_Float16);
  static_assert(lds <= 163840, "LDS budget");
  static bool once = hipFuncSetAttribute((const void*)conv3x3_rw_kernel<KCH, NP, F16IN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
  if (!once) return egne::fail(EGNE_ERR_LAUNCH, "conv3x3_rw: cannot raise the dynamic LDS limit to %zu", lds);
  // 256 workgroups = 8 XCDs x 32; the ncb blocks of a worker sit on one XCD: 32 / ncb workers per XCD
  hipLaunchKernelGGL((conv3x3_rw_kernel<KCH, NP, F16IN>), dim3(256), dim3(512), lds, st, d, fhi, flo, a_scale, os, tiles_x, tiles_y, ntiles, ncb, nrun);
  return egne::check_launch("egne_conv3x3_rw_f16_fwd");
}

}  // namespace

// Same descriptor and weight pack as egne_conv3x3_rs_f16_fwd (one input slice with optional fused affine, 3x3 / pad 1 / dilation 1,
// Ktot = slice width rounded up to 32 (resident weights up to 64, streamed per chunk above), weights from
// egne_pack_conv_weight_f16frag); CoutP = 32, 64, 128 or 256; Cout_store a
// multiple of 8; 16-byte aligned output / residual / pooled slices; no statistics (egne_conv3x3_rs_f16_fwd writes those); optional
// pooled second output as there.
extern "C" int egne_conv3x3_rw_f16_fwd(const egne_conv_desc* dp, const void* fhi, const void* flo, float a_scale, float w_scale,
                                       void* stream) {
  EGNE_REQUIRE(dp && fhi && flo, "conv3x3_rw: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad_mode == 0 && d.ngroups == 1 && d.nseg == 1 && d.pad_h == 1 &&
               d.pad_w == 1 && d.dil[0] == 1 && d.Ho == d.H && d.Wo == d.W && !d.stats_ws, "conv3x3_rw: geometry / options not supported");
  const egne_seg& g = d.seg[0];
  EGNE_REQUIRE(g.ptr && g.Cp % 8 == 0 && (g.Cp + 31) / 32 * 32 == d.Ktot && d.Ktot >= 32 && d.Ktot <= 1024 && g.ch_off % 4 == 0 &&
               g.pix_stride % 4 == 0 && ((uintptr_t)g.ptr & 15) == 0 && (g.scale == nullptr) == (g.shift == nullptr), "conv3x3_rw: input slice");
  EGNE_REQUIRE(d.CoutP % 32 == 0 && d.CoutP <= 256 && d.Cout_store >= 8 && d.Cout_store <= d.CoutP && d.Cout_store % 8 == 0 && d.out &&
               ((uintptr_t)d.out & 15) == 0 && d.out_pix_stride % 4 == 0 && d.out_ch_off % 4 == 0 &&
               d.out_ch_off + d.Cout_store <= d.out_pix_stride && (!d.bias || ((uintptr_t)d.bias & 15) == 0), "conv3x3_rw: output");
  EGNE_REQUIRE(!d.residual || (((uintptr_t)d.residual & 15) == 0 && d.res_pix_stride % 4 == 0 && d.res_ch_off % 4 == 0), "conv3x3_rw: residual alignment");
  EGNE_REQUIRE(!d.post_scale || (((uintptr_t)d.post_scale & 15) == 0 && ((uintptr_t)d.post_shift & 15) == 0), "conv3x3_rw: post affine alignment");
  EGNE_REQUIRE(!d.pool_out || (((uintptr_t)d.pool_out & 15) == 0 && d.pool_pix_stride % 4 == 0 && d.pool_ch_off % 4 == 0 && !d.post_scale &&
                                d.pool_ch_off + d.Cout_store <= d.pool_pix_stride &&
                                (long long)((d.H + 1) / 2) * ((d.W + 1) / 2) * d.pool_pix_stride * 4 < (1ll << 31)), "conv3x3_rw: pooled output");
  EGNE_REQUIRE(((uintptr_t)fhi & 15) == 0 && ((uintptr_t)flo & 15) == 0 && a_scale > 0.f && w_scale > 0.f, "conv3x3_rw: weights / scales");
  EGNE_REQUIRE(d.out_split != 1 || (d.out_split_scale > 0.f && d.Cout_store % 32 == 0 && d.out_ch_off % 32 == 0 && !d.pool_out && !d.post_scale && !d.residual),
               "conv3x3_rw: split-pair output needs whole 32-channel blocks, a positive scale and no pooled / post-affine / residual options");
  EGNE_REQUIRE(d.out_split == 0 || d.out_split == 1 || (d.out_split == 2 && d.f16_products == 1 && d.out_split_scale > 0.f && !d.post_scale && !d.residual),
               "conv3x3_rw: f16 output (out_split = 2) needs f16_products = 1, a positive scale and no post-affine / residual options");
  const bool in16 = g.presplit == 2;
  EGNE_REQUIRE(g.presplit == 0 || (in16 && d.f16_products == 1 && !g.scale && g.ch_off % 8 == 0 && g.pix_stride % 8 == 0 &&
                                   (long long)d.H * d.W * g.pix_stride * 2 < (1ll << 31)),
               "conv3x3_rw: an f16 input slice (presplit = 2) needs f16_products = 1, no fused affine and 16-byte aligned pixels");
  EGNE_REQUIRE((long long)d.H * d.W * g.pix_stride * 4 < (1ll << 31) && (long long)d.H * d.W * d.out_pix_stride * 4 < (1ll << 31) &&
               (!d.residual || (long long)d.H * d.W * d.res_pix_stride * 4 < (1ll << 31)), "conv3x3_rw: frame too large for 32-bit byte offsets");
  const float os = 1.0f / (a_scale * w_scale);
  hipStream_t st = (hipStream_t)stream;
  const _Float16 *h = (const _Float16*)fhi, *l = (const _Float16*)flo;
  if (d.f16_products == 1 && in16) {       // ... read from f16 storage
    if (d.Ktot == 32) return launch_rw<1, 1, true>(d, h, l, a_scale, os, st);
    if (d.Ktot == 64) return launch_rw<2, 1, true>(d, h, l, a_scale, os, st);
    return launch_rw<0, 1, true>(d, h, l, a_scale, os, st);
  }
  if (d.f16_products == 1) {       // plain f16 operands (egne_conv_desc.f16_products; of a split-pair OUTPUT only the hi plane is written: no plain-f16 consumer reads the other)
    if (d.Ktot == 32) return launch_rw<1, 1>(d, h, l, a_scale, os, st);
    if (d.Ktot == 64) return launch_rw<2, 1>(d, h, l, a_scale, os, st);
    return launch_rw<0, 1>(d, h, l, a_scale, os, st);
  }
  if (d.Ktot == 32) return launch_rw<1>(d, h, l, a_scale, os, st);
  if (d.Ktot == 64) return launch_rw<2>(d, h, l, a_scale, os, st);
  return launch_rw<0>(d, h, l, a_scale, os, st);
}

extern "C" int egne_rw_debug(int dbg, void* out_stamps) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wdbg), &dbg, sizeof(int)) != hipSuccess) return -2;
  if (out_stamps && hipMemcpyFromSymbol(out_stamps, HIP_SYMBOL(g_wstamps), sizeof(unsigned long long) * 256 * 8 * 4) != hipSuccess) return -2;
  return 0;
}
