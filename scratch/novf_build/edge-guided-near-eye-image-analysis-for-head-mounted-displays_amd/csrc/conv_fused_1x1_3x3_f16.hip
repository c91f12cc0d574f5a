// A 1x1 convolution over a concatenation of raw NHWC slices FUSED with the 3x3 "same" convolution that consumes it, on
// the split-f16 MFMA path (fp32 tensors, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate; numerics:
// conv_f16x3.hip).  models/RITnet_v2.py:59-62 (conv22(conv21(cat(x, x1))), conv32(conv31(cat(x, x1, x22)))) and :84-87
// (conv12(conv11(cat(up, skip))), conv22(conv21(cat(up, skip, x1)))): the 1x1 result is read exactly once, by that 3x3,
// and the pair is HBM-bound (ESF-Net block 0: 3 + 2 tensor passes for conv21/conv22).  Here the 1x1 output never leaves
// the CU.  A workgroup is 8 waves with FIXED ROLES and two LDS halo images:
//
//   producers (waves 0-3)  evaluate the 1x1 on the (TH+2) x 34 halo of tile i+1 the way the streaming 1x1 kernel
//             (conv1x1_f16.hip) walks an image: a lane's MFMA operand (8 channels of one pixel) comes straight from
//             HBM / L2 through a per-frame BUFFER resource (out-of-image halo pixels carry the offset 0x80000000 and
//             load zeros), 4 channel groups (8 KB per wave) per batch, the next batch in flight while the current one is
//             converted and multiplied; the product is computed transposed, so that a lane ends up with 4 consecutive
//             channels of its pixel: bias added, ZERO outside the image (the 3x3's zero padding pads the 1x1 OUTPUT),
//             scaled, split into hi / lo halves and written to image (i+1)&1 (80-B pixel pitch per 32 channels);
//   consumers (waves 4-7)  run the 9 taps of the 3x3 on tile i from image i&1 exactly as conv_halo_f16.hip does (weights
//             through a register ring from L2), all 32-channel chunks resident, then the epilogue.
//
// One s_barrier per tile.  The producers' vector-memory queue holds nothing but their own loads, so a wait for one batch
// never covers the latency of a later one, and the HBM latency of tile i+1 hides behind the matrix work of tile i.
// Tiles are dealt so that the workgroups of one XCD work on neighbouring tiles (halo overlap served by that XCD's L2).
#include "common.h"
#include "epilogue32.h"
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
__device__ int g_dbg = 0;
__device__ unsigned long long g_stamps[256 * 8 * 4];

constexpr int LDH = 40, TW = 32, HWd = TW + 2, MAXG = 48;
constexpr unsigned OOB = 0x80000000u;

struct GroupTab {
  int v[MAXG];       // per 16-channel group: (slice << 16) | (8-channel tail << 15) | group index inside the slice
  int off[MAXG];     // UNI: byte offset of the group's first channel inside a pixel of the common buffer
  int dn;            // UNI: groups flagged 0x4000 in v[] start dn samples further into that buffer (the edge pass' half of a 2B batch)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void split8(const u32x4 a, const u32x4 b, float s, h8& hi, h8& lo) {
  const f32x4 va = __builtin_bit_cast(f32x4, a), vb = __builtin_bit_cast(f32x4, b);
  const float x[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h2 h, l;
    egne::split2(x[2 * q], x[2 * q + 1], s, h, l);     // plain (unpacked) VALU: split_f16.h
    hi[2 * q] = h[0]; hi[2 * q + 1] = h[1];
    lo[2 * q] = l[0]; lo[2 * q + 1] = l[1];
  }
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// NCH: 32-channel chunks of the intermediate (1x1 output = 3x3 input); WN: 32-wide output tiles of the 3x3 (CoutP2 = 32*WN);
// TH: tile rows (8: two per consumer wave, 4: one per wave -- what two images of a 64-channel intermediate leave room for);
// NB: batches of 4 channel groups of the 1x1's K (ceil(groups / 4)).
// C4: the first convolution is not a 1x1 but a 3x3 / pad 1 on <= 4 input channels (utils.py:1047 convBlock: conv2(leaky(conv1(x))))
//     with its 9 taps folded into K as in conv3x3_c4_f16.hip: group s of a pixel = the 4-channel vectors of taps 4s + 2h and
//     4s + 2h + 1 (h = lane half), three groups, weights from egne_pack_conv3x3_c4_weight_f16; its activation is applied
//     before the result is split into the LDS image.
// UPADD: p1.residual names a LOW-resolution tensor [B][p1.Ho][p1.Wo] whose bilinear x2 upsampling (F.interpolate, scale 2,
//     align_corners False) is added to the 1x1 result: conv11(cat(up(x), skip)) = up(W_up x) + W_skip skip (RITnet_v2.py:84-86;
//     the 1x1 and the interpolation are both linear and the interpolation weights sum to one), so the up-sampled tensor is
//     never materialised.  The 6 x 18 low-resolution pixels a tile's halo needs are staged in LDS one tile ahead.
// p1: the 1x1 (slices, bias, CoutP = 32*NCH); p2: the 3x3 (bias, act, post affine, residual, output).
// w1hi / w1lo: fragments of egne_pack_conv1x1_weight_f16; f2hi / f2lo: fragments of egne_pack_conv_weight_f16frag.
// UNI: every slice of the 1x1's input is a channel range of ONE buffer (a dense block's x | x1 | x22: models/RITnet_v2.py:57-62):
//     one buffer resource per tile instead of one per channel group (whose 64-bit frame pointers, kept across the statically
//     unrolled item schedule, overflowed the scalar register file: ~350 v_readlane / v_writelane per tile and producer wave).
// C1V (with C4, one-channel planar input): the first convolution runs on the VECTOR ALU in exact fp32 -- 9 fused multiply-adds per
//     (halo pixel, channel) on a 12 x 36 input patch staged in LDS, weights as scalar operands (p1.w = float [32][12]: nine taps, the
//     bias, two zeros per channel) -- instead of gathering nine dwords per halo pixel from memory (96 tiny loads per tile), splitting
//     them and spending a K = 48 MFMA on nine taps: the producers were 12.4 k cycles per tile against the consumers' 6.3 k.
// GL: channel groups in the LAST batch when known at compile time (0: decided from G1 at run time, one branch per group).  With
//     it an item is straight-line code: all its weight fragments are requested from LDS up front and the three MFMAs of group u
//     are interleaved with the fp32 -> hi / lo split of group u + 1 (sched_group_barrier), instead of a ds_read latency and
//     three back-to-back MFMAs per group in a wave that issues in order.
template <int NCH, int WN, int TH, int NB, bool C4 = false, bool UPADD = false, bool UNI = false, int GL = 0, bool C1V = false>
__global__ __launch_bounds__(512)
void fused_1x1_3x3_kernel(const egne_conv_desc p1, const egne_conv_desc p2, const GroupTab gt, const _Float16* __restrict__ w1hi,
                          const _Float16* __restrict__ w1lo, int G1, const _Float16* __restrict__ f2hi,
                          const _Float16* __restrict__ f2lo, float a1, float os1, float a2, float os2, int tiles_x, int tiles_y,
                          int ntiles) {
  constexpr int GB = NCH == 1 ? 4 : 2;              // 16-channel groups per producer item (8 / 4 KB of loads per wave)
  constexpr bool WLDS = NCH == 1;                   // the 1x1's weights in LDS (all of them fit beside two 32-channel images only)
  constexpr int LG = WLDS ? MAXG : 7;               // 64-channel intermediate: the first LG channel groups' fragments are LDS resident (28 KB is what
                                                    // two 64-channel images leave), the rest comes from L2 per item.  Each producer wave pulling every
                                                    // fragment from L2 for every 32-pixel block was 256-384 KB per tile and CU on top of the consumers'
                                                    // 288 KB and the activations: the launch ran at the CU's L2 port (~30 B/clk), not at the MFMA pipe
  constexpr int HHd = TH + 2, NPX = HHd * HWd, NMT = (NPX + 31) / 32;
  constexpr int IMG = 2 * NCH * NPX * LDH;          // halfs per image: [hi | lo][NCH][NPX][LDH]
  extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
  float* lbias = (float*)(ldsh + 2 * IMG);          // the 1x1's bias: read at every job end through LDS, so that the read is
                                                    // not queued (vmcnt retires in order) behind the next item's prefetch

  const int dbg = g_dbg;
  unsigned long long t_work = 0, t_wait = 0, t_mma = 0, t_last = __builtin_amdgcn_s_memtime(), r_first = __builtin_amdgcn_s_memrealtime();
  auto stamp = [&](unsigned long long& accum) {
    if (dbg & 64) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      accum += t - t_last; t_last = t;
    }
  };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform for the compiler too: descriptors and loop counters derived from it stay scalar
  const int li = lane & 31, lh = lane >> 5;
  const int H = p2.H, W = p2.W;

  // tile sequence of this workgroup: blocks b and b + 8 share an XCD (round-robin dispatch, a speed assumption only), so
  // chunk c of `per` consecutive tiles goes to the blocks with b % 8 == c % 8
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
  while (tile_at(nmine) < ntiles) ++nmine;          // contiguous prefix: tile_at is increasing in i
  if (tid < 32 * NCH) lbias[tid] = p1.bias ? p1.bias[tid] : 0.f;
  // the 1x1's weight fragments live in LDS for the whole launch ([group][tile][hi | lo][lane][8]): read through the LDS
  // queue they never wait behind the producers' prefetched activations (vmcnt retires in order)
  _Float16* lw = (_Float16*)(lbias + 32 * NCH);
  const int GLDS = G1 < LG ? G1 : LG;
  for (int it = tid; it < GLDS * NCH * 2 * 64; it += 512) {        // 16-byte items
    const int l = it & 63, hl = (it >> 6) & 1, q = it >> 7;      // q = group * NCH + tile
    *(u32x4*)&lw[(long long)it * 8] = *(const u32x4*)((hl ? w1lo : w1hi) + ((long long)q * 64 + l) * 8);
  }
  // UPADD: two tiles of the low-resolution addend, [6 rows][18 columns][32 * NCH + 4] floats each.  The pixel pitch is 144 (272) bytes,
  // not 128 (256): the 32 lanes of a half wave read 16-byte vectors of 16-17 DIFFERENT patch pixels at the same channel offset, and
  // with a power-of-two pitch all of them start in the same four banks (measured: 45-47 % of the LDS cycles of these launches were
  // bank conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_ACTIVE); 36 k mod 64 walks all sixteen 4-bank slots
  constexpr int PP = 32 * NCH + 4;
  constexpr int PROWS = TH / 2 + 2, PCOLS = TW / 2 + 2, PTILE = PROWS * PCOLS * PP;
  float* const lp = (float*)(lw + GLDS * NCH * 2 * 512);
  __syncthreads();

  if constexpr (C1V) if (wave < 4) {
    // =================================================================== producers, one-channel first layer on the vector ALU
    static_assert(!C1V || (C4 && NCH == 1 && TH == 8 && !UPADD), "C1V: 1 -> 32 channels in front of a 32-channel 3x3, 8-row tiles");
    constexpr int PW = TW + 4, PH = TH + 4, PN = PW * PH;      // input patch of a tile: 12 x 36 pixels (halo of the halo)
    float* const lpatch = lp;                                  // two patches
    const egne_seg sg = p1.seg[0];
    const float sl = p1.act == EGNE_ACT_RELU ? 0.f : (p1.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
    const int hq = wave >> 1;                                  // wave-uniform: channels 16 hq .. 16 hq + 15 (scalar weight operands)
    unsigned preg[2];
    auto patch_issue = [&](const Tile& tl, bool on) {
      const __amdgpu_buffer_rsrc_t r = make_rsrc(sg.ptr + (long long)tl.b * H * W, (unsigned)H * W * 4u);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int idx = tid + 256 * k, py = idx / PW, px = idx - py * PW;
        const int y = tl.y0 - 2 + py, x = tl.x0 - 2 + px;
        const bool ok = on && idx < PN && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && !(dbg & 1);
        preg[k] = __builtin_amdgcn_raw_buffer_load_b32(r, ok ? (y * W + x) * 4 : (int)OOB, 0, 0);      // zero padding of the first convolution
      }
    };
    auto patch_store = [&](float* dst) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int idx = tid + 256 * k;
        if (idx < PN) dst[idx] = __builtin_bit_cast(float, preg[k]);
      }
    };
    // the weight table in LDS ([32][12] floats behind the patches): a uniform-address ds_read_b128 is a broadcast, and unlike a load
    // from memory it neither queues behind the patch prefetch nor costs a vector-memory instruction per weight
    float* const lwt = lpatch + 2 * PN;
    for (int e = tid; e < 32 * 12; e += 256) lwt[e] = p1.w[e];
    auto produce = [&](int i) {           // tile i into image i & 1 from patch i & 1; patch i + 1 stored, patch i + 2 requested
      const Tile tl = decode(tile_at(i));
      _Float16* Thi = ldsh + (i & 1) * IMG;
      _Float16* Tlo = Thi + NPX * LDH;
      const float* pt = lpatch + (i & 1) * PN;
      // a lane's three halo pixels (hp = 64 (wave & 1) + lane + 128 r) side by side, so that a channel's weights are read once for them
      float t[3][9], vs[3];
      int hpv[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int hp = (wave & 1) * 64 + lane + 128 * r;       // halo pixel of the 10 x 34 intermediate
        const int hpc = hp < NPX ? hp : 0;
        const int hy = hpc / HWd, hx = hpc - hy * HWd;
        const int y = tl.y0 - 1 + hy, x = tl.x0 - 1 + hx;
        vs[r] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? a2 : 0.f;    // zero OUTSIDE the image: the 3x3's padding
        hpv[r] = hp;
#pragma unroll
        for (int k = 0; k < 9; ++k) t[r][k] = pt[(hy + k / 3) * PW + hx + k % 3];
      }
      h8 hi[3][2], lo[3][2];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const f32x4* wc = (const f32x4*)(lwt + (hq * 16 + j) * 12);
        const f32x4 w0 = wc[0], w1 = wc[1], w2 = wc[2];
        const float w[10] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0], w2[1]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          float acc = w[9];
#pragma unroll
          for (int k = 0; k < 9; ++k) acc = __builtin_fmaf(w[k], t[r][k], acc);
          acc = fmaxf(acc, acc * sl) * vs[r];
          const _Float16 h = (_Float16)acc;
          hi[r][j >> 3][j & 7] = h;
          lo[r][j >> 3][j & 7] = (_Float16)(acc - (float)h);
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
        if (hpv[r] < NPX) {
          const int o = hpv[r] * LDH + hq * 16;
          *(h8*)&Thi[o] = hi[r][0]; *(h8*)&Thi[o + 8] = hi[r][1];
          *(h8*)&Tlo[o] = lo[r][0]; *(h8*)&Tlo[o + 8] = lo[r][1];
        }
      patch_store(lpatch + ((i + 1) & 1) * PN);
      const bool on2 = i + 2 < nmine;
      patch_issue(decode(tile_at(on2 ? i + 2 : i)), on2);
    };
    patch_issue(decode(tile_at(0)), nmine > 0);
    patch_store(lpatch);
    patch_issue(decode(tile_at(nmine > 1 ? 1 : 0)), nmine > 1);
    lds_barrier();                          // patch 0 visible to every producer wave
    if (nmine > 0) produce(0);
    lds_barrier();
    stamp(t_wait); t_work = 0; t_wait = 0;
    for (int i = 0; i < nmine; ++i) {
      if (i + 1 < nmine) produce(i + 1);
      stamp(t_work);
      lds_barrier();
      stamp(t_wait);
    }
  }
  if (!C1V && wave < 4) {
    // =================================================================== producers: 1x1 on the halo -> LDS image
    // Every producer wave runs a STATIC schedule of N = JOBS * NB items per tile (item = 4 channel groups of one 32-pixel
    // block; wave w owns blocks w, w + 4, ...; a block past the halo is all out-of-range lanes: no traffic, nothing
    // written).  Loads run DIST = 2 items ahead through a ring of 3 register buffers and wrap into the NEXT tile, so the
    // pipeline never drains at a tile boundary; with no data-dependent control flow inside, every wait is a counted vmcnt.
    // Without room for the weights in LDS (64-channel intermediate) they come from L2 into a register double buffer, one
    // item EARLY and in front of that step's activation prefetch, which costs one level of prefetch distance: DIST = 3.
    constexpr int JOBS = (NMT + 3) / 4, N = JOBS * NB, NBUF = WLDS ? 3 : 4, DIST = NBUF - 1;
    static_assert(N % NBUF == 0 && (WLDS || N % 2 == 0) && N >= DIST, "buffer index of an item must not depend on the tile");
    u32x4 xa[NBUF][GB], xb[NBUF][GB];
    const float* ptile = lp;                         // UPADD: the current tile's staged addend
    u32x4 wq[WLDS ? 1 : 2][GB][NCH][2];
    f32x16 acc[NCH];
    const unsigned w1bytes = (unsigned)G1 * NCH * 64u * 16u;
    const __amdgpu_buffer_rsrc_t rw1h = make_rsrc(w1hi, w1bytes), rw1l = make_rsrc(w1lo, w1bytes);
    auto load_w = [&](auto kc) {          // weight fragments of item K (they depend on its batch index only)
      constexpr int K = decltype(kc)::value, bi = K % NB;
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        const int gg = bi * GB + u;
        if (gg < LG) continue;                       // LDS resident
        const int wo = gg < G1 ? (gg * NCH * 64 + lane) * 16 : (int)OOB;
#pragma unroll
        for (int tn = 0; tn < NCH; ++tn) {
          wq[K & 1][u][tn][0] = __builtin_amdgcn_raw_buffer_load_b128(rw1h, wo, tn * 1024, 0);
          wq[K & 1][u][tn][1] = __builtin_amdgcn_raw_buffer_load_b128(rw1l, wo, tn * 1024, 0);
        }
      }
    };

    int pyy, pxx;                                   // image coordinates of the lane's halo pixel (set by pixel())
    auto pixel = [&](const Tile& tl, int job, int& hp, bool& valid, int& pix) {
      hp = (wave + 4 * job) * 32 + li;
      const int hy = hp / HWd, hx = hp - hy * HWd;
      const int y = tl.y0 - 1 + hy, x = tl.x0 - 1 + hx;
      valid = hp < NPX && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      pix = y * W + x;
      pyy = y; pxx = x;
    };
    // Loads are issued UNCONDITIONALLY (groups past the end and tiles past the last one carry the out-of-range offset
    // and return zeros without touching memory): a conditionally issued load makes the compiler's vmcnt bookkeeping
    // assume the shortest queue at every join and wait for everything.
    auto issue = [&](const Tile& tl, bool on, auto kc) {
      constexpr int K = decltype(kc)::value, BUF = K % NBUF, job = K / NB, bi = K % NB;
      int hp, pix; bool valid;
      pixel(tl, job, hp, valid, pix);
      valid = valid && on && !(dbg & 1);
      if constexpr (C4) {
        const egne_seg sg = p1.seg[0];
        const __amdgpu_buffer_rsrc_t r =
            make_rsrc(sg.ptr + (long long)tl.b * H * W * sg.pix_stride, (unsigned)H * W * (unsigned)sg.pix_stride * 4u);
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int gg = bi * GB + u;
          int offs[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int tap = 4 * gg + 2 * lh + t, ky = (tap * 11) >> 5, kx = tap - 3 * ky;
            const int yy = pyy + ky - 1, xx = pxx + kx - 1;
            const bool ok = valid && tap < 9 && gg < G1 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            offs[t] = ok ? ((yy * W + xx) * (int)sg.pix_stride + sg.ch_off) * 4 : (int)OOB;
          }
          if (sg.pix_stride == 1) {      // one-channel planar input ([B][H][W] = NCHW with C = 1): a dword per tap, channels 1-3 are zero
            xa[BUF][u] = u32x4{__builtin_amdgcn_raw_buffer_load_b32(r, offs[0], 0, 0), 0u, 0u, 0u};
            xb[BUF][u] = u32x4{__builtin_amdgcn_raw_buffer_load_b32(r, offs[1], 0, 0), 0u, 0u, 0u};
          } else {
            xa[BUF][u] = __builtin_amdgcn_raw_buffer_load_b128(r, offs[0], 0, 0);
            xb[BUF][u] = __builtin_amdgcn_raw_buffer_load_b128(r, offs[1], 0, 0);
          }
        }
        return;
      }
      if constexpr (UNI) {
        const egne_seg sg = p1.seg[0];
        const unsigned fbytes = (unsigned)H * W * (unsigned)sg.pix_stride * 4u;
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc(sg.ptr + (long long)tl.b * H * W * sg.pix_stride, fbytes);
        const __amdgpu_buffer_rsrc_t r1 = make_rsrc(sg.ptr + (long long)(tl.b + gt.dn) * H * W * sg.pix_stride, fbytes);
        const int voff = valid ? (pix * (int)sg.pix_stride + 4 * lh) * 4 : (int)OOB;
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int gg = bi * GB + u;
          const bool on_g = gg < G1;
          const int e = gt.v[gg < MAXG ? gg : 0];
          const int so = gt.off[gg < MAXG ? gg : 0];
          const __amdgpu_buffer_rsrc_t r = (e & 0x4000) ? r1 : r0;          // wave-uniform select
          xa[BUF][u] = __builtin_amdgcn_raw_buffer_load_b128(r, on_g ? voff : (int)OOB, so, 0);                                   // channels 16g + 4lh .. +3
          xb[BUF][u] = __builtin_amdgcn_raw_buffer_load_b128(r, (on_g && !(e & 0x8000)) ? voff + 32 : (int)OOB, so, 0);   // 16g + 8 + 4lh .. +3
        }
        return;
      }
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        const int gg = bi * GB + u;
        const int e = gt.v[gg < G1 ? gg : 0];
        const egne_seg sg = p1.seg[e >> 16];
        const __amdgpu_buffer_rsrc_t r =
            make_rsrc(sg.ptr + (long long)tl.b * H * W * sg.pix_stride, (unsigned)H * W * (unsigned)sg.pix_stride * 4u);
        const int voff = (valid && gg < G1) ? (pix * (int)sg.pix_stride + sg.ch_off + 4 * lh) * 4 : (int)OOB;
        const int lg = e & 0x3fff;
        xa[BUF][u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, lg * 64, 0);                              // channels 16g + 4lh .. +3
        xb[BUF][u] = __builtin_amdgcn_raw_buffer_load_b128(r, (e & 0x8000) ? (int)OOB : voff + 32, lg * 64, 0);   // 16g + 8 + 4lh .. +3
      }
    };
    auto compute = [&](const Tile& tl, const Tile& nx, bool nx_on, _Float16* img, auto kc) {
      constexpr int K = decltype(kc)::value, BUF = K % NBUF, job = K / NB, bi = K % NB;
      if (bi == 0) {
#pragma unroll
        for (int tn = 0; tn < NCH; ++tn) acc[tn] = (f32x16)(0.f);
      }
      if constexpr (!WLDS) load_w(std::integral_constant<int, (K + 1) % N>{});
      if constexpr (K + DIST < N) issue(tl, true, std::integral_constant<int, K + DIST>{});
      else issue(nx, nx_on, std::integral_constant<int, K + DIST - N>{});
      if constexpr (GL > 0) {
        constexpr int NG = bi == NB - 1 ? GL : GB;
        constexpr int NLDS = (bi * GB + NG <= LG) ? NG : (bi * GB >= LG ? 0 : LG - bi * GB);     // groups of this item read from LDS
        h8 bh[NG][NCH], bl[NG][NCH];
#pragma unroll
        for (int u = 0; u < NG; ++u)
#pragma unroll
          for (int tn = 0; tn < NCH; ++tn) {
            if (bi * GB + u < LG) {
              const _Float16* wp = lw + (((bi * GB + u) * NCH + tn) * 128 + lane) * 8;
              bh[u][tn] = *(const h8*)wp; bl[u][tn] = *(const h8*)(wp + 512);
            } else if constexpr (!WLDS) {
              bh[u][tn] = __builtin_bit_cast(h8, wq[K & 1][u][tn][0]); bl[u][tn] = __builtin_bit_cast(h8, wq[K & 1][u][tn][1]);
            }
          }
#pragma unroll
        for (int u = 0; u < NG; ++u) {
          h8 ah, al;
          split8(xa[BUF][u], xb[BUF][u], a1, ah, al);
#pragma unroll
          for (int tn = 0; tn < NCH; ++tn) {
            acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[u][tn], al, acc[tn], 0, 0, 0);
            acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[u][tn], ah, acc[tn], 0, 0, 0);
            acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[u][tn], ah, acc[tn], 0, 0, 0);
          }
        }
        // desired issue order: split of group 0, then each MFMA followed by a share of the next group's split
        if constexpr (NLDS > 0) __builtin_amdgcn_sched_group_barrier(0x100, NLDS * NCH * 2, 0);     // the LDS reads
        __builtin_amdgcn_sched_group_barrier(0x002, 30, 0);
#pragma unroll
        for (int m = 0; m < NG * NCH * 3; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, NCH == 1 ? 10 : 5, 0);
        }
      } else {
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        if (bi * GB + u < G1) {
          h8 ah, al;
          split8(xa[BUF][u], xb[BUF][u], a1, ah, al);
#pragma unroll
          for (int tn = 0; tn < NCH; ++tn) {
            h8 bh = {}, bl = {};
            if (bi * GB + u < LG) {
              const _Float16* wp = lw + (((bi * GB + u) * NCH + tn) * 128 + lane) * 8;
              bh = *(const h8*)wp; bl = *(const h8*)(wp + 512);
            } else if constexpr (!WLDS) {
              bh = __builtin_bit_cast(h8, wq[K & 1][u][tn][0]); bl = __builtin_bit_cast(h8, wq[K & 1][u][tn][1]);
            }
            acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc[tn], 0, 0, 0);
            acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc[tn], 0, 0, 0);
            acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[tn], 0, 0, 0);
          }
        }
      }
      }
      if (bi == NB - 1) {
        // transposed product: the lane holds channels n = 32*tn + 8*j + 4*lh + e (register 4*j + e) of halo pixel hp
        _Float16* Thi = img;
        _Float16* Tlo = img + NCH * NPX * LDH;
        int hp, pix; bool valid;
        pixel(tl, job, hp, valid, pix);
        if (hp < NPX) {
          const float vs = valid ? a2 : 0.f, vo = valid ? os1 * a2 : 0.f;
          // UPADD: ATen's area_pixel_compute_source_index(scale 0.5, align_corners false): s = max(0.5 (d + 0.5) - 0.5, 0)
          int o00 = 0, o01 = 0, o10 = 0, o11 = 0;
          float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
          if constexpr (UPADD) {
            const int ph = p1.Ho, pw = p1.Wo;
            float sy = 0.5f * (pyy + 0.5f) - 0.5f, sx = 0.5f * (pxx + 0.5f) - 0.5f;
            sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
            const int y0 = (int)sy, x0 = (int)sx, y1 = y0 + (y0 < ph - 1 ? 1 : 0), x1 = x0 + (x0 < pw - 1 ? 1 : 0);
            const float ly = sy - y0, lx = sx - x0;
            const int ry = tl.y0 / 2 - 1, rx = tl.x0 / 2 - 1;
            o00 = ((y0 - ry) * PCOLS + x0 - rx) * PP; o01 = ((y0 - ry) * PCOLS + x1 - rx) * PP;
            o10 = ((y1 - ry) * PCOLS + x0 - rx) * PP; o11 = ((y1 - ry) * PCOLS + x1 - rx) * PP;
            w00 = (1.f - ly) * (1.f - lx) * vs; w01 = (1.f - ly) * lx * vs; w10 = ly * (1.f - lx) * vs; w11 = ly * lx * vs;
            if (!valid) o00 = o01 = o10 = o11 = 0;
          }
#pragma unroll
          for (int tn = 0; tn < NCH; ++tn)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              f32x4 b4 = *(const f32x4*)(lbias + tn * 32 + 8 * j + 4 * lh) * vs;
              if constexpr (UPADD) {
                const int c = tn * 32 + 8 * j + 4 * lh;
                b4 += w00 * *(const f32x4*)(ptile + o00 + c) + w01 * *(const f32x4*)(ptile + o01 + c) +
                      w10 * *(const f32x4*)(ptile + o10 + c) + w11 * *(const f32x4*)(ptile + o11 + c);
              }
              f32x2 v0 = {acc[tn][4 * j] * vo + b4[0], acc[tn][4 * j + 1] * vo + b4[1]};
              f32x2 v1 = {acc[tn][4 * j + 2] * vo + b4[2], acc[tn][4 * j + 3] * vo + b4[3]};
              if constexpr (C4) {                    // activation of the first convolution (scaling by a2 > 0 commutes with it)
                const float sl = p1.act == EGNE_ACT_RELU ? 0.f : (p1.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
                v0[0] = fmaxf(v0[0], v0[0] * sl); v0[1] = fmaxf(v0[1], v0[1] * sl);
                v1[0] = fmaxf(v1[0], v1[0] * sl); v1[1] = fmaxf(v1[1], v1[1] * sl);
              }
              const h2 h0 = __builtin_convertvector(v0, h2), h1 = __builtin_convertvector(v1, h2);
              const h2 l0 = __builtin_convertvector(v0 - __builtin_convertvector(h0, f32x2), h2);
              const h2 l1 = __builtin_convertvector(v1 - __builtin_convertvector(h1, f32x2), h2);
              const h4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
              const int o = (tn * NPX + hp) * LDH + 8 * j + 4 * lh;
              *(h4*)&Thi[o] = hi;
              *(h4*)&Tlo[o] = lo;
            }
        }
      }
    };
    // UPADD staging: the low-resolution rows y0/2 - 1 .. y0/2 + TH/2 and columns x0/2 - 1 .. x0/2 + 16 of the addend (clamped to
    // the tensor: replicated border, as the interpolation clamps), 16 bytes per lane
    constexpr int NPI = UPADD ? (PROWS * PCOLS * 8 * NCH + 255) / 256 : 1;
    u32x4 pst[NPI];
    auto issue_p = [&](const Tile& tl, bool on) {
      const int ph = p1.Ho, pw = p1.Wo;
      const __amdgpu_buffer_rsrc_t r = make_rsrc(p1.residual, (unsigned)((long long)p1.B * ph * pw * p1.res_pix_stride * 4));
#pragma unroll
      for (int i = 0; i < NPI; ++i) {
        const int it = tid + 256 * i, px = it / (8 * NCH), pc = it - px * (8 * NCH);
        const int ry = px / PCOLS, rx = px - ry * PCOLS;
        const int gy = min(max(tl.y0 / 2 - 1 + ry, 0), ph - 1), gx = min(max(tl.x0 / 2 - 1 + rx, 0), pw - 1);
        const int off = (on && px < PROWS * PCOLS) ? (((tl.b * ph + gy) * pw + gx) * (int)p1.res_pix_stride + p1.res_ch_off + pc * 4) * 4 : (int)OOB;
        pst[i] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
      }
    };
    auto store_p = [&](float* dstp) {
#pragma unroll
      for (int i = 0; i < NPI; ++i) {
        const int it = tid + 256 * i, px = it / (8 * NCH), pc = it - px * (8 * NCH);
        if (it < PROWS * PCOLS * 8 * NCH) *(u32x4*)&dstp[px * PP + pc * 4] = pst[i];
      }
    };
    auto produce = [&](int i) {           // tile i of this workgroup into image i & 1; prefetches the head of tile i + 1
      const Tile tl = decode(tile_at(i));
      const bool nx_on = i + 1 < nmine;
      const Tile nx = decode(tile_at(nx_on ? i + 1 : i));
      _Float16* img = ldsh + (i & 1) * IMG;
      if constexpr (UPADD) issue_p(nx, nx_on);            // the next tile's addend: lands long before the store below
      ptile = lp + (i & 1) * PTILE;
      [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
        (compute(tl, nx, nx_on, img, std::integral_constant<int, Ks>{}), ...);
      }(std::make_integer_sequence<int, N>{});
      if constexpr (UPADD) store_p(lp + ((i + 1) & 1) * PTILE);
    };

    if constexpr (UPADD) {            // tile 0's addend, visible to all producer waves before they use it
      issue_p(decode(tile_at(0)), nmine > 0);
      store_p(lp);
      lds_barrier();
    }
    if (nmine > 0) {
      const Tile t0 = decode(tile_at(0));
      if constexpr (!WLDS) load_w(std::integral_constant<int, 0>{});
      [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
        (issue(t0, true, std::integral_constant<int, Ks>{}), ...);
      }(std::make_integer_sequence<int, DIST>{});
      produce(0);
    }
    lds_barrier();
    stamp(t_wait); t_work = 0; t_wait = 0;
    for (int i = 0; i < nmine; ++i) {
      if (i + 1 < nmine) produce(i + 1);
      stamp(t_work);
      lds_barrier();
      stamp(t_wait);
    }
  }
  if (wave >= 4) {
    // =================================================================== consumers: 9 taps from the LDS image
    if (dbg & 128) __builtin_amdgcn_s_setprio(1);
    // Wave -> (rows, output tiles): 8-row tiles give every wave two rows and all WN output tiles; with 4-row tiles and
    // 64 output channels a wave takes two rows and ONE of the two output tiles, so that every weight fragment it pulls
    // from L2 feeds two row blocks and the register ring runs two taps ahead (a 64-wide wave tile on one row would
    // pull twice the weights per pixel with one tap of look-ahead: measured L2-latency bound).
    constexpr int NSPLIT = (TH == 4 && WN == 2) ? 2 : 1;
    constexpr int WNW = WN / NSPLIT, WMW = TH * NSPLIT / 4;
    const int cw = wave - 4;
    const int row0 = (cw / NSPLIT) * WMW, nt0 = (cw % NSPLIT) * WNW;
    const unsigned frame_out = (unsigned)H * W * (unsigned)p2.out_pix_stride * 4u;
    const unsigned frame_res = (unsigned)H * W * (unsigned)p2.res_pix_stride * 4u;
    const float slope_out = p2.act == EGNE_ACT_RELU ? 0.f : (p2.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
    constexpr int KT16 = NCH * 2, NT2 = WN;
    const unsigned w2bytes = 9u * (unsigned)(KT16 * 16) * (unsigned)(NT2 * 32) * 2u;
    const __amdgpu_buffer_rsrc_t rwh = make_rsrc(f2hi, w2bytes), rwl = make_rsrc(f2lo, w2bytes);
    constexpr int stride_k16 = NT2 * 1024, stride_tap = KT16 * NT2 * 1024;
    const int wlane = lane * 16 + nt0 * 1024;
    const int abase = (row0 * HWd + li) * LDH + lh * 8;
    const int out_step = (int)p2.out_pix_stride * 4, res_step = (int)p2.res_pix_stride * 4;

    // One step = 16 intermediate channels of one tap: 3 * WMW * WNW MFMAs.  The operand rows of step j + 1 are requested from LDS
    // BEFORE the MFMAs of step j are issued (left alone, hipcc sinks every ds_read next to its use and each step pays the LDS
    // latency: one consumer wave per SIMD has nothing else to hide it with); the 3x3's weights are the same for every tile,
    // so their register ring runs on across tiles (32 -> 32 channels: all 36 fragments stay in registers for the whole launch).
    constexpr int NS = NCH * 18;                         // steps per tile
    constexpr bool WREG = NCH == 1 && WNW == 1;
    constexpr int NR = WREG ? NS : (WNW == 1 ? 6 : (NS % 4 == 0 ? 4 : 3));   // ring slots in steps; NS % NR == 0
    static_assert(NS % NR == 0, "ring slot of a step must not depend on the tile");
    auto w_off = [&](int j) { const int ch = j / 18, tap = (j % 18) >> 1, ks = j & 1; return (ch * 2 + ks) * stride_k16 + tap * stride_tap; };
    auto a_off = [&](int j) { const int ch = j / 18, tap = (j % 18) >> 1, ks = j & 1; return ch * NPX * LDH + ((tap / 3) * HWd + tap % 3) * LDH + ks * 16; };
    const bool st_on = p2.stats_ws != nullptr;
    const bool full_epi = p2.post_scale != nullptr || p2.residual != nullptr;
    egne::EpiLane ek[WNW];                               // per-lane output channel constants
#pragma unroll
    for (int tn = 0; tn < WNW; ++tn) {
      const int n = (nt0 + tn) * 32 + li;
      const bool nok = n < p2.Cout_store;
      ek[tn].bias = (p2.bias && nok) ? p2.bias[n] : 0.f;
      ek[tn].post_scale = (p2.post_scale && nok) ? p2.post_scale[n] : 1.f;
      ek[tn].post_shift = (p2.post_scale && nok) ? p2.post_shift[n] : 0.f;
    }
    u32x4 qh[NR][WNW], ql[NR][WNW];
#pragma unroll
    for (int s = 0; s < NR; ++s)
#pragma unroll
      for (int tn = 0; tn < WNW; ++tn) {
        qh[s][tn] = __builtin_amdgcn_raw_buffer_load_b128(rwh, wlane, w_off(s) + tn * 1024, 0);
        ql[s][tn] = __builtin_amdgcn_raw_buffer_load_b128(rwl, wlane, w_off(s) + tn * 1024, 0);
      }
    if constexpr (UPADD || C1V) lds_barrier();      // matches the producers' staging barrier
    lds_barrier();
    stamp(t_wait); t_work = 0; t_wait = 0;
    for (int i = 0; i < nmine; ++i) {
      const Tile tl = decode(tile_at(i));
      const _Float16* Thi = ldsh + (i & 1) * IMG + abase;
      const _Float16* Tlo = Thi + NCH * NPX * LDH;
      f32x16 acc[WMW][WNW];
#pragma unroll
      for (int a = 0; a < WMW; ++a)
#pragma unroll
        for (int n = 0; n < WNW; ++n) acc[a][n] = (f32x16)(0.f);
      h8 ah[2][WMW], al[2][WMW];
#pragma unroll
      for (int tm = 0; tm < WMW; ++tm) {
        ah[0][tm] = *(const h8*)&Thi[a_off(0) + tm * HWd * LDH];
        al[0][tm] = *(const h8*)&Tlo[a_off(0) + tm * HWd * LDH];
      }
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        if (j + 1 < NS) {
#pragma unroll
          for (int tm = 0; tm < WMW; ++tm) {
            ah[(j + 1) & 1][tm] = *(const h8*)&Thi[a_off(j + 1) + tm * HWd * LDH];
            al[(j + 1) & 1][tm] = *(const h8*)&Tlo[a_off(j + 1) + tm * HWd * LDH];
          }
        }
        h8 bh[WNW], bl[WNW];
#pragma unroll
        for (int tn = 0; tn < WNW; ++tn) {
          bh[tn] = __builtin_bit_cast(h8, qh[j % NR][tn]);
          bl[tn] = __builtin_bit_cast(h8, ql[j % NR][tn]);
          if constexpr (!WREG) {
            qh[j % NR][tn] = __builtin_amdgcn_raw_buffer_load_b128(rwh, wlane, w_off((j + NR) % NS) + tn * 1024, 0);
            ql[j % NR][tn] = __builtin_amdgcn_raw_buffer_load_b128(rwl, wlane, w_off((j + NR) % NS) + tn * 1024, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);              // the reads and refills above are issued before this step's MFMAs
#pragma unroll
        for (int tm = 0; tm < WMW; ++tm)
#pragma unroll
          for (int tn = 0; tn < WNW; ++tn) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j & 1][tm], bh[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j & 1][tm], bl[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j & 1][tm], bh[tn], acc[tm][tn], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (dbg & 64) { asm volatile("" : "+v"(acc[0][0])); stamp(t_mma); }

      // ---- epilogue (epilogue32.h): lane holds channel n of 16 pixels x = x_lane + c_r, c_r = (r&3) + 8*(r>>2), of tile row tm ----
      {
        const __amdgpu_buffer_rsrc_t rout = make_rsrc(p2.out + (long long)tl.b * H * W * p2.out_pix_stride, frame_out);
        const __amdgpu_buffer_rsrc_t rres =
            make_rsrc(p2.residual ? p2.residual + (long long)tl.b * H * W * p2.res_pix_stride : nullptr, p2.residual ? frame_res : 0u);
        const int xl = tl.x0 + 4 * lh;
        const int cmax = xl < W ? W - xl : 0;      // c_r < cmax  <=>  x < W
        const bool edge = tl.x0 + TW > W || tl.y0 + TH > H || (nt0 + WNW) * 32 > p2.Cout_store;     // wave-uniform
        bool bad = false;
#pragma unroll
        for (int tn = 0; tn < WNW; ++tn) {
          const int n = (nt0 + tn) * 32 + li;
          const bool nok = n < p2.Cout_store;
          double st_s = 0., st_q = 0.;       // sum / sum of squares of this wave's stored values of channel n (stats_ws)
#pragma unroll
          for (int tm = 0; tm < WMW; ++tm) {
            const int y = tl.y0 + row0 + tm;
            const int cm = (nok && y < H) ? cmax : 0;
            const int pix = y * W + xl;
            const int o0 = (pix * (int)p2.out_pix_stride + p2.out_ch_off + n) * 4;
            const int r0 = (pix * (int)p2.res_pix_stride + p2.res_ch_off + n) * 4;
            if (st_on) egne::epi_row32_select<true>(edge, p2.post_scale != nullptr, p2.residual != nullptr, acc[tm][tn], rout, rres, o0, r0, out_step, res_step, cm, os2, slope_out, ek[tn], st_s, st_q, bad, egne_ovf_row(y, H));
            else egne::epi_row32_select<false>(edge, p2.post_scale != nullptr, p2.residual != nullptr, acc[tm][tn], rout, rres, o0, r0, out_step, res_step, cm, os2, slope_out, ek[tn], st_s, st_q, bad, egne_ovf_row(y, H));
          }
          if (st_on) {                 // one chunk = this wave's rows of this tile (fixed order: deterministic)
            st_s += __shfl_xor(st_s, 32); st_q += __shfl_xor(st_q, 32);
            if (lh == 0 && n < p2.Cout_store) {
              const int tile_in_frame = (tl.y0 / TH) * tiles_x + tl.x0 / TW;
              double2* w = (double2*)p2.stats_ws + ((long long)tl.b * p2.stats_nchunk + tile_in_frame * (4 / NSPLIT) + cw / NSPLIT) * p2.Cout_store + n;
              *w = make_double2(st_s, st_q);
            }
          }
        }
        egne_ovf_commit(bad, p2.ovf_flag);
      }
      stamp(t_work);
      lds_barrier();      // image i&1 may be overwritten, image (i+1)&1 is complete
      stamp(t_wait);
    }
  }
  if ((dbg & 64) && lane == 0) {
    unsigned long long* o = g_stamps + ((long long)blockIdx.x * 8 + wave) * 4;
    o[0] = t_work; o[1] = t_wait; o[2] = nmine | ((unsigned long long)(t_mma / (nmine ? nmine : 1)) << 32); o[3] = __builtin_amdgcn_s_memrealtime() - r_first;
  }
}

template <int NCH, int WN, int TH, int NB, bool C4 = false, bool UPADD = false, bool UNI = false, int GL = 0, bool C1V = false>
int launch_fused(const egne_conv_desc& d1, const egne_conv_desc& d2, const GroupTab& gt, const _Float16* w1hi, const _Float16* w1lo,
                 int G1, const _Float16* f2hi, const _Float16* f2lo, float a1, float os1, float a2, float os2, hipStream_t st) {
  const int tiles_x = (d2.W + TW - 1) / TW, tiles_y = (d2.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y * d2.B;
  const size_t lds = (size_t)2 * 2 * NCH * (TH + 2) * HWd * LDH * sizeof(_Float16) + 32 * NCH * sizeof(float) + (NCH == 1 ? (size_t)G1 * 2048 : (size_t)(G1 < 7 ? G1 : 7) * 4096) +
                     (UPADD ? (size_t)2 * (TH / 2 + 2) * (TW / 2 + 2) * (32 * NCH + 4) * sizeof(float) : 0) +
                     (C1V ? ((size_t)2 * (TH + 4) * (TW + 4) + 32 * 12) * sizeof(float) : 0);
  static bool once = hipFuncSetAttribute((const void*)fused_1x1_3x3_kernel<NCH, WN, TH, NB, C4, UPADD, UNI, GL, C1V>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024) == hipSuccess;
  if (!once || lds > 160 * 1024) return egne::fail(EGNE_ERR_LAUNCH, "conv_fused_1x1_3x3: %zu bytes of LDS", lds);
  int gx = 256;
  if (gx > ntiles) gx = ntiles;
  hipLaunchKernelGGL((fused_1x1_3x3_kernel<NCH, WN, TH, NB, C4, UPADD, UNI, GL, C1V>), dim3(gx), dim3(512), lds, st, d1, d2, gt, w1hi, w1lo, G1, f2hi, f2lo, a1, os1,
                     a2, os2, tiles_x, tiles_y, ntiles);
  return egne::check_launch("egne_conv1x1_3x3_fused_f16_fwd");
}

}  // namespace

// d1: the 1x1 convolution (kh = kw = 1, stride 1, raw slices without fused affine, no activation / residual / post affine,
//     CoutP = 32 or 64 = d2's input width rounded up to 32; its output pointer is ignored -- the result is not stored);
// d2: the 3x3 / stride 1 / pad 1 / dilation 1 convolution on that result (CoutP 32 or 64; seg[] ignored).
// w1*: egne_pack_conv1x1_weight_f16 fragments, f2*: egne_pack_conv_weight_f16frag fragments (Ktot = d1.CoutP).
// a1 pre-scales the slices, a2 the 1x1 result (both powers of two; |x|*a must stay inside the f16 range).
extern "C" int egne_conv1x1_3x3_fused_f16_fwd(const egne_conv_desc* dp1, const egne_conv_desc* dp2, const void* w1hi, const void* w1lo,
                                              float a1, float w1_scale, const void* f2hi, const void* f2lo, float a2, float w2_scale,
                                              void* stream) {
  EGNE_REQUIRE(dp1 && dp2 && w1hi && w1lo && f2hi && f2lo, "conv_fused_1x1_3x3: null pointer");
  const egne_conv_desc& d1 = *dp1;
  const egne_conv_desc& d2 = *dp2;
  EGNE_REQUIRE(d1.kh == 1 && d1.kw == 1 && d1.stride == 1 && d1.pad_h == 0 && d1.pad_w == 0 && d1.ngroups == 1 && d1.nseg >= 1 &&
               d1.nseg <= EGNE_MAXSEG && !d1.post_scale && d1.act == EGNE_ACT_NONE && (d1.CoutP == 32 || d1.CoutP == 64),
               "conv_fused_1x1_3x3: 1x1 descriptor");
  EGNE_REQUIRE(d2.kh == 3 && d2.kw == 3 && d2.stride == 1 && d2.pad_mode == 0 && d2.ngroups == 1 && d2.pad_h == 1 && d2.pad_w == 1 &&
               d2.dil[0] == 1 && d2.Ho == d2.H && d2.Wo == d2.W && d2.B == d1.B && d2.H == d1.H && d2.W == d1.W && d2.Ktot == d1.CoutP &&
               (d2.CoutP == 32 || d2.CoutP == 64), "conv_fused_1x1_3x3: 3x3 descriptor");
  EGNE_REQUIRE(!d1.bias || ((uintptr_t)d1.bias & 15) == 0, "conv_fused_1x1_3x3: bias alignment");
  GroupTab gt;
  gt.dn = 0;
  int G = 0;
  bool uni = true;
  for (int s = 0; s < d1.nseg; ++s) {
    const egne_seg& g = d1.seg[s];
    EGNE_REQUIRE(g.ptr && !g.scale && !g.shift && g.act_in == EGNE_ACT_NONE && g.Cp % 8 == 0 && g.ch_off % 4 == 0 && g.pix_stride % 4 == 0 &&
                 ((uintptr_t)g.ptr & 15) == 0 && g.ch_off + g.Cp <= g.pix_stride && (long long)d1.H * d1.W * g.pix_stride * 4 < (1ll << 31),
                 "conv_fused_1x1_3x3: slice %d", s);
    const int n16 = (g.Cp + 15) / 16;
    EGNE_REQUIRE(G + n16 <= MAXG, "conv_fused_1x1_3x3: more than %d channel groups", MAXG);
    // UNI: same buffer = same pixel pitch and a start that is a whole number of frames (0 or ONE common count dn) after seg[0]'s
    const long long fbytes = (long long)d1.H * d1.W * d1.seg[0].pix_stride * 4, delta = (const char*)g.ptr - (const char*)d1.seg[0].ptr;
    int far = 0;
    if (g.pix_stride != d1.seg[0].pix_stride || delta < 0 || delta % fbytes != 0) uni = false;
    else if (delta > 0) {
      const long long q = delta / fbytes;
      if (q > 0x7fffffff || (gt.dn != 0 && gt.dn != (int)q)) uni = false;
      else { gt.dn = (int)q; far = 0x4000; }
    }
    for (int k = 0; k < n16; ++k) {
      gt.v[G + k] = (s << 16) | ((k == n16 - 1 && (g.Cp & 15)) ? 0x8000 : 0) | far | k;
      gt.off[G + k] = (g.ch_off + 16 * k) * 4;
    }
    G += n16;
  }
  for (int k = G; k < MAXG; ++k) gt.v[k] = gt.off[k] = 0;
  EGNE_REQUIRE(d2.out && d2.Cout_store <= d2.CoutP && d2.out_ch_off + d2.Cout_store <= d2.out_pix_stride &&
               (long long)d2.H * d2.W * d2.out_pix_stride * 4 < (1ll << 31) &&
               (!d2.residual || (long long)d2.H * d2.W * d2.res_pix_stride * 4 < (1ll << 31)), "conv_fused_1x1_3x3: output");
  EGNE_REQUIRE(((uintptr_t)w1hi & 15) == 0 && ((uintptr_t)w1lo & 15) == 0 && ((uintptr_t)f2hi & 15) == 0 && ((uintptr_t)f2lo & 15) == 0 &&
               a1 > 0.f && a2 > 0.f && w1_scale > 0.f && w2_scale > 0.f, "conv_fused_1x1_3x3: weights / scales");
  {
    const int th = d1.CoutP == 32 ? 8 : 4, rg = (d1.CoutP == 64 && d2.CoutP == 64) ? 2 : 4;
    EGNE_REQUIRE(!d2.stats_ws || (((uintptr_t)d2.stats_ws & 15) == 0 && d2.stats_nchunk == ((d2.W + 31) / 32) * ((d2.H + th - 1) / th) * rg),
                 "conv_fused_1x1_3x3: stats_nchunk must be tiles * %d for this shape", rg);
  }
  const float os1 = 1.0f / (a1 * w1_scale), os2 = 1.0f / (a2 * w2_scale);
  hipStream_t st = (hipStream_t)stream;
  const _Float16 *a = (const _Float16*)w1hi, *b = (const _Float16*)w1lo, *c = (const _Float16*)f2hi, *e = (const _Float16*)f2lo;
  EGNE_REQUIRE(G <= 12, "conv_fused_1x1_3x3: at most 192 input channels (12 groups of 16) are built, got %d groups", G);
  if (d1.residual) {
    EGNE_REQUIRE(d1.CoutP == 32 && d2.CoutP == 32 && G <= 8 && d1.Ho * 2 == d1.H && d1.Wo * 2 == d1.W && d1.res_ch_off % 4 == 0 &&
                 d1.res_pix_stride % 4 == 0 && ((uintptr_t)d1.residual & 15) == 0 && d1.res_ch_off + 32 <= d1.res_pix_stride &&
                 (long long)d1.B * d1.Ho * d1.Wo * d1.res_pix_stride * 4 < (1ll << 31),
                 "conv_fused_1x1_3x3: the up-sampled addend needs 32 -> 32 channels, <= 8 groups and a half-resolution tensor");
    if (uni && G == 4) return launch_fused<1, 1, 8, 1, false, true, true, 4>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st);
    if (uni && G == 6) return launch_fused<1, 1, 8, 2, false, true, true, 2>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st);
    if (uni)
      return G <= 4 ? launch_fused<1, 1, 8, 1, false, true, true>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st)
                    : launch_fused<1, 1, 8, 2, false, true, true>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st);
    return G <= 4 ? launch_fused<1, 1, 8, 1, false, true>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st)
                  : launch_fused<1, 1, 8, 2, false, true>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st);
  }
  if (d1.CoutP == 32) {
    const int nb = (G + 3) / 4;
#define EGNE_FUSED(WN_, U_) \
  (nb == 1 ? launch_fused<1, WN_, 8, 1, false, false, U_>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st) \
           : nb == 2 ? launch_fused<1, WN_, 8, 2, false, false, U_>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st) \
                     : launch_fused<1, WN_, 8, 3, false, false, U_>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st))
    if (uni && d2.CoutP == 32 && (G == 4 || G == 6 || G == 8)) {     // the dense-block shapes: straight-line items
      if (G == 4) return launch_fused<1, 1, 8, 1, false, false, true, 4>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st);
      if (G == 6) return launch_fused<1, 1, 8, 2, false, false, true, 2>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st);
      return launch_fused<1, 1, 8, 2, false, false, true, 4>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st);
    }
    if (uni) return d2.CoutP == 32 ? EGNE_FUSED(1, true) : EGNE_FUSED(2, true);
    return d2.CoutP == 32 ? EGNE_FUSED(1, false) : EGNE_FUSED(2, false);
#undef EGNE_FUSED
  }
  // 64-channel intermediate: 4-row tiles, items of 2 groups, an even number of batches per 32-pixel block
  const int nb2 = ((G + 1) / 2 + 1) / 2 * 2;
#define EGNE_FUSED2(WN_, U_) \
  (nb2 == 2 ? launch_fused<2, WN_, 4, 2, false, false, U_>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st) \
            : nb2 == 4 ? launch_fused<2, WN_, 4, 4, false, false, U_>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st) \
                       : launch_fused<2, WN_, 4, 6, false, false, U_>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st))
  if (uni && d2.CoutP == 64 && G == 7) return launch_fused<2, 2, 4, 4, false, false, true, 1>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st);    // ESF-Net block 1:
  if (uni && d2.CoutP == 64 && G == 11) return launch_fused<2, 2, 4, 6, false, false, true, 1>(d1, d2, gt, a, b, G, c, e, a1, os1, a2, os2, st);   // straight-line items
  if (uni && d2.CoutP == 64) return EGNE_FUSED2(2, true);     // dense block with 64-channel intermediates (one buffer)
  return d2.CoutP == 32 ? EGNE_FUSED2(1, false) : EGNE_FUSED2(2, false);
#undef EGNE_FUSED2
}

// The same with a 3x3 / pad 1 convolution on <= 4 input channels in front (utils.py:1047-1048 convBlock: conv1 -> LeakyReLU ->
// conv2 -> LeakyReLU [-> eval BatchNorm as d2's post affine]): d1 = that convolution (one slice of >= 4 padded channels,
// CoutP = 32, its activation applied before the 3x3), c4hi / c4lo = egne_pack_conv3x3_c4_weight_f16 with CoutP 32.
extern "C" int egne_conv3x3c4_3x3_fused_f16_fwd(const egne_conv_desc* dp1, const egne_conv_desc* dp2, const void* c4hi, const void* c4lo,
                                                float a1, float w1_scale, const void* f2hi, const void* f2lo, float a2, float w2_scale,
                                                void* stream) {
  EGNE_REQUIRE(dp1 && dp2 && c4hi && c4lo && f2hi && f2lo, "conv_fused_c4_3x3: null pointer");
  const egne_conv_desc& d1 = *dp1;
  const egne_conv_desc& d2 = *dp2;
  EGNE_REQUIRE(d1.kh == 3 && d1.kw == 3 && d1.stride == 1 && d1.pad_h == 1 && d1.pad_w == 1 && d1.pad_mode == 0 && d1.ngroups == 1 &&
               d1.dil[0] == 1 && d1.nseg == 1 && !d1.residual && !d1.post_scale && d1.CoutP == 32, "conv_fused_c4_3x3: first descriptor");
  const egne_seg& g = d1.seg[0];
  const bool planar = g.pix_stride == 1;      // [B][H][W] one-channel input read in place (no NHWC staging copy)
  EGNE_REQUIRE(g.ptr && !g.scale && !g.shift && g.act_in == EGNE_ACT_NONE && ((uintptr_t)g.ptr & 15) == 0 &&
               (planar ? g.ch_off == 0 : (g.Cp >= 4 && g.ch_off % 4 == 0 && g.pix_stride % 4 == 0 && g.ch_off + 4 <= g.pix_stride)) &&
               (long long)d1.H * d1.W * g.pix_stride * 4 < (1ll << 31), "conv_fused_c4_3x3: input slice");
  EGNE_REQUIRE(d2.kh == 3 && d2.kw == 3 && d2.stride == 1 && d2.pad_mode == 0 && d2.ngroups == 1 && d2.pad_h == 1 && d2.pad_w == 1 &&
               d2.dil[0] == 1 && d2.Ho == d2.H && d2.Wo == d2.W && d2.B == d1.B && d2.H == d1.H && d2.W == d1.W && d2.Ktot == 32 &&
               d2.CoutP == 32, "conv_fused_c4_3x3: 3x3 descriptor");
  EGNE_REQUIRE(d2.out && d2.Cout_store <= d2.CoutP && d2.out_ch_off + d2.Cout_store <= d2.out_pix_stride &&
               (long long)d2.H * d2.W * d2.out_pix_stride * 4 < (1ll << 31) &&
               (!d2.residual || (long long)d2.H * d2.W * d2.res_pix_stride * 4 < (1ll << 31)), "conv_fused_c4_3x3: output");
  EGNE_REQUIRE(!d2.stats_ws || (((uintptr_t)d2.stats_ws & 15) == 0 && d2.stats_nchunk == ((d2.W + 31) / 32) * ((d2.H + 7) / 8) * 4),
               "conv_fused_c4_3x3: stats_nchunk must be tiles * 4");
  EGNE_REQUIRE(((uintptr_t)c4hi & 15) == 0 && ((uintptr_t)c4lo & 15) == 0 && ((uintptr_t)f2hi & 15) == 0 && ((uintptr_t)f2lo & 15) == 0 &&
               a1 > 0.f && a2 > 0.f && w1_scale > 0.f && w2_scale > 0.f && (!d1.bias || ((uintptr_t)d1.bias & 15) == 0), "conv_fused_c4_3x3: weights / scales");
  GroupTab gt;
  gt.dn = 0;
  for (int k = 0; k < MAXG; ++k) gt.v[k] = gt.off[k] = 0;
  // one-channel planar input with the compact fp32 weight table in d1.w ([32][12] floats: nine taps, bias, two zeros): first layer on
  // the vector ALU in exact fp32 (no weight fragments in LDS: G1 = 0)
  if (planar && d1.w && ((uintptr_t)d1.w & 15) == 0)
    return launch_fused<1, 1, 8, 1, true, false, false, 3, true>(d1, d2, gt, (const _Float16*)c4hi, (const _Float16*)c4lo, 0, (const _Float16*)f2hi,
                                                                 (const _Float16*)f2lo, a1, 1.0f / (a1 * w1_scale), a2, 1.0f / (a2 * w2_scale), (hipStream_t)stream);
  return launch_fused<1, 1, 8, 1, true, false, false, 3>(d1, d2, gt, (const _Float16*)c4hi, (const _Float16*)c4lo, 3, (const _Float16*)f2hi, (const _Float16*)f2lo,
                                        a1, 1.0f / (a1 * w1_scale), a2, 1.0f / (a2 * w2_scale), (hipStream_t)stream);
}

// diagnostics (not part of the product interface): set the debug word of the fused kernel, read its per-wave stamps
extern "C" int egne_fused_debug(int dbg, void* out_stamps) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &dbg, sizeof(int)) != hipSuccess) return -2;
  if (out_stamps && hipMemcpyFromSymbol(out_stamps, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 256 * 8 * 4) != hipSuccess) return -2;
  return 0;
}
