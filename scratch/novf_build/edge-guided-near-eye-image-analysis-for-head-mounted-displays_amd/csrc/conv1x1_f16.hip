// Streaming 1x1 convolution over a concatenation of NHWC slices on the split-f16 MFMA path (fp32 tensors, three
// v_mfma_f32_32x32x16_f16 per product, fp32 accumulate; numerics: conv_f16x3.hip).
//
// The 1x1 convs of ESF-Net's dense blocks (models/RITnet_v2.py:59,61 conv21 / conv31, :84,86 conv11 / conv21,
// :38 Transition_down) read 64..160 channels and write 32..64: 6 FLOP per byte, an HBM stream.  Nothing is
// reused between pixels, so nothing is staged: every wave walks 32-pixel blocks on its own, each lane loads its
// MFMA operand (8 channels of one pixel, two 16-byte pieces) straight from HBM through a BUFFER resource built
// per block (rows past the tensor end and padded channel groups get the out-of-range offset 0x80000000 and load
// zeros), splits it into hi / lo halves in registers and multiplies with weight fragments that sit in LDS for the
// whole launch.  No barriers after the weight copy, no address arithmetic in the loop; the occupancy (4 waves per
// SIMD) keeps ~150 KB of loads in flight per CU.
//
// K order: the slot (half h = lane>>5, j) of 16-channel group g holds channel 16*g + (j < 4 ? 4*h + j : 8 + 4*h + j - 4),
// so that the two lanes of a pixel read ADJACENT 16-byte pieces (one full 32-byte sector per instruction); the
// pack (egne_pack_conv1x1_weight_f16) uses the same order.  The product is computed transposed (weights as the
// A operand): a lane ends up with 4 consecutive output channels of one pixel -> 16-byte stores.
#include "common.h"
#include "split_f16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned OOB = 0x80000000u;

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

// fhi / flo: [G][CoutP/32][64 lanes][8 halfs]; G = sum over slices of ceil(Cp/16)
// UP: p.residual names a HALF-resolution tensor [B][p.Ho][p.Wo] whose bilinear x2 up-sampling (F.interpolate, scale 2, align_corners
//     False) is added to the result: conv11(cat(up(x), skip)) = up(W_up x) + W_skip skip (models/RITnet_v2.py:84-86; the 1x1 and the
//     interpolation are linear and the interpolation weights sum to one) -- the up-sampled operand of an up block is never
//     materialised, the 1x1 reads the skip slices only (conv_fused_1x1_3x3_f16.hip does the same for the 32-channel block).
template <int TN, bool UP = false>
__global__ __launch_bounds__(256) void conv1x1_f16x3_kernel(const egne_conv_desc p, const _Float16* __restrict__ fhi,
                                                            const _Float16* __restrict__ flo, float a_scale, float out_scale,
                                                            int G, long long M, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) _Float16 wl[];   // [G][TN][hi|lo][64][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kq = lane >> 5;
  const int NT = p.CoutP >> 5, nt0 = blockIdx.y * TN;
  for (int it = tid; it < G * TN * 2 * 64; it += 256) {        // 16-byte items
    const int l = it & 63, hl = (it >> 6) & 1;
    const int tn = (it >> 7) % TN, g = (it >> 7) / TN;
    const _Float16* src = (hl ? flo : fhi) + (((long long)g * NT + nt0 + tn) * 64 + l) * 8;
    *(u32x4*)&wl[(long long)it * 8] = *(const u32x4*)src;
  }
  __syncthreads();
  // UP: this wave's patch of the half-resolution addend behind the weights: [UPC columns][32 TN channels], the two tapped rows
  // already blended (every pixel of a block shares them and their weight); pixel pitch 32 TN + 8 floats (the 16 (column,
  // channel-quad) pairs of a 16-lane read group fall on 16 different 16-byte slots)
  constexpr int UPC = 18, UPP = 32 * TN + 8;
  [[maybe_unused]] float* const patch = (float*)(wl + (long long)G * TN * 2 * 64 * 8) + wave * (UPC * UPP);

  const float slope = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  // transposed product: lane holds output channels n = 32*nt + 8*j + 4*kq + e (r = 4*j + e) of pixel li
  f32x4 bias[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = (nt0 + tn) * 32 + 8 * j + 4 * kq;
      bias[tn][j] = (p.bias && n < p.Cout_store) ? *(const f32x4*)(p.bias + n) : (f32x4)(0.f);
    }

  for (int blk = blockIdx.x * 4 + wave; blk < nblocks; blk += gridDim.x * 4) {
    const long long m0 = (long long)blk * 32;
    const long long rows = M - m0 < 32 ? M - m0 : 32;
    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tn] = (f32x16)(0.f);
    int g = 0;
    for (int s = 0; s < p.nseg; ++s) {
      const egne_seg sg = p.seg[s];
      const __amdgpu_buffer_rsrc_t r = make_rsrc(sg.ptr + m0 * sg.pix_stride, (unsigned)(rows * sg.pix_stride * 4));
      const int voff = (li * (int)sg.pix_stride + sg.ch_off + 4 * kq) * 4;
      const int n16 = (sg.Cp + 15) >> 4;
      const bool tail8 = (sg.Cp & 15) != 0;                      // last group of the slice holds 8 channels only
      for (int g0 = 0; g0 < n16; g0 += 4) {
        u32x4 xa[4], xb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int gg = g0 + u;
          const int oa = gg < n16 ? voff : (int)OOB;
          const int ob = (gg < n16 && !(tail8 && gg == n16 - 1)) ? voff : (int)OOB;
          xa[u] = __builtin_amdgcn_raw_buffer_load_b128(r, oa, gg * 64, 0);        // channels 16g + 4kq .. +3
          xb[u] = __builtin_amdgcn_raw_buffer_load_b128(r, ob + 32, gg * 64, 0);   // channels 16g + 8 + 4kq .. +3
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (g0 + u < n16) {
            h8 ah, al;
            split8(xa[u], xb[u], a_scale, ah, al);
            const _Float16* wp = wl + ((long long)(g + g0 + u) * TN * 2 * 64 + lane) * 8;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
              const h8 wh = *(const h8*)(wp + (tn * 2 + 0) * 512), wo = *(const h8*)(wp + (tn * 2 + 1) * 512);
              acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al, acc[tn], 0, 0, 0);
              acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wo, ah, acc[tn], 0, 0, 0);
              acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah, acc[tn], 0, 0, 0);
            }
          }
        }
      }
      g += n16;
    }
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + m0 * p.out_pix_stride, (unsigned)(rows * p.out_pix_stride * 4));
    bool bad = false;          // a non-finite result of this block (egne_conv_desc.ovf_flag)
    // UP: the lane's pixel (b, y, x) and its four low-resolution taps -- ATen's area_pixel_compute_source_index(scale 0.5,
    // align_corners false): s = max(0.5 (d + 0.5) - 0.5, 0)
    [[maybe_unused]] int o00 = 0, o01 = 0, o10 = 0, o11 = 0;
    [[maybe_unused]] float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rres = ro;
    [[maybe_unused]] bool pix_ok = false, staged = false;
    if constexpr (UP) {
      const int ph = p.Ho, pw = p.Wo, hw = p.H * p.W;
      const long long m = m0 + li;
      pix_ok = m < M;
      const int b = pix_ok ? (int)(m / hw) : 0;
      const int r = pix_ok ? (int)(m - (long long)b * hw) : 0;
      const int y = r / p.W, x = r - y * p.W;
      float sy = 0.5f * (y + 0.5f) - 0.5f, sx = 0.5f * (x + 0.5f) - 0.5f;
      sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
      const int y0 = (int)sy, x0 = (int)sx, y1 = y0 + (y0 < ph - 1 ? 1 : 0), x1 = x0 + (x0 < pw - 1 ? 1 : 0);
      const float ly = sy - y0, lx = sx - x0;
      w00 = (1.f - ly) * (1.f - lx); w01 = (1.f - ly) * lx; w10 = ly * (1.f - lx); w11 = ly * lx;
      const int rs = (int)p.res_pix_stride, base = b * ph * pw;
      rres = make_rsrc(p.residual, (unsigned)((long long)p.B * ph * pw * rs * 4));
      // a block that lies inside ONE image row (wave-uniform): its pixels tap the same two low-resolution rows (blended while they are
      // staged) and <= 18 columns -- whole-pixel loads (lanes side by side, 128 TN bytes per pixel) into LDS instead of 4 x 4 TN
      // gathers of 16 bytes per lane, every one of which touches 32 different cache lines
      const int r0 = (int)(m0 % hw), xb0 = r0 % p.W;
      staged = rows == 32 && xb0 + 32 <= p.W;
      if (staged) {
        const int bb = (int)(m0 / hw), yb = r0 / p.W;
        float syb = 0.5f * (yb + 0.5f) - 0.5f; syb = syb < 0.f ? 0.f : syb;
        const int yb0 = (int)syb, yb1 = yb0 + (yb0 < ph - 1 ? 1 : 0);
        const float lyb = syb - yb0;
        float sxb = 0.5f * (xb0 + 0.5f) - 0.5f; sxb = sxb < 0.f ? 0.f : sxb;
        const int cb = (int)sxb;                                  // first column tapped by the block
        constexpr int VPP = 8 * TN;                              // 16-byte vectors per pixel
        for (int it = lane; it < UPC * VPP; it += 64) {
          const int cc = it / VPP, c4 = it - cc * VPP;
          const int gx = cb + cc < pw ? cb + cc : pw - 1;
          const int n = nt0 * 32 + c4 * 4;
          const int o0 = n < p.Cout_store ? (((bb * ph + yb0) * pw + gx) * rs + p.res_ch_off + n) * 4 : (int)OOB;
          const int o1 = n < p.Cout_store ? (((bb * ph + yb1) * pw + gx) * rs + p.res_ch_off + n) * 4 : (int)OOB;
          const f32x4 ra = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, o0, 0, 0));
          const f32x4 rb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, o1, 0, 0));
          *(f32x4*)&patch[cc * UPP + c4 * 4] = (1.f - lyb) * ra + lyb * rb;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // one wave: its own LDS writes are visible to its own reads
        o00 = ((x0 - cb) * UPP) * 4; o01 = ((x1 - cb) * UPP) * 4;
        w00 = 1.f - lx; w01 = lx;
      } else {
        o00 = ((base + y0 * pw + x0) * rs + p.res_ch_off) * 4; o01 = ((base + y0 * pw + x1) * rs + p.res_ch_off) * 4;
        o10 = ((base + y1 * pw + x0) * rs + p.res_ch_off) * 4; o11 = ((base + y1 * pw + x1) * rs + p.res_ch_off) * 4;
      }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = (nt0 + tn) * 32 + 8 * j + 4 * kq;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = acc[tn][4 * j + e] * out_scale + bias[tn][j][e];
          v[e] = fmaxf(t, t * slope);
        }
        if (tn == 0 && j == 0) bad |= egne_nonfinite(v[0]);       // lane = pixel: one channel per pixel (common.h)
        if constexpr (UP) {
          if (staged) {
            const char* pb = (const char*)patch + (tn * 32 + 8 * j + 4 * kq) * 4;
            v += w00 * *(const f32x4*)(pb + o00) + w01 * *(const f32x4*)(pb + o01);
          } else {
          const bool on = pix_ok && n < p.Cout_store;
          const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, on ? o00 + n * 4 : (int)OOB, 0, 0));
          const f32x4 b_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, on ? o01 + n * 4 : (int)OOB, 0, 0));
          const f32x4 c = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, on ? o10 + n * 4 : (int)OOB, 0, 0));
          const f32x4 d_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, on ? o11 + n * 4 : (int)OOB, 0, 0));
          v += w00 * a + w01 * b_ + w10 * c + w11 * d_;
          }
        }
        const int off = n < p.Cout_store ? (li * (int)p.out_pix_stride + p.out_ch_off + n) * 4 : (int)OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, off, 0, 0);
      }
    egne_ovf_commit(bad, p.ovf_flag);
  }
}


// Transition_down of an eval plan in ONE pass (models/RITnet_v2.py:32-44: avg_pool2d(conv1x1(leaky(IN(cat(out, x)))), 2); the
// pooling and the 1x1 are both linear, so the 2x2 average moves in front of the convolution): the lane's MFMA operand is the
// average over the 2x2 window of leaky(x * scale + shift) -- 8 loads of 16 bytes per 16-channel group instead of 2, no pooled
// tensor in HBM (it was written by norm_act_pool2_k and read back by the streaming kernel: one launch and 2 x the pooled bytes
// less).  Same weights and K order as conv1x1_f16x3_kernel; p.H / p.W are the INPUT size, p.Ho / p.Wo = H/2, W/2.
template <int TN>
__global__ __launch_bounds__(256) void conv1x1_pool_f16x3_kernel(const egne_conv_desc p, const _Float16* __restrict__ fhi,
                                                                 const _Float16* __restrict__ flo, float a_scale, float out_scale,
                                                                 int G, long long M, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) _Float16 wl[];   // [G][TN][hi|lo][64][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kq = lane >> 5;
  const int NT = p.CoutP >> 5, nt0 = blockIdx.y * TN;
  for (int it = tid; it < G * TN * 2 * 64; it += 256) {        // 16-byte items
    const int l = it & 63, hl = (it >> 6) & 1;
    const int tn = (it >> 7) % TN, g = (it >> 7) / TN;
    const _Float16* src = (hl ? flo : fhi) + (((long long)g * NT + nt0 + tn) * 64 + l) * 8;
    *(u32x4*)&wl[(long long)it * 8] = *(const u32x4*)src;
  }
  __syncthreads();

  const float slope = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  f32x4 bias[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = (nt0 + tn) * 32 + 8 * j + 4 * kq;
      bias[tn][j] = (p.bias && n < p.Cout_store) ? *(const f32x4*)(p.bias + n) : (f32x4)(0.f);
    }
  const int H = p.H, W = p.W, Hp = p.Ho, Wp = p.Wo, hwp = Hp * Wp;
  const float s4 = 0.25f * a_scale;

  for (int blk = blockIdx.x * 4 + wave; blk < nblocks; blk += gridDim.x * 4) {
    const long long m0 = (long long)blk * 32;
    const long long rows = M - m0 < 32 ? M - m0 : 32;
    const int b0 = (int)(m0 / hwp);                       // frame of the block's first pixel; a block spans at most two frames
    const long long m = m0 + li;
    const bool valid = m < M;
    const int b = valid ? (int)(m / hwp) : b0;
    const int r = (int)(m - (long long)b * hwp);
    const int yp = valid ? r / Wp : 0, xp = valid ? r - yp * Wp : 0;
    const int pin = ((b - b0) * H + 2 * yp) * W + 2 * xp;   // first pixel of the window, relative to frame b0
    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tn] = (f32x16)(0.f);
    int g = 0;
    for (int s = 0; s < p.nseg; ++s) {
      const egne_seg sg = p.seg[s];
      const long long frame = (long long)H * W * sg.pix_stride;
      const long long left = ((long long)p.B - b0) * frame * 4;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(sg.ptr + (long long)b0 * frame, (unsigned)(left < 2 * frame * 4 ? left : 2 * frame * 4));
      const int ps4 = (int)sg.pix_stride * 4;
      const int voff = valid ? (pin * (int)sg.pix_stride + sg.ch_off + 4 * kq) * 4 : (int)OOB;
      const float slope_in = sg.act_in == EGNE_ACT_RELU ? 0.f : (sg.act_in == EGNE_ACT_LEAKY ? 0.01f : 1.f);
      const float* tsc = sg.scale + (long long)b * sg.Cp + 4 * kq;
      const float* tsh = sg.shift + (long long)b * sg.Cp + 4 * kq;
      const int n16 = (sg.Cp + 15) >> 4;
      const bool tail8 = (sg.Cp & 15) != 0;                      // last group of the slice holds 8 channels only
      for (int gg = 0; gg < n16; ++gg) {
        const bool has_b = !(tail8 && gg == n16 - 1);
        const int ob = has_b ? voff + 32 : (int)OOB;
        u32x4 xa[4], xb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int d = ((t >> 1) * W + (t & 1)) * ps4;
          xa[t] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff == (int)OOB ? (int)OOB : voff + d, gg * 64, 0);
          xb[t] = __builtin_amdgcn_raw_buffer_load_b128(rs, ob == (int)OOB ? (int)OOB : ob + d, gg * 64, 0);
        }
        const f32x4 sca = *(const f32x4*)(tsc + gg * 16), sha = *(const f32x4*)(tsh + gg * 16);
        const f32x4 scb = has_b ? *(const f32x4*)(tsc + gg * 16 + 8) : (f32x4)(0.f), shb = has_b ? *(const f32x4*)(tsh + gg * 16 + 8) : (f32x4)(0.f);
        f32x4 sa = (f32x4)(0.f), sb = (f32x4)(0.f);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f32x4 va = __builtin_bit_cast(f32x4, xa[t]) * sca + sha, vb = __builtin_bit_cast(f32x4, xb[t]) * scb + shb;
#pragma unroll
          for (int e = 0; e < 4; ++e) { sa[e] += fmaxf(va[e], va[e] * slope_in); sb[e] += fmaxf(vb[e], vb[e] * slope_in); }
        }
        if (!valid) { sa = (f32x4)(0.f); sb = (f32x4)(0.f); }
        h8 ah, al;
        split8(__builtin_bit_cast(u32x4, sa), __builtin_bit_cast(u32x4, sb), s4, ah, al);
        const _Float16* wp = wl + ((long long)(g + gg) * TN * 2 * 64 + lane) * 8;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const h8 wh = *(const h8*)(wp + (tn * 2 + 0) * 512), wo = *(const h8*)(wp + (tn * 2 + 1) * 512);
          acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al, acc[tn], 0, 0, 0);
          acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wo, ah, acc[tn], 0, 0, 0);
          acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah, acc[tn], 0, 0, 0);
        }
      }
      g += n16;
    }
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + m0 * p.out_pix_stride, (unsigned)(rows * p.out_pix_stride * 4));
    bool bad = false;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = (nt0 + tn) * 32 + 8 * j + 4 * kq;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = acc[tn][4 * j + e] * out_scale + bias[tn][j][e];
          v[e] = fmaxf(t, t * slope);
        }
        if (tn == 0 && j == 0) bad |= egne_nonfinite(v[0]);       // lane = pixel: one channel per pixel (common.h)
        const int off = n < p.Cout_store ? (li * (int)p.out_pix_stride + p.out_ch_off + n) * 4 : (int)OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, off, 0, 0);
      }
    egne_ovf_commit(bad, p.ovf_flag);
  }
}

// OIHW (kh = kw = 1) fp32 -> hi / lo f16 fragments [G][CoutP/32][lane = h*32 + n%32][8]; kmap[g*16 + h*8 + j] names
// the logical input channel of that K slot (or -1: padding)
__global__ void pack_w1x1_f16_k(const float* __restrict__ w, int Cout, int Cin, const int* __restrict__ kmap, int G, int CoutP,
                                float wscale, _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
  const long long total = (long long)G * CoutP * 16;
  const int NT = CoutP >> 5;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), nn = (int)((i >> 3) & 31), h = (int)((i >> 8) & 1);
    long long q = i >> 9;
    const int nt = (int)(q % NT);
    const int g = (int)(q / NT);
    const int n = nt * 32 + nn, ci = kmap[g * 16 + h * 8 + j];
    const float v = (n < Cout && ci >= 0) ? w[(long long)n * Cin + ci] * wscale : 0.f;
    const _Float16 hh = (_Float16)v;
    hi[i] = hh;
    lo[i] = (_Float16)(v - (float)hh);
  }
}

}  // namespace

extern "C" int egne_pack_conv1x1_weight_f16(const float* w_oihw, int Cout, int Cin, const int32_t* kmap, int G, int CoutP,
                                            float wscale, void* fhi, void* flo, void* stream) {
  EGNE_REQUIRE(w_oihw && kmap && fhi && flo && Cout > 0 && Cin > 0 && G > 0 && CoutP >= Cout && CoutP % 32 == 0 && wscale > 0.f,
               "pack_conv1x1_f16: bad sizes Cout %d Cin %d G %d CoutP %d", Cout, Cin, G, CoutP);
  long long total = (long long)G * CoutP * 16, g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pack_w1x1_f16_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, kmap, G, CoutP, wscale,
                     (_Float16*)fhi, (_Float16*)flo);
  return egne::check_launch("egne_pack_conv1x1_weight_f16");
}

// 1x1 / stride 1 / no padding over up to EGNE_MAXSEG raw slices (no fused affine), CoutP 32 or a multiple of 64, no
// residual, no post affine.  Weights in the order of egne_pack_conv1x1_weight_f16 (G = sum of ceil(Cp/16)).
// With d.residual set AND d.Ho * 2 == d.H, d.Wo * 2 == d.W (otherwise Ho / Wo equal H / W): the residual is a HALF-resolution
// tensor [B][Ho][Wo] whose bilinear x2 up-sampling is added to the (activation-free) result, see conv1x1_f16x3_kernel<TN, UP>.
extern "C" int egne_conv1x1_f16x3_fwd(const egne_conv_desc* dp, const void* fhi, const void* flo, float a_scale, float w_scale,
                                      void* stream) {
  EGNE_REQUIRE(dp && fhi && flo, "conv1x1_f16: null pointer");
  const egne_conv_desc& d = *dp;
  const bool up = d.residual && d.Ho * 2 == d.H && d.Wo * 2 == d.W;
  EGNE_REQUIRE(d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0 && d.ngroups == 1 && (up || (d.Ho == d.H && d.Wo == d.W)) &&
               d.nseg >= 1 && d.nseg <= EGNE_MAXSEG && (up || !d.residual) && !d.post_scale, "conv1x1_f16: unsupported descriptor");
  EGNE_REQUIRE(!up || (d.act == EGNE_ACT_NONE && ((uintptr_t)d.residual & 15) == 0 && d.res_pix_stride % 4 == 0 && d.res_ch_off % 4 == 0 &&
                       d.res_ch_off + d.Cout_store <= d.res_pix_stride && (long long)d.B * d.Ho * d.Wo * d.res_pix_stride * 4 < (1ll << 31) &&
                       (long long)d.B * d.H * d.W < (1ll << 31)),
               "conv1x1_f16: up-sampled addend (no activation, 16-byte aligned channel vectors, < 2 GB)");
  EGNE_REQUIRE(d.CoutP % 32 == 0 && (d.CoutP == 32 || d.CoutP % 64 == 0) && d.Cout_store <= d.CoutP && d.Cout_store % 4 == 0 && d.out &&
               ((uintptr_t)d.out & 15) == 0 && d.out_pix_stride % 4 == 0 && d.out_ch_off % 4 == 0 &&
               d.out_ch_off + d.Cout_store <= d.out_pix_stride, "conv1x1_f16: output");
  EGNE_REQUIRE(!d.bias || ((uintptr_t)d.bias & 15) == 0, "conv1x1_f16: bias alignment");
  int G = 0;
  for (int s = 0; s < d.nseg; ++s) {
    const egne_seg& g = d.seg[s];
    EGNE_REQUIRE(g.ptr && !g.scale && !g.shift && g.act_in == EGNE_ACT_NONE && g.Cp % 8 == 0 && g.ch_off % 4 == 0 &&
                 g.pix_stride % 4 == 0 && ((uintptr_t)g.ptr & 15) == 0 && g.ch_off + g.Cp <= g.pix_stride && g.pix_stride * 128 < (1ll << 31),
                 "conv1x1_f16: slice %d", s);
    G += (g.Cp + 15) / 16;
  }
  EGNE_REQUIRE(d.out_pix_stride * 128 < (1ll << 31) && ((uintptr_t)fhi & 15) == 0 && ((uintptr_t)flo & 15) == 0 && a_scale > 0.f && w_scale > 0.f,
               "conv1x1_f16: strides / weights / scales");
  const long long M = (long long)d.B * d.H * d.W;
  const long long nb = (M + 31) / 32;
  EGNE_REQUIRE(nb < (1ll << 31), "conv1x1_f16: too many pixels");
  const int TN = d.CoutP == 32 ? 1 : 2;
  const size_t lds = (size_t)G * TN * 2 * 64 * 8 * sizeof(_Float16);
  EGNE_REQUIRE(lds <= 80 * 1024, "conv1x1_f16: K = %d groups of 16 does not fit the LDS weight image", G);
  static bool once = [] {
    return hipFuncSetAttribute((const void*)conv1x1_f16x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess &&
           hipFuncSetAttribute((const void*)conv1x1_f16x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
  }();
  if (!once) return egne::fail(EGNE_ERR_LAUNCH, "conv1x1_f16: cannot raise the dynamic LDS limit");
  const int ny = d.CoutP / (32 * TN);
  long long gx = (nb + 3) / 4;
  const long long cap = 256 * 8;
  if (gx > cap) gx = cap;
  const float os = 1.0f / (a_scale * w_scale);
  hipStream_t st = (hipStream_t)stream;
  if (up) {
    const size_t lds_up = lds + (size_t)4 * 18 * (32 * TN + 8) * sizeof(float);       // + a patch of the addend per wave
    EGNE_REQUIRE(lds_up <= 80 * 1024, "conv1x1_f16: K = %d groups of 16 leaves no room for the addend patches", G);
    static bool once_up = [] {
      return hipFuncSetAttribute((const void*)conv1x1_f16x3_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess &&
             hipFuncSetAttribute((const void*)conv1x1_f16x3_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
    }();
    if (!once_up) return egne::fail(EGNE_ERR_LAUNCH, "conv1x1_f16: cannot raise the dynamic LDS limit");
    if (TN == 1)
      hipLaunchKernelGGL((conv1x1_f16x3_kernel<1, true>), dim3((unsigned)gx, ny), dim3(256), lds_up, st, d, (const _Float16*)fhi, (const _Float16*)flo,
                         a_scale, os, G, M, (int)nb);
    else
      hipLaunchKernelGGL((conv1x1_f16x3_kernel<2, true>), dim3((unsigned)gx, ny), dim3(256), lds_up, st, d, (const _Float16*)fhi, (const _Float16*)flo,
                         a_scale, os, G, M, (int)nb);
  } else if (TN == 1)
    hipLaunchKernelGGL((conv1x1_f16x3_kernel<1>), dim3((unsigned)gx, ny), dim3(256), lds, st, d, (const _Float16*)fhi, (const _Float16*)flo,
                       a_scale, os, G, M, (int)nb);
  else
    hipLaunchKernelGGL((conv1x1_f16x3_kernel<2>), dim3((unsigned)gx, ny), dim3(256), lds, st, d, (const _Float16*)fhi, (const _Float16*)flo,
                       a_scale, os, G, M, (int)nb);
  return egne::check_launch("egne_conv1x1_f16x3_fwd");
}

// Transition_down of an eval plan: avg_pool2d(., 2) folded in front of the 1x1 (see conv1x1_pool_f16x3_kernel).  d: the 1x1 with
// H / W = the INPUT size, Ho / Wo = H / 2, W / 2 (floor), every slice WITH its per-(frame, channel) scale / shift (InstanceNorm of the
// consumer, models/RITnet_v2.py:40) and act_in; output [B][Ho][Wo].  Weights as for egne_conv1x1_f16x3_fwd.
extern "C" int egne_conv1x1_pool2_f16x3_fwd(const egne_conv_desc* dp, const void* fhi, const void* flo, float a_scale, float w_scale,
                                            void* stream) {
  EGNE_REQUIRE(dp && fhi && flo, "conv1x1_pool2_f16: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0 && d.ngroups == 1 && d.Ho == d.H / 2 && d.Wo == d.W / 2 &&
               d.Ho > 0 && d.Wo > 0 && d.nseg >= 1 && d.nseg <= EGNE_MAXSEG && !d.residual && !d.post_scale, "conv1x1_pool2_f16: unsupported descriptor");
  EGNE_REQUIRE(d.CoutP % 32 == 0 && d.CoutP <= 96 && d.Cout_store <= d.CoutP && d.Cout_store % 4 == 0 && d.out &&
               ((uintptr_t)d.out & 15) == 0 && d.out_pix_stride % 4 == 0 && d.out_ch_off % 4 == 0 &&
               d.out_ch_off + d.Cout_store <= d.out_pix_stride && (!d.bias || ((uintptr_t)d.bias & 15) == 0), "conv1x1_pool2_f16: output");
  int G = 0;
  for (int s = 0; s < d.nseg; ++s) {
    const egne_seg& g = d.seg[s];
    EGNE_REQUIRE(g.ptr && g.scale && g.shift && ((uintptr_t)g.scale & 15) == 0 && ((uintptr_t)g.shift & 15) == 0 && g.Cp % 8 == 0 &&
                 g.ch_off % 4 == 0 && g.pix_stride % 4 == 0 && ((uintptr_t)g.ptr & 15) == 0 && g.ch_off + g.Cp <= g.pix_stride &&
                 (long long)d.H * d.W * g.pix_stride * 8 < (1ll << 31), "conv1x1_pool2_f16: slice %d", s);
    G += (g.Cp + 15) / 16;
  }
  EGNE_REQUIRE(d.out_pix_stride * 128 < (1ll << 31) && ((uintptr_t)fhi & 15) == 0 && ((uintptr_t)flo & 15) == 0 && a_scale > 0.f && w_scale > 0.f,
               "conv1x1_pool2_f16: strides / weights / scales");
  const long long M = (long long)d.B * d.Ho * d.Wo;
  const long long nb = (M + 31) / 32;
  EGNE_REQUIRE(nb < (1ll << 31), "conv1x1_pool2_f16: too many pixels");
  const int TN = d.CoutP / 32;        // all output channels in one workgroup: the (4x larger) input is read once
  const size_t lds = (size_t)G * TN * 2 * 64 * 8 * sizeof(_Float16);
  EGNE_REQUIRE(lds <= 80 * 1024, "conv1x1_pool2_f16: K = %d groups of 16 does not fit the LDS weight image", G);
  static bool once = [] {
    return hipFuncSetAttribute((const void*)conv1x1_pool_f16x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess &&
           hipFuncSetAttribute((const void*)conv1x1_pool_f16x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess &&
           hipFuncSetAttribute((const void*)conv1x1_pool_f16x3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
  }();
  if (!once) return egne::fail(EGNE_ERR_LAUNCH, "conv1x1_pool2_f16: cannot raise the dynamic LDS limit");
  const int ny = 1;
  long long gx = (nb + 3) / 4;
  const long long cap = 256 * 8;
  if (gx > cap) gx = cap;
  const float os = 1.0f / (a_scale * w_scale);
  hipStream_t st = (hipStream_t)stream;
  if (TN == 1) hipLaunchKernelGGL((conv1x1_pool_f16x3_kernel<1>), dim3((unsigned)gx, (unsigned)ny), dim3(256), lds, st, d, (const _Float16*)fhi, (const _Float16*)flo, a_scale, os, G, M, (int)nb);
  else if (TN == 2) hipLaunchKernelGGL((conv1x1_pool_f16x3_kernel<2>), dim3((unsigned)gx, (unsigned)ny), dim3(256), lds, st, d, (const _Float16*)fhi, (const _Float16*)flo, a_scale, os, G, M, (int)nb);
  else hipLaunchKernelGGL((conv1x1_pool_f16x3_kernel<3>), dim3((unsigned)gx, (unsigned)ny), dim3(256), lds, st, d, (const _Float16*)fhi, (const _Float16*)flo, a_scale, os, G, M, (int)nb);
  return egne::check_launch("egne_conv1x1_pool2_f16x3_fwd");
}
