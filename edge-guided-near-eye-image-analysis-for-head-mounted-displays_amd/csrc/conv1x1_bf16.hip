// 1x1 convolution over up to EGNE_MAXSEG raw bf16 slices on v_mfma_f32_16x16x32_bf16 (fp32 accumulate, bf16 output): the concat-free
// 1x1 convolutions of ESF-Net (models/RITnet_v2.py:59-61 conv21 / conv31, :38-41 Transition_down behind its pooling, :85-86
// conv11 / conv21 of the up blocks) and their merged data gradients in training plans with bf16 activation storage.
//
// Streaming form: the tensor IS the MFMA operand.  With the product transposed (weights as the A operand) a lane's B operand of a
// 32-channel k-step is "8 consecutive channels of one pixel" = ONE 16-byte global load, the four k-groups of a pixel are four
// neighbouring lanes -- 64 contiguous bytes per pixel and instruction, which is all a 32-channel slice has --, so activations never
// touch LDS or the vector ALU on the way in; the weights of the workgroup's output channels stay in LDS for the whole launch.
// What the first version got wrong (measured: 1.8 TB/s on the merged data gradients, behind the exact-fp32 implicit GEMM): it stored
// each lane's 4 result channels as they come out of the MFMA -- 16 contiguous bytes per pixel and instruction, and the same for the
// accumulated residual.  The memory pipeline works per cache line touched, not per byte: here a wave's 32 x CW result tile goes
// through LDS once (fp32) and leaves as 16-byte vectors of 8 channels with the lanes of a pixel side by side: whole 128-byte lines
// per pixel for 64 output channels, residual read the same way.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned OOB = 0x80000000u;
constexpr int KC = 4;              // k-steps (of 32 channels) requested together: 2 x 4 loads of 16 bytes per lane in flight
constexpr int MAXKS = 48;          // k-steps of a launch (table in the kernel arguments)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// per k-step: slice index and first channel of the step inside the slice
struct KTab { unsigned char seg[MAXKS]; unsigned short c0[MAXKS]; };

// NB16: 16-channel output blocks of this workgroup (CW = 16 NB16 channels; blockIdx.y selects the group)
// NW: waves per workgroup (4; 8 where the resident weights leave room for one workgroup per compute unit only: twice the waves share them)
// UP: the "residual" is a HALF-resolution tensor P [B][Ho = H/2][Wo = W/2] whose bilinear x2 up-sampling (F.interpolate(scale_factor=2,
// mode='bilinear', align_corners=False): models/RITnet_v2.py:80-83) is added to the result -- conv11(cat(up(x), skip)) = up(W_up x) +
// W_skip skip, the 1x1 and the interpolation commute -- so training plans never hold the up-sampled operand either (round 5; the
// split-f16 inference kernel has done this since round 4, conv1x1_f16.hip).  Four 16-byte taps per output vector, requested before the
// products like the plain residual; P is a quarter of the output's pixels and stays in L2.
template <int NB16, int NW = 4, bool UP = false>
__global__ __launch_bounds__(64 * NW)
void conv1x1_bf16_kernel(const egne_conv_desc p, const egne_bf16* __restrict__ wfrag, int nks, int nb16_total, KTab tab, long long M) {
  static_assert(NB16 == 2 || NB16 == 4, "the pixel-major store pattern needs 8 CW to divide 64 lanes");
  constexpr int CW = 16 * NB16, LDP = CW + 4;                           // result tile row pitch (floats)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  egne_bf16* const lw = (egne_bf16*)smem;                               // [nks][NB16][64 lanes][8]
  float* const tile = (float*)(smem + (size_t)nks * NB16 * 1024) + (threadIdx.x >> 6) * 32 * LDP;     // this wave's [32 px][LDP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int l15 = lane & 15, kg = lane >> 4;
  const int b0 = blockIdx.y * NB16;
  for (int it = tid; it < nks * NB16 * 64; it += 64 * NW) {             // 16-byte items
    const int l = it & 63, r = it >> 6, j = r % NB16, ks = r / NB16;
    const bool ok = b0 + j < nb16_total;
    const u32x4 v = ok ? *(const u32x4*)(wfrag + (((long long)ks * nb16_total + b0 + j) * 64 + l) * 8) : u32x4{0u, 0u, 0u, 0u};
    *(u32x4*)&lw[(long long)it * 8] = v;
  }
  __syncthreads();
  const float slope = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  egne_bf16* const outp = (egne_bf16*)p.out;
  const egne_bf16* const resp = (const egne_bf16*)p.residual;
  const int cw0 = 16 * b0;                                               // first output channel of this workgroup
  constexpr int G = CW / 8;                                              // lanes per pixel on the way out (8 channels each)
  constexpr int PPI = 64 / G;                                            // pixels per store instruction
  const long long ngroups = (M + 31) / 32;
  const long long wave_id = (long long)blockIdx.x * NW + (tid >> 6), nwaves = (long long)gridDim.x * NW;
  // KC k-steps of a group's operands: 2 x KC loads of 16 bytes per lane.  The NEXT chunk (of this group, or the first of the wave's next
  // group) is requested before the products of the current one (round 5: one chunk at a time left every chunk's latency exposed --
  // 1.0-1.6 TB/s on the 10-16 k-step layers of the wider model), the accumulated residual before the group's first product.
  auto load_chunk = [&](long long gq, int k0, u32x4 (&xq)[KC][2]) {
    const long long m0 = gq * 32;
    const int rows = (int)(M - m0 < 32 ? (M - m0 > 0 ? M - m0 : 0) : 32);      // (past the last group: nothing in bounds, zeros)
#pragma unroll
    for (int u = 0; u < KC; ++u) {
      const int ks = k0 + u;
      const bool on = ks < nks;
      const egne_seg& sg = p.seg[on ? tab.seg[ks] : 0];
      const int c = (on ? tab.c0[ks] : 0) + 8 * kg;
      const __amdgpu_buffer_rsrc_t r = make_rsrc((const egne_bf16*)sg.ptr + (rows ? m0 : 0) * sg.pix_stride, (unsigned)rows * (unsigned)sg.pix_stride * 2u);
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {      // rows past M fall outside the resource and read zeros
        const int off = (on && c < sg.Cp) ? ((16 * ph + l15) * (int)sg.pix_stride + sg.ch_off + c) * 2 : (int)OOB;
        xq[u][ph] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
      }
    }
  };
  // (UP: the whole half-resolution tensor as one resource -- a group's pixels may straddle two frames; the host checks it is < 2 GB)
  const __amdgpu_buffer_rsrc_t rup = make_rsrc(resp, UP ? (unsigned)((long long)p.B * p.Ho * p.Wo * p.res_pix_stride * 2) : 0u);
  u32x4 xa[KC][2], xn[KC][2];
  if (wave_id < ngroups) load_chunk(wave_id, 0, xa);
  for (long long g = wave_id; g < ngroups; g += nwaves) {
    const long long m0 = g * 32;
    const int rows = (int)(M - m0 < 32 ? M - m0 : 32);
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(outp + m0 * p.out_pix_stride, (unsigned)rows * (unsigned)p.out_pix_stride * 2u);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(resp ? resp + m0 * p.res_pix_stride : nullptr, resp ? (unsigned)rows * (unsigned)p.res_pix_stride * 2u : 0u);
    u32x4 rw[UP ? 4 * (32 / PPI) : 32 / PPI];
    float uly[32 / PPI], ulx[32 / PPI];
#pragma unroll
    for (int i = 0; i < 32 / PPI; ++i) {
      const int px = i * PPI + lane / G, n = cw0 + 8 * (lane % G);
      if constexpr (UP) {
        // pixel m0 + px = (b, oy, ox) of the H x W map; taps (y0 | y1) x (x0 | x1) of the Ho x Wo map (ATen area_pixel_compute_source_index,
        // scale 0.5, align_corners = false: max(0.5 (d + 0.5) - 0.5, 0)), weights ly / lx towards the second tap
        const unsigned m = (unsigned)(m0 + px), hw = (unsigned)(p.H * p.W);
        const unsigned b = m / hw, r = m - b * hw, oy = r / (unsigned)p.W, ox = r - oy * (unsigned)p.W;
        float sy = 0.5f * ((float)oy + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
        float sx = 0.5f * ((float)ox + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < p.Ho - 1 ? 1 : 0), x1 = x0 + (x0 < p.Wo - 1 ? 1 : 0);
        uly[i] = sy - (float)y0; ulx[i] = sx - (float)x0;
        const bool ok = n < p.Cout_store && px < rows;
        const int fb = (int)b * p.Ho * p.Wo, co = p.res_ch_off + n;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int yy = (t >> 1) ? y1 : y0, xx = (t & 1) ? x1 : x0;
          rw[4 * i + t] = __builtin_amdgcn_raw_buffer_load_b128(rup, ok ? ((fb + yy * p.Wo + xx) * (int)p.res_pix_stride + co) * 2 : (int)OOB, 0, 0);
        }
      } else {
        rw[i] = resp ? __builtin_amdgcn_raw_buffer_load_b128(rres, n < p.Cout_store ? (px * (int)p.res_pix_stride + p.res_ch_off + n) * 2 : (int)OOB, 0, 0) : u32x4{0u, 0u, 0u, 0u};
      }
    }
    f32x4 acc[2][NB16];
#pragma unroll
    for (int a = 0; a < 2 * NB16; ++a) (&acc[0][0])[a] = (f32x4)(0.f);
    for (int k0 = 0; k0 < nks; k0 += KC) {
      if (k0 + KC < nks) load_chunk(g, k0 + KC, xn);
      else load_chunk(g + nwaves, 0, xn);
#pragma unroll
      for (int u = 0; u < KC; ++u) {
        if (k0 + u < nks) {
#pragma unroll
          for (int j = 0; j < NB16; ++j) {
            const egne_bf16x8 a = *(const egne_bf16x8*)&lw[(((k0 + u) * NB16 + j) * 64 + lane) * 8];
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
              acc[ph][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(egne_bf16x8, xa[u][ph]), acc[ph][j], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < KC; ++u) { xa[u][0] = xn[u][0]; xa[u][1] = xn[u][1]; }
    }
    // lane holds pixel 16 ph + l15, channels 16 j + 4 kg + r: through the wave's LDS tile into pixel-major 8-channel vectors
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int j = 0; j < NB16; ++j) {
        const int n = cw0 + 16 * j + 4 * kg;
        const f32x4 bv = (p.bias && n < p.Cout_store) ? *(const f32x4*)(p.bias + n) : (f32x4)(0.f);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = acc[ph][j][e] + bv[e];
          v[e] = fmaxf(t, t * slope);
        }
        *(f32x4*)&tile[(16 * ph + l15) * LDP + 16 * j + 4 * kg] = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (one wave: its own LDS writes are visible to its own reads after this)
#pragma unroll
    for (int i = 0; i < 32 / PPI; ++i) {
      const int px = i * PPI + lane / G, cg = lane % G;
      const int n = cw0 + 8 * cg;
      const bool nok = n < p.Cout_store;                     // Cout_store is a multiple of 8
      egne_fv<8> v;
      const f32x4 t0 = *(const f32x4*)&tile[px * LDP + 8 * cg], t1 = *(const f32x4*)&tile[px * LDP + 8 * cg + 4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v.v[e] = t0[e]; v.v[4 + e] = t1[e]; }
      if constexpr (UP) {
        const float ly = uly[i], lx = ulx[i], hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a[2], bq[2], cq[2], dq[2];
          a[0] = __builtin_bit_cast(float, rw[4 * i][e] << 16); a[1] = __builtin_bit_cast(float, rw[4 * i][e] & 0xffff0000u);
          bq[0] = __builtin_bit_cast(float, rw[4 * i + 1][e] << 16); bq[1] = __builtin_bit_cast(float, rw[4 * i + 1][e] & 0xffff0000u);
          cq[0] = __builtin_bit_cast(float, rw[4 * i + 2][e] << 16); cq[1] = __builtin_bit_cast(float, rw[4 * i + 2][e] & 0xffff0000u);
          dq[0] = __builtin_bit_cast(float, rw[4 * i + 3][e] << 16); dq[1] = __builtin_bit_cast(float, rw[4 * i + 3][e] & 0xffff0000u);
#pragma unroll
          for (int h = 0; h < 2; ++h) v.v[2 * e + h] += hy * (hx * a[h] + lx * bq[h]) + ly * (hx * cq[h] + lx * dq[h]);      // (the order of egne_upsample2x)
        }
      } else if (resp) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v.v[2 * e] += __builtin_bit_cast(float, rw[i][e] << 16);
          v.v[2 * e + 1] += __builtin_bit_cast(float, rw[i][e] & 0xffff0000u);
        }
      }
      const f32x4 lo = {v.v[0], v.v[1], v.v[2], v.v[3]}, hi = {v.v[4], v.v[5], v.v[6], v.v[7]};
      const egne_bf16x4 l4 = __builtin_convertvector(lo, egne_bf16x4), h4 = __builtin_convertvector(hi, egne_bf16x4);
      const egne_bf16x8 pk = {l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pk), rout,
                                             nok ? (px * (int)p.out_pix_stride + p.out_ch_off + n) * 2 : (int)OOB, 0, 0);
    }
  }
}

// ---- several destinations in one launch (round 5) -----------------------------------------------------------------------------------
// The data gradient of a 1x1 over a would-be torch.cat (models/RITnet_v2.py:59-61,85-86) is one 1x1 per member of the cat, every one of
// them over the SAME input gz: bf16 plans keep each member in a buffer of its own, so the plan ran one launch per member and read gz
// once per launch (three or four times per layer).  Here a wave loads the 32-pixel group's operands ONCE (all k-steps: at most 4 x 2
// registers of 16 bytes) and walks the destinations: per destination its weight fragments (all resident in LDS), up to 64 output
// channels at a time through the wave's LDS tile, out as whole 8-channel vectors with the optional accumulated residual.
//
// Two more things ride on the last writer of a gradient slice (egne_dst.mask_y): the activation mask of the layer whose OUTPUT the
// slice is the gradient of -- gz = g * act'(y), what egne_act_bwd_bias would do in a pass of its own (read g, read y, write g) -- and,
// for ONE destination of the launch (at most 128 channels), the channel sums of gz for that layer's bias gradient: every wave keeps
// them in registers (fp64) over all the pixel groups it walks and writes one row of egne_dst.sums [wave][C] at the end; the
// assignment of groups to waves is fixed by the grid, so egne_group_sums_reduce adds the rows in a fixed order (deterministic).
// (A row per GROUP, reduced by one block, took longer than the pass it replaced: 1.2 M rows per B=256 layer.)
struct DstTab {
  void* ptr[EGNE_MAXDST]; const void* res[EGNE_MAXDST]; const void* mask[EGNE_MAXDST]; float* sums[EGNE_MAXDST];
  int stride[EGNE_MAXDST], off[EGNE_MAXDST], C[EGNE_MAXDST], nb16[EGNE_MAXDST], wofs[EGNE_MAXDST];       // wofs: first 1-KB fragment row of the destination in LDS (per k-step: + nbt * ks)
  int rstride[EGNE_MAXDST], roff[EGNE_MAXDST], mstride[EGNE_MAXDST], moff[EGNE_MAXDST], act[EGNE_MAXDST];
  const egne_bf16* wfrag[EGNE_MAXDST];
  long long rpix[EGNE_MAXDST];        // pixels [0, rpix) accumulate onto the residual, the rest are stored (egne_dst.res_pixels; M: all)
};

// One chunk of NBQ 16-channel blocks (32 or 64 output channels) of destination d for the wave's 32-pixel group: the residual and mask
// vectors of all its pixels are requested BEFORE the MFMAs (their latency hides behind the products and the LDS round trip: with the
// pixel loop's trip count a run-time value every load used to wait for its own reply, 2-4 serialised round trips per chunk -- the
// kernel streamed at 2.6 TB/s), then the products, the wave's LDS tile, and out as 16-byte vectors of 8 channels.
template <int NKS, int NBQ>
__device__ __forceinline__ void multi_chunk(const DstTab& dt, int d, int b0, int nks, int nbt, const egne_bf16* lw, float* tile, const u32x4 (&xb)[NKS][2],
                                            const __amdgpu_buffer_rsrc_t rout, const __amdgpu_buffer_rsrc_t rres, const __amdgpu_buffer_rsrc_t rmsk,
                                            int rows, int lane, bool summing, double (&wsum)[8]) {
  constexpr int LDP = 64 + 4;
  constexpr int G = 2 * NBQ, PPI = 64 / G, IT = 32 / PPI;      // lanes per pixel on the way out (8 channels each), pixels per instruction
  const int l15 = lane & 15, kg = lane >> 4;
  const int cg = lane % G, pl = lane / G;
  const int C = dt.C[d];
  const int n = 16 * b0 + 8 * cg;                              // first channel of the lane's vector inside the destination
  const bool nok = n < C;
  const bool has_res = dt.res[d] != nullptr, has_mask = dt.mask[d] != nullptr;
  u32x4 rw[IT], yw[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int px = i * PPI + pl;
    rw[i] = has_res ? __builtin_amdgcn_raw_buffer_load_b128(rres, nok ? (px * dt.rstride[d] + dt.roff[d] + n) * 2 : (int)OOB, 0, 0) : u32x4{0u, 0u, 0u, 0u};
    yw[i] = has_mask ? __builtin_amdgcn_raw_buffer_load_b128(rmsk, nok ? (px * dt.mstride[d] + dt.moff[d] + n) * 2 : (int)OOB, 0, 0) : u32x4{0u, 0u, 0u, 0u};
  }
  f32x4 acc[2][NBQ];
#pragma unroll
  for (int a = 0; a < 2 * NBQ; ++a) (&acc[0][0])[a] = (f32x4)(0.f);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if (ks >= nks) break;
#pragma unroll
    for (int j = 0; j < NBQ; ++j) {
      const egne_bf16x8 a = *(const egne_bf16x8*)&lw[((long long)(ks * nbt + dt.wofs[d] + b0 + j) * 64 + lane) * 8];
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
        acc[ph][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(egne_bf16x8, xb[ks][ph]), acc[ph][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int j = 0; j < NBQ; ++j) *(f32x4*)&tile[(16 * ph + l15) * LDP + 16 * j + 4 * kg] = acc[ph][j];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (one wave: its own LDS writes are visible to its own reads after this)
  const float slope = dt.act[d] == EGNE_ACT_RELU ? 0.f : (dt.act[d] == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  float csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int px = i * PPI + pl;
    float v[8];
    const f32x4 t0 = *(const f32x4*)&tile[px * LDP + 8 * cg], t1 = *(const f32x4*)&tile[px * LDP + 8 * cg + 4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = t0[e]; v[4 + e] = t1[e]; }
    if (has_res) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] += __builtin_bit_cast(float, rw[i][e] << 16);
        v[2 * e + 1] += __builtin_bit_cast(float, rw[i][e] & 0xffff0000u);
      }
    }
    if (has_mask) {         // gz = g * act'(y): the slice is the gradient of a layer's activated output y
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y0 = __builtin_bit_cast(float, yw[i][e] << 16), y1 = __builtin_bit_cast(float, yw[i][e] & 0xffff0000u);
        v[2 * e] = y0 > 0.f ? v[2 * e] : slope * v[2 * e];
        v[2 * e + 1] = y1 > 0.f ? v[2 * e + 1] : slope * v[2 * e + 1];
      }
    }
    const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
    const egne_bf16x4 l4 = __builtin_convertvector(lo, egne_bf16x4), h4 = __builtin_convertvector(hi, egne_bf16x4);
    const egne_bf16x8 pk = {l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
    const bool pok = nok && px < rows;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pk), rout, pok ? (px * dt.stride[d] + dt.off[d] + n) * 2 : (int)OOB, 0, 0);
    if (summing) {            // sums of what was STORED (bf16-rounded), as a pass over the stored tensor would see it
#pragma unroll
      for (int e = 0; e < 4; ++e) { csum[e] += pok ? (float)l4[e] : 0.f; csum[4 + e] += pok ? (float)h4[e] : 0.f; }
    }
  }
  if (summing) {              // (the group's <= 16 values per lane and channel in fp32, the running sums in fp64)
#pragma unroll
    for (int e = 0; e < 8; ++e) wsum[e] += (double)csum[e];
  }
}

template <int NKS>
__device__ __forceinline__ void multi_load_x(const egne_conv_desc& p, const KTab& tab, int nks, long long m0, long long M, int l15, int kg, u32x4 (&xb)[NKS][2]) {
  const int rows = (int)(M - m0 < 32 ? (M - m0 > 0 ? M - m0 : 0) : 32);      // (a group past the end: nothing in bounds, zeros)
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const bool on = ks < nks;                                   // (NKS = 6 / 8 also serve 5 / 7 k-steps)
    const egne_seg& sg = p.seg[on ? tab.seg[ks] : 0];
    const int c = on ? tab.c0[ks] + 8 * kg : 1 << 20;
    const __amdgpu_buffer_rsrc_t r = make_rsrc((const egne_bf16*)sg.ptr + (rows ? m0 : 0) * sg.pix_stride, (unsigned)rows * (unsigned)sg.pix_stride * 2u);
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int off = c < sg.Cp ? ((16 * ph + l15) * (int)sg.pix_stride + sg.ch_off + c) * 2 : (int)OOB;
      xb[ks][ph] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    }
  }
}

template <int NKS>
__global__ __launch_bounds__(256)
void conv1x1_bf16_multi_kernel(const egne_conv_desc p, DstTab dt, int ndst, int nbt, KTab tab, long long M, int sum_dst, int nks) {
  constexpr int LDP = 64 + 4;
  constexpr bool PREFETCH = NKS <= 4;                                   // the next group's operands requested while this one is worked on
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  egne_bf16* const lw = (egne_bf16*)smem;                               // [nks][nbt][64 lanes][8]
  // (behind the nks k-steps the launch allocated, not the NKS of the instantiation: NKS = 6 / 8 also serve 5 / 7 k-steps)
  float* const tile = (float*)(smem + (size_t)nks * nbt * 1024) + (threadIdx.x >> 6) * 32 * LDP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int l15 = lane & 15, kg = lane >> 4;
  for (int d = 0; d < ndst; ++d) {
    const int nb = dt.nb16[d];
    for (int it = tid; it < nks * nb * 64; it += 256) {
      const int l = it & 63, r = it >> 6, j = r % nb, ks = r / nb;
      *(u32x4*)&lw[((long long)(ks * nbt + dt.wofs[d] + j) * 64 + l) * 8] = *(const u32x4*)(dt.wfrag[d] + (((long long)ks * nb + j) * 64 + l) * 8);
    }
  }
  __syncthreads();
  const long long ngroups = (M + 31) / 32;
  const long long wave_id = (long long)blockIdx.x * 4 + (tid >> 6), nwaves = (long long)gridDim.x * 4;
  double wsum[2][8];                  // destination sum_dst: this wave's sums of the lane's 8 channels, first / second 64-channel chunk
#pragma unroll
  for (int a = 0; a < 16; ++a) (&wsum[0][0])[a] = 0.;
  u32x4 xb[NKS][2], xn[PREFETCH ? NKS : 1][2];
  if (PREFETCH && wave_id < ngroups) multi_load_x<NKS>(p, tab, nks, wave_id * 32, M, l15, kg, xb);
  for (long long g = wave_id; g < ngroups; g += nwaves) {
    const long long m0 = g * 32;
    const int rows = (int)(M - m0 < 32 ? M - m0 : 32);
    if constexpr (PREFETCH) multi_load_x<NKS>(p, tab, nks, (g + nwaves) * 32, M, l15, kg, xn);
    else multi_load_x<NKS>(p, tab, nks, m0, M, l15, kg, xb);
    for (int d = 0; d < ndst; ++d) {
      const __amdgpu_buffer_rsrc_t rout = make_rsrc((egne_bf16*)dt.ptr[d] + m0 * dt.stride[d], (unsigned)rows * (unsigned)dt.stride[d] * 2u);
      // (rows of the group beyond the residual's pixel limit read out of bounds: zero)
      const long long rleft = dt.rpix[d] - m0;
      const int rrows = rleft < rows ? (rleft > 0 ? (int)rleft : 0) : rows;
      const __amdgpu_buffer_rsrc_t rres = make_rsrc(dt.res[d] ? (const egne_bf16*)dt.res[d] + m0 * dt.rstride[d] : nullptr, dt.res[d] ? (unsigned)rrows * (unsigned)dt.rstride[d] * 2u : 0u);
      const __amdgpu_buffer_rsrc_t rmsk = make_rsrc(dt.mask[d] ? (const egne_bf16*)dt.mask[d] + m0 * dt.mstride[d] : nullptr, dt.mask[d] ? (unsigned)rows * (unsigned)dt.mstride[d] * 2u : 0u);
      const bool summing = d == sum_dst;
      for (int b0 = 0; b0 < dt.nb16[d]; b0 += 4) {                     // 64 output channels at a time (nb16 is even)
        if (dt.nb16[d] - b0 >= 4) multi_chunk<NKS, 4>(dt, d, b0, nks, nbt, lw, tile, xb, rout, rres, rmsk, rows, lane, summing, wsum[b0 ? 1 : 0]);
        else multi_chunk<NKS, 2>(dt, d, b0, nks, nbt, lw, tile, xb, rout, rres, rmsk, rows, lane, summing, wsum[b0 ? 1 : 0]);
      }
    }
    if constexpr (PREFETCH) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) { xb[ks][0] = xn[ks][0]; xb[ks][1] = xn[ks][1]; }
    }
  }
  if (sum_dst >= 0) {
    // over the pixel lanes that share a channel vector (lanes cg, cg + G, cg + 2 G, ...: fixed order), then one row [C] per wave
    const int C = dt.C[sum_dst], nb16 = dt.nb16[sum_dst];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const int b0 = 4 * ch;
      if (b0 < nb16) {
        const int nbq = nb16 - b0 < 4 ? nb16 - b0 : 4, G = 2 * nbq;
        for (int o = G; o < 64; o <<= 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) wsum[ch][e] += __shfl_xor(wsum[ch][e], o);
        }
        const int n = 16 * b0 + 8 * (lane % G);
        if (lane < G && n < C) {
          float* w = dt.sums[sum_dst] + wave_id * C + n;
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = (float)wsum[ch][e];
        }
      }
    }
  }
}

// out[c] (+)= sum over rows of sums[row][c], fixed order (egne_dst.sums -> a bias gradient); 32 channels x 32 interleaved row ranges
__global__ __launch_bounds__(1024) void group_sums_reduce_k(const float* __restrict__ sums, long long ngroups, int ld, int C, float* __restrict__ out,
                                                            double* __restrict__ total, int accumulate) {
  __shared__ double part[32][32];
  const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + c;
  double s = 0;
  if (i < C)
    for (long long k = q; k < ngroups; k += 32) s += (double)sums[k * ld + i];
  part[q][c] = s;
  __syncthreads();
  if (q == 0 && i < C) {
    double t = 0;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += part[r][c];
    if (out) out[i] = accumulate ? out[i] + (float)t : (float)t;
    if (total) total[i] = t;
  }
}

// flat fp32 pack [CoutP][Ktot] (egne_pack_conv_weight / egne_pack_conv_weight_dgrad with kh = kw = 1) -> bf16 fragments
// [k-step][CoutP/16][lane = kg*16 + n%16][8]: element j of lane (n, kg) = W[n][kofs(step) + 8 kg + j], zero beyond the slice
__global__ void pack_conv1x1_bf16_k(const float* __restrict__ wflat, int CoutP, int Ktot, int nks, KTab tab, const int* __restrict__ kofs,
                                    const int* __restrict__ segcp, egne_bf16* __restrict__ out) {
  const long long total = (long long)nks * CoutP * 32;
  const int nb16 = CoutP >> 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
    long long q = i >> 9;
    const int cb = (int)(q % nb16), ks = (int)(q / nb16);
    const int n = cb * 16 + (l & 15), c = tab.c0[ks] + 8 * (l >> 4) + j;
    const int s = tab.seg[ks];
    out[i] = (egne_bf16)(c < segcp[s] ? wflat[(long long)n * Ktot + kofs[s] + c] : 0.f);
  }
}

bool make_tab(const egne_conv_desc& d, KTab* tab, int* nks_) {
  int nks = 0;
  for (int s = 0; s < d.nseg; ++s)
    for (int c0 = 0; c0 < d.seg[s].Cp; c0 += 32) {
      if (nks >= MAXKS) return false;
      tab->seg[nks] = (unsigned char)s; tab->c0[nks] = (unsigned short)c0; ++nks;
    }
  *nks_ = nks;
  return true;
}

// 16-channel output blocks per workgroup and the LDS bytes that takes: all of them up to 64 channels, else pieces of 64 or 32
// (2 or 4: the pixel-major store pattern needs 8 CW to divide 64 lanes; CoutP is a multiple of 32, so nb16 is even)
int blocks_per_wg(int nb16) { return nb16 <= 4 ? nb16 : (nb16 % 4 == 0 ? 4 : 2); }
size_t lds_bytes(int nks, int nb, int nw = 4) { return (size_t)nks * nb * 1024 + (size_t)nw * 32 * (16 * nb + 4) * sizeof(float); }
// workgroups of `lds` bytes that fit the 160 KB of a compute unit (1-KB granules)
int wgs_per_cu(size_t lds) { const int n = (int)(163840 / ((lds + 1023) / 1024 * 1024)); return n < 1 ? 1 : (n > 8 ? 8 : n); }

}  // namespace

// number of bf16 elements of the fragment pack of a descriptor (k-steps x CoutP x 32), or -1 if the launch is not supported
extern "C" int64_t egne_conv1x1_bf16_pack_elems(const egne_conv_desc* dp) {
  if (!dp) return -1;
  KTab tab; int nks = 0;
  if (!make_tab(*dp, &tab, &nks)) return -1;
  if (lds_bytes(nks, blocks_per_wg(dp->CoutP / 16)) > 120 * 1024) return -1;
  return (int64_t)nks * dp->CoutP * 32;
}

// wflat: device pointer to the fp32 pack [CoutP][Ktot] of the same descriptor; seginfo: device int32 [2 * nseg] = the K offset of
// every slice, then the padded channel count of every slice
extern "C" int egne_pack_conv1x1_bf16(const egne_conv_desc* dp, const float* wflat, const int32_t* seginfo, void* wfrag, void* stream) {
  EGNE_REQUIRE(dp && wflat && seginfo && wfrag, "pack_conv1x1_bf16: null pointer");
  const egne_conv_desc& d = *dp;
  KTab tab; int nks = 0;
  EGNE_REQUIRE(d.nseg >= 1 && d.nseg <= EGNE_MAXSEG && d.CoutP % 32 == 0 && make_tab(d, &tab, &nks), "pack_conv1x1_bf16: too many k-steps");
  long long total = (long long)nks * d.CoutP * 32, g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(pack_conv1x1_bf16_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, wflat, d.CoutP, d.Ktot, nks, tab, seginfo,
                     seginfo + d.nseg, (egne_bf16*)wfrag);
  return egne::check_launch("egne_pack_conv1x1_bf16");
}

extern "C" int egne_conv1x1_bf16_fwd(const egne_conv_desc* dp, const void* wfrag, void* stream) {
  EGNE_REQUIRE(dp && wfrag, "conv1x1_bf16: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.dtype == 1, "conv1x1_bf16: the descriptor must say bf16 tensors (dtype 1)");
  // Ho = H / 2, Wo = W / 2 with a residual: the residual is a half-resolution tensor to up-sample and add (as egne_conv1x1_f16x3_fwd)
  const bool up = d.residual && 2 * d.Ho == d.H && 2 * d.Wo == d.W;
  EGNE_REQUIRE(d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0 && d.ngroups == 1 && ((d.Ho == d.H && d.Wo == d.W) || up) &&
               d.nseg >= 1 && d.nseg <= EGNE_MAXSEG && !d.post_scale && !d.stats_ws && !d.pool_out && !d.dyn_scale && !d.absmax_out,
               "conv1x1_bf16: geometry / options not supported");
  EGNE_REQUIRE(!up || (d.act == EGNE_ACT_NONE && (long long)d.B * d.Ho * d.Wo * d.res_pix_stride * 2 < (1ll << 31) && (long long)d.B * d.H * d.W < (1ll << 31)),
               "conv1x1_bf16: up-sampled addend needs no activation and a half-resolution tensor below 2 GB");
  int ktot = 0;
  for (int s = 0; s < d.nseg; ++s) {
    const egne_seg& g = d.seg[s];
    EGNE_REQUIRE(g.ptr && !g.scale && !g.shift && g.Cp % 8 == 0 && g.ch_off % 8 == 0 && g.pix_stride % 8 == 0 && ((uintptr_t)g.ptr & 15) == 0 &&
                 g.ch_off + g.Cp <= g.pix_stride && g.pix_stride * 64 < (1ll << 31), "conv1x1_bf16: slice %d (raw, 16-byte groups of 8 channels)", s);
    ktot += g.Cp;
  }
  EGNE_REQUIRE(ktot == d.Ktot && d.CoutP % 32 == 0 && d.Cout_store % 8 == 0 && d.Cout_store <= d.CoutP && d.out && ((uintptr_t)d.out & 15) == 0 &&
               d.out_pix_stride % 8 == 0 && d.out_ch_off % 8 == 0 && d.out_ch_off + d.Cout_store <= d.out_pix_stride &&
               d.out_pix_stride * 64 < (1ll << 31) && (!d.bias || ((uintptr_t)d.bias & 15) == 0), "conv1x1_bf16: output (16-byte groups of 8 channels)");
  EGNE_REQUIRE(!d.residual || (((uintptr_t)d.residual & 15) == 0 && d.res_pix_stride % 8 == 0 && d.res_ch_off % 8 == 0 && d.res_pix_stride * 64 < (1ll << 31)),
               "conv1x1_bf16: residual alignment");
  KTab tab; int nks = 0;
  EGNE_REQUIRE(make_tab(d, &tab, &nks), "conv1x1_bf16: more than %d k-steps", MAXKS);
  const int nb16 = d.CoutP / 16, nb = blocks_per_wg(nb16);
  EGNE_REQUIRE(lds_bytes(nks, nb) <= 120 * 1024, "conv1x1_bf16: %zu bytes of LDS per workgroup exceed the budget", lds_bytes(nks, nb));
  // waves per compute unit: as many workgroups of four waves as fit its LDS next to each other, or -- where the resident weights leave
  // room for one only -- eight waves around one copy of them (round 5: the 11-16 k-step layers of the wider model ran one wave per SIMD)
  const int nw = (nb == 4 && 8 * wgs_per_cu(lds_bytes(nks, nb, 8)) > 4 * wgs_per_cu(lds_bytes(nks, nb, 4)) && lds_bytes(nks, nb, 8) <= 156 * 1024) ? 8 : 4;
  const size_t lds = lds_bytes(nks, nb, nw);
  const long long M = (long long)d.B * d.H * d.W;
  const int gy = (nb16 + nb - 1) / nb;
  long long gx = ((M + 31) / 32 + nw - 1) / nw;
  int per_cu = wgs_per_cu(lds);
  const int max_waves = (nb == 2 && !up) ? 12 : 8;            // (142 / 170-240 registers: three / two waves per SIMD)
  if (per_cu * nw > max_waves) per_cu = max_waves / nw;
  long long cap = 256ll * per_cu / gy;                        // workgroups: as many as stay resident
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  hipStream_t st = (hipStream_t)stream;
  auto go = [&](auto kern) -> int {
    const bool raised = egne::raise_lds((const void*)kern, 156 * 1024);
    if (!raised) return egne::fail(EGNE_ERR_LAUNCH, "conv1x1_bf16: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy), dim3(64 * nw), lds, st, d, (const egne_bf16*)wfrag, nks, nb16, tab, M);
    return egne::check_launch("egne_conv1x1_bf16_fwd");
  };
  if (up) {
    if (nb == 2) return go(conv1x1_bf16_kernel<2, 4, true>);
    if (nw == 8) return go(conv1x1_bf16_kernel<4, 8, true>);
    return go(conv1x1_bf16_kernel<4, 4, true>);
  }
  if (nb == 2) return go(conv1x1_bf16_kernel<2>);
  if (nw == 8) return go(conv1x1_bf16_kernel<4, 8>);
  return go(conv1x1_bf16_kernel<4>);
}

// Several 1x1 convolutions over the SAME input slices in one launch, each with its own weights (fragments of egne_pack_conv1x1_bf16 for a
// descriptor with that destination's CoutP) and destination slice: the per-member data gradients of a 1x1 over a would-be torch.cat.
// d: input slices (seg[]), B, H, W, Ktot; its output fields are ignored.  Returns EGNE_ERR_ARG if the shapes do not fit (the caller
// falls back to one egne_conv1x1_bf16_fwd per destination): more than 8 k-steps, or more weight fragments than fit LDS.
extern "C" int egne_conv1x1_bf16_multi_supported(const egne_conv_desc* dp, int ndst, const egne_dst* dsts) {
  if (!dp || !dsts || ndst < 1 || ndst > EGNE_MAXDST) return 0;
  KTab tab; int nks = 0;
  if (!make_tab(*dp, &tab, &nks) || nks > 8) return 0;
  int nbt = 0;
  for (int i = 0; i < ndst; ++i) {
    if (dsts[i].CoutP % 32 || dsts[i].C % 8 || dsts[i].C > dsts[i].CoutP) return 0;
    nbt += dsts[i].CoutP / 16;
  }
  return (size_t)nks * nbt * 1024 + (size_t)4 * 32 * 68 * sizeof(float) <= 120 * 1024;
}

static long long multi_grid(const egne_conv_desc& d, int nks, int nbt) {
  const size_t lds = (size_t)nks * nbt * 1024 + (size_t)4 * 32 * 68 * sizeof(float);
  const long long M = (long long)d.B * d.H * d.W;
  long long gx = ((M + 31) / 32 + 3) / 4;
  int per_cu = wgs_per_cu(lds);
  if (per_cu > 2) per_cu = 2;                                 // (138-208 registers + accumulators: two waves per SIMD)
  const long long cap = 256ll * per_cu;
  return gx > cap ? cap : gx;
}

// rows of egne_dst.sums this launch writes (one per wave): what the caller allocates (x C floats) and hands to egne_group_sums_reduce
extern "C" int64_t egne_conv1x1_bf16_multi_waves(const egne_conv_desc* dp, int ndst, const egne_dst* dsts) {
  if (!egne_conv1x1_bf16_multi_supported(dp, ndst, dsts)) return -1;
  KTab tab; int nks = 0;
  make_tab(*dp, &tab, &nks);
  int nbt = 0;
  for (int i = 0; i < ndst; ++i) nbt += dsts[i].CoutP / 16;
  return multi_grid(*dp, nks, nbt) * 4;
}

extern "C" int egne_conv1x1_bf16_multi_fwd(const egne_conv_desc* dp, int ndst, const egne_dst* dsts, void* stream) {
  EGNE_REQUIRE(dp && dsts && egne_conv1x1_bf16_multi_supported(dp, ndst, dsts), "conv1x1_bf16_multi: shapes not supported");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.dtype == 1 && d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0 && d.ngroups == 1 && d.nseg >= 1 && d.nseg <= EGNE_MAXSEG,
               "conv1x1_bf16_multi: bf16 1x1 descriptors only");
  for (int s = 0; s < d.nseg; ++s) {
    const egne_seg& g = d.seg[s];
    EGNE_REQUIRE(g.ptr && !g.scale && !g.shift && g.Cp % 8 == 0 && g.ch_off % 8 == 0 && g.pix_stride % 8 == 0 && ((uintptr_t)g.ptr & 15) == 0 &&
                 g.ch_off + g.Cp <= g.pix_stride && g.pix_stride * 64 < (1ll << 31), "conv1x1_bf16_multi: slice %d (raw, 16-byte groups of 8 channels)", s);
  }
  KTab tab; int nks = 0;
  make_tab(d, &tab, &nks);
  DstTab dt{};
  int nbt = 0, sum_dst = -1;
  for (int i = 0; i < ndst; ++i) {
    const egne_dst& q = dsts[i];
    if (q.sums) {
      EGNE_REQUIRE(sum_dst < 0 && q.CoutP <= 128, "conv1x1_bf16_multi: channel sums for ONE destination of at most 128 channels per launch");
      sum_dst = i;
    }
    EGNE_REQUIRE(q.out && q.wfrag && ((uintptr_t)q.out & 15) == 0 && q.out_pix_stride % 8 == 0 && q.out_ch_off % 8 == 0 && q.out_ch_off + q.C <= q.out_pix_stride &&
                 q.out_pix_stride * 64 < (1ll << 31) && ((uintptr_t)q.wfrag & 15) == 0, "conv1x1_bf16_multi: destination %d", i);
    EGNE_REQUIRE(!q.residual || (((uintptr_t)q.residual & 15) == 0 && q.res_pix_stride % 8 == 0 && q.res_ch_off % 8 == 0 && q.res_pix_stride * 64 < (1ll << 31)),
                 "conv1x1_bf16_multi: residual of destination %d", i);
    EGNE_REQUIRE(!q.mask_y || (((uintptr_t)q.mask_y & 15) == 0 && q.mask_pix_stride % 8 == 0 && q.mask_ch_off % 8 == 0 && q.mask_pix_stride * 64 < (1ll << 31)),
                 "conv1x1_bf16_multi: mask tensor of destination %d", i);
    EGNE_REQUIRE(!q.sums || ((uintptr_t)q.sums & 15) == 0, "conv1x1_bf16_multi: sums of destination %d", i);
    dt.ptr[i] = q.out; dt.res[i] = q.residual; dt.mask[i] = q.mask_y; dt.sums[i] = q.sums;
    dt.stride[i] = (int)q.out_pix_stride; dt.off[i] = q.out_ch_off; dt.C[i] = q.C; dt.nb16[i] = q.CoutP / 16; dt.wofs[i] = nbt;
    dt.rstride[i] = (int)q.res_pix_stride; dt.roff[i] = q.res_ch_off; dt.mstride[i] = (int)q.mask_pix_stride; dt.moff[i] = q.mask_ch_off; dt.act[i] = q.act;
    dt.wfrag[i] = (const egne_bf16*)q.wfrag;
    EGNE_REQUIRE(q.res_pixels >= 0 && (q.residual || q.res_pixels == 0), "conv1x1_bf16_multi: res_pixels of destination %d", i);
    dt.rpix[i] = q.res_pixels > 0 ? (long long)q.res_pixels : (long long)d.B * d.H * d.W;
    nbt += q.CoutP / 16;
  }
  const size_t lds = (size_t)nks * nbt * 1024 + (size_t)4 * 32 * 68 * sizeof(float);
  const long long M = (long long)d.B * d.H * d.W;
  const long long gx = multi_grid(d, nks, nbt);
  hipStream_t st = (hipStream_t)stream;
  auto go = [&](auto kern) -> int {
    const bool raised = egne::raise_lds((const void*)kern, 120 * 1024);
    if (!raised) return egne::fail(EGNE_ERR_LAUNCH, "conv1x1_bf16_multi: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(256), lds, st, d, dt, ndst, nbt, tab, M, sum_dst, nks);
    return egne::check_launch("egne_conv1x1_bf16_multi_fwd");
  };
  switch (nks) {
    case 1: return go(conv1x1_bf16_multi_kernel<1>);
    case 2: return go(conv1x1_bf16_multi_kernel<2>);
    case 3: return go(conv1x1_bf16_multi_kernel<3>);
    case 4: return go(conv1x1_bf16_multi_kernel<4>);
    case 5: case 6: return go(conv1x1_bf16_multi_kernel<6>);
    default: return go(conv1x1_bf16_multi_kernel<8>);
  }
}

// out[c] (+)= sum over rows of sums[row * ld + c], c < C (egne_dst.sums of a launch: nrows = egne_conv1x1_bf16_multi_waves, ld = that
// destination's C); total (optional): the same sums as doubles
extern "C" int egne_group_sums_reduce(const float* sums, int64_t nrows, int ld, int C, float* out, double* total, int accumulate, void* stream) {
  EGNE_REQUIRE(sums && (out || total) && nrows > 0 && C > 0 && ld >= C, "group_sums_reduce: bad arguments");
  hipLaunchKernelGGL(group_sums_reduce_k, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, sums, (long long)nrows, ld, C, out, total, accumulate);
  return egne::check_launch("egne_group_sums_reduce");
}
