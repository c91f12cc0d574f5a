// 3x3 / stride 1 / pad 1 convolution with a NARROW output (<= 4 channels) over one fp32 slice of 32 or 64 channels, in EXACT fp32 on the
// vector ALU: the logits layer of ESF-Net (models/RITnet_v2.py:249 `final`: 32 -> 3 classes at full resolution; utils.py:1047 convBlock).
//
// On the matrix path that layer computes a 32-wide output block for 3 channels (90 GFLOP of padding per 64 frames: 369 us on the
// resident-weights kernel, MFMA bound), while the tensor traffic is 0.7 GB (~130 us).  Here a thread owns one output pixel and its <= 4
// accumulators: 9 taps x 32 channels x 3 outputs = 864 fused multiply-adds per pixel -- 62-124 us of VALU time for 64 frames, below
// the memory time.  The (8 + 2) x (32 + 2) halo of a tile is staged once in LDS as fp32 (144-byte pixel pitch: eight consecutive lanes
// cover all 32 banks with their 16-byte reads), the next tile's halo is in flight in registers while the current one is evaluated,
// and the weights are SCALAR operands: every address is a launch constant, so they arrive through the scalar cache (s_load) and feed
// v_fmac directly -- no LDS reads, no vector registers.  One rounding per operation, fp32 accumulation in a fixed order (tap-major,
// then channel): closer to the reference's fp32 convolution than the split-f16 products of the other layers.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// the weight pack read through the CONSTANT address space: uniform addresses then load through the scalar cache (s_load_dwordx16)
// and the values feed v_fmac as scalar operands.  (Through a plain global pointer hipcc issued a vector global_load_dwordx4 and a
// full vmcnt(0) wait per four multiply-adds: the pack might alias the output.)  Nothing in a launch writes the pack.
typedef __attribute__((address_space(4))) const float cfloat;

namespace {

constexpr int TW = 32, TH = 8, HWd = TW + 2, HHd = TH + 2, NPX = HHd * HWd, NT = 256;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// NCH: 32-channel chunks of the input slice.  p.w = [9 taps][32 NCH channels][4 outputs] fp32 (egne_pack_conv3x3_narrow_weight): the
// four outputs of a (tap, channel) are neighbours, so a 64-bit scalar pair is the operand of one packed multiply-add (from the
// [tap][CoutP][Ktot] pack the compiler spent three scalar moves per multiply-add on building those pairs)
template <int NCH>
__global__ __launch_bounds__(NT, NCH == 1 ? 3 : 2)
void conv3x3_narrow_f32_kernel(const egne_conv_desc p, int tiles_x, int tiles_y, int ntiles) {
  constexpr int CP = 32 * NCH, PITCH = CP + 4;                        // floats per halo pixel
  constexpr int NI = (NPX * (CP / 4) + NT - 1) / NT;                    // 16-byte items per thread and tile
  extern __shared__ __attribute__((aligned(16))) float halo[];          // [NPX][PITCH]
  const int tid = threadIdx.x;
  const int py = tid >> 5, px = tid & 31;                               // the thread's pixel of the tile
  const egne_seg sg = p.seg[0];
  const unsigned frame_in = (unsigned)p.H * p.W * (unsigned)sg.pix_stride * 4u;
  constexpr int NO = 4;
  cfloat* const w = (cfloat*)(unsigned long long)p.w;
  const bool vec_out = (p.Cout_store & 3) == 0 && (p.out_ch_off & 3) == 0 && (p.out_pix_stride & 3) == 0 && ((uintptr_t)p.out & 15) == 0;
  const float slope = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  float bias[NO], ps[NO], pt[NO];          // v = act(acc + bias) * post_scale + post_shift (eval BatchNorm behind the activation, utils.py:1049)
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const bool on = o < p.Cout_store;
    bias[o] = (p.bias && on) ? p.bias[o] : 0.f;
    ps[o] = (p.post_scale && on) ? p.post_scale[o] : 1.f;
    pt[o] = (p.post_scale && on) ? p.post_shift[o] : 0.f;
  }

  struct Tile { int b, y0, x0; };
  auto decode = [&](int t) {
    Tile r;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    r.b = t; r.y0 = ty * TH; r.x0 = tx * TW;
    return r;
  };
  // item i of a thread: halo pixel q = (tid + NT i) / (CP/4), 16-byte piece c4 of its channels; everything that does not depend on
  // the tile is computed once (halo coordinates, byte offset relative to the tile's first halo pixel, LDS slot)
  u32x4 st[NI];
  int hyx[NI], rel[NI], slot[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int it = tid + NT * i, q = it / (CP / 4), c4 = it - q * (CP / 4);
    const int hy = q / HWd, hx = q - hy * HWd;
    const bool in = q < NPX && c4 * 4 < sg.Cp;
    hyx[i] = in ? (hy << 16) | hx : 0x7fff7fff;
    rel[i] = ((hy * p.W + hx) * (int)sg.pix_stride + sg.ch_off + c4 * 4) * 4;
    slot[i] = q < NPX ? q * PITCH + c4 * 4 : -1;
  }
  auto issue = [&](const Tile& tl) {
    const __amdgpu_buffer_rsrc_t r = make_rsrc(sg.ptr + (long long)tl.b * p.H * p.W * sg.pix_stride, frame_in);
    const int tbase = (((tl.y0 - 1) * p.W + tl.x0 - 1) * (int)sg.pix_stride) * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned y = (unsigned)(tl.y0 - 1 + (hyx[i] >> 16)), x = (unsigned)(tl.x0 - 1 + (hyx[i] & 0xffff));
      st[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (y < (unsigned)p.H && x < (unsigned)p.W) ? tbase + rel[i] : (int)OOB, 0, 0);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (slot[i] >= 0) *(u32x4*)&halo[slot[i]] = st[i];
  };

  int t = blockIdx.x;
  if (t >= ntiles) return;
  Tile cur = decode(t);
  issue(cur);
  while (true) {
    __syncthreads();                 // the previous tile's reads of the image are done
    stage();
    __syncthreads();
    const int tn = t + gridDim.x;
    const bool more = tn < ntiles;
    Tile nx = cur;
    if (more) { nx = decode(tn); issue(nx); }
    float acc[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] = 0.f;
    // a ROLLED loop over (tap, 16-channel group): 4 reads of 16 bytes, 64 scalar weights (four s_load_dwordx16) and 32 packed
    // multiply-adds per trip, two trips per pass with the next trip's pixel data requested before this trip's arithmetic.
    // (Unrolled, the compiler hoists the invariant scalar loads of all 1152 weights to the top and spills them into vector
    // registers; 8-channel trips with the weights requested a trip ahead as well measured slower: 316 vs 273 us.)
    constexpr int G = CP / 16, NTRIP = 9 * G;
    const float* hrow = &halo[(py * HWd + px) * PITCH];
    auto xptr = [&](int it) { const int tap = it / G, g = it - tap * G; return hrow + ((tap / 3) * HWd + tap % 3) * PITCH + g * 16; };
    f32x4 xa[4], xb[4];
    auto rd = [&](f32x4 (&x)[4], int it) {
      const float* hp = xptr(it);
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) x[c4] = *(const f32x4*)(hp + c4 * 4);
    };
    auto mac = [&](const f32x4 (&x)[4], int it) {
      cfloat* wt = w + it * 64;                    // trip `it` = (tap, 16-channel group): 16 channels x 4 outputs, in the pack's order
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int o = 0; o < NO; ++o) acc[o] = __builtin_fmaf(x[c4][e], wt[(c4 * 4 + e) * 4 + o], acc[o]);
    };
    static_assert(NTRIP % 2 == 0, "two trips per loop pass");
    rd(xa, 0);
#pragma unroll 1
    for (int it = 0; it < NTRIP; it += 2) {
      rd(xb, it + 1);
      mac(xa, it);
      rd(xa, it + 2 < NTRIP ? it + 2 : it);
      mac(xb, it + 1);
    }
    const int y = cur.y0 + py, x = cur.x0 + px;
    if (y < p.H && x < p.W) {
      float* const op = p.out + (long long)cur.b * p.H * p.W * p.out_pix_stride + ((long long)y * p.W + x) * p.out_pix_stride + p.out_ch_off;
      float r[NO];
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const float v = acc[o] + bias[o];
        r[o] = fmaxf(v, v * slope) * ps[o] + pt[o];
      }
      if (vec_out) {                           // stored channels past the fourth are the slice's padding: zeros
        *(f32x4*)op = f32x4{r[0], r[1], r[2], r[3]};
        if (p.Cout_store == 8) *(f32x4*)(op + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int o = 0; o < 8; ++o)
          if (o < p.Cout_store) op[o] = o < NO ? r[o < NO ? o : 0] : 0.f;
      }
    }
    if (!more) break;
    t = tn; cur = nx;
  }
}

bool supported(const egne_conv_desc& d) {
  if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad_h != 1 || d.pad_w != 1 || d.pad_mode != 0 || d.ngroups != 1 || d.dil[0] != 1 ||
      d.nseg != 1 || d.dtype != 0 || d.Ho != d.H || d.Wo != d.W)
    return false;
  const egne_seg& g = d.seg[0];
  if (!g.ptr || g.scale || g.shift || g.act_in != EGNE_ACT_NONE || g.presplit || g.Cp % 4 || g.Cp > 64 || g.Cp < 4 || g.ch_off % 4 || g.pix_stride % 4 ||
      ((uintptr_t)g.ptr & 15) || g.ch_off + g.Cp > g.pix_stride)
    return false;
  if (!d.w || ((uintptr_t)d.w & 63) || !d.out || d.Cout_store < 1 || d.Cout_store > 8 || d.out_ch_off + d.Cout_store > d.out_pix_stride)
    return false;
  if (d.residual || (d.post_scale && !d.post_shift) || d.stats_ws || d.pool_out || d.dyn_scale || d.absmax_out || d.out_split) return false;
  if (d.act != EGNE_ACT_NONE && d.act != EGNE_ACT_RELU && d.act != EGNE_ACT_LEAKY) return false;
  if ((long long)d.H * d.W * g.pix_stride * 4 >= (1ll << 31)) return false;
  return true;
}

}  // namespace

namespace {
// OIHW [Cout <= 4][Cin][3][3] -> [9][CP][4] (CP = Cin rounded up to 32; zeros beyond Cout / Cin)
__global__ void pack_narrow_k(const float* __restrict__ w, int Cout, int Cin, int CP, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * CP * 4) return;
  const int o = i & 3, c = (i >> 2) % CP, tap = (i >> 2) / CP;
  out[i] = (o < Cout && c < Cin) ? w[((long long)o * Cin + c) * 9 + tap] : 0.f;
}
}  // namespace

extern "C" int egne_pack_conv3x3_narrow_weight(const float* w_oihw, int Cout, int Cin, float* out, void* stream) {
  EGNE_REQUIRE(w_oihw && out && Cout >= 1 && Cout <= 4 && Cin >= 1 && Cin <= 64, "pack_conv3x3_narrow: 1..4 outputs over 1..64 channels");
  const int CP = (Cin + 31) / 32 * 32;
  hipLaunchKernelGGL(pack_narrow_k, dim3((9 * CP * 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, CP, out);
  return egne::check_launch("egne_pack_conv3x3_narrow_weight");
}

extern "C" int egne_conv3x3_narrow_supported(const egne_conv_desc* d) { return d && supported(*d) ? 1 : 0; }

// d.w: [9][32 or 64][4] fp32 (egne_pack_conv3x3_narrow_weight), 64-byte aligned; one raw fp32 slice of 4..64 channels, 1..4 stored
// output channels, 5..8 when the slice's padding channels are to be written as zeros (d.Ktot / d.CoutP are not read)
extern "C" int egne_conv3x3_narrow_fwd(const egne_conv_desc* dp, void* stream) {
  EGNE_REQUIRE(dp && supported(*dp), "conv3x3_narrow: descriptor not supported (3x3 / pad 1, one raw fp32 slice of <= 64 channels, <= 4 outputs + padding)");
  const egne_conv_desc& d = *dp;
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH, ntiles = tiles_x * tiles_y * d.B;
  const int nch = (d.seg[0].Cp + 31) / 32;
  const size_t lds = (size_t)NPX * (32 * nch + 4) * sizeof(float);
  int gx = 256 * (nch == 1 ? 3 : 1);                  // 48 KB of LDS per workgroup with 32 channels: three per CU
  if (gx > ntiles) gx = ntiles;
  hipStream_t st = (hipStream_t)stream;
  static bool once = hipFuncSetAttribute((const void*)conv3x3_narrow_f32_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
  if (!once) return egne::fail(EGNE_ERR_LAUNCH, "conv3x3_narrow: %zu bytes of LDS", lds);
  if (nch == 1) hipLaunchKernelGGL(conv3x3_narrow_f32_kernel<1>, dim3(gx), dim3(NT), lds, st, d, tiles_x, tiles_y, ntiles);
  else hipLaunchKernelGGL(conv3x3_narrow_f32_kernel<2>, dim3(gx), dim3(NT), lds, st, d, tiles_x, tiles_y, ntiles);
  return egne::check_launch("egne_conv3x3_narrow_fwd");
}
