// 3x3 (dilation d, stride 1, "same") convolution with an LDS-resident halo tile, fp32 MFMA.
//
// The flat implicit-GEMM kernel (conv_igemm.hip) re-gathers the input pixels from L2 for each of the
// 9 taps and pays two barriers per (tap, 32 channels).  Here a workgroup owns a TH x 32 block of output
// pixels of one frame: per 32-channel chunk it stages the (TH+2d) x (32+2d) input halo ONCE
// (global -> registers -> LDS, zero padding and the optional fused InstanceNorm affine applied on the
// way) and then issues all 9 taps x 4 k-steps of MFMAs from LDS -- a tap is just an address offset into
// the halo.  One MFMA M-tile (32 rows) is one image-row segment of 32 pixels, so the A fragment read is
// the same conflict-free ds_read_b128 pattern (144-B pixel pitch) as in the flat kernel.
// The weights do not go through LDS at all: they are pre-packed in MFMA-fragment order
// [tap][k/8][n/32][lane][4] so that each wave fetches its B fragment with one fully coalesced
// 1-KiB global load (L2/L1 resident, prefetched one k-step ahead).
#include "common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KC = 32;
constexpr int LDK = 36;
constexpr int TW = 32;
constexpr int NI_MAX = 14;  // staged float4 per thread: ceil((TH+2d)*(TW+2d)*8/256) for TH=8, d<=2

__device__ __forceinline__ float act1(float v, int act) {
  if (act == EGNE_ACT_RELU) return fmaxf(v, 0.f);
  if (act == EGNE_ACT_LEAKY) return v > 0.f ? v : 0.01f * v;
  return v;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// All global traffic goes through BUFFER instructions on per-frame resources (see conv_halo_f16.hip): tile-independent
// per-lane byte offsets plus one scalar per tile, out-of-image / padded lanes carry 0x80000000 and read zeros (or drop
// their store) in the buffer unit's range check; activations are max(v, slope*v) -- no branches, no 64-bit address math.
template <int WM, int WN, int D>
__global__ __launch_bounds__(256, (WN <= 2 ? 2 : 1)) void conv3x3_halo_kernel(const egne_conv_desc p, const float* __restrict__ wf,
                                                                              int tiles_x, int tiles_y, int ntiles) {
  constexpr int TH = 4 * WM;
  constexpr int PF = WN >= 4 ? 1 : 4;   // B-fragment prefetch depth in k-steps
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int d = D;
  constexpr int HWd = TW + 2 * d, HHd = TH + 2 * d, npx = HHd * HWd;
  constexpr int nitems = npx * 8;
  constexpr int NI = (nitems + 255) / 256;
  static_assert(HWd >= 32, "one wrap per 32-pixel step");
  const int nt0 = blockIdx.y * WN;          // first 32-wide N tile
  const int NT = p.CoutP >> 5, KT8 = p.Ktot >> 3;
  const egne_seg sg = p.seg[0];
  const int Cp = sg.Cp;
  const int c4 = tid & 7;
  const unsigned frame_in = (unsigned)p.H * p.W * (unsigned)sg.pix_stride * 4u;
  const unsigned frame_out = (unsigned)p.H * p.W * (unsigned)p.out_pix_stride * 4u;
  const unsigned frame_res = (unsigned)p.H * p.W * (unsigned)p.res_pix_stride * 4u;

  int hyx[NI];
  {
    int px = tid >> 3;
    int hy = px / HWd, hx = px - hy * HWd;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      hyx[i] = tid + 256 * i < nitems ? (hy << 16) | hx : 0x7fff7fff;
      hx += 32;
      if (hx >= HWd) { hx -= HWd; ++hy; }
    }
  }
  const int lofs0 = (tid >> 3) * LDK + c4 * 4;
  const int ps4 = (int)sg.pix_stride * 4;

  // The workgroup walks over output tiles (grid-stride) and, inside a tile, over 32-channel chunks.
  // The halo of the NEXT (tile, chunk) is prefetched into registers while the MFMAs of the current one run.
  struct Tile { int b, y0, x0; };
  auto tile_of = [&](int t) {
    Tile r;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    r.b = t / tiles_y; r.y0 = ty * TH; r.x0 = tx * TW;
    return r;
  };

  unsigned goff[NI];
  __amdgpu_buffer_rsrc_t rin = make_rsrc(sg.ptr, 0);
  int stage_b = 0;
  auto map_tile = [&](const Tile& tl) {
    const int ybase = tl.y0 - d, xbase = tl.x0 - d;
    const int tbase = ((ybase * p.W + xbase) * (int)sg.pix_stride + sg.ch_off + c4 * 4) * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int hy = hyx[i] >> 16, hx = hyx[i] & 0xffff;
      const unsigned iy = (unsigned)(ybase + hy), ix = (unsigned)(xbase + hx);
      goff[i] = (iy < (unsigned)p.H && ix < (unsigned)p.W) ? (unsigned)(tbase + (hy * p.W + hx) * ps4) : OOB;
    }
    rin = make_rsrc(sg.ptr + (long long)tl.b * p.H * p.W * sg.pix_stride, frame_in);
    stage_b = tl.b;
  };

  u32x4 st[NI];
  f32x4 st_sc = {1.f, 1.f, 1.f, 1.f}, st_sh = {0.f, 0.f, 0.f, 0.f};
  unsigned st_cmask = 0;
  auto load_chunk = [&](int c0) {
    const bool cok = c0 + c4 * 4 < Cp;
    st_cmask = cok ? 0u : OOB;
    if (sg.scale) {
      st_sc = *(const f32x4*)(cok ? sg.scale + (long long)stage_b * Cp + c0 + c4 * 4 : egne_zero_page);
      st_sh = *(const f32x4*)(cok ? sg.shift + (long long)stage_b * Cp + c0 + c4 * 4 : egne_zero_page);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) st[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)(goff[i] | st_cmask), c0 * 4, 0);
  };
  const float slope_in = sg.act_in == EGNE_ACT_RELU ? 0.f : (sg.act_in == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  const float slope_out = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  auto store_chunk = [&]() {
    if (sg.scale) {   // zero padding AFTER the normalisation
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        f32x4 v = __builtin_bit_cast(f32x4, st[i]) * st_sc + st_sh;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope_in);
        if ((goff[i] | st_cmask) & OOB) v = (f32x4)(0.f);
        st[i] = __builtin_bit_cast(u32x4, v);
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (i < NI - 1 || tid + 256 * i < nitems) *(u32x4*)&lds[lofs0 + i * 32 * LDK] = st[i];
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int n = 0; n < WN; ++n) acc[a][n] = (f32x16)(0.f);

  // fragment-order weights [tap][k/8][n/32][lane][4]: byte offsets, lane part in the VGPR, the rest scalar
  const unsigned wbytes = 9u * (unsigned)p.Ktot * (unsigned)p.CoutP * 4u;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(wf, wbytes);
  const int stride_k8 = NT * 1024, stride_tap = KT8 * NT * 1024;
  const int wlane = lane * 16;
  const int out_step = (int)p.out_pix_stride * 4, res_step = (int)p.res_pix_stride * 4;

  int t = blockIdx.x;
  if (t >= ntiles) return;
  Tile cur = tile_of(t);
  map_tile(cur);
  load_chunk(0);
  int c0 = 0;
  while (true) {
    __syncthreads();           // previous chunk's MFMA reads are done
    store_chunk();
    __syncthreads();
    // prefetch the next (tile, chunk)
    const bool last_chunk = c0 + KC >= Cp;
    const int tn_ = t + gridDim.x;
    if (!last_chunk) {
      load_chunk(c0 + KC);
    } else if (tn_ < ntiles) {
      const Tile nx = tile_of(tn_);
      map_tile(nx);
      load_chunk(0);
    }
    const int rem = Cp - c0;
    const int nk8 = rem >= KC ? 4 : (rem >> 3);
    const int wchunk = nt0 * 1024 + (c0 >> 3) * stride_k8;
    // software pipeline over the 9 x nk8 k-steps: B fragments ride a register ring PF steps ahead
    // (slot = k-step index inside the tap), the A fragment of the next step is read from LDS before
    // the MFMAs of the current one are issued.
    u32x4 bq[PF == 4 ? 4 : 1][WN];
    if (PF == 4) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (s < nk8) {
#pragma unroll
          for (int tn = 0; tn < WN; ++tn) bq[s][tn] = __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, wchunk + s * stride_k8 + tn * 1024, 0);
        }
    } else {
#pragma unroll
      for (int tn = 0; tn < WN; ++tn) bq[0][tn] = __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, wchunk + tn * 1024, 0);
    }
    const float* abase = &lds[(wave * WM * HWd + li) * LDK + lh * 4];
    f32x4 an[WM];
#pragma unroll
    for (int tm = 0; tm < WM; ++tm) an[tm] = *(const f32x4*)(abase + tm * HWd * LDK);
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const float* arow = abase + (ky * d * HWd + kx * d) * LDK;
      const int tap1 = tap + 1;
      const int ky1 = tap1 / 3, kx1 = tap1 - ky1 * 3;
      const float* arow1 = abase + (ky1 * d * HWd + kx1 * d) * LDK;   // only dereferenced when tap < 8
      const int wtap1 = wchunk + tap1 * stride_tap;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < nk8) {
          f32x4 a[WM], bcur[WN];
#pragma unroll
          for (int tm = 0; tm < WM; ++tm) a[tm] = an[tm];
#pragma unroll
          for (int tn = 0; tn < WN; ++tn) bcur[tn] = __builtin_bit_cast(f32x4, bq[PF == 4 ? s : 0][tn]);
          // next A fragment
          if (s + 1 < nk8) {
#pragma unroll
            for (int tm = 0; tm < WM; ++tm) an[tm] = *(const f32x4*)(arow + tm * HWd * LDK + (s + 1) * 8);
          } else if (tap < 8) {
#pragma unroll
            for (int tm = 0; tm < WM; ++tm) an[tm] = *(const f32x4*)(arow1 + tm * HWd * LDK);
          }
          // B prefetch
          if (PF == 4) {
            if (tap < 8) {
#pragma unroll
              for (int tn = 0; tn < WN; ++tn) bq[s][tn] = __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, wtap1 + s * stride_k8 + tn * 1024, 0);
            }
          } else {
            const int wn = (s + 1 < nk8) ? wtap1 - stride_tap + (s + 1) * stride_k8 : wtap1;
            if (s + 1 < nk8 || tap < 8) {
#pragma unroll
              for (int tn = 0; tn < WN; ++tn) bq[0][tn] = __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, wn + tn * 1024, 0);
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < WM; ++tm)
#pragma unroll
              for (int tn = 0; tn < WN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], bcur[tn][j], acc[tm][tn], 0, 0, 0);
        }
      }
    }
    if (!last_chunk) { c0 += KC; continue; }

    // ---- epilogue of tile `cur`: lane holds channel n of 16 pixels x = x_lane + c_r, c_r = (r&3) + 8*(r>>2) ----
    {
      const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out + (long long)cur.b * p.H * p.W * p.out_pix_stride, frame_out);
      const __amdgpu_buffer_rsrc_t rres =
          make_rsrc(p.residual ? p.residual + (long long)cur.b * p.H * p.W * p.res_pix_stride : nullptr, p.residual ? frame_res : 0u);
      const int xl = cur.x0 + 4 * lh;
      const int cmax = xl < p.W ? p.W - xl : 0;
#pragma unroll
      for (int tn = 0; tn < WN; ++tn) {
        const int n = (nt0 + tn) * 32 + li;
        const bool nok = n < p.Cout_store;
        float bv = 0.f, ps = 1.f, pt = 0.f;
        if (p.bias) bv = p.bias[n];
        if (p.post_scale) { ps = p.post_scale[n]; pt = p.post_shift[n]; }
#pragma unroll
        for (int tm = 0; tm < WM; ++tm) {
          const int y = cur.y0 + wave * WM + tm;
          const int cm = (nok && y < p.H) ? cmax : 0;
          const int pix = y * p.W + xl;
          const unsigned o0 = (unsigned)((pix * (int)p.out_pix_stride + p.out_ch_off + n) * 4);
          float rv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = 0.f;
          if (p.residual) {
            const unsigned r0 = (unsigned)((pix * (int)p.res_pix_stride + p.res_ch_off + n) * 4);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int c = (r & 3) + 8 * (r >> 2);
              rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, (int)(c < cm ? r0 + c * res_step : OOB), 0, 0));
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2);
            float v = acc[tm][tn][r] + bv;
            v = fmaxf(v, v * slope_out) * ps + pt + rv[r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, (int)(c < cm ? o0 + c * out_step : OOB), 0, 0);
          }
          acc[tm][tn] = (f32x16)(0.f);
        }
      }
    }
    t = tn_;
    if (t >= ntiles) break;
    cur = tile_of(t);
    c0 = 0;
  }
}

// OIHW -> fragment order [tap][k/8][n/32][h][n%32][4]  (k = 8*kg + 4*h + e)
__global__ void pack_weight_frag_k(const float* __restrict__ w, int Cout, int Cin, int T, const int* __restrict__ kinv,
                                   int CoutP, int Ktot, float* __restrict__ out) {
  const long long total = (long long)T * CoutP * Ktot;
  const int NT = CoutP >> 5;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3);
    const int nn = (int)((i >> 2) & 31);
    const int h = (int)((i >> 7) & 1);
    long long q = i >> 8;
    const int nt = (int)(q % NT); q /= NT;
    const int kg = (int)(q % (Ktot >> 3));
    const int t = (int)(q / (Ktot >> 3));
    const int n = nt * 32 + nn, k = kg * 8 + h * 4 + e;
    const int ci = kinv[k];
    out[i] = (n < Cout && ci >= 0) ? w[((long long)n * Cin + ci) * T + t] : 0.f;
  }
}

template <int WM, int WN, int D>
int launch_halo(const egne_conv_desc& d, const float* wf, hipStream_t st) {
  constexpr int TH = 4 * WM;
  const int dd = D;
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const size_t lds = (size_t)(TH + 2 * dd) * (TW + 2 * dd) * LDK * sizeof(float);
  const int ntiles = tiles_x * tiles_y * d.B, ny = d.CoutP / (32 * WN);
  // persistent-style grid: ~3 workgroups per CU in total, each walks over tiles grid-stride
  int gx = (256 * 3 + ny - 1) / ny;
  if (gx > ntiles) gx = ntiles;
  dim3 grid((unsigned)gx, (unsigned)ny);
  hipLaunchKernelGGL((conv3x3_halo_kernel<WM, WN, D>), grid, dim3(256), lds, st, d, wf, tiles_x, tiles_y, ntiles);
  return egne::check_launch("egne_conv3x3_halo_fwd");
}

}  // namespace

extern "C" int egne_pack_conv_weight_frag(const float* w_oihw, int Cout, int Cin, int kh, int kw, const int32_t* kinv,
                                          int CoutP, int Ktot, float* w_packed, void* stream) {
  EGNE_REQUIRE(w_oihw && kinv && w_packed, "pack_frag: null pointer");
  EGNE_REQUIRE(Cout > 0 && Cin > 0 && CoutP >= Cout && CoutP % 32 == 0 && Ktot >= Cin && Ktot % 8 == 0,
               "pack_frag: bad sizes Cout %d Cin %d CoutP %d Ktot %d", Cout, Cin, CoutP, Ktot);
  long long total = (long long)kh * kw * CoutP * Ktot, g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pack_weight_frag_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, kh * kw,
                     kinv, CoutP, Ktot, w_packed);
  return egne::check_launch("egne_pack_conv_weight_frag");
}

extern "C" int egne_conv3x3_halo_supported(const egne_conv_desc* dp) {
  if (!dp) return 0;
  const egne_conv_desc& d = *dp;
  return d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad_mode == 0 && d.ngroups == 1 && d.nseg == 1 &&
         d.pad_h == 1 && d.pad_w == 1 && d.dil[0] >= 1 && d.dil[0] <= 2 && d.Ho == d.H && d.Wo == d.W;
}

// Same descriptor as egne_conv2d_fwd; `d->w` must point to the FRAGMENT-order pack.
extern "C" int egne_conv3x3_halo_fwd(const egne_conv_desc* dp, void* stream) {
  EGNE_REQUIRE(dp != nullptr, "conv_halo: null descriptor");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(egne_conv3x3_halo_supported(dp), "conv_halo: geometry not supported (3x3, stride 1, same, dilation<=2, one slice)");
  const egne_seg& g = d.seg[0];
  EGNE_REQUIRE(g.ptr && g.Cp > 0 && g.Cp % 8 == 0 && g.Cp == d.Ktot, "conv_halo: bad input slice");
  EGNE_REQUIRE(g.ch_off % 4 == 0 && g.pix_stride % 4 == 0 && g.ch_off + g.Cp <= g.pix_stride && ((uintptr_t)g.ptr & 15) == 0,
               "conv_halo: input slice alignment");
  EGNE_REQUIRE((g.scale == nullptr) == (g.shift == nullptr), "conv_halo: scale/shift mismatch");
  EGNE_REQUIRE(d.CoutP > 0 && d.CoutP % 32 == 0 && d.Cout_store > 0 && d.Cout_store <= d.CoutP, "conv_halo: Cout");
  EGNE_REQUIRE(d.w && d.out && ((uintptr_t)d.w & 15) == 0, "conv_halo: null/unaligned weight or output");
  EGNE_REQUIRE(d.out_ch_off + d.Cout_store <= d.out_pix_stride, "conv_halo: output slice exceeds pixel stride");
  EGNE_REQUIRE((d.post_scale == nullptr) == (d.post_shift == nullptr), "conv_halo: post affine mismatch");
  EGNE_REQUIRE((long long)d.H * d.W * g.pix_stride * 4 < (1ll << 31) && (long long)d.H * d.W * d.out_pix_stride * 4 < (1ll << 31) &&
               (!d.residual || (long long)d.H * d.W * d.res_pix_stride * 4 < (1ll << 31)), "conv_halo: frame too large for 32-bit byte offsets");
  hipStream_t st = (hipStream_t)stream;
  const int c = d.CoutP;
  static const int gen = [] { const char* e = getenv("EGNE_HALO_V"); return e ? atoi(e) : 2; }();   // generation 3 (conv_halo3.hip) is opt-in: measured slower so far
  if (gen >= 3 && egne::halo3_supported(d)) return egne::halo3_launch(d, st);
  if (d.dil[0] == 1) {
    if (c % 128 == 0) return launch_halo<2, 4, 1>(d, d.w, st);
    if (c % 64 == 0) return launch_halo<2, 2, 1>(d, d.w, st);
    return launch_halo<2, 1, 1>(d, d.w, st);
  }
  if (c % 128 == 0) return launch_halo<2, 4, 2>(d, d.w, st);
  if (c % 64 == 0) return launch_halo<2, 2, 2>(d, d.w, st);
  return launch_halo<2, 1, 2>(d, d.w, st);
}
