// 3x3 halo convolution, generation 3: weights AND halo in LDS, one workgroup per CU, deferred epilogue.
//
// PMC read of the previous halo kernel on a 32->32 layer: MFMA pipe 43 % busy, waves 40 % of their
// cycles in s_waitcnt.  Cause: the B fragments came from global memory inside the MFMA loop, and vmcnt
// retires in order -- the first B wait of a tile also waited for the NEXT tile's halo prefetch (HBM
// latency), every tile.  Here nothing in the MFMA loop touches vmcnt:
//   * the weight fragments of the current 32-channel chunk sit in LDS in MFMA-fragment order
//     ([tap][k8][n-tile][lane][4]; one conflict-free ds_read_b128 per lane); for layers with <= 32
//     input channels they are loaded once per workgroup and stay resident over all its tiles;
//   * the (TH+2d) x (32+2d) input halo of the NEXT (tile, chunk) is fetched global -> registers while
//     the 9 taps x 4 k-steps of the current one run, and written to LDS between two barriers;
//   * the epilogue of tile t (bias / activation / affine / residual / store) is deferred: its
//     16*WM*WN store instructions are spread over the MFMA steps of tile t+1, so the matrix pipe never
//     waits for the store phase.
// LDS = 49 KB halo + 36*WN KB weights  ->  one 256-thread workgroup per CU (1 wave per SIMD), which is
// what the register budget of the deferred epilogue wants anyway.
#include "common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KC = 32, LDK = 36, TW = 32;

__device__ __forceinline__ float act1(float v, int act) {
  if (act == EGNE_ACT_RELU) return fmaxf(v, 0.f);
  if (act == EGNE_ACT_LEAKY) return v > 0.f ? v : 0.01f * v;
  return v;
}

template <int WM, int WN, int D>
__global__ __launch_bounds__(256, 1) void conv3x3_halo3_kernel(const egne_conv_desc p, const float* __restrict__ wf,
                                                               int tiles_x, int tiles_y, int ntiles) {
  constexpr int TH = 4 * WM;
  constexpr int d = D;
  constexpr int HWd = TW + 2 * d, HHd = TH + 2 * d, npx = HHd * HWd;
  constexpr int nitems = npx * 8;
  constexpr int NI = (nitems + 255) / 256;        // staged halo float4 per thread
  constexpr int NB = 9 * WN;                      // staged weight float4 per thread (9*4*WN*64 / 256)
  constexpr int NSTORE = 16 * WM * WN;            // epilogue store instructions per tile
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Al = lds;
  float* Bl = lds + npx * LDK;                    // [tap*4 + s][tn][lane][4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nt0 = blockIdx.y * WN;
  const int NT = p.CoutP >> 5, KT8 = p.Ktot >> 3;
  const egne_seg sg = p.seg[0];
  const int Cp = sg.Cp;
  const int c4 = tid & 7;
  const bool single_chunk = Cp <= KC;

  struct Tile { int b, y0, x0; };
  auto tile_of = [&](int t) {
    Tile r;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    r.b = t / tiles_y; r.y0 = ty * TH; r.x0 = tx * TW;
    return r;
  };

  int goff[NI];
  const float* src = sg.ptr;
  int stage_b = 0;
  auto map_tile = [&](const Tile& tl) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int item = tid + 256 * i;
      goff[i] = -1;
      if (item < nitems) {
        const int px = item >> 3;
        const int hy = px / HWd, hx = px - hy * HWd;
        const int iy = tl.y0 - d + hy, ix = tl.x0 - d + hx;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
          goff[i] = (int)((((long long)iy * p.W + ix) * sg.pix_stride) + sg.ch_off + c4 * 4);
      }
    }
    src = sg.ptr + (long long)tl.b * p.H * p.W * sg.pix_stride;
    stage_b = tl.b;
  };

  f32x4 sta[NI], stb[NB];
  // load phase: unconditional loads only (invalid lanes read the zero page); the fused affine /
  // activation and the zero padding are applied in the store phase, after the MFMAs of the current chunk.
  f32x4 st_sc = {1.f, 1.f, 1.f, 1.f}, st_sh = {0.f, 0.f, 0.f, 0.f};
  bool st_cok = true;
  auto load_A = [&](int c0) {
    const bool cok = c0 + c4 * 4 < Cp;
    st_cok = cok;
    if (sg.scale) {
      const float* sp = cok ? sg.scale + (long long)stage_b * Cp + c0 + c4 * 4 : egne_zero_page;
      const float* hp = cok ? sg.shift + (long long)stage_b * Cp + c0 + c4 * 4 : egne_zero_page;
      st_sc = *(const f32x4*)sp;
      st_sh = *(const f32x4*)hp;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float* q = (goff[i] >= 0 && cok) ? src + goff[i] + c0 : egne_zero_page;
      sta[i] = *(const f32x4*)q;
    }
  };
  auto store_A = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int item = tid + 256 * i;
      if (item < nitems) {
        f32x4 v = sta[i];
        if (sg.scale) {
          v = v * st_sc + st_sh;
          if (sg.act_in == EGNE_ACT_LEAKY) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
          } else if (sg.act_in == EGNE_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (!(goff[i] >= 0 && st_cok)) v = (f32x4)(0.f);   // zero padding AFTER the normalisation
        }
        *(f32x4*)&Al[(item >> 3) * LDK + c4 * 4] = v;
      }
    }
  };
  // weight fragments of chunk c0: item = tid + 256*j -> (block = item/64 = (tap*4+s)*WN + tn, lane = item%64)
  auto load_B = [&](int c0) {
    const int nk8 = (Cp - c0) >= KC ? 4 : ((Cp - c0) >> 3);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int item = tid + 256 * j;
      const int blk = item >> 6, ln = item & 63;
      const int tn = blk % WN, ts = blk / WN;
      const int s = ts & 3, tap = ts >> 2;
      const float* q = (s < nk8) ? wf + (((long long)tap * KT8 + (c0 >> 3) + s) * NT + nt0 + tn) * 256 + ln * 4 : egne_zero_page;
      stb[j] = *(const f32x4*)q;
    }
  };
  auto store_B = [&]() {
#pragma unroll
    for (int j = 0; j < NB; ++j) *(f32x4*)&Bl[(tid + 256 * j) * 4] = stb[j];
  };

  f32x16 acc[WM][WN], pacc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int n = 0; n < WN; ++n) { acc[a][n] = (f32x16)(0.f); pacc[a][n] = (f32x16)(0.f); }

  // per-lane epilogue constants (column n of each N tile)
  float bv[WN], ps[WN], pt[WN];
  bool nok[WN];
#pragma unroll
  for (int tn = 0; tn < WN; ++tn) {
    const int n = (nt0 + tn) * 32 + li;
    nok[tn] = n < p.Cout_store;
    bv[tn] = p.bias ? p.bias[n] : 0.f;
    ps[tn] = p.post_scale ? p.post_scale[n] : 1.f;
    pt[tn] = p.post_scale ? p.post_shift[n] : 0.f;
  }
  // store #q (0..NSTORE-1) of the tile `tl` held in `A`
  auto emit_store = [&](const f32x16 (&A)[WM][WN], const Tile& tl, int q) {
    const int r = q & 15, tm = (q >> 4) % WM, tn = (q >> 4) / WM;
    const int y = tl.y0 + wave * WM + tm;
    const int x = tl.x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (nok[tn] && y < p.H && x < p.W) {
      const long long m = ((long long)tl.b * p.H + y) * p.W + x;
      const int n = (nt0 + tn) * 32 + li;
      float v = act1(A[tm][tn][r] + bv[tn], p.act);
      if (p.post_scale) v = v * ps[tn] + pt[tn];
      if (p.residual) v += p.residual[m * p.res_pix_stride + p.res_ch_off + n];
      p.out[m * p.out_pix_stride + p.out_ch_off + n] = v;
    }
  };

  int t = blockIdx.x;
  if (t >= ntiles) return;
  Tile cur = tile_of(t), prev = cur;
  bool have_prev = false;
  map_tile(cur);
  load_A(0);
  load_B(0);
  bool b_pending = true;       // stb holds weights that still have to go to LDS
  int c0 = 0;
  const float* abase = &Al[(wave * WM * HWd + li) * LDK + lh * 4];
  const float* bbase = &Bl[lane * 4];

  while (true) {
    __syncthreads();           // all waves are done reading the previous chunk from LDS
    store_A();
    if (b_pending) store_B();
    __syncthreads();
    const bool last_chunk = c0 + KC >= Cp;
    const int tnext = t + gridDim.x;
    b_pending = false;
    if (!last_chunk) {
      load_A(c0 + KC);
      load_B(c0 + KC);
      b_pending = true;
    } else if (tnext < ntiles) {
      const Tile nx = tile_of(tnext);
      map_tile(nx);
      load_A(0);
      if (!single_chunk) { load_B(0); b_pending = true; }
    }
    const int rem = Cp - c0;
    const int nk8 = rem >= KC ? 4 : (rem >> 3);
    const bool defer = have_prev && c0 == 0;   // spread the previous tile's stores over this chunk's steps

    f32x4 an[WM];
#pragma unroll
    for (int tm = 0; tm < WM; ++tm) an[tm] = *(const f32x4*)(abase + tm * HWd * LDK);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const float* arow = abase + (ky * d * HWd + kx * d) * LDK;
      const int ky1 = (tap + 1) / 3, kx1 = (tap + 1) - ky1 * 3;
      const float* arow1 = abase + (ky1 * d * HWd + kx1 * d) * LDK;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < nk8) {
          f32x4 a[WM], b[WN];
#pragma unroll
          for (int tm = 0; tm < WM; ++tm) a[tm] = an[tm];
#pragma unroll
          for (int tn = 0; tn < WN; ++tn) b[tn] = *(const f32x4*)(bbase + ((tap * 4 + s) * WN + tn) * 256);
          if (s + 1 < nk8) {
#pragma unroll
            for (int tm = 0; tm < WM; ++tm) an[tm] = *(const f32x4*)(arow + tm * HWd * LDK + (s + 1) * 8);
          } else if (tap < 8) {
#pragma unroll
            for (int tm = 0; tm < WM; ++tm) an[tm] = *(const f32x4*)(arow1 + tm * HWd * LDK);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < WM; ++tm)
#pragma unroll
              for (int tn = 0; tn < WN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
          // deferred epilogue: stores of the previous tile ride under this step's MFMAs
          if (defer) {
            constexpr int PER = (NSTORE + 35) / 36;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
              const int q = (tap * 4 + s) * PER + u;
              if (q < NSTORE) emit_store(pacc, prev, q);
            }
          }
        }
      }
    }
    if (!last_chunk) { c0 += KC; continue; }

    // tile finished: its accumulators become the deferred set
#pragma unroll
    for (int tm = 0; tm < WM; ++tm)
#pragma unroll
      for (int tn = 0; tn < WN; ++tn) { pacc[tm][tn] = acc[tm][tn]; acc[tm][tn] = (f32x16)(0.f); }
    prev = cur;
    have_prev = true;
    t = tnext;
    if (t >= ntiles) break;
    cur = tile_of(t);
    c0 = 0;
  }
  // last tile of this workgroup: nothing left to hide the stores under
  if (have_prev) {
#pragma unroll
    for (int q = 0; q < NSTORE; ++q) emit_store(pacc, prev, q);
  }
}

template <int WM, int WN, int D>
int launch3(const egne_conv_desc& d, hipStream_t st) {
  constexpr int TH = 4 * WM;
  const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
  const size_t lds = ((size_t)(TH + 2 * D) * (TW + 2 * D) * LDK + (size_t)9 * 4 * WN * 256) * sizeof(float);
  const int ntiles = tiles_x * tiles_y * d.B, ny = d.CoutP / (32 * WN);
  int gx = (256 + ny - 1) / ny;      // one workgroup per CU in total
  if (gx > ntiles) gx = ntiles;
  if (gx < 1) gx = 1;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)conv3x3_halo3_kernel<WM, WN, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_halo3_kernel<WM, WN, D>), dim3(gx, ny), dim3(256), lds, st, d, d.w, tiles_x, tiles_y, ntiles);
  return egne::check_launch("egne_conv3x3_halo_fwd(v3)");
}

}  // namespace

namespace egne {
// Partial first chunks (Cp < 32 with deferred stores) are not handled by the deferred epilogue; the
// dispatcher keeps such layers (Cp = 8, 16, 24) on the generation-2 kernel.
bool halo3_supported(const egne_conv_desc& d) {
  static const int wn2 = [] { const char* e = getenv("EGNE_HALO3_WN2"); return e ? atoi(e) : 0; }();
  return d.seg[0].Cp >= 32 && (d.CoutP % 64 != 0 || wn2);
}

int halo3_launch(const egne_conv_desc& d, hipStream_t st) {
  const int c = d.CoutP;
  if (d.dil[0] == 1) {
    if (c % 64 == 0) return launch3<2, 2, 1>(d, st);
    return launch3<2, 1, 1>(d, st);
  }
  if (c % 64 == 0) return launch3<2, 2, 2>(d, st);
  return launch3<2, 1, 2>(d, st);
}
}  // namespace egne
