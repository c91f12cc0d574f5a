// BDCN side-output path (bdcn_new.py:118-191): per-stage 1x1 "down" convs + score heads, then the
// transposed-conv upsampling / crop / cascades / fuse / sigmoid tail.  HBM-bound byte work.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MSC = 32;   // MSBlock output channels
constexpr int DNC = 21;   // "down" conv channels

// 8 lanes per pixel (4 channels each).  Everything is linear, so each lane carries its partial of
// the two scores and the 8 partials are summed with three xor-shuffles at the end.
__global__ __launch_bounds__(256) void stage_scores_k(const float* m0, const float* m1, const float* m2, int nblk,
                                                      long long pix_stride, long long npix,
                                                      const float* __restrict__ wd, const float* __restrict__ bd,
                                                      const float* __restrict__ ws, const float* __restrict__ bs,
                                                      const float* __restrict__ ws1, const float* __restrict__ bs1,
                                                      float* __restrict__ s, float* __restrict__ s1) {
  __shared__ float lw[3 * DNC * MSC];
  __shared__ float lws[2 * DNC];
  __shared__ float lb[2];
  for (int i = threadIdx.x; i < nblk * DNC * MSC; i += blockDim.x) lw[i] = wd[i];
  if (threadIdx.x < DNC) { lws[threadIdx.x] = ws[threadIdx.x]; lws[DNC + threadIdx.x] = ws1[threadIdx.x]; }
  if (threadIdx.x == 0) {
    // constant part: heads applied to the summed down-conv biases, plus the head biases
    float a = 0.f, b = 0.f;
    for (int j = 0; j < DNC; ++j) {
      float t = 0.f;
      for (int k = 0; k < nblk; ++k) t += bd[k * DNC + j];
      a += ws[j] * t; b += ws1[j] * t;
    }
    lb[0] = a + bs[0]; lb[1] = b + bs1[0];
  }
  __syncthreads();
  const int v = threadIdx.x & 7;
  const float* ms[3] = {m0, m1, m2};
  const long long stride = (long long)gridDim.x * (blockDim.x >> 3);
  for (long long p = (long long)blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3); p < npix + ((blockDim.x >> 3) - 1);
       p += stride) {
    const bool ok = p < npix;
    float d[DNC];
#pragma unroll
    for (int j = 0; j < DNC; ++j) d[j] = 0.f;
    for (int k = 0; k < nblk; ++k) {
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (ok) x = *(const f32x4*)(ms[k] + p * pix_stride + v * 4);
      const float* w = lw + k * DNC * MSC + v * 4;
#pragma unroll
      for (int j = 0; j < DNC; ++j)
        d[j] += w[j * MSC] * x[0] + w[j * MSC + 1] * x[1] + w[j * MSC + 2] * x[2] + w[j * MSC + 3] * x[3];
    }
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < DNC; ++j) { a += lws[j] * d[j]; b += lws[DNC + j] * d[j]; }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (ok && v == 0) { s[p] = a + lb[0]; s1[p] = b + lb[1]; }
  }
}

// ConvTranspose2d(1,1,k=2*stride,stride) of the two score maps of a stage ([h,w] each, same taps) sampled at cropped output pixel
// (y,x): the <= 2 x 2 contributing inputs and their taps are found once for both maps; kw_ = the k x k table (in LDS).
__device__ __forceinline__ void convT2_at(const float* __restrict__ ma, const float* __restrict__ mb, int h, int w,
                                          const float* kw_, int stride, int crop, int y, int x, float& ra, float& rb) {
  const int k = 2 * stride;
  const int yy = y + crop, xx = x + crop;
  float acc_a = 0.f, acc_b = 0.f;
  // in[iy] contributes with tap ky = yy - iy*stride in [0,k)
  // (stride is 2 / 4 / 8 in bdcn_new.py:91-97: a shift; the general case keeps the division)
  const bool p2 = (stride & (stride - 1)) == 0;
  const int sh = 31 - __builtin_clz((unsigned)stride);
  const int iy1 = p2 ? yy >> sh : yy / stride, ix1 = p2 ? xx >> sh : xx / stride;
#pragma unroll
  for (int a = 1; a >= 0; --a) {
    const int iy = iy1 - a;
    if (iy < 0 || iy >= h) continue;
    const int ky = yy - iy * stride;
    if (ky >= k) continue;
#pragma unroll
    for (int b = 1; b >= 0; --b) {
      const int ix = ix1 - b;
      if (ix < 0 || ix >= w) continue;
      const int kx = xx - ix * stride;
      if (kx >= k) continue;
      const float t = kw_[ky * k + kx];
      acc_a += ma[iy * w + ix] * t;
      acc_b += mb[iy * w + ix] * t;
    }
  }
  ra = acc_a; rb = acc_b;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// grid (ceil(W / 64), ceil(H / 4), B): a thread per pixel of a 64 x 4 block (320-wide frames fill every lane; 256 x 1 blocks left
// 37 % of them idle), the four upsampler tables (k x k, k = 2 stride <= 32) staged in LDS once per block, both score maps of a
// stage interpolated with one set of indices.  No 64-bit division per element.
constexpr int UPMAX = 32 * 32;
__global__ __launch_bounds__(256) void bdcn_tail_k(const egne_bdcn_tail_desc d) {
  __shared__ float lup[4][UPMAX];
#pragma unroll
  for (int k = 1; k < 5; ++k) {
    const int kk = 4 * d.stride[k] * d.stride[k];
    for (int i = threadIdx.x; i < kk; i += 256) lup[k - 1][i] = d.up[k][i];
  }
  __syncthreads();
  const long long HW = (long long)d.H * d.W;
  const int b = blockIdx.z, y = blockIdx.y * 4 + (threadIdx.x >> 6), x = blockIdx.x * 64 + (threadIdx.x & 63);
  if (x < d.W && y < d.H) {
    const long long i = (long long)b * HW + (long long)y * d.W + x;
    float sa[5], sb[5];
    sa[0] = d.s[0][i];
    sb[0] = d.s1[0][i];
#pragma unroll
    for (int k = 1; k < 5; ++k) {
      const long long off = (long long)b * d.h[k] * d.w[k];
      convT2_at(d.s[k] + off, d.s1[k] + off, d.h[k], d.w[k], lup[k - 1], d.stride[k], d.crop[k], y, x, sa[k], sb[k]);
    }
    // cascades, same association as bdcn_new.py:167-176
    float p[10];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float t = sa[k];
      for (int j = k - 1; j >= 0; --j) t += sa[j];
      p[k] = t;
      t = sb[k];
      for (int j = k + 1; j < 5; ++j) t += sb[j];
      p[5 + k] = t;
    }
    float f = 0.f;
#pragma unroll
    for (int k = 0; k < 10; ++k) f += d.fuse_w[k] * p[k];
    f += d.fuse_b[0];
#pragma unroll
    for (int k = 0; k < 10; ++k)
      if (d.out[k]) d.out[k][i] = sigmoidf_(p[k]);
    if (d.out[10]) {
      float e = sigmoidf_(f);
      if (d.edge_thres == 1 && e >= 0.1f) e = 1.f;
      d.out[10][i] = e;
    }
  }
}

// OIHW -> [tap][CoutP][Ktot]; kinv[k] = input channel stored at padded K position k, or -1
__global__ void pack_weight_k(const float* __restrict__ w, int Cout, int Cin, int T, const int* __restrict__ kinv,
                              int CoutP, int Ktot, float* __restrict__ out) {
  const long long total = (long long)T * CoutP * Ktot;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Ktot);
    long long q = i / Ktot;
    const int n = (int)(q % CoutP);
    const int t = (int)(q / CoutP);
    const int ci = kinv[k];
    out[i] = (n < Cout && ci >= 0) ? w[((long long)n * Cin + ci) * T + t] : 0.f;
  }
}

}  // namespace

extern "C" int egne_bdcn_stage_scores(const float* const* ms, int nblk, int64_t ms_pix_stride, int64_t npix,
                                      const float* wd, const float* bd, const float* ws, const float* bs,
                                      const float* ws1, const float* bs1, float* s, float* s1, void* stream) {
  EGNE_REQUIRE(ms && nblk >= 1 && nblk <= 3, "stage_scores: nblk %d", nblk);
  for (int k = 0; k < nblk; ++k)
    EGNE_REQUIRE(ms[k] && ((uintptr_t)ms[k] & 15) == 0, "stage_scores: input %d null/unaligned", k);
  EGNE_REQUIRE(ms_pix_stride >= MSC && ms_pix_stride % 4 == 0 && npix > 0, "stage_scores: bad stride/npix");
  EGNE_REQUIRE(wd && bd && ws && bs && ws1 && bs1 && s && s1, "stage_scores: null pointer");
  long long g = (npix + 31) / 32;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(stage_scores_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, ms[0], nblk > 1 ? ms[1] : ms[0],
                     nblk > 2 ? ms[2] : ms[0], nblk, (long long)ms_pix_stride, (long long)npix, wd, bd, ws, bs, ws1, bs1,
                     s, s1);
  return egne::check_launch("egne_bdcn_stage_scores");
}

extern "C" int egne_bdcn_tail(const egne_bdcn_tail_desc* dp, void* stream) {
  EGNE_REQUIRE(dp, "bdcn_tail: null descriptor");
  const egne_bdcn_tail_desc& d = *dp;
  EGNE_REQUIRE(d.B > 0 && d.H > 0 && d.W > 0 && d.fuse_w && d.fuse_b, "bdcn_tail: bad arguments");
  EGNE_REQUIRE(d.h[0] == d.H && d.w[0] == d.W, "bdcn_tail: stage 1 must be at input resolution");
  for (int k = 0; k < 5; ++k) {
    EGNE_REQUIRE(d.s[k] && d.s1[k], "bdcn_tail: null score map %d", k);
    if (k == 0) continue;
    EGNE_REQUIRE(d.up[k] && d.stride[k] >= 1 && d.stride[k] <= 16 && d.crop[k] >= 0, "bdcn_tail: stage %d upsampler (stride 1..16)", k);
    // the cropped window must lie inside the transposed-conv output (bdcn_new.py:7-12 crop assert)
    const int oh = (d.h[k] - 1) * d.stride[k] + 2 * d.stride[k], ow = (d.w[k] - 1) * d.stride[k] + 2 * d.stride[k];
    EGNE_REQUIRE(d.crop[k] + d.H <= oh && d.crop[k] + d.W <= ow, "bdcn_tail: stage %d upsampled %dx%d smaller than crop+%dx%d", k, oh, ow, d.H, d.W);
  }
  EGNE_REQUIRE(d.H <= 4 * 65535 && d.B <= 65535, "bdcn_tail: grid limits");
  hipLaunchKernelGGL(bdcn_tail_k, dim3((unsigned)((d.W + 63) / 64), (unsigned)((d.H + 3) / 4), (unsigned)d.B), dim3(256), 0, (hipStream_t)stream, d);
  return egne::check_launch("egne_bdcn_tail");
}

extern "C" int egne_pack_conv_weight(const float* w_oihw, int Cout, int Cin, int kh, int kw, const int32_t* kinv,
                                     int CoutP, int Ktot, float* w_packed, void* stream) {
  EGNE_REQUIRE(w_oihw && kinv && w_packed, "pack: null pointer");
  EGNE_REQUIRE(Cout > 0 && Cin > 0 && kh > 0 && kw > 0 && CoutP >= Cout && CoutP % 32 == 0 && Ktot >= Cin && Ktot % 8 == 0,
               "pack: bad sizes Cout %d Cin %d CoutP %d Ktot %d", Cout, Cin, CoutP, Ktot);
  long long total = (long long)kh * kw * CoutP * Ktot, g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pack_weight_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, kh * kw, kinv,
                     CoutP, Ktot, w_packed);
  return egne::check_launch("egne_pack_conv_weight");
}
