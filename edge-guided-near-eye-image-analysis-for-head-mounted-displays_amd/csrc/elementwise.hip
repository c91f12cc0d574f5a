// HBM-bound helper kernels on NHWC channel slices: normalisation statistics, pooling, bilinear
// upsampling, layout conversion, small activations.  All loads/stores are 16 B per lane along the
// channel axis (slices are 16-B aligned and padded to multiples of 8 channels).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// max |x| over a channel slice, as the bit pattern of the float (non-negative floats order like unsigned integers; a NaN
// anywhere yields a pattern above +inf).  Calibration of the split-f16 pre-scale: the caller zeroes out_bits first.
__global__ __launch_bounds__(256) void absmax_k(const float* __restrict__ x, long long pix_stride, int ch_off, int Cp,
                                                long long npix, unsigned* __restrict__ out_bits) {
  const int nv = Cp >> 2;
  const long long total = npix * nv;
  unsigned m = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / nv;
    const int c = (int)(i - p * nv) * 4;
    const f32x4 v = *(const f32x4*)(x + p * pix_stride + ch_off + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned b = __float_as_uint(v[e]) & 0x7fffffffu;
      m = b > m ? b : m;
    }
  }
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned t = (unsigned)__shfl_xor((int)m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(out_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out_bits, m);
}

// the same over a slice held as f16 (egne_conv_desc.out_split = 2): the pattern written is that of the value AS A FLOAT
__global__ __launch_bounds__(256) void absmax_f16_k(const _Float16* __restrict__ x, long long pix_stride, int ch_off, int Cp,
                                                    long long npix, unsigned* __restrict__ out_bits) {
  typedef _Float16 h8_ __attribute__((ext_vector_type(8)));
  const int nv = Cp >> 3;
  const long long total = npix * nv;
  unsigned m = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / nv;
    const int c = (int)(i - p * nv) * 8;
    const h8_ v = *(const h8_*)(x + p * pix_stride + ch_off + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned b = __float_as_uint((float)v[e]) & 0x7fffffffu;
      m = b > m ? b : m;
    }
  }
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned t = (unsigned)__shfl_xor((int)m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(out_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out_bits, m);
}

// ------------------------------------------------------------------------------------------------
// InstanceNorm / BatchNorm statistics: two deterministic stages, fp64 accumulation.
//   stage 1: grid (nchunk, ceil(Cp/32), Bn); block = 8 channel-vectors x 32 pixel rows
//   stage 2: one thread per (n, c)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void norm_stats_partial(const T* __restrict__ x, long long pix_stride,
                                                          int ch_off, int Cp, long long npix_per_n, int nchunk,
                                                          double* __restrict__ ws) {
  constexpr int N = egne_vt<T>::N, CV = 32 / N, ROWS = 256 / CV;      // 16-byte vectors: 8 x 32 rows (fp32) or 4 x 64 rows (bf16)
  const int chunk = blockIdx.x, cg = blockIdx.y, n = blockIdx.z;
  const int v = threadIdx.x % CV, row = threadIdx.x / CV;
  const int c = cg * 32 + v * N;
  const long long per = (npix_per_n + nchunk - 1) / nchunk;
  const long long p0 = (long long)chunk * per;
  const long long p1 = p0 + per < npix_per_n ? p0 + per : npix_per_n;
  double s[N], q[N];
#pragma unroll
  for (int e = 0; e < N; ++e) { s[e] = 0; q[e] = 0; }
  if (c < Cp) {
    const T* base = x + (long long)n * npix_per_n * pix_stride + ch_off + c;
    for (long long p = p0 + row; p < p1; p += 4 * ROWS) {        // four rows per trip: loads issued together (same summation order)
      egne_fv<N> t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = p + ROWS * u < p1 ? ldv(base + (p + ROWS * u) * pix_stride) : fv_fill<N>(0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < N; ++e) { s[e] += t[u].v[e]; q[e] += (double)t[u].v[e] * t[u].v[e]; }
    }
  }
  __shared__ double sh[ROWS][32][2];
#pragma unroll
  for (int e = 0; e < N; ++e) { sh[row][v * N + e][0] = s[e]; sh[row][v * N + e][1] = q[e]; }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc_ = threadIdx.x >> 1, w_ = threadIdx.x & 1;
    double a = 0;
    for (int r = 0; r < ROWS; ++r) a += sh[r][cc_][w_];
    const int cc = cg * 32 + cc_;
    if (cc < Cp) ws[(((long long)n * nchunk + chunk) * Cp + cc) * 2 + w_] = a;
  }
}

// stage 2: one block per (32 channels, n): 32 partial-sum streams per channel (independent loads in flight), combined in a fixed order.
// (One thread per (n, c) walking its nchunk partial sums serially was 45-60 us per call for batch statistics -- 32 threads, thousands
// of dependent loads each -- and 31 such calls per training step.)
__global__ __launch_bounds__(1024) void norm_stats_final(const double* __restrict__ ws, int Cp, int Bn, int nchunk, long long npix_per_n,
                                                        float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                        float* __restrict__ mean_out, float* __restrict__ var_out) {
  const int n = blockIdx.y, cl = threadIdx.x & 31, c = blockIdx.x * 32 + cl, kg = threadIdx.x >> 5;
  double s = 0, q = 0;
  if (c < Cp) {
    const double* w = ws + ((long long)n * nchunk * Cp + c) * 2;
    for (int k = kg; k < nchunk; k += 32) {
      const double2 v = *(const double2*)(w + (long long)k * Cp * 2);
      s += v.x; q += v.y;
    }
  }
  __shared__ double sh[32][32][2];
  sh[kg][cl][0] = s; sh[kg][cl][1] = q;
  __syncthreads();
  if (kg == 0 && c < Cp) {
    for (int g = 1; g < 32; ++g) { s += sh[g][cl][0]; q += sh[g][cl][1]; }
    const int i = n * Cp + c;
    const double mean = s / (double)npix_per_n;
    double var = q / (double)npix_per_n - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    scale[i] = rstd;
    shift[i] = (float)(-mean) * rstd;
    if (mean_out) mean_out[i] = (float)mean;
    if (var_out) var_out[i] = (float)var;
  }
}

// Finish of the statistics whose partial sums came out of a convolution's epilogue: one block per (32 channels, frame),
// 32 partial-sum streams per channel (independent loads in flight), fixed combination order.
__global__ __launch_bounds__(1024) void norm_stats_finish_k(const double* __restrict__ ws, int Cp, int nchunk, long long npix_per_n,
                                                           float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                           float* __restrict__ mean_out = nullptr, float* __restrict__ var_out = nullptr,
                                                           long long row_stride = 1) {
  const int n = blockIdx.y, c = blockIdx.x * 32 + (threadIdx.x & 31), kg = threadIdx.x >> 5;       // 32 partial-sum streams
  double s = 0, q = 0;
  if (c < Cp) {
    const double* w = ws + ((long long)n * nchunk * row_stride * Cp + c) * 2;
    for (int k = kg; k < nchunk; k += 32) {
      const double2 v = *(const double2*)(w + (long long)k * row_stride * Cp * 2);
      s += v.x; q += v.y;
    }
  }
  __shared__ double sh[32][32][2];
  sh[kg][threadIdx.x & 31][0] = s; sh[kg][threadIdx.x & 31][1] = q;
  __syncthreads();
  if (kg == 0 && c < Cp) {
    for (int g = 1; g < 32; ++g) { s += sh[g][threadIdx.x][0]; q += sh[g][threadIdx.x][1]; }
    const double mean = s / (double)npix_per_n;
    double var = q / (double)npix_per_n - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    scale[n * Cp + c] = rstd;
    shift[n * Cp + c] = (float)(-mean) * rstd;
    if (mean_out) { mean_out[n * Cp + c] = (float)mean; var_out[n * Cp + c] = (float)var; }
  }
}

template <typename T>
__global__ void affine_k(const T* x, long long pix_stride, int ch_off, T* y, long long ys, int yo, int Cp,
                         long long npix, const float* __restrict__ scale, const float* __restrict__ shift) {
  const int nv = Cp >> 2;
  const long long total = npix * nv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / nv;
    const int c = (int)(i - p * nv) * 4;
    const f32x4 v = ld4(x + p * pix_stride + ch_off + c);
    const f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
    st4(y + p * ys + yo + c, v * sc + sh);
  }
}

// y = act(x * scale[c] + shift[c]): BatchNorm followed by its activation (models/deepvog_pytorch.py:36-41)
__global__ void affine_act_k(const float* x, long long pix_stride, int ch_off, float* y, long long ys, int yo, int Cp,
                             long long npix, const float* __restrict__ scale, const float* __restrict__ shift, float slope) {
  const int nv = Cp >> 2;
  const long long total = npix * nv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / nv;
    const int c = (int)(i - p * nv) * 4;
    f32x4 v = ld4(x + p * pix_stride + ch_off + c) * *(const f32x4*)(scale + c) + *(const f32x4*)(shift + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
    st4(y + p * ys + yo + c, v);
  }
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void avgpool2_k(const T* __restrict__ x, long long xs, int xo, T* __restrict__ y, long long ys,
                           int yo, int B, int H, int W, int Cp) {
  const int Ho = H >> 1, Wo = W >> 1, nv = Cp >> 2;
  const long long total = (long long)B * Ho * Wo * nv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % nv) * 4;
    long long p = i / nv;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const T* s = x + (((long long)b * H + 2 * oy) * W + 2 * ox) * xs + xo + c;
    const f32x4 a = ld4(s), bb = ld4(s + xs);
    const f32x4 cc = ld4(s + (long long)W * xs), d = ld4(s + (long long)W * xs + xs);
    f32x4 r = ((a + bb) + cc) + d;  // row-major accumulation order of ATen's avg_pool2d
    r = r * 0.25f;
    st4(y + (((long long)b * Ho + oy) * Wo + ox) * ys + yo + c, r);
  }
}

// avgpool2(act(x*scale+shift)): Transition_down with the pooling commuted in front of its 1x1 conv
// (both are linear, models/RITnet_v2.py:40-44), which quarters the conv's work and traffic.
template <typename T>
__global__ void norm_act_pool2_k(const T* __restrict__ x, long long xs, int xo, const float* __restrict__ scale,
                                 const float* __restrict__ shift, int act, T* __restrict__ y, long long ys, int yo,
                                 int B, int H, int W, int Cp) {
  constexpr int N = egne_vt<T>::N;
  const int Ho = H >> 1, Wo = W >> 1, nv = Cp / N;
  const long long total = (long long)B * Ho * Wo * nv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % nv) * N;
    long long p = i / nv;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const egne_fv<N> sc = ldf<N>(scale + (long long)b * Cp + c), sh = ldf<N>(shift + (long long)b * Cp + c);
    const T* s = x + (((long long)b * H + 2 * oy) * W + 2 * ox) * xs + xo + c;
    const egne_fv<N> v[4] = {ldv(s), ldv(s + xs), ldv(s + (long long)W * xs), ldv(s + (long long)W * xs + xs)};
    egne_fv<N> r = fv_fill<N>(0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < N; ++e) {
        const float t = v[k].v[e] * sc.v[e] + sh.v[e];
        r.v[e] += act == EGNE_ACT_LEAKY ? (t > 0.f ? t : 0.01f * t) : (act == EGNE_ACT_RELU ? fmaxf(t, 0.f) : t);
      }
#pragma unroll
    for (int e = 0; e < N; ++e) r.v[e] *= 0.25f;
    stv(y + (((long long)b * Ho + oy) * Wo + ox) * ys + yo + c, r);
  }
}

// grid = (tiles over Wo * Cp/4, Ho, B): no 64-bit div / mod per element (they cost more issue time than the 4 loads)
__global__ void maxpool2_k(const float* __restrict__ x, long long xs, int xo, float* __restrict__ y, long long ys,
                           int yo, int B, int H, int W, int Ho, int Wo, int stride, int Cp) {
  const unsigned nv = Cp >> 2;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)Wo * nv) return;
  const int ox = (int)(t / nv), c = (int)(t - ox * nv) * 4;
  const int b = blockIdx.z;
  const int x0 = ox * stride, x1 = x0 + 1 < W ? x0 + 1 : x0;  // ceil_mode: clipped window
  const float* s = x + ((long long)b * H * W) * xs + xo + c;
  // 4 output rows per thread: 16 independent loads in flight
  f32x4 v[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = blockIdx.y * 4 + j;
    const int y0 = oy < Ho ? oy * stride : 0, y1 = y0 + 1 < H ? y0 + 1 : y0;
    v[j][0] = *(const f32x4*)(s + ((long long)y0 * W + x0) * xs);
    v[j][1] = *(const f32x4*)(s + ((long long)y0 * W + x1) * xs);
    v[j][2] = *(const f32x4*)(s + ((long long)y1 * W + x0) * xs);
    v[j][3] = *(const f32x4*)(s + ((long long)y1 * W + x1) * xs);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = blockIdx.y * 4 + j;
    if (oy < Ho) {
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = fmaxf(fmaxf(v[j][0][e], v[j][1][e]), fmaxf(v[j][2][e], v[j][3][e]));
      *(f32x4*)(y + (((long long)b * Ho + oy) * Wo + ox) * ys + yo + c) = r;
    }
  }
}

// the same pooling over a slice held as f16 (egne_conv_desc.out_split = 2 of its producer; the maximum commutes with the storage scale):
// eight channels (16 bytes) per thread
__global__ void maxpool2_f16_k(const _Float16* __restrict__ x, long long xs, int xo, _Float16* __restrict__ y, long long ys,
                               int yo, int B, int H, int W, int Ho, int Wo, int stride, int Cp) {
  typedef _Float16 h8_ __attribute__((ext_vector_type(8)));
  const unsigned nv = Cp >> 3;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)Wo * nv) return;
  const int ox = (int)(t / nv), c = (int)(t - ox * nv) * 8;
  const int b = blockIdx.z;
  const int x0 = ox * stride, x1 = x0 + 1 < W ? x0 + 1 : x0;
  const _Float16* s = x + ((long long)b * H * W) * xs + xo + c;
  h8_ v[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = blockIdx.y * 4 + j;
    const int y0 = oy < Ho ? oy * stride : 0, y1 = y0 + 1 < H ? y0 + 1 : y0;
    v[j][0] = *(const h8_*)(s + ((long long)y0 * W + x0) * xs);
    v[j][1] = *(const h8_*)(s + ((long long)y0 * W + x1) * xs);
    v[j][2] = *(const h8_*)(s + ((long long)y1 * W + x0) * xs);
    v[j][3] = *(const h8_*)(s + ((long long)y1 * W + x1) * xs);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = blockIdx.y * 4 + j;
    if (oy < Ho) {
      h8_ r;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float m = fmaxf(fmaxf((float)v[j][0][e], (float)v[j][1][e]), fmaxf((float)v[j][2][e], (float)v[j][3][e]));
        r[e] = (_Float16)m;
      }
      *(h8_*)(y + (((long long)b * Ho + oy) * Wo + ox) * ys + yo + c) = r;
    }
  }
}

template <typename T>
__global__ void upsample2x_k(const T* __restrict__ x, long long xs, int xo, T* __restrict__ y, long long ys,
                             int yo, int B, int H, int W, int Cp) {
  const int Ho = 2 * H, Wo = 2 * W;
  const unsigned nv = Cp >> 2;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)Wo * nv) return;
  const int ox = (int)(t / nv), c = (int)(t - ox * nv) * 4;
  const int oy = blockIdx.y, b = blockIdx.z;
  // ATen area_pixel_compute_source_index(scale=0.5, align_corners=false): max(0.5*(d+0.5)-0.5, 0)
  float sy = 0.5f * (oy + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
  float sx = 0.5f * (ox + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
  const T* s = x + ((long long)b * H * W) * xs + xo + c;
  const f32x4 a = ld4(s + ((long long)y0 * W + x0) * xs);
  const f32x4 bb = ld4(s + ((long long)y0 * W + x1) * xs);
  const f32x4 cc = ld4(s + ((long long)y1 * W + x0) * xs);
  const f32x4 d = ld4(s + ((long long)y1 * W + x1) * xs);
  const f32x4 r = hy * (hx * a + lx * bb) + ly * (hx * cc + lx * d);
  st4(y + (((long long)b * Ho + oy) * Wo + ox) * ys + yo + c, r);
}

// F.interpolate(scale_factor=2, mode='nearest') (models/RITnet_v1.py:89): y[oy][ox] = x[oy >> 1][ox >> 1]
template <typename T>
__global__ void upsample2x_nearest_k(const T* __restrict__ x, long long xs, int xo, T* __restrict__ y, long long ys, int yo, int B, int H, int W, int Cp) {
  const int Wo = 2 * W, Ho = 2 * H;
  const unsigned nv = Cp >> 2;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)Wo * nv) return;
  const int ox = (int)(t / nv), c = (int)(t - ox * nv) * 4;
  const int oy = blockIdx.y, b = blockIdx.z;
  st4(y + (((long long)b * Ho + oy) * Wo + ox) * ys + yo + c, ld4(x + (((long long)b * H + (oy >> 1)) * W + (ox >> 1)) * xs + xo + c));
}

template <typename T>
__global__ void nchw_to_nhwc_k(const float* __restrict__ x, int B, int C, int H, int W, T* __restrict__ y,
                               long long ys, int yo, int Cp) {
  const long long HW = (long long)H * W, total = (long long)B * HW;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
    const long long b = p / HW, r = p - b * HW;
    T* dst = y + p * ys + yo;
    for (int c = 0; c < Cp; ++c) st1(dst + c, c < C ? x[(b * C + c) * HW + r] : 0.f);
  }
}

__global__ void nhwc_to_nchw_k(const float* __restrict__ x, long long xs, int xo, int B, int C, int H, int W,
                               float* __restrict__ y) {
  const long long HW = (long long)H * W, total = (long long)B * HW;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
    const long long b = p / HW, r = p - b * HW;
    const float* src = x + p * xs + xo;
    for (int c = 0; c < C; ++c) y[(b * C + c) * HW + r] = src[c];
  }
}

template <typename T>
__global__ void ellipse_head_act_k(T* __restrict__ x, int B, int ld) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 10) return;
  const int b = i / 10, j = i - b * 10, k = j % 5;
  float v = ld1(x + (long long)b * ld + j);
  if (k < 2) v = tanhf(v);
  else if (k < 4) v = 1.f / (1.f + expf(-v));
  st1(x + (long long)b * ld + j, v);
}

template <typename T>
__global__ void selu_k(T* __restrict__ x, long long n) {
  const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = ld1(x + i);
    st1(x + i, scale * (v > 0.f ? v : alpha * (expf(v) - 1.f)));
  }
}

// one block per sample; thread = channel; fp32 pairwise-ish (per-thread serial over <= a few hundred px)
template <typename T, typename TO>
__global__ void spatial_mean_k(const T* __restrict__ x, long long pix_stride, int ch_off, int C, int HW,
                               TO* __restrict__ out) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const T* s = x + (long long)b * HW * pix_stride + ch_off + c;
    double a = 0;
    int p = 0;
    for (; p + 8 <= HW; p += 8) {          // eight loads in flight, summed in pixel order (one dependent load at a time took 76 us)
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = ld1(s + (long long)(p + u) * pix_stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) a += t[u];
    }
    for (; p < HW; ++p) a += ld1(s + (long long)p * pix_stride);
    st1(out + (long long)b * C + c, (float)(a / HW));
  }
}

inline int grid_for(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

template <typename T> inline bool vec_ok(long long stride, int off, int Cp) {     // 16-byte vectors of T
  constexpr int N = egne_vt<T>::N;
  return stride % N == 0 && off % N == 0 && Cp % N == 0;
}
inline bool slice_ok(const void* p, long long stride, int off, int Cp) {     // (4-element vectors: 16 bytes of fp32, 8 of bf16)
  return p && ((uintptr_t)p & 15) == 0 && stride % 4 == 0 && off % 4 == 0 && Cp > 0 && Cp % 4 == 0 && off + Cp <= stride;
}

}  // namespace

static int norm_nchunk(int Bn, long long npix, int Cp) {
  const int cgroups = (Cp + 31) / 32;
  // ~4096 workgroups of four waves: sixteen per CU (256 chunks of a 32-channel tensor = one workgroup per CU ran at 2.7 TB/s)
  long long want = 4096 / ((long long)Bn * cgroups);
  if (want < 1) want = 1;
  long long maxchunk = npix / 256 > 0 ? npix / 256 : 1;
  long long n = want < maxchunk ? want : maxchunk;
  return (int)(n > 1024 ? 1024 : n);
}

extern "C" int egne_absmax(const float* x, int64_t pix_stride, int ch_off, int Cp, int64_t npix, void* out_bits, void* stream) {
  EGNE_REQUIRE(x && out_bits && Cp > 0 && Cp % 4 == 0 && ch_off % 4 == 0 && pix_stride % 4 == 0 && npix > 0, "absmax: bad arguments");
  long long total = npix * (Cp >> 2), g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(absmax_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, (long long)pix_stride, ch_off, Cp,
                     (long long)npix, (unsigned*)out_bits);
  return egne::check_launch("egne_absmax");
}

extern "C" int egne_absmax_f16(const void* x, int64_t pix_stride, int ch_off, int Cp, int64_t npix, void* out_bits, void* stream) {
  EGNE_REQUIRE(x && out_bits && Cp > 0 && Cp % 8 == 0 && ch_off % 8 == 0 && pix_stride % 8 == 0 && npix > 0 && ((uintptr_t)x & 15) == 0,
               "absmax_f16: bad arguments");
  long long total = npix * (Cp >> 3), g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(absmax_f16_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (long long)pix_stride, ch_off, Cp,
                     (long long)npix, (unsigned*)out_bits);
  return egne::check_launch("egne_absmax_f16");
}

// ------------------------------------------------------------------------------------------------
// Zero a list of device buffers in ONE launch (the gradient twins of a backward plan: torch._foreach_zero_ took 75 launches).
// table[i] = {address, bytes, first block}: 16-byte aligned, bytes a multiple of 16; a block clears up to 64 KB of one buffer.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zero_many_k(const unsigned long long* __restrict__ table, int nseg) {
  int lo = 0, hi = nseg - 1;                          // last segment whose first block <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[3 * mid + 2] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  egne_f32x4* base = (egne_f32x4*)table[3 * lo];
  const unsigned long long nvec = table[3 * lo + 1] >> 4, v0 = (blockIdx.x - table[3 * lo + 2]) * 4096ull;
  const egne_f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const unsigned long long v = v0 + i * 256 + threadIdx.x;
    if (v < nvec) base[v] = z;
  }
}

extern "C" int egne_zero_many(const void* table, int nseg, int64_t nblocks, void* stream) {
  EGNE_REQUIRE(table && nseg > 0 && nblocks > 0 && nblocks < (1ll << 31), "zero_many: bad arguments");
  hipLaunchKernelGGL(zero_many_k, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)table, nseg);
  return egne::check_launch("egne_zero_many");
}

extern "C" int64_t egne_norm_stats_workspace_bytes(int B, int HW, int Cp, int per_sample) {
  const int Bn = per_sample ? B : 1;
  const long long npix = per_sample ? HW : (long long)B * HW;
  return (int64_t)Bn * norm_nchunk(Bn, npix, Cp) * Cp * 2 * sizeof(double);
}

template <typename T>
static int norm_stats_impl(const T* x, int64_t pix_stride, int ch_off, int Cp, int B, int HW, int per_sample,
                           float eps, float* scale, float* shift, float* mean_out, float* var_out, void* ws, void* stream) {
  EGNE_REQUIRE(slice_ok(x, pix_stride, ch_off, Cp) && vec_ok<T>(pix_stride, ch_off, Cp), "norm_stats: bad slice (stride %lld off %d Cp %d)", (long long)pix_stride, ch_off, Cp);
  EGNE_REQUIRE(B > 0 && HW > 0 && scale && shift && ws && ((uintptr_t)ws & 15) == 0, "norm_stats: bad arguments (ws must be 16-byte aligned)");
  const int Bn = per_sample ? B : 1;
  const long long npix = per_sample ? HW : (long long)B * HW;
  const int cgroups = (Cp + 31) / 32;
  const int nchunk = norm_nchunk(Bn, npix, Cp);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(norm_stats_partial<T>, dim3(nchunk, cgroups, Bn), dim3(256), 0, st, x, (long long)pix_stride, ch_off,
                     Cp, npix, nchunk, (double*)ws);
  hipLaunchKernelGGL(norm_stats_final, dim3(cgroups, Bn), dim3(1024), 0, st, (const double*)ws, Cp, Bn, nchunk,
                     npix, eps, scale, shift, mean_out, var_out);
  return egne::check_launch("egne_norm_stats");
}
extern "C" int egne_norm_stats(const float* x, int64_t pix_stride, int ch_off, int Cp, int B, int HW, int per_sample,
                               float eps, float* scale, float* shift, float* mean_out, float* var_out, void* ws, void* stream) {
  return norm_stats_impl(x, pix_stride, ch_off, Cp, B, HW, per_sample, eps, scale, shift, mean_out, var_out, ws, stream);
}
extern "C" int egne_norm_stats_bf16(const void* x, int64_t pix_stride, int ch_off, int Cp, int B, int HW, int per_sample,
                                    float eps, float* scale, float* shift, float* mean_out, float* var_out, void* ws, void* stream) {
  return norm_stats_impl((const egne_bf16*)x, pix_stride, ch_off, Cp, B, HW, per_sample, eps, scale, shift, mean_out, var_out, ws, stream);
}

extern "C" int egne_norm_stats_finish(const void* ws, int Cp, int B, int nchunk, int HW, float eps, float* scale, float* shift,
                                      void* stream) {
  EGNE_REQUIRE(ws && scale && shift && Cp > 0 && B > 0 && nchunk > 0 && HW > 0 && ((uintptr_t)ws & 15) == 0, "norm_stats_finish: bad arguments");
  hipLaunchKernelGGL(norm_stats_finish_k, dim3((Cp + 31) / 32, B), dim3(1024), 0, (hipStream_t)stream, (const double*)ws, Cp, nchunk,
                     (long long)HW, eps, scale, shift);
  return egne::check_launch("egne_norm_stats_finish");
}

// first level of a long reduction (batch statistics: one "sample" of B x nchunk rows): block (32 channels, group g) sums the rows
// [g R, (g + 1) R) and leaves the sum IN row g R (every block touches its own rows only; fixed order: deterministic)
__global__ __launch_bounds__(1024) void norm_stats_groups_k(double* __restrict__ ws, int Cp, int nrows, int R) {
  const int g = blockIdx.y, c = blockIdx.x * 32 + (threadIdx.x & 31), kg = threadIdx.x >> 5;
  const int r0 = g * R, r1 = r0 + R < nrows ? r0 + R : nrows;
  double s = 0, q = 0;
  if (c < Cp) {
    for (int k = r0 + kg; k < r1; k += 32) {
      const double2 v = *(const double2*)(ws + ((long long)k * Cp + c) * 2);
      s += v.x; q += v.y;
    }
  }
  __shared__ double sh[32][32][2];
  sh[kg][threadIdx.x & 31][0] = s; sh[kg][threadIdx.x & 31][1] = q;
  __syncthreads();
  if (kg == 0 && c < Cp) {
    for (int j = 1; j < 32; ++j) { s += sh[j][threadIdx.x][0]; q += sh[j][threadIdx.x][1]; }
    *(double2*)(ws + ((long long)r0 * Cp + c) * 2) = make_double2(s, q);
  }
}

// the same with the moments themselves (biased variance) next to rstd / -mean rstd: a training-mode BatchNorm (utils.py:1049) whose batch
// statistics come out of the producing convolution's epilogue -- its B samples' chunks are one "sample" of B nchunk chunks and B HW pixels
extern "C" int egne_norm_stats_finish_moments(const void* ws, int Cp, int B, int nchunk, int HW, float eps, float* scale, float* shift,
                                              float* mean_out, float* var_out, void* stream) {
  EGNE_REQUIRE(ws && scale && shift && mean_out && var_out && Cp > 0 && B > 0 && nchunk > 0 && HW > 0 && ((uintptr_t)ws & 15) == 0,
               "norm_stats_finish_moments: bad arguments");
  // (ws is reduced IN PLACE when a sample has many rows -- one block per (32 channels, sample) would walk tens of megabytes alone: the
  //  first row of every group of R rows then holds the group's sum; the partial sums are the producing launch's to rewrite next step)
  constexpr int R = 256;
  if (nchunk > 4 * R) {
    const int G = (nchunk + R - 1) / R;
    EGNE_REQUIRE(nchunk % R == 0 || B == 1, "norm_stats_finish_moments: %d rows per sample are not a multiple of %d (several samples)", nchunk, R);
    for (int b = 0; b < B; ++b)
      hipLaunchKernelGGL(norm_stats_groups_k, dim3((Cp + 31) / 32, G), dim3(1024), 0, (hipStream_t)stream, (double*)ws + (long long)b * nchunk * Cp * 2, Cp, nchunk, R);
    hipLaunchKernelGGL(norm_stats_finish_k, dim3((Cp + 31) / 32, B), dim3(1024), 0, (hipStream_t)stream, (const double*)ws, Cp, G,
                       (long long)HW, eps, scale, shift, mean_out, var_out, (long long)R);
    return egne::check_launch("egne_norm_stats_finish_moments");
  }
  hipLaunchKernelGGL(norm_stats_finish_k, dim3((Cp + 31) / 32, B), dim3(1024), 0, (hipStream_t)stream, (const double*)ws, Cp, nchunk,
                     (long long)HW, eps, scale, shift, mean_out, var_out);
  return egne::check_launch("egne_norm_stats_finish_moments");
}

template <typename T>
static int affine_impl(const T* x, int64_t xs, int xo, T* y, int64_t ys, int yo, int Cp, int64_t npix, const float* scale,
                       const float* shift, void* stream) {
  EGNE_REQUIRE(slice_ok(x, xs, xo, Cp) && slice_ok(y, ys, yo, Cp) && scale && shift && npix > 0, "affine: bad arguments");
  hipLaunchKernelGGL(affine_k<T>, dim3(grid_for(npix * (Cp / 4))), dim3(256), 0, (hipStream_t)stream, x, (long long)xs, xo, y,
                     (long long)ys, yo, Cp, (long long)npix, scale, shift);
  return egne::check_launch("egne_affine");
}
extern "C" int egne_affine_inplace(float* x, int64_t pix_stride, int ch_off, int Cp, int64_t npix, const float* scale,
                                   const float* shift, void* stream) {
  return affine_impl<float>(x, pix_stride, ch_off, x, pix_stride, ch_off, Cp, npix, scale, shift, stream);
}
extern "C" int egne_affine(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo, int Cp, int64_t npix,
                           const float* scale, const float* shift, void* stream) {
  return affine_impl<float>(x, xs, xo, y, ys, yo, Cp, npix, scale, shift, stream);
}
extern "C" int egne_affine_act(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo, int Cp, int64_t npix,
                               const float* scale, const float* shift, int act, void* stream) {
  EGNE_REQUIRE(slice_ok(x, xs, xo, Cp) && slice_ok(y, ys, yo, Cp) && scale && shift && npix > 0, "affine_act: bad arguments");
  EGNE_REQUIRE(act == EGNE_ACT_NONE || act == EGNE_ACT_RELU || act == EGNE_ACT_LEAKY, "affine_act: activation %d", act);
  const float slope = act == EGNE_ACT_RELU ? 0.f : (act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  hipLaunchKernelGGL(affine_act_k, dim3(grid_for(npix * (Cp / 4))), dim3(256), 0, (hipStream_t)stream, x, (long long)xs, xo, y,
                     (long long)ys, yo, Cp, (long long)npix, scale, shift, slope);
  return egne::check_launch("egne_affine_act");
}
extern "C" int egne_affine_bf16(const void* x, int64_t xs, int xo, void* y, int64_t ys, int yo, int Cp, int64_t npix,
                                const float* scale, const float* shift, void* stream) {
  return affine_impl<egne_bf16>((const egne_bf16*)x, xs, xo, (egne_bf16*)y, ys, yo, Cp, npix, scale, shift, stream);
}

template <typename T>
static int avgpool2_impl(const T* x, int64_t xs, int xo, T* y, int64_t ys, int yo, int B, int H, int W, int Cp, void* stream) {
  EGNE_REQUIRE(slice_ok(x, xs, xo, Cp) && slice_ok(y, ys, yo, Cp), "avgpool2: bad slices");
  EGNE_REQUIRE(B > 0 && H >= 2 && W >= 2, "avgpool2: bad shape");
  hipLaunchKernelGGL(avgpool2_k<T>, dim3(grid_for((long long)B * (H / 2) * (W / 2) * (Cp / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, (long long)xs, xo, y, (long long)ys, yo, B, H, W, Cp);
  return egne::check_launch("egne_avgpool2");
}
extern "C" int egne_avgpool2(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo, int B, int H, int W,
                             int Cp, void* stream) {
  return avgpool2_impl(x, xs, xo, y, ys, yo, B, H, W, Cp, stream);
}
extern "C" int egne_avgpool2_bf16(const void* x, int64_t xs, int xo, void* y, int64_t ys, int yo, int B, int H, int W,
                                  int Cp, void* stream) {
  return avgpool2_impl((const egne_bf16*)x, xs, xo, (egne_bf16*)y, ys, yo, B, H, W, Cp, stream);
}

template <typename T>
static int norm_act_pool2_impl(const T* x, int64_t xs, int xo, const float* scale, const float* shift, int act,
                               T* y, int64_t ys, int yo, int B, int H, int W, int Cp, void* stream) {
  EGNE_REQUIRE(slice_ok(x, xs, xo, Cp) && slice_ok(y, ys, yo, Cp) && scale && shift, "norm_act_pool2: bad slices");
  EGNE_REQUIRE(vec_ok<T>(xs, xo, Cp) && vec_ok<T>(ys, yo, Cp), "norm_act_pool2: slices must be 16-byte vectors (8 bf16 channels)");
  EGNE_REQUIRE(B > 0 && H >= 2 && W >= 2, "norm_act_pool2: bad shape");
  hipLaunchKernelGGL(norm_act_pool2_k<T>, dim3(grid_for((long long)B * (H / 2) * (W / 2) * (Cp / egne_vt<T>::N))), dim3(256), 0,
                     (hipStream_t)stream, x, (long long)xs, xo, scale, shift, act, y, (long long)ys, yo, B, H, W, Cp);
  return egne::check_launch("egne_norm_act_pool2");
}
extern "C" int egne_norm_act_pool2(const float* x, int64_t xs, int xo, const float* scale, const float* shift, int act,
                                   float* y, int64_t ys, int yo, int B, int H, int W, int Cp, void* stream) {
  return norm_act_pool2_impl(x, xs, xo, scale, shift, act, y, ys, yo, B, H, W, Cp, stream);
}
extern "C" int egne_norm_act_pool2_bf16(const void* x, int64_t xs, int xo, const float* scale, const float* shift, int act,
                                        void* y, int64_t ys, int yo, int B, int H, int W, int Cp, void* stream) {
  return norm_act_pool2_impl((const egne_bf16*)x, xs, xo, scale, shift, act, (egne_bf16*)y, ys, yo, B, H, W, Cp, stream);
}

extern "C" int egne_maxpool2_f16(const void* x, int64_t xs, int xo, void* y, int64_t ys, int yo, int B, int H, int W,
                                 int Ho, int Wo, int stride, int Cp, void* stream) {
  EGNE_REQUIRE(x && y && Cp > 0 && Cp % 8 == 0 && xo % 8 == 0 && yo % 8 == 0 && xs % 8 == 0 && ys % 8 == 0 && ((uintptr_t)x & 15) == 0 &&
               ((uintptr_t)y & 15) == 0 && xo + Cp <= xs && yo + Cp <= ys, "maxpool2_f16: bad slices");
  EGNE_REQUIRE(stride == 1 || stride == 2, "maxpool2_f16: stride %d", stride);
  auto osz = [&](int n) { int o = (n - 2 + stride - 1) / stride + 1; if ((o - 1) * stride >= n) --o; return o; };
  EGNE_REQUIRE(B > 0 && H >= 2 && W >= 2 && Ho == osz(H) && Wo == osz(W), "maxpool2_f16: output %dx%d != %dx%d", Ho, Wo, osz(H), osz(W));
  EGNE_REQUIRE(Ho <= 65535 && B <= 65535, "maxpool2_f16: grid limits");
  hipLaunchKernelGGL(maxpool2_f16_k, dim3((Wo * (Cp / 8) + 255) / 256, (Ho + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x,
                     (long long)xs, xo, (_Float16*)y, (long long)ys, yo, B, H, W, Ho, Wo, stride, Cp);
  return egne::check_launch("egne_maxpool2_f16");
}

extern "C" int egne_maxpool2(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo, int B, int H, int W,
                             int Ho, int Wo, int stride, int Cp, void* stream) {
  EGNE_REQUIRE(slice_ok(x, xs, xo, Cp) && slice_ok(y, ys, yo, Cp), "maxpool2: bad slices");
  EGNE_REQUIRE(stride == 1 || stride == 2, "maxpool2: stride %d", stride);
  // ceil_mode output size; the last window must start inside the input
  auto osz = [&](int n) { int o = (n - 2 + stride - 1) / stride + 1; if ((o - 1) * stride >= n) --o; return o; };
  EGNE_REQUIRE(B > 0 && H >= 2 && W >= 2 && Ho == osz(H) && Wo == osz(W), "maxpool2: output %dx%d != %dx%d", Ho, Wo, osz(H), osz(W));
  EGNE_REQUIRE(Ho <= 65535 && B <= 65535, "maxpool2: grid limits");
  hipLaunchKernelGGL(maxpool2_k, dim3((Wo * (Cp / 4) + 255) / 256, (Ho + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)xs, xo, y, (long long)ys, yo, B, H, W, Ho, Wo, stride, Cp);
  return egne::check_launch("egne_maxpool2");
}

template <typename T>
static int upsample2x_impl(const T* x, int64_t xs, int xo, T* y, int64_t ys, int yo, int B, int H, int W, int Cp, void* stream) {
  EGNE_REQUIRE(slice_ok(x, xs, xo, Cp) && slice_ok(y, ys, yo, Cp), "upsample2x: bad slices");
  EGNE_REQUIRE(B > 0 && H > 0 && W > 0, "upsample2x: bad shape");
  EGNE_REQUIRE(2 * H <= 65535 && B <= 65535, "upsample2x: grid limits");
  hipLaunchKernelGGL(upsample2x_k<T>, dim3((2 * W * (Cp / 4) + 255) / 256, 2 * H, B), dim3(256), 0,
                     (hipStream_t)stream, x, (long long)xs, xo, y, (long long)ys, yo, B, H, W, Cp);
  return egne::check_launch("egne_upsample2x");
}
extern "C" int egne_upsample2x(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo, int B, int H, int W,
                               int Cp, void* stream) {
  return upsample2x_impl(x, xs, xo, y, ys, yo, B, H, W, Cp, stream);
}
extern "C" int egne_upsample2x_bf16(const void* x, int64_t xs, int xo, void* y, int64_t ys, int yo, int B, int H, int W,
                                    int Cp, void* stream) {
  return upsample2x_impl((const egne_bf16*)x, xs, xo, (egne_bf16*)y, ys, yo, B, H, W, Cp, stream);
}

template <typename T>
static int upsample2x_nearest_impl(const T* x, int64_t xs, int xo, T* y, int64_t ys, int yo, int B, int H, int W, int Cp, void* stream) {
  EGNE_REQUIRE(slice_ok(x, xs, xo, Cp) && slice_ok(y, ys, yo, Cp), "upsample2x_nearest: bad slices");
  EGNE_REQUIRE(B > 0 && H > 0 && W > 0 && 2 * H <= 65535 && B <= 65535, "upsample2x_nearest: bad shape");
  hipLaunchKernelGGL(upsample2x_nearest_k<T>, dim3((2 * W * (Cp / 4) + 255) / 256, 2 * H, B), dim3(256), 0, (hipStream_t)stream, x, (long long)xs, xo,
                     y, (long long)ys, yo, B, H, W, Cp);
  return egne::check_launch("egne_upsample2x_nearest");
}
extern "C" int egne_upsample2x_nearest(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo, int B, int H, int W, int Cp, void* stream) {
  return upsample2x_nearest_impl(x, xs, xo, y, ys, yo, B, H, W, Cp, stream);
}
extern "C" int egne_upsample2x_nearest_bf16(const void* x, int64_t xs, int xo, void* y, int64_t ys, int yo, int B, int H, int W, int Cp, void* stream) {
  return upsample2x_nearest_impl((const egne_bf16*)x, xs, xo, (egne_bf16*)y, ys, yo, B, H, W, Cp, stream);
}

template <typename T>
static int nchw_to_nhwc_impl(const float* x, int B, int C, int H, int W, T* y, int64_t ys, int yo, int Cp, void* stream) {
  EGNE_REQUIRE(x && y && B > 0 && C > 0 && C <= Cp && yo + Cp <= ys, "nchw_to_nhwc: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_k<T>, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, x, B, C, H,
                     W, y, (long long)ys, yo, Cp);
  return egne::check_launch("egne_nchw_to_nhwc");
}
extern "C" int egne_nchw_to_nhwc(const float* x, int B, int C, int H, int W, float* y, int64_t ys, int yo, int Cp,
                                 void* stream) {
  return nchw_to_nhwc_impl(x, B, C, H, W, y, ys, yo, Cp, stream);
}
/* fp32 NCHW in (the caller's frames), bf16 NHWC out */
extern "C" int egne_nchw_to_nhwc_bf16(const float* x, int B, int C, int H, int W, void* y, int64_t ys, int yo, int Cp,
                                      void* stream) {
  return nchw_to_nhwc_impl(x, B, C, H, W, (egne_bf16*)y, ys, yo, Cp, stream);
}

extern "C" int egne_nhwc_to_nchw(const float* x, int64_t xs, int xo, int B, int C, int H, int W, float* y,
                                 void* stream) {
  EGNE_REQUIRE(x && y && B > 0 && C > 0 && xo + C <= xs, "nhwc_to_nchw: bad arguments");
  hipLaunchKernelGGL(nhwc_to_nchw_k, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)xs, xo, B, C, H, W, y);
  return egne::check_launch("egne_nhwc_to_nchw");
}

extern "C" int egne_ellipse_head_act(float* x, int B, int ld, void* stream) {
  EGNE_REQUIRE(x && B > 0 && ld >= 10, "ellipse_head_act: bad arguments");
  hipLaunchKernelGGL(ellipse_head_act_k<float>, dim3((B * 10 + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, B, ld);
  return egne::check_launch("egne_ellipse_head_act");
}
extern "C" int egne_ellipse_head_act_bf16(void* x, int B, int ld, void* stream) {
  EGNE_REQUIRE(x && B > 0 && ld >= 10, "ellipse_head_act: bad arguments");
  hipLaunchKernelGGL(ellipse_head_act_k<egne_bf16>, dim3((B * 10 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (egne_bf16*)x, B, ld);
  return egne::check_launch("egne_ellipse_head_act_bf16");
}

extern "C" int egne_selu_inplace(float* x, int64_t n, void* stream) {
  EGNE_REQUIRE(x && n > 0, "selu: bad arguments");
  hipLaunchKernelGGL(selu_k<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n);
  return egne::check_launch("egne_selu_inplace");
}
extern "C" int egne_selu_inplace_bf16(void* x, int64_t n, void* stream) {
  EGNE_REQUIRE(x && n > 0, "selu: bad arguments");
  hipLaunchKernelGGL(selu_k<egne_bf16>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (egne_bf16*)x, (long long)n);
  return egne::check_launch("egne_selu_inplace_bf16");
}

template <typename T, typename TO>
static int spatial_mean_impl(const T* x, int64_t pix_stride, int ch_off, int C, int B, int HW, TO* out, void* stream) {
  EGNE_REQUIRE(x && out && B > 0 && HW > 0 && C > 0 && ch_off + C <= pix_stride, "spatial_mean: bad arguments");
  hipLaunchKernelGGL((spatial_mean_k<T, TO>), dim3(B), dim3(256), 0, (hipStream_t)stream, x, (long long)pix_stride, ch_off, C, HW, out);
  return egne::check_launch("egne_spatial_mean");
}
extern "C" int egne_spatial_mean(const float* x, int64_t pix_stride, int ch_off, int C, int B, int HW, float* out,
                                 void* stream) {
  return spatial_mean_impl(x, pix_stride, ch_off, C, B, HW, out, stream);
}
/* bf16 in, bf16 out (the mean is one more activation tensor of the plan) */
extern "C" int egne_spatial_mean_bf16(const void* x, int64_t pix_stride, int ch_off, int C, int B, int HW, void* out,
                                      void* stream) {
  return spatial_mean_impl((const egne_bf16*)x, pix_stride, ch_off, C, B, HW, (egne_bf16*)out, stream);
}
