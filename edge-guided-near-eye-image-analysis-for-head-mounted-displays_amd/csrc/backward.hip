// Backward kernels of ESF-Net (BDCN is frozen: train.py:129, utils.py:646).  Data gradients of the
// convolutions reuse the forward MFMA kernels with flipped / transposed weight packs
// (egne_pack_conv_weight_dgrad); this file holds the weight-gradient GEMM, the loss-head gradient
// and the HBM-bound elementwise / reduction backward ops.  Everything is deterministic (two-stage
// reductions, no atomics).
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

inline int grid_for(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------------------------------------
// loss head backward (models/RITnet_v2.py:372-432, loss.py:16-137): d total / d logits, d total / d elOut
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sgn(float v) { return (v > 0.f) - (v < 0.f); }

template <typename T>
__global__ __launch_bounds__(256) void loss_bwd_k(const egne_loss_desc d, const float* __restrict__ gscale_p,
                                                  T* __restrict__ g_logits, long long gs, int go,
                                                  float* __restrict__ g_elOut) {
  const float gscale = gscale_p[0];
  const int b = blockIdx.y;
  const int HW = d.H * d.W;
  const float* cf = d.coef + b * 32;
  const float nmask = d.out_terms[5];
  const float mp = cf[0];
  const float segw = (mp == 1.f && nmask > 0.f) ? 20.f * gscale / nmask : 0.f;
  const float A = cf[4], Bq = cf[5], dact = cf[6], msw = cf[7];
  const float cpx = cf[8], cpy = cf[9], cix = cf[10], ciy = cf[11];
  const float pm = cf[12], ps = cf[13], im = cf[14], is = cf[15];
  const float spx = sgn(cpx - cf[16]), spy = sgn(cpy - cf[17]);
  const float six = sgn(cix - d.elNorm[b * 10 + 0]), siy = sgn(ciy - d.elNorm[b * 10 + 1]);
  const float kp = gscale * 0.5f / (2.f * (float)d.B) * 4.f;
  const float ki = (nmask > 0.f) ? gscale * 0.5f * mp / (2.f * nmask) * (-4.f) : 0.f;
  const float fHW = (float)HW;
  const long long base = (long long)b * HW;
  // upstream gradient w.r.t. the soft-argmax centres (temperature 4 / -4 folded in); the iris centre is a function of the logits only
  // when some sample has a mask (RITnet_v2.py:392-404)
  const bool iris_up = d.g_pred_c && nmask > 0.f;
  const float upx = d.g_pred_c ? 4.f * d.g_pred_c[b * 4 + 2] : 0.f, upy = d.g_pred_c ? 4.f * d.g_pred_c[b * 4 + 3] : 0.f;
  const float uix = iris_up ? -4.f * d.g_pred_c[b * 4 + 0] : 0.f, uiy = iris_up ? -4.f * d.g_pred_c[b * 4 + 1] : 0.f;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const T* lp = (const T*)d.logits + (base + p) * d.pix_stride + d.ch_off;
    const float l0 = ld1(lp), l1 = ld1(lp + 1), l2 = ld1(lp + 2);
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (segw != 0.f) {
      const int t = (int)d.target[base + p];
      const float mx = fmaxf(l0, fmaxf(l1, l2));
      const float e0 = expf(l0 - mx), e1 = expf(l1 - mx), e2 = expf(l2 - mx);
      const float inv = 1.f / (e0 + e1 + e2);
      const float pr[3] = {e0 * inv, e1 * inv, e2 * inv};
      float u[3], dot = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float tc = (t == c) ? 1.f : 0.f;
        u[c] = d.alpha * d.distMap[((long long)b * 3 + c) * HW + p] / (3.f * fHW)
             - (1.f - d.alpha) * dact * 2.f * cf[1 + c] * (tc * Bq - A) / (Bq * Bq);
        dot += pr[c] * u[c];
      }
      const float ce = msw / fHW;
      g0 = segw * (pr[0] * (u[0] - dot) + ce * (pr[0] - (t == 0 ? 1.f : 0.f)));
      g1 = segw * (pr[1] * (u[1] - dot) + ce * (pr[1] - (t == 1 ? 1.f : 0.f)));
      g2 = segw * (pr[2] * (u[2] - dot) + ce * (pr[2] - (t == 2 ? 1.f : 0.f)));
    }
    const int y = p / d.W, x = p - y * d.W;
    const float gx = d.grid_x[x], gy = d.grid_y[y];
    const float wp = expf(4.f * l2 - pm) / ps;
    // d c / d l_j = T p_j (x_j - c) for c = sum_j p_j x_j, p = softmax(T l): the loss term's own upstream (sign / count) plus what the
    // caller back-propagates through pred_c (elPred's centre entries)
    g2 += kp * wp * (spx * (gx - cpx) + spy * (gy - cpy));
    if (d.g_pred_c) g2 += wp * (upx * (gx - cpx) + upy * (gy - cpy));
    if (ki != 0.f || iris_up) {
      const float wi = expf(-4.f * l0 - im) / is;
      if (ki != 0.f) g0 += ki * wi * (six * (gx - cix) + siy * (gy - ciy));
      if (iris_up) g0 += wi * (uix * (gx - cix) + uiy * (gy - ciy));
    }
    if (d.g_op_nchw) {
      const float* q = d.g_op_nchw + (long long)b * 3 * HW + p;
      g0 += q[0]; g1 += q[HW]; g2 += q[2 * (long long)HW];
    }
    T* o = g_logits + (base + p) * gs + go;
    st1(o, g0); st1(o + 1, g1); st1(o + 2, g2);
  }
  if (blockIdx.x == 0 && threadIdx.x < 10) {
    const int j = threadIdx.x;
    const float nabs = (float)d.B - nmask;
    float g = 0.f;
    if (mp == 1.f) {
      if (nmask > 0.f) g = gscale * sgn(d.elOut[b * 10 + j] - d.elNorm[b * 10 + j]) / nmask;   // 10 * (1/10) / nmask
    } else if (j == 5 || j == 6) {
      g = 10.f * gscale * sgn(d.elOut[b * 10 + j] - cf[16 + (j - 5)]) / (2.f * nabs);
    }
    if (d.g_elOut_up) g += d.g_elOut_up[b * 10 + j];
    g_elOut[b * 10 + j] = g;
  }
}

// ------------------------------------------------------------------------------------------------
// gz = gy * act'(y) in place + bias gradient (sum over pixels), two deterministic stages
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_bias_partial(T* __restrict__ g, long long gs, int go,
                                                            const T* __restrict__ y, long long ys, int yo, int act,
                                                            int Cp, long long npix, int nchunk, double* __restrict__ ws,
                                                            unsigned* __restrict__ absmax_bits) {
  constexpr int N = egne_vt<T>::N, CV = 32 / N, ROWS = 256 / CV;      // 16-byte vectors: 8 x 32 rows (fp32) or 4 x 64 rows (bf16)
  const int chunk = blockIdx.x, cg = blockIdx.y;
  unsigned mb = 0;
  const int v = threadIdx.x % CV, row = threadIdx.x / CV;
  const int c = cg * 32 + v * N;
  const long long per = (npix + nchunk - 1) / nchunk;
  const long long p0 = (long long)chunk * per, p1 = p0 + per < npix ? p0 + per : npix;
  double s[N];
#pragma unroll
  for (int e = 0; e < N; ++e) s[e] = 0;
  if (c < Cp) {
    // four pixel rows per trip, all eight loads issued before the first use (one row at a time ran at 3.0 TB/s: latency bound)
    for (long long p = p0 + row; p < p1; p += 4 * ROWS) {
      egne_fv<N> t[4], yy[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long q = p + ROWS * u;
        t[u] = q < p1 ? ldv(g + q * gs + go + c) : fv_fill<N>(0.f);
        yy[u] = (q < p1 && act != EGNE_ACT_NONE) ? ldv(y + q * ys + yo + c) : fv_fill<N>(1.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long q = p + ROWS * u;
        if (act != EGNE_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < N; ++e) t[u].v[e] = yy[u].v[e] > 0.f ? t[u].v[e] : (act == EGNE_ACT_LEAKY ? 0.01f * t[u].v[e] : 0.f);
          if (q < p1) stv(g + q * gs + go + c, t[u]);
        }
#pragma unroll
        for (int e = 0; e < N; ++e) {
          s[e] += t[u].v[e];
          const unsigned b = __float_as_uint(t[u].v[e]) & 0x7fffffffu;
          mb = b > mb ? b : mb;
        }
      }
    }
  }
  if (absmax_bits) {
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned t = (unsigned)__shfl_xor((int)mb, o);
      mb = t > mb ? t : mb;
    }
    if ((threadIdx.x & 63) == 0 && mb > __hip_atomic_load(absmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(absmax_bits, mb);
  }
  __shared__ double sh[ROWS][32];
#pragma unroll
  for (int e = 0; e < N; ++e) sh[row][v * N + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 32) {
    double a = 0;
    for (int r = 0; r < ROWS; ++r) a += sh[r][threadIdx.x];
    const int cc = cg * 32 + threadIdx.x;
    if (cc < Cp) ws[(long long)chunk * Cp + cc] = a;
  }
}

// 1024 threads = 32 channels x 32 interleaved chunk ranges, summed through LDS in a fixed order (deterministic).  (8 ranges were 128
// dependent loads per thread for the 1024 chunks of a full-resolution tensor: 27 us per call, 54 calls per training step.)
__global__ __launch_bounds__(1024) void reduce_chunks_k(const double* __restrict__ ws, int stride, int n, int nchunk, float* __restrict__ out,
                                                        int accumulate) {
  __shared__ double part[32][32];
  const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + c;
  double s = 0;
  if (i < n)
    for (int k = q; k < nchunk; k += 32) s += ws[(long long)k * stride + i];
  part[q][c] = s;
  __syncthreads();
  if (q == 0 && i < n) {
    double t = 0;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += part[r][c];
    out[i] = accumulate ? out[i] + (float)t : (float)t;
  }
}

// ------------------------------------------------------------------------------------------------
// InstanceNorm / BatchNorm backward.  xh = x*scale + shift (scale = rstd, shift = -mean*rstd);
// g = gy * act'(xh) [* gamma];  gx += rstd * (g - mean(g) - xh * mean(g*xh));  dgamma = sum gy*xh, dbeta = sum gy
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void norm_bwd_partial(const T* __restrict__ x, long long xs, int xo,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const T* __restrict__ gy, long long gs, int go, int act_in,
                                                        int Cp, long long npix_per_n, int nchunk, int per_sample,
                                                        double* __restrict__ ws, int poolW) {
  // poolW > 0: gy is the gradient of the 2x2-average-POOLED tensor ([n][H/2][W/2]); pixel p = (y, x) of a W = poolW wide map
  // receives a quarter of its pooled cell's gradient (egne_norm_pool2_bwd)
  constexpr int N = egne_vt<T>::N, CV = 32 / N, ROWS = 256 / CV;
  const int chunk = blockIdx.x, cg = blockIdx.y, n = blockIdx.z;
  const int v = threadIdx.x % CV, row = threadIdx.x / CV;
  const int c = cg * 32 + v * N;
  const long long per = (npix_per_n + nchunk - 1) / nchunk;
  const long long p0 = (long long)chunk * per, p1 = p0 + per < npix_per_n ? p0 + per : npix_per_n;
  double s1[N], s2[N];
#pragma unroll
  for (int e = 0; e < N; ++e) { s1[e] = 0; s2[e] = 0; }
  if (c < Cp) {
    const egne_fv<N> sc = ldf<N>(scale + (long long)(per_sample ? n : 0) * Cp + c);
    const egne_fv<N> sh = ldf<N>(shift + (long long)(per_sample ? n : 0) * Cp + c);
    const long long nb = (long long)n * npix_per_n;
    for (long long p = p0 + row; p < p1; p += 4 * ROWS) {        // four rows per trip: loads issued together (same summation order)
      egne_fv<N> xv4[4], g4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long q = p + ROWS * u;
        const bool ok = q < p1;
        xv4[u] = ok ? ldv(x + (nb + q) * xs + xo + c) : fv_fill<N>(0.f);
        if (poolW) {       // (q < H W < 2^31: 32-bit division -- the 64-bit one was most of this kernel's instructions)
          const unsigned uq = (unsigned)q, py = uq / (unsigned)poolW, px = uq - py * (unsigned)poolW;
          g4[u] = ok ? ldv(gy + ((nb >> 2) + (long long)(py >> 1) * (poolW >> 1) + (px >> 1)) * gs + go + c) : fv_fill<N>(0.f);
#pragma unroll
          for (int e = 0; e < N; ++e) g4[u].v[e] *= 0.25f;
        } else {
          g4[u] = ok ? ldv(gy + (nb + q) * gs + go + c) : fv_fill<N>(0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          float g = g4[u].v[e];
          const float xh = xv4[u].v[e] * sc.v[e] + sh.v[e];
          if (act_in == EGNE_ACT_LEAKY) g = xh > 0.f ? g : 0.01f * g;
          else if (act_in == EGNE_ACT_RELU) g = xh > 0.f ? g : 0.f;
          s1[e] += g; s2[e] += (double)g * xh;
        }
      }
    }
  }
  __shared__ double sh_[ROWS][32][2];
#pragma unroll
  for (int e = 0; e < N; ++e) { sh_[row][v * N + e][0] = s1[e]; sh_[row][v * N + e][1] = s2[e]; }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc_ = threadIdx.x >> 1, w_ = threadIdx.x & 1;
    double a = 0;
    for (int r = 0; r < ROWS; ++r) a += sh_[r][cc_][w_];
    const int cc = cg * 32 + cc_;
    if (cc < Cp) ws[(((long long)n * nchunk + chunk) * Cp + cc) * 2 + w_] = a;
  }
}

// ------------------------------------------------------------------------------------------------
// Fused form (round 5).  A down block's input x (and its output `out`) is normalised ONCE (scale / shift per sample) and consumed by two
// readers: conv1 behind IN(x) (models/RITnet_v2.py:57) and Transition_down behind avg_pool(leaky(IN(.))) (:40-44).  The InstanceNorm
// backward is linear in its upstream gradient, so both go through ONE backward with
//   G = a1 + leaky'(xh) * up(gq) / 4        (a1: conv1's data gradient, full resolution; gq: the pooled cell's gradient)
//   gx = rstd * (G - mean(G) - xh * mean(G xh)),
// and gx is not accumulated into the gradient tensor by a pass of its own: it is added where the tensor's producing layer masks its
// output gradient anyway (egne_act_norm_bwd: gz = act'(y) * (g + gx), y = x, + that layer's bias sums) -- the pass egne_act_bwd_bias
// makes.  Before: two statistics passes + two apply passes (each a read-modify-write of the gradient) + the masking pass.
template <typename T>
struct NormAddends {
  const T* a1; long long a1s; int a1o;       // full-resolution addend (may be null)
  const T* gq; long long gqs; int gqo;       // pooled addend (may be null): [n][H/2][W/2], poolW = W of the full-resolution map
  int poolW, act_q;                          // activation between the normalisation and the pooling (LeakyReLU of Transition_down)
};

template <typename T, int N>
__device__ __forceinline__ egne_fv<N> norm_addend_G(const NormAddends<T>& A, const egne_fv<N>& xh, long long nb, unsigned q, int c, bool ok) {
  egne_fv<N> G = fv_fill<N>(0.f);
  if (A.a1 && ok) G = ldv(A.a1 + (nb + q) * A.a1s + A.a1o + c);
  if (A.gq && ok) {
    const unsigned py = q / (unsigned)A.poolW, px = q - py * (unsigned)A.poolW;
    const egne_fv<N> g = ldv(A.gq + ((nb >> 2) + (long long)(py >> 1) * (A.poolW >> 1) + (px >> 1)) * A.gqs + A.gqo + c);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      float ge = 0.25f * g.v[e];
      if (A.act_q == EGNE_ACT_LEAKY) ge = xh.v[e] > 0.f ? ge : 0.01f * ge;
      else if (A.act_q == EGNE_ACT_RELU) ge = xh.v[e] > 0.f ? ge : 0.f;
      G.v[e] += ge;
    }
  }
  return G;
}

// sums of G and G xh per (n, chunk, c): the layout of norm_bwd_partial (finished by norm_bwd_final)
template <typename T>
__global__ __launch_bounds__(256) void norm_fuse_partial(const T* __restrict__ x, long long xs, int xo, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, NormAddends<T> A, int Cp, long long npix_per_n,
                                                         int nchunk, double* __restrict__ ws, int per_sample) {
  // per_sample = 0: batch statistics (BatchNorm): one (scale, shift) row for all samples; the rows of ws are then summed over n too
  constexpr int N = egne_vt<T>::N, CV = 32 / N, ROWS = 256 / CV;
  const int chunk = blockIdx.x, cg = blockIdx.y, n = blockIdx.z;
  const int v = threadIdx.x % CV, row = threadIdx.x / CV;
  const int c = cg * 32 + v * N;
  const long long per = (npix_per_n + nchunk - 1) / nchunk;
  const long long p0 = (long long)chunk * per, p1 = p0 + per < npix_per_n ? p0 + per : npix_per_n;
  double s1[N], s2[N];
#pragma unroll
  for (int e = 0; e < N; ++e) { s1[e] = 0; s2[e] = 0; }
  if (c < Cp) {
    const long long tr = per_sample ? n : 0;
    const egne_fv<N> sc = ldf<N>(scale + tr * Cp + c), sh = ldf<N>(shift + tr * Cp + c);
    const long long nb = (long long)n * npix_per_n;
    const T* const tag = nullptr;
    for (long long p = p0 + row; p < p1; p += 2 * ROWS) {        // two rows per trip, loads kept packed (16 bytes = 4 registers per tensor and row)
      egne_u32x4 xv[2], a1v[2], gqv[2];
      bool okk[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long long q = p + ROWS * u;
        const bool ok = okk[u] = q < p1;
        xv[u] = ok ? ldraw(x + (nb + q) * xs + xo + c) : egne_u32x4{0u, 0u, 0u, 0u};
        a1v[u] = (A.a1 && ok) ? ldraw(A.a1 + (nb + q) * A.a1s + A.a1o + c) : egne_u32x4{0u, 0u, 0u, 0u};
        if (A.gq && ok) {
          const unsigned uq = (unsigned)q, py = uq / (unsigned)A.poolW, px = uq - py * (unsigned)A.poolW;
          gqv[u] = ldraw(A.gq + ((nb >> 2) + (long long)(py >> 1) * (A.poolW >> 1) + (px >> 1)) * A.gqs + A.gqo + c);
        } else {
          gqv[u] = egne_u32x4{0u, 0u, 0u, 0u};
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const egne_fv<N> xf = unpackv(tag, xv[u]), af = unpackv(tag, a1v[u]), gf = unpackv(tag, gqv[u]);
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const float xh = xf.v[e] * sc.v[e] + sh.v[e];
          float ge = 0.25f * gf.v[e];
          if (A.act_q == EGNE_ACT_LEAKY) ge = xh > 0.f ? ge : 0.01f * ge;
          else if (A.act_q == EGNE_ACT_RELU) ge = xh > 0.f ? ge : 0.f;
          const float G = okk[u] ? af.v[e] + ge : 0.f;
          s1[e] += G; s2[e] += (double)G * xh;
        }
      }
    }
  }
  __shared__ double sh_[ROWS][32][2];
#pragma unroll
  for (int e = 0; e < N; ++e) { sh_[row][v * N + e][0] = s1[e]; sh_[row][v * N + e][1] = s2[e]; }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc_ = threadIdx.x >> 1, w_ = threadIdx.x & 1;
    double a = 0;
    for (int r = 0; r < ROWS; ++r) a += sh_[r][cc_][w_];
    const int cc = cg * 32 + cc_;
    if (cc < Cp) ws[(((long long)n * nchunk + chunk) * Cp + cc) * 2 + w_] = a;
  }
}

// gz = act'(y) * (g + rstd * (G - m1 - xh m2)) in place + bias partial sums: one row [Cp] of ws per (sample, chunk) -- every row of ws
// is summed by the second stages (reduce_chunks_k, egne_pair_bias_bwd), so any partition of the pixels will do; y = x: the tensor
// the gradient belongs to is the one that was normalised.  Grid (chunks per sample, channel groups, samples): the per-(n, c)
// coefficients are loaded once per thread.
template <typename T>
__global__ __launch_bounds__(256) void act_norm_bwd_partial(T* __restrict__ g, long long gs, int go, const T* __restrict__ y, long long ys, int yo,
                                                            int act, const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ sums, NormAddends<T> A, int Cp, long long HW,
                                                            int nps, double* __restrict__ ws, int per_sample, const float* __restrict__ gamma,
                                                            float invN, int acc_n) {
  // per_sample = 0 (BatchNorm): one row of scale / shift / sums for the whole batch, invN = 1 / (B HW), gamma scales the result;
  // acc_n: g of samples n < acc_n is read and accumulated onto, g of the others is written only (the normalisation's backward is
  // the only source of their gradient: 0 for a BatchNorm's input, the image half of the batch for an encoder tensor the decoder reads)
  constexpr int N = egne_vt<T>::N, CV = 32 / N, ROWS = 256 / CV;
  const int chunk = blockIdx.x, cg = blockIdx.y, n = blockIdx.z;
  const int v = threadIdx.x % CV, row = threadIdx.x / CV;
  const int c = cg * 32 + v * N;
  const long long per = (HW + nps - 1) / nps;
  const long long p0 = (long long)chunk * per, p1 = p0 + per < HW ? p0 + per : HW;
  double s[N];
#pragma unroll
  for (int e = 0; e < N; ++e) s[e] = 0;
  if (c < Cp) {
    const long long tr = per_sample ? n : 0;
    const bool accumulate = n < acc_n;
    const egne_fv<N> sc = ldf<N>(scale + tr * Cp + c), sh = ldf<N>(shift + tr * Cp + c);
    float k0[N], k1[N], k2[N];     // sc gamma, sc gamma mean(G), sc gamma mean(G xh): r = g + k0 (a1 + ge) - k1 - xh k2
#pragma unroll
    for (int e = 0; e < N; ++e) {
      k0[e] = sc.v[e] * (gamma ? gamma[c + e] : 1.f);
      k1[e] = k0[e] * sums[2 * (tr * Cp + c + e)] * invN; k2[e] = k0[e] * sums[2 * (tr * Cp + c + e) + 1] * invN;
    }
    const long long nb = (long long)n * HW;
    const T* const tag = nullptr;
    for (long long p = p0 + row; p < p1; p += 2 * ROWS) {        // two rows per trip, loads kept packed
      egne_u32x4 t[2], yy[2], a1v[2], gqv[2];
      bool okk[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long long q = p + ROWS * u;
        const bool ok = okk[u] = q < p1;
        t[u] = (ok && accumulate) ? ldraw(g + (nb + q) * gs + go + c) : egne_u32x4{0u, 0u, 0u, 0u};
        yy[u] = ok ? ldraw(y + (nb + q) * ys + yo + c) : egne_u32x4{0u, 0u, 0u, 0u};
        a1v[u] = (A.a1 && ok) ? ldraw(A.a1 + (nb + q) * A.a1s + A.a1o + c) : egne_u32x4{0u, 0u, 0u, 0u};
        if (A.gq && ok) {
          const unsigned uq = (unsigned)q, py = uq / (unsigned)A.poolW, px = uq - py * (unsigned)A.poolW;
          gqv[u] = ldraw(A.gq + ((nb >> 2) + (long long)(py >> 1) * (A.poolW >> 1) + (px >> 1)) * A.gqs + A.gqo + c);
        } else {
          gqv[u] = egne_u32x4{0u, 0u, 0u, 0u};
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long long q = p + ROWS * u;
        const egne_fv<N> tf = unpackv(tag, t[u]), yf = unpackv(tag, yy[u]), af = unpackv(tag, a1v[u]), gf = unpackv(tag, gqv[u]);
        egne_fv<N> o;
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const float xh = yf.v[e] * sc.v[e] + sh.v[e];
          float ge = 0.25f * gf.v[e];
          if (A.act_q == EGNE_ACT_LEAKY) ge = xh > 0.f ? ge : 0.01f * ge;
          else if (A.act_q == EGNE_ACT_RELU) ge = xh > 0.f ? ge : 0.f;
          float r = tf.v[e] + k0[e] * (af.v[e] + ge) - k1[e] - xh * k2[e];
          if (act == EGNE_ACT_LEAKY) r = yf.v[e] > 0.f ? r : 0.01f * r;
          else if (act == EGNE_ACT_RELU) r = yf.v[e] > 0.f ? r : 0.f;
          o.v[e] = okk[u] ? r : 0.f;
        }
        if (okk[u]) stv(g + (nb + q) * gs + go + c, o);
#pragma unroll
        for (int e = 0; e < N; ++e) s[e] += o.v[e];
      }
    }
  }
  __shared__ double sh[ROWS][32];
#pragma unroll
  for (int e = 0; e < N; ++e) sh[row][v * N + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 32) {
    double a = 0;
    for (int r = 0; r < ROWS; ++r) a += sh[r][threadIdx.x];
    const int cc = cg * 32 + threadIdx.x;
    if (cc < Cp) ws[((long long)n * nps + chunk) * Cp + cc] = a;
  }
}

// one block per (32 channels, n), 32 partial-sum streams per channel, fixed combination order (see norm_stats_final)
__global__ __launch_bounds__(1024) void norm_bwd_final(const double* __restrict__ ws, int Cp, int Bn, int nchunk, float* __restrict__ sums,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
  const int n = blockIdx.y, cl = threadIdx.x & 31, c = blockIdx.x * 32 + cl, kg = threadIdx.x >> 5;
  double a = 0, b = 0;
  if (c < Cp) {
    const double* w = ws + ((long long)n * nchunk * Cp + c) * 2;
    for (int k = kg; k < nchunk; k += 32) {
      const double2 v = *(const double2*)(w + (long long)k * Cp * 2);
      a += v.x; b += v.y;
    }
  }
  __shared__ double sh[32][32][2];
  sh[kg][cl][0] = a; sh[kg][cl][1] = b;
  __syncthreads();
  if (kg == 0 && c < Cp) {
    for (int g = 1; g < 32; ++g) { a += sh[g][cl][0]; b += sh[g][cl][1]; }
    const int i = n * Cp + c;
    sums[2 * i] = (float)a; sums[2 * i + 1] = (float)b;
    if (dgamma && c < C) { dgamma[c] += (float)b; dbeta[c] += (float)a; }   // Bn == 1 for BatchNorm
  }
}

template <typename T>
__global__ void norm_bwd_apply(const T* __restrict__ x, long long xs, int xo, const float* __restrict__ scale,
                               const float* __restrict__ shift, const float* __restrict__ gamma,
                               const T* __restrict__ gy, long long gs, int go, int act_in, int Cp,
                               long long npix_per_n, int Bn, int per_sample, const float* __restrict__ sums,
                               T* __restrict__ gx, long long gxs, int gxo, int poolW, int accumulate) {
  constexpr int N = egne_vt<T>::N;
  // grid.y = sample (1 for batch statistics): inside a sample every index fits 32 bits (checked by the entry point) -- the flat
  // 64-bit index cost three 64-bit divisions per vector, most of the instructions of the pooled variant
  const unsigned nv = (unsigned)(Cp / N);
  const unsigned total = (unsigned)npix_per_n * nv;
  const float invN = 1.f / (float)npix_per_n;
  const int n = blockIdx.y;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned p = i / nv;
    const int c = (int)(i - p * nv) * N;
    const long long pp = (long long)n * npix_per_n + p;               // global pixel index over (n, p)
    const egne_fv<N> sc = ldf<N>(scale + (long long)n * Cp + c);
    const egne_fv<N> sh = ldf<N>(shift + (long long)n * Cp + c);
    const egne_fv<N> xv = ldv(x + pp * xs + xo + c);
    egne_fv<N> g;
    float gsc = 1.f;
    if (poolW) {      // per_sample mode, even H and W (checked by the entry point)
      const unsigned py = p / (unsigned)poolW, px = p - py * (unsigned)poolW;
      g = ldv(gy + (((long long)n * npix_per_n >> 2) + (long long)(py >> 1) * (poolW >> 1) + (px >> 1)) * gs + go + c);
      gsc = 0.25f;
    } else {
      g = ldv(gy + pp * gs + go + c);
    }
    T* dst = gx + pp * gxs + gxo + c;
    egne_fv<N> o = fv_fill<N>(0.f);
    if (accumulate) o = ldv(dst);          // (the first writer of a gradient slice stores: no read)
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float xh = xv.v[e] * sc.v[e] + sh.v[e];
      float ge = g.v[e] * gsc;
      if (act_in == EGNE_ACT_LEAKY) ge = xh > 0.f ? ge : 0.01f * ge;
      else if (act_in == EGNE_ACT_RELU) ge = xh > 0.f ? ge : 0.f;
      const float gm = gamma ? gamma[c + e] : 1.f;
      const float m1 = sums[2 * ((long long)n * Cp + c + e)] * invN, m2 = sums[2 * ((long long)n * Cp + c + e) + 1] * invN;
      o.v[e] += sc.v[e] * gm * (ge - m1 - xh * m2);
    }
    stv(dst, o);
  }
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void avgpool2_bwd_k(const T* __restrict__ gy, long long gs, int go, T* __restrict__ gx, long long xs,
                               int xo, int B, int H, int W, int Cp) {
  const int Ho = H >> 1, Wo = W >> 1, nv = Cp >> 2;
  const long long total = (long long)B * Ho * Wo * nv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % nv) * 4;
    long long p = i / nv;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    f32x4 g = ld4(gy + (((long long)b * Ho + oy) * Wo + ox) * gs + go + c);
    g = g * 0.25f;
    T* d = gx + (((long long)b * H + 2 * oy) * W + 2 * ox) * xs + xo + c;
    st4(d, ld4(d) + g); st4(d + xs, ld4(d + xs) + g);
    st4(d + (long long)W * xs, ld4(d + (long long)W * xs) + g); st4(d + (long long)W * xs + xs, ld4(d + (long long)W * xs + xs) + g);
  }
}

// transpose of F.interpolate(bilinear, x2, align_corners=False): gather over the <=4x4 output pixels
__device__ __forceinline__ int up_taps(int y, int H, int* oy, float* w) {
  int n = 0;
  if (y >= 1) { oy[n] = 2 * y - 1; w[n++] = 0.25f; }
  oy[n] = 2 * y; w[n++] = (y == 0) ? 1.0f : 0.75f;
  oy[n] = 2 * y + 1; w[n++] = (y == H - 1) ? 1.0f : 0.75f;
  if (y + 1 <= H - 1) { oy[n] = 2 * y + 2; w[n++] = 0.25f; }
  return n;
}
// grid (tiles over W * Cp/N, H, B): one 16-byte channel vector of one input pixel per thread, no 64-bit division per element
// (the flat-index form spent four of them per 8-byte vector: 459 us for the 120x160 -> 240x320 block)
template <typename T, bool ACC = true>
__global__ void upsample2x_bwd_k(const T* __restrict__ gy, long long gs, int go, T* __restrict__ gx, long long xs,
                                 int xo, int B, int H, int W, int Cp) {
  constexpr int N = egne_vt<T>::N;
  const unsigned nv = (unsigned)(Cp / N), t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)W * nv) return;
  const int x = (int)(t / nv), c = (int)(t - (unsigned)x * nv) * N;
  const int y = blockIdx.y, b = blockIdx.z;
  const int Wo = 2 * W, Ho = 2 * H;
  int oys[4], oxs[4]; float wy[4], wx[4];
  const int ny = up_taps(y, H, oys, wy), nx = up_taps(x, W, oxs, wx);
  egne_fv<N> acc = fv_fill<N>(0.f);
  for (int a = 0; a < ny; ++a)
    for (int q = 0; q < nx; ++q) {
      const egne_fv<N> g = ldv(gy + (((long long)b * Ho + oys[a]) * Wo + oxs[q]) * gs + go + c);
      const float w = wy[a] * wx[q];
#pragma unroll
      for (int e = 0; e < N; ++e) acc.v[e] += w * g.v[e];
    }
  T* dp = gx + (((long long)b * H + y) * W + x) * xs + xo + c;
  if constexpr (ACC) {
    const egne_fv<N> o = ldv(dp);
#pragma unroll
    for (int e = 0; e < N; ++e) acc.v[e] += o.v[e];
  }
  stv(dp, acc);
}

// transpose of the nearest-neighbour x2 up-sampling: gx[y][x] += the four output pixels that copied it
template <typename T>
__global__ void upsample2x_nearest_bwd_k(const T* __restrict__ gy, long long gs, int go, T* __restrict__ gx, long long xs, int xo, int B, int H, int W, int Cp) {
  const int nv = Cp >> 2, Wo = 2 * W, Ho = 2 * H;
  const long long total = (long long)B * H * W * nv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % nv) * 4;
    long long p = i / nv;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    const T* s = gy + (((long long)b * Ho + 2 * y) * Wo + 2 * x) * gs + go + c;
    const f32x4 acc = (ld4(s) + ld4(s + gs)) + (ld4(s + (long long)Wo * gs) + ld4(s + (long long)Wo * gs + gs));
    T* dp = gx + (((long long)b * H + y) * W + x) * xs + xo + c;
    st4(dp, ld4(dp) + acc);
  }
}

template <typename T>
__global__ void head_act_bwd_k(T* __restrict__ g, const T* __restrict__ y, int B, int ld) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 10) return;
  const int b = i / 10, j = i - b * 10, k = j % 5;
  const float v = ld1(y + (long long)b * ld + j);
  float d = 1.f;
  if (k < 2) d = 1.f - v * v; else if (k < 4) d = v * (1.f - v);
  st1(g + (long long)b * ld + j, ld1(g + (long long)b * ld + j) * d);
}

template <typename T>
__global__ void selu_bwd_k(T* __restrict__ g, const T* __restrict__ y, long long n) {
  const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    { const float yv = ld1(y + i); st1(g + i, ld1(g + i) * (yv > 0.f ? scale : yv + scale * alpha)); }
}

template <typename T>
__global__ void spatial_mean_bwd_k(const T* __restrict__ g, int gld, T* __restrict__ gx, long long xs, int xo,
                                   int C, int HW) {
  const int b = blockIdx.x;
  const float inv = 1.f / (float)HW;
  for (long long i = threadIdx.x; i < (long long)HW * C; i += blockDim.x) {
    const int c = (int)(i % C);
    const long long p = i / C;
    { T* q = gx + ((long long)b * HW + p) * xs + xo + c; st1(q, ld1(q) + ld1(g + (long long)b * gld + c) * inv); }
  }
}

// d(weight * mean|softmax(x) - 1/C|)/dx  (loss.py:150) or d CE/dx (:153)
template <typename T>
__global__ void conf_loss_bwd_k(const T* __restrict__ x, int ld, const long long* __restrict__ gt, int B, int C, int flag,
                                const float* __restrict__ gscale_p, T* __restrict__ gx, int gld) {
  const float gscale = gscale_p[0];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const T* r = x + (long long)b * ld;
  float m = -INFINITY;
  for (int k = 0; k < C; ++k) m = fmaxf(m, ld1(r + k));
  float se = 0.f;
  for (int k = 0; k < C; ++k) se += expf(ld1(r + k) - m);
  if (flag) {
    float dot = 0.f;
    for (int k = 0; k < C; ++k) { const float p = expf(ld1(r + k) - m) / se; dot += p * sgn(p - 1.0f / C); }
    for (int k = 0; k < C; ++k) {
      const float p = expf(ld1(r + k) - m) / se;
      st1(gx + (long long)b * gld + k, gscale / (float)(B * C) * p * (sgn(p - 1.0f / C) - dot));
    }
  } else {
    for (int k = 0; k < C; ++k)
      st1(gx + (long long)b * gld + k, gscale / (float)B * (expf(ld1(r + k) - m) / se - (gt[b] == k ? 1.f : 0.f)));
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient: dW[tap][co][k] = sum over pixels gz[m][co] * xin[m shifted by tap][k]  (k over the padded
// concatenated input channels, with the same fused load transform as the forward conv).
// One workgroup = one 32(co) x 32(k) tile over a pixel range; the 4 waves split each 32-pixel chunk
// (K-split) and are reduced through LDS at the end.  Partials go to ws[split][tap][CoutP][Ktot].
// ------------------------------------------------------------------------------------------------
constexpr int WPX = 128;  // pixels per chunk (two barriers per 16 MFMAs of every wave)
constexpr int WLD = 33;   // LDS row pitch (floats): lanes read consecutive floats of one pixel row
constexpr int WNR = WPX / 32;

// BFM (bf16 tensors): the contraction over pixels on v_mfma_f32_32x32x16_bf16 -- both tiles staged as plain [pixel][32] bf16 rows and
// read back transposed ("eight consecutive pixels of one channel") with ds_read_b64_tr_b16, as wgrad_bf16.hip does: two MFMAs of 32
// cycles per wave and chunk instead of sixteen of 64.  The generic form serves the reflect-padded 7x7 and the 4x4 / stride-2
// convolutions of the StyleEncoder (RITnet_v2.py:91-107), where it was 35 % of a configs[3] training step.
typedef __attribute__((address_space(3))) egne_bf16x4* wg_lds_bf4_ptr;
// FOLD (one slice of 8 padded channels, one group): a column tile holds FOUR taps x 8 channels instead of one tap x 32 channels of
// which 8 exist -- a quarter of the tiles, each reading gz once for four taps (the 7x7 on three channels: 13 x 2 tiles instead of 49 x 2).
template <typename TS, bool BFM = false, bool FOLD = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const egne_conv_desc p, const TS* __restrict__ gz, long long gzs,
                                                         int gzo, int nsplit, float* __restrict__ ws) {
  static_assert(!BFM || sizeof(TS) == 2, "bf16 MFMA form: bf16 tensors");
  __shared__ __attribute__((aligned(16))) float As[WPX * WLD];   // gz chunk   [pixel][co]
  __shared__ __attribute__((aligned(16))) float Bs[WPX * WLD];   // x chunk    [pixel][k]
  __shared__ float red[4][16][64];
  egne_bf16* const Ah = (egne_bf16*)As;      // BFM: [pixel][32] bf16, 64-byte rows
  egne_bf16* const Bh = (egne_bf16*)Bs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int T = p.kh * p.kw;
  // column tile -> (group, tap, segment, c0)
  int ct = blockIdx.z;
  int seg = 0, kofs = 0, g = 0, tap = 0, c0 = 0;
  {
    int per_tap = 0;
    for (int s = 0; s < p.nseg; ++s) per_tap += (p.seg[s].Cp + 31) / 32;
    g = ct / (per_tap * T); ct -= g * per_tap * T;
    tap = ct / per_tap; ct -= tap * per_tap;
    for (seg = 0; seg < p.nseg; ++seg) {
      const int n = (p.seg[seg].Cp + 31) / 32;
      if (ct < n) break;
      ct -= n; kofs += p.seg[seg].Cp;
    }
    c0 = ct * 32;
  }
  if constexpr (FOLD) { seg = 0; kofs = 0; g = 0; c0 = 0; tap = 4 * (int)blockIdx.z + ((tid & 7) >> 1); }      // (this thread's tap of the group)
  const egne_seg sg = p.seg[seg];
  const int co0 = blockIdx.y * 32;
  const long long M = (long long)p.B * p.Ho * p.Wo;
  const long long per = ((M + nsplit - 1) / nsplit + WPX - 1) / WPX * WPX;
  const long long m_begin = (long long)blockIdx.x * per, m_end = m_begin + per < M ? m_begin + per : M;
  const int dil = p.dil[g];
  const int ky = tap / p.kw, kx = tap - ky * p.kw;
  const int dy = (ky - p.pad_h) * dil, dx = (kx - p.pad_w) * dil;
  // 1x1 / stride 1 / no padding: the input pixel of output pixel m is pixel m (no div / mod per row)
  const bool simple = T == 1 && p.stride == 1 && p.pad_h == 0 && p.pad_w == 0 && p.H == p.Ho && p.W == p.Wo;
  const float slope_in = sg.act_in == EGNE_ACT_RELU ? 0.f : (sg.act_in == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  const int hw = p.Ho * p.Wo;

  // loader: thread -> (pixel rows tid>>3 + 32*i, float4 column = tid&7) of both tiles
  const int lr = tid >> 3, lc = (tid & 7) * 4;
  const int xc = FOLD ? ((tid & 7) & 1) * 4 : c0 + lc;         // first of this thread's four input channels
  const bool aok = co0 + lc < p.Cout_store, bok = FOLD ? tap < T : c0 + lc < sg.Cp;
  f32x16 acc = (f32x16)(0.f);
  // the loads of chunk c + 1 are requested before the MFMAs of chunk c (they used to be requested and waited for inside one chunk: a
  // full memory latency per 128 pixels, 937 times per workgroup for the StyleEncoder's first layer)
  f32x4 av[WNR], bv[WNR];
  int bb[WNR];
  // (frame, row, column) of this thread's rows, advanced by 128 pixels per chunk: the 64-bit division per row and chunk this replaces
  // was most of the kernel's instructions on small-channel layers
  int cb[WNR], cy[WNR], cx[WNR];
  {
#pragma unroll
    for (int i = 0; i < WNR; ++i) {
      const long long m = m_begin + lr + 32 * i;
      cb[i] = (int)(m / hw);
      const int r = (int)(m - (long long)cb[i] * hw);
      cy[i] = r / p.Wo; cx[i] = r - cy[i] * p.Wo;
    }
  }
  const int adv_y = WPX / p.Wo, adv_x = WPX - adv_y * p.Wo;
  auto issue = [&](long long mc) {
#pragma unroll
    for (int i = 0; i < WNR; ++i) {
      const long long m = mc + lr + 32 * i;
      const bool in = m < m_end;
      const TS* ap = (in && aok) ? gz + m * gzs + gzo + co0 + lc : zero_page<TS>();
      const TS* bp = zero_page<TS>();
      bb[i] = -1;
      const int b = cb[i], oy = cy[i], ox = cx[i];
      cx[i] += adv_x; cy[i] += adv_y;                     // the next chunk's coordinates
      if (cx[i] >= p.Wo) { cx[i] -= p.Wo; ++cy[i]; }
      while (cy[i] >= p.Ho) { cy[i] -= p.Ho; ++cb[i]; }
      if (simple) {
        if (in && bok) bp = (const TS*)sg.ptr + m * sg.pix_stride + sg.ch_off + xc;
        if (sg.scale && in && bok) bb[i] = b;
      } else if (in && bok) {
        int iy = oy * p.stride + dy, ix = ox * p.stride + dx;
        bool ok = true;
        if (p.pad_mode == 1) {
          iy = iy < 0 ? -iy : (iy >= p.H ? 2 * p.H - 2 - iy : iy);
          ix = ix < 0 ? -ix : (ix >= p.W ? 2 * p.W - 2 - ix : ix);
        } else {
          ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        }
        if (ok) {
          bp = (const TS*)sg.ptr + (((long long)b * p.H + iy) * p.W + ix) * sg.pix_stride + sg.ch_off + xc;
          bb[i] = b;
        }
      }
      av[i] = ld4(ap);
      bv[i] = ld4(bp);
    }
  };
  if (m_begin < m_end) issue(m_begin);
  for (long long mc = m_begin; mc < m_end; mc += WPX) {
    if (sg.scale) {
#pragma unroll
      for (int i = 0; i < WNR; ++i) {
        const bool ok = bb[i] >= 0;
        const f32x4 sc = *(const f32x4*)(ok ? sg.scale + (long long)bb[i] * sg.Cp + xc : egne_zero_page);
        const f32x4 sh = *(const f32x4*)(ok ? sg.shift + (long long)bb[i] * sg.Cp + xc : egne_zero_page);
        f32x4 v = bv[i] * sc + sh;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope_in);
        bv[i] = ok ? v : (f32x4)(0.f);
      }
    } else if (slope_in != 1.f) {
#pragma unroll
      for (int i = 0; i < WNR; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[i][e] = fmaxf(bv[i][e], bv[i][e] * slope_in);
    }
    __syncthreads();
    if constexpr (BFM) {
#pragma unroll
      for (int i = 0; i < WNR; ++i) {
        *(egne_bf16x4*)&Ah[(lr + 32 * i) * 32 + lc] = __builtin_convertvector(av[i], egne_bf16x4);      // (gz: exact, the values were bf16)
        *(egne_bf16x4*)&Bh[(lr + 32 * i) * 32 + lc] = __builtin_convertvector(bv[i], egne_bf16x4);
      }
      __syncthreads();
      if (mc + WPX < m_end) issue(mc + WPX);
      // wave w contracts pixels 32 w .. 32 w + 31 of the chunk in two 16-pixel steps.  Transposing read: 16-lane group g takes channels
      // 16 (g & 1) .. + 15 and the pixel octet g >> 1 of the step; lane 4 q + c of the group supplies pixel q, channels 4 c .. 4 c + 3
      const int g16 = lane >> 4, i16 = lane & 15;
      const int lbase = (8 * (g16 >> 1) + (i16 >> 2)) * 32 + 16 * (g16 & 1) + 4 * (i16 & 3);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const egne_bf16* ga = Ah + lbase + (32 * wave + 16 * s) * 32;
        const egne_bf16* xa = Bh + lbase + (32 * wave + 16 * s) * 32;
        const egne_bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((wg_lds_bf4_ptr)ga);
        const egne_bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((wg_lds_bf4_ptr)(ga + 4 * 32));
        const egne_bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((wg_lds_bf4_ptr)xa);
        const egne_bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((wg_lds_bf4_ptr)(xa + 4 * 32));
        const egne_bf16x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const egne_bf16x8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int i = 0; i < WNR; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { As[(lr + 32 * i) * WLD + lc + e] = av[i][e]; Bs[(lr + 32 * i) * WLD + lc + e] = bv[i][e]; }
    __syncthreads();
    if (mc + WPX < m_end) issue(mc + WPX);
    // wave w consumes pixel pairs 16w..16w+15 of the chunk: D[co][k] += A[co][px] * B[px][k]
#pragma unroll
    for (int s = 0; s < WPX / 8; ++s) {
      const int px = (wave * (WPX / 8) + s) * 2 + lh;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[px * WLD + li], Bs[px * WLD + li], acc, 0, 0, 0);
    }
    }
  }
  // cross-wave reduction; lane holds column k = li of rows co = (r&3) + 8*(r>>2) + 4*lh
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  if (wave == 0) {
    const int otap = FOLD ? 4 * (int)blockIdx.z + (li >> 3) : tap;       // FOLD: column li = (tap of the group, channel li & 7)
    float* dst = ws + (((long long)blockIdx.x * p.ngroups + g) * T + otap) * (long long)p.CoutP * p.Ktot;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
      const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int k = FOLD ? (li & 7) : c0 + li;
      if (FOLD ? otap < T : k < sg.Cp) dst[(long long)co * p.Ktot + kofs + k] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Generic weight gradient, bf16 tensors, WIDE form: a workgroup owns one (tap, 32-channel input chunk) column for FOUR 32-channel output
// blocks -- wave w contracts every pixel of a 128-pixel chunk for output block w (eight v_mfma_f32_32x32x16_bf16 per chunk, no
// cross-wave reduction), so an x tile staged once feeds four times the matrix work of the tile-per-workgroup form above, whose
// 16 KB of loads and two barriers bought two MFMAs per wave (the 4x4 / stride-2 blocks of the StyleEncoder, RITnet_v2.py:96-103, ran
// at 73 TFLOP/s).  Operands by ds_read_b64_tr_b16 from plain [pixel][32] bf16 rows; the next chunk's loads fly during the MFMAs.
// Plain inputs only (no fused affine / activation on load), 16-byte aligned slices.  Partials: ws[split][tap][CoutP][Ktot].
// FOLD (one slice of 8 padded channels, 64 output channels): a column tile holds four taps x 8 channels (as conv_wgrad_kernel<.., FOLD>);
// the workgroup takes TWO such tap groups and both output blocks -- wave w = (output block w & 1, tap group w >> 1).
// ------------------------------------------------------------------------------------------------------------------------
template <bool FOLD>
__global__ __launch_bounds__(256) void conv_wgrad_wide_kernel(const egne_conv_desc p, const egne_bf16* __restrict__ gz, long long gzs,
                                                              int gzo, int nsplit, float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) egne_bf16 Gh[4 * WPX * 32];     // [output block][pixel][32 co]
  __shared__ __attribute__((aligned(16))) egne_bf16 Xh[(FOLD ? 2 : 1) * WPX * 32];         // [tap group][pixel][32 k]
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.kh * p.kw;
  int ct = blockIdx.z, seg = 0, kofs = 0, tap = 0, c0 = 0;
  if constexpr (!FOLD) {
    int per_tap = 0;
    for (int s = 0; s < p.nseg; ++s) per_tap += (p.seg[s].Cp + 31) / 32;
    tap = ct / per_tap; ct -= tap * per_tap;
    for (seg = 0; seg < p.nseg; ++seg) {
      const int n = (p.seg[seg].Cp + 31) / 32;
      if (ct < n) break;
      ct -= n; kofs += p.seg[seg].Cp;
    }
    c0 = ct * 32;
  }
  const egne_seg sg = p.seg[seg];
  const egne_bf16* const xin = (const egne_bf16*)sg.ptr;
  const int cog = FOLD ? 0 : blockIdx.y * 128;
  const long long M = (long long)p.B * p.Ho * p.Wo;
  const long long per = ((M + nsplit - 1) / nsplit + WPX - 1) / WPX * WPX;
  const long long m_begin = (long long)blockIdx.x * per, m_end = m_begin + per < M ? m_begin + per : M;
  const int hw = p.Ho * p.Wo;
  // gz loader: rows (tid >> 3) + 32 i, 16-byte pieces (tid & 7) [and (tid & 7) + 8] of the row's 64 [128] output channels
  const int glr = tid >> 3, gpc = tid & 7;
  constexpr int GH = FOLD ? 1 : 2;
  // x loader: rows (tid >> 2) + 64 j, piece tid & 3 of the row's 32 columns: 8 input channels, or (FOLD) all 8 channels of one tap
  const int xlr = tid >> 2, xpc = tid & 3;
  constexpr int XG = FOLD ? 2 : 1;
  int dyv[XG], dxv[XG];
  bool xokv[XG];
#pragma unroll
  for (int q = 0; q < XG; ++q) {
    const int tp = FOLD ? 4 * (2 * (int)blockIdx.z + q) + xpc : tap;
    const int ky = tp / p.kw, kx = tp - ky * p.kw;
    dyv[q] = (ky - p.pad_h) * p.dil[0]; dxv[q] = (kx - p.pad_w) * p.dil[0];
    xokv[q] = FOLD ? tp < T : c0 + xpc * 8 < sg.Cp;
  }
  const int xch = FOLD ? 0 : c0 + xpc * 8;
  u32x4_ rg[4][GH], rx[2][XG];
  int cb[2], cy[2], cx[2];         // (frame, row, column) of this thread's two x rows, advanced by 128 pixels per chunk (no division per chunk)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long long m = m_begin + xlr + 64 * j;
    cb[j] = (int)(m / hw);
    const int r = (int)(m - (long long)cb[j] * hw);
    cy[j] = r / p.Wo; cx[j] = r - cy[j] * p.Wo;
  }
  const int adv_y = WPX / p.Wo, adv_x = WPX - adv_y * p.Wo;
  auto issue = [&](long long mc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long m = mc + glr + 32 * i;
#pragma unroll
      for (int h = 0; h < GH; ++h) {
        const int co = cog + (gpc + 8 * h) * 8;
        const egne_bf16* ap = (m < m_end && co < p.Cout_store) ? gz + m * gzs + gzo + co : zero_page<egne_bf16>();
        rg[i][h] = *(const u32x4_*)ap;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long long m = mc + xlr + 64 * j;
      const int b = cb[j], oy = cy[j], ox = cx[j];
      cx[j] += adv_x; cy[j] += adv_y;
      if (cx[j] >= p.Wo) { cx[j] -= p.Wo; ++cy[j]; }
      while (cy[j] >= p.Ho) { cy[j] -= p.Ho; ++cb[j]; }
#pragma unroll
      for (int q = 0; q < XG; ++q) {
        const egne_bf16* bp = zero_page<egne_bf16>();
        if (m < m_end && xokv[q]) {
          int iy = oy * p.stride + dyv[q], ix = ox * p.stride + dxv[q];
          bool ok = true;
          if (p.pad_mode == 1) {
            iy = iy < 0 ? -iy : (iy >= p.H ? 2 * p.H - 2 - iy : iy);
            ix = ix < 0 ? -ix : (ix >= p.W ? 2 * p.W - 2 - ix : ix);
          } else {
            ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
          }
          if (ok) bp = xin + (((long long)b * p.H + iy) * p.W + ix) * sg.pix_stride + sg.ch_off + xch;
        }
        rx[j][q] = *(const u32x4_*)bp;
      }
    }
  };
  const int g16 = lane >> 4, i16 = lane & 15;
  const int lbase = (8 * (g16 >> 1) + (i16 >> 2)) * 32 + 16 * (g16 & 1) + 4 * (i16 & 3);      // transposing read, as conv_wgrad_kernel<BFM>
  f32x16 acc = (f32x16)(0.f);
  if (m_begin < m_end) issue(m_begin);
  for (long long mc = m_begin; mc < m_end; mc += WPX) {
    __syncthreads();                 // every wave is done with the previous chunk's tiles
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int h = 0; h < GH; ++h) {
        const int pc = gpc + 8 * h;
        *(u32x4_*)&Gh[((pc >> 2) * WPX + glr + 32 * i) * 32 + (pc & 3) * 8] = rg[i][h];
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < XG; ++q) *(u32x4_*)&Xh[(q * WPX + xlr + 64 * j) * 32 + xpc * 8] = rx[j][q];
    __syncthreads();
    if (mc + WPX < m_end) issue(mc + WPX);
    const egne_bf16* ga = Gh + (FOLD ? (wave & 1) : wave) * WPX * 32 + lbase;
    const egne_bf16* xa = Xh + (FOLD ? (wave >> 1) : 0) * WPX * 32 + lbase;
#pragma unroll
    for (int s = 0; s < WPX / 16; ++s) {
      const egne_bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((wg_lds_bf4_ptr)(ga + 16 * s * 32));
      const egne_bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((wg_lds_bf4_ptr)(ga + (16 * s + 4) * 32));
      const egne_bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((wg_lds_bf4_ptr)(xa + 16 * s * 32));
      const egne_bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((wg_lds_bf4_ptr)(xa + (16 * s + 4) * 32));
      const egne_bf16x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      const egne_bf16x8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
  }
  // lane holds column k = lane & 31 of rows co = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of this wave's output block
  const int cob = FOLD ? 32 * (wave & 1) : cog + 32 * wave;
  const int kcol = lane & 31;
  const int otap = FOLD ? 4 * (2 * (int)blockIdx.z + (wave >> 1)) + (kcol >> 3) : tap;
  const int k = FOLD ? (kcol & 7) : c0 + kcol;
  if (cob < p.CoutP && (FOLD ? otap < T : k < sg.Cp)) {
    float* dst = ws + ((long long)blockIdx.x * T + otap) * (long long)p.CoutP * p.Ktot;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cob + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      dst[(long long)co * p.Ktot + kofs + k] = acc[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// 1x1 / stride 1 weight gradient with every (32 co, 32 k) block of the layer in ONE workgroup (dense-block conv21 / conv31,
// decoder conv11 / conv21, Transition_down: models/RITnet_v2.py:38,59,61,86).  The tile-per-workgroup kernel above restages gz for
// every input chunk and x for every output block (12 blocks: 24 tile loads per pixel chunk where 8 would do, L2-bound at ~30
// TFLOP/s).  Here a workgroup walks its pixel range in chunks of CH pixels, stages all nco gz tiles and all nkc x tiles of the chunk
// once ([pixel][36] floats each, the fused InstanceNorm affine + activation applied while staging), and its four waves share
// the (co block, k chunk) pairs: wave w owns pairs w, w+4, ... for ALL pixels, so there is no cross-wave reduction and every HBM
// byte is read once.  Exact fp32 (v_mfma_f32_32x32x2_f32).  Partials: ws[split][CoutP][Ktot] as for the other forms.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int W1_MAXT = 12;      // tiles (gz + x) of a chunk
constexpr int W1LD = 36;         // LDS row pitch (floats): 16-byte stores, conflict-free operand reads
struct W1Tab { short seg[W1_MAXT]; short c0[W1_MAXT]; short kofs[W1_MAXT]; };

template <int PPW, int CH, typename T>
__global__ __launch_bounds__(256) void conv1x1_wgrad_allpairs_kernel(const egne_conv_desc p, const T* __restrict__ gz, long long gzs,
                                                                     int gzo, int nsplit, int nco, int nkc, W1Tab tab,
                                                                     float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float w1lds[];     // [nco + nkc][CH][W1LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nt = nco + nkc, npairs = nco * nkc;
  const long long M = (long long)p.B * p.Ho * p.Wo;
  const long long per = ((M + nsplit - 1) / nsplit + CH - 1) / CH * CH;
  const long long m_begin = (long long)blockIdx.x * per, m_end = m_begin + per < M ? m_begin + per : M;
  const int hw = p.Ho * p.Wo;
  const int lr = tid >> 3, lc = (tid & 7) * 4;
  constexpr int NR = CH / 32;       // pixel rows of a tile per thread
  f32x16 acc[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) acc[i] = (f32x16)(0.f);

  f32x4 v[W1_MAXT][NR];
  auto load_chunk = [&](long long mc) {
#pragma unroll
    for (int j = 0; j < W1_MAXT; ++j) {
      if (j < nt) {
        const bool isg = j < nco;
        const egne_seg& sg = p.seg[isg ? 0 : tab.seg[j]];
        const int c = (isg ? j * 32 : tab.c0[j]) + lc;
        const bool cok = isg ? c < p.Cout_store : c < sg.Cp;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const long long m = mc + lr + 32 * i;
          const bool ok = cok && m < m_end;
          const T* ptr = !ok ? zero_page<T>() : (isg ? gz + m * gzs + gzo + c : (const T*)sg.ptr + m * sg.pix_stride + sg.ch_off + c);
          v[j][i] = ld4(ptr);
        }
      }
    }
  };
  if (m_begin < m_end) load_chunk(m_begin);
  for (long long mc = m_begin; mc < m_end; mc += CH) {
    // fused load transform of the x tiles (as in the forward convolution)
#pragma unroll
    for (int j = 0; j < W1_MAXT; ++j) {
      if (j >= nco && j < nt) {
        const egne_seg& sg = p.seg[tab.seg[j]];
        const float slope_in = sg.act_in == EGNE_ACT_RELU ? 0.f : (sg.act_in == EGNE_ACT_LEAKY ? 0.01f : 1.f);
        const int c = tab.c0[j] + lc;
        if (sg.scale) {
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            const long long m = mc + lr + 32 * i;
            const bool ok = c < sg.Cp && m < m_end;
            const int bb = ok ? (int)(m / hw) : 0;
            const f32x4 sc = *(const f32x4*)(ok ? sg.scale + (long long)bb * sg.Cp + c : egne_zero_page);
            const f32x4 sh = *(const f32x4*)(ok ? sg.shift + (long long)bb * sg.Cp + c : egne_zero_page);
            f32x4 t = v[j][i] * sc + sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = fmaxf(t[e], t[e] * slope_in);
            v[j][i] = ok ? t : (f32x4)(0.f);
          }
        } else if (slope_in != 1.f) {
#pragma unroll
          for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][i][e] = fmaxf(v[j][i][e], v[j][i][e] * slope_in);
        }
      }
    }
    __syncthreads();      // every wave is done with the previous chunk's tiles
#pragma unroll
    for (int j = 0; j < W1_MAXT; ++j)
      if (j < nt) {
#pragma unroll
        for (int i = 0; i < NR; ++i) *(f32x4*)&w1lds[(j * CH + lr + 32 * i) * W1LD + lc] = v[j][i];
      }
    __syncthreads();
    if (mc + CH < m_end) load_chunk(mc + CH);       // next chunk's loads fly during this chunk's MFMAs
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int q = wave + 4 * i;
      if (q < npairs) {
        const int kc = q / nco, ct = q - kc * nco;
        const float* As = w1lds + (ct * CH) * W1LD;
        const float* Bs = w1lds + ((nco + kc) * CH) * W1LD;
#pragma unroll 8
        for (int s = 0; s < CH / 2; ++s)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(2 * s + lh) * W1LD + li], Bs[(2 * s + lh) * W1LD + li], acc[i], 0, 0, 0);
      }
    }
  }
  // lane holds column k = li of rows co = (r&3) + 8*(r>>2) + 4*lh of its pairs' blocks
  float* dst = ws + (long long)blockIdx.x * p.ngroups * (long long)p.CoutP * p.Ktot;
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int q = wave + 4 * i;
    if (q < npairs) {
      const int kc = q / nco, ct = q - kc * nco;
      const int j = nco + kc;
      const int k = tab.c0[j] + li;
      if (k < p.seg[tab.seg[j]].Cp) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          dst[(long long)co * p.Ktot + tab.kofs[j] + k] = acc[i][r];
        }
      }
    }
  }
}

// sum the split partials and add into the OIHW gradient of group g
// sum the split partials and add into the OIHW gradient of group g: a block = 32 elements x 8 interleaved split ranges,
// combined through LDS in a fixed order (deterministic)
// clean: the partial sums are cleared as they are read -- the forms whose kernels do not write every partial (1x1 over slices, the
// generic implicit GEMM) need a zero-filled workspace, and clearing here replaces a fill in front of every such launch
__global__ __launch_bounds__(256) void wgrad_reduce_k(float* __restrict__ ws, int nsplit, int G, int g, int T, int Cout, int Cin,
                                                      const int* __restrict__ kinv, int CoutP, int Ktot, float* __restrict__ gw, int clean) {
  __shared__ float part[8][32];
  const long long total = (long long)T * CoutP * Ktot;
  const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
  for (long long base = (long long)blockIdx.x * 32; base < total; base += (long long)gridDim.x * 32) {
    const long long i = base + c;
    float s = 0.f;
    if (i < total)
      for (int sp = q; sp < nsplit; sp += 8) {
        float* e = ws + ((long long)sp * G + g) * total + i;
        s += *e;
        if (clean) *e = 0.f;
      }
    __syncthreads();
    part[q][c] = s;
    __syncthreads();
    if (q == 0 && i < total) {
      const int k = (int)(i % Ktot);
      const long long r = i / Ktot;
      const int n = (int)(r % CoutP);
      const int t = (int)(r / CoutP);
      const int ci = kinv[k];
      if (n < Cout && ci >= 0) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += part[j][c];
        gw[((long long)n * Cin + ci) * T + t] += v;
      }
    }
  }
}


// dgrad pack: forward OIHW w -> weights of the transposed conv for input channels [ci0, ci0+Cpiece):
// out[tap'][n = ci - ci0][k = co] = w[co][ci][T-1-tap'], flat ([tap][CoutP'][Ktot']) or fragment order.
__global__ void pack_weight_dgrad_k(const float* __restrict__ w, int Cout, int Cin, int T, int ci0, int Cpiece, int CoutPp,
                                    int Ktotp, int frag, float* __restrict__ out) {
  const long long total = (long long)T * CoutPp * Ktotp;
  const int NT = CoutPp >> 5;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int t, n, k;
    if (frag) {
      const int e = (int)(i & 3), nn = (int)((i >> 2) & 31), h = (int)((i >> 7) & 1);
      long long q = i >> 8;
      const int nt = (int)(q % NT); q /= NT;
      const int kg = (int)(q % (Ktotp >> 3));
      t = (int)(q / (Ktotp >> 3));
      n = nt * 32 + nn; k = kg * 8 + h * 4 + e;
    } else {
      k = (int)(i % Ktotp);
      long long q = i / Ktotp;
      n = (int)(q % CoutPp); t = (int)(q / CoutPp);
    }
    out[i] = (n < Cpiece && k < Cout) ? w[((long long)k * Cin + ci0 + n) * T + (T - 1 - t)] : 0.f;
  }
}

inline bool slice_ok(const void* p, long long stride, int off, int Cp) {
  return p && ((uintptr_t)p & 15) == 0 && stride % 4 == 0 && off % 4 == 0 && Cp > 0 && Cp % 4 == 0 && off + Cp <= stride;
}
template <typename T> inline bool vec_ok(long long stride, int off, int Cp) {     // 16-byte vectors of T
  constexpr int N = egne_vt<T>::N;
  return stride % N == 0 && off % N == 0 && Cp % N == 0;
}
int chunks_for(long long npix, int Cp, int Bn) {
  const int cgroups = (Cp + 31) / 32;
  // ~4096 workgroups of four waves: sixteen per CU (256 chunks of a 32-channel tensor = one workgroup per CU ran at 2.7 TB/s)
  long long want = 4096 / ((long long)Bn * cgroups);
  if (want < 1) want = 1;
  long long maxchunk = npix / 256 > 0 ? npix / 256 : 1;
  long long n = want < maxchunk ? want : maxchunk;
  return (int)(n > 1024 ? 1024 : n);
}

}  // namespace

template <typename T>
static int loss_bwd_impl(const egne_loss_desc* dp, const float* gscale, T* g_logits, int64_t gs, int go, float* g_elOut, void* stream) {
  EGNE_REQUIRE(dp && g_logits && g_elOut && gscale, "loss_bwd: null pointer");
  const egne_loss_desc& d = *dp;
  EGNE_REQUIRE(d.coef && d.grid_x && d.grid_y && d.out_terms, "loss_bwd: forward state (coef/grid) missing");
  EGNE_REQUIRE(go + 3 <= gs, "loss_bwd: bad gradient slice");
  const int HW = d.H * d.W;
  int gx = (HW + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(loss_bwd_k<T>, dim3(gx, d.B), dim3(256), 0, (hipStream_t)stream, d, gscale, g_logits, (long long)gs, go, g_elOut);
  return egne::check_launch("egne_loss_bwd");
}
extern "C" int egne_loss_bwd(const egne_loss_desc* dp, const float* gscale, float* g_logits, int64_t gs, int go, float* g_elOut,
                             void* stream) {
  EGNE_REQUIRE(dp && dp->dtype == 0, "loss_bwd: descriptor says bf16 logits (use egne_loss_bwd_bf16)");
  return loss_bwd_impl(dp, gscale, g_logits, gs, go, g_elOut, stream);
}
/* logits (descriptor, dtype = 1) and their gradient in bf16 */
extern "C" int egne_loss_bwd_bf16(const egne_loss_desc* dp, const float* gscale, void* g_logits, int64_t gs, int go, float* g_elOut,
                                  void* stream) {
  EGNE_REQUIRE(dp && dp->dtype == 1, "loss_bwd_bf16: descriptor says fp32 logits");
  return loss_bwd_impl(dp, gscale, (egne_bf16*)g_logits, gs, go, g_elOut, stream);
}

extern "C" int64_t egne_act_bwd_bias_workspace_bytes(int64_t npix, int Cp) {
  return (int64_t)chunks_for(npix, Cp, 1) * Cp * sizeof(double);
}

template <typename T>
static int act_bwd_bias_impl(T* g, int64_t gs, int go, const T* y, int64_t ys, int yo, int act, int Cp,
                             int64_t npix, float* dbias, int C, int accumulate, void* ws, uint32_t* absmax_bits, void* stream) {
  EGNE_REQUIRE(slice_ok(g, gs, go, Cp) && npix > 0 && ws, "act_bwd_bias: bad gradient slice");
  EGNE_REQUIRE(vec_ok<T>(gs, go, Cp) && (act == EGNE_ACT_NONE || vec_ok<T>(ys, yo, Cp)), "act_bwd_bias: slices must be 16-byte vectors (8 bf16 channels)");
  EGNE_REQUIRE(act == EGNE_ACT_NONE || slice_ok(y, ys, yo, Cp), "act_bwd_bias: bad output slice");
  const int nchunk = chunks_for(npix, Cp, 1);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(act_bwd_bias_partial<T>, dim3(nchunk, (Cp + 31) / 32), dim3(256), 0, st, g, (long long)gs, go, y,
                     (long long)ys, yo, act, Cp, (long long)npix, nchunk, (double*)ws, (unsigned*)absmax_bits);
  if (dbias)
    hipLaunchKernelGGL(reduce_chunks_k, dim3((Cp + 31) / 32), dim3(1024), 0, st, (const double*)ws, Cp, C < Cp ? C : Cp,
                       nchunk, dbias, accumulate);
  return egne::check_launch("egne_act_bwd_bias");
}
extern "C" int egne_act_bwd_bias_absmax(float* g, int64_t gs, int go, const float* y, int64_t ys, int yo, int act, int Cp,
                                        int64_t npix, float* dbias, int C, int accumulate, void* ws, uint32_t* absmax_bits,
                                        void* stream) {
  return act_bwd_bias_impl(g, gs, go, y, ys, yo, act, Cp, npix, dbias, C, accumulate, ws, absmax_bits, stream);
}
extern "C" int egne_act_bwd_bias(float* g, int64_t gs, int go, const float* y, int64_t ys, int yo, int act, int Cp,
                                 int64_t npix, float* dbias, int C, int accumulate, void* ws, void* stream) {
  return act_bwd_bias_impl(g, gs, go, y, ys, yo, act, Cp, npix, dbias, C, accumulate, ws, (uint32_t*)nullptr, stream);
}
extern "C" int egne_act_bwd_bias_bf16(void* g, int64_t gs, int go, const void* y, int64_t ys, int yo, int act, int Cp,
                                      int64_t npix, float* dbias, int C, int accumulate, void* ws, void* stream) {
  return act_bwd_bias_impl((egne_bf16*)g, gs, go, (const egne_bf16*)y, ys, yo, act, Cp, npix, dbias, C, accumulate, ws, (uint32_t*)nullptr, stream);
}

// ------------------------------------------------------------------------------------------------
// Bias gradient of the 1x1 in front of a 3x3 (the dense blocks' activation-free 'a' layer): db_a = sum_q g_tmp[q] with g_tmp the 3x3's
// data gradient.  That sum is linear in gz_b (the 3x3's masked output gradient):
//     sum_q g_tmp[q][c] = sum_{co,tap} W_b[co][c][tap] * S_tap[co],    S_tap[co] = sum of gz_b[.][co] over the pixels p with p + tap
// inside the image = the per-channel total T (which act_bwd_bias of the 3x3 has just left in its chunk workspace) minus one border
// row, one border column, plus their corner.  Two small launches replace a pass over the full-resolution g_tmp.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pair_border_k(const T* __restrict__ g, long long gs, int go, int Cs, int H, int W,
                                                     double* __restrict__ part) {       // part[b][8][Cs]
  // blockIdx.y = 0: first / last row (+ the four corners), 1: first / last column.  One 16-byte vector per lane and pixel.
  constexpr int N = egne_vt<T>::N;
  __shared__ float red[2 * 256 * N];
  const int cv = Cs / N, np = 256 / cv, v = threadIdx.x % cv, pt = threadIdx.x / cv, cols = blockIdx.y;
  const T* fb = g + (long long)blockIdx.x * H * W * gs + go;
  const long long step = cols ? (long long)W * gs : gs, last = cols ? (long long)(W - 1) * gs : (long long)(H - 1) * W * gs;
  const int n = cols ? H : W;
  egne_fv<N> a = fv_fill<N>(0.f), b = fv_fill<N>(0.f);
  if (pt < np) {
#pragma unroll 2
    for (int i = pt; i < n; i += np) {
      const egne_fv<N> u = ldv(fb + i * step + v * N), w = ldv(fb + i * step + last + v * N);
#pragma unroll
      for (int k = 0; k < N; ++k) { a.v[k] += u.v[k]; b.v[k] += w.v[k]; }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      red[(pt * 2) * Cs + v * N + k] = a.v[k];
      red[(pt * 2 + 1) * Cs + v * N + k] = b.v[k];
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < Cs) {
    const int co = threadIdx.x;
    double s0 = 0, s1 = 0;
    for (int q = 0; q < np; ++q) { s0 += red[(q * 2) * Cs + co]; s1 += red[(q * 2 + 1) * Cs + co]; }
    double* o = part + (long long)blockIdx.x * 8 * Cs + co;
    o[(2 * cols) * Cs] = s0;
    o[(2 * cols + 1) * Cs] = s1;
    if (!cols) {
      o[4 * Cs] = ld1(fb + co);
      o[5 * Cs] = ld1(fb + (long long)(W - 1) * gs + co);
      o[6 * Cs] = ld1(fb + (long long)(H - 1) * W * gs + co);
      o[7 * Cs] = ld1(fb + ((long long)H * W - 1) * gs + co);
    }
  }
}

__global__ __launch_bounds__(1024) void pair_bias_final_k(const double* __restrict__ ws, int nchunk, const double* __restrict__ part,
                                                          int B, int Cs, int Cout, int Ca, const float* __restrict__ w,
                                                          float* __restrict__ db_b, float* __restrict__ db_a) {
  __shared__ double red[32][32];
  __shared__ double Tl[256], E[8 * 256], S[9 * 256];
  const int tid = threadIdx.x, c = tid & 31, q = tid >> 5;
  for (int c0 = 0; c0 < Cs; c0 += 32) {                 // T = the 3x3's per-channel total, reduced as reduce_chunks_k does
    double s = 0;
    if (c0 + c < Cs) {
      const double* wc = ws + c0 + c;
      int k = q;
      for (; k + 224 < nchunk; k += 256) {              // eight independent loads in flight (the workspace is L2 / HBM latency bound)
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = wc[(long long)(k + 32 * j) * Cs];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += t[j];
      }
      for (; k < nchunk; k += 32) s += wc[(long long)k * Cs];
    }
    red[q][c] = s;
    __syncthreads();
    if (q == 0 && c0 + c < Cs) {
      double t = 0;
#pragma unroll
      for (int r = 0; r < 32; ++r) t += red[r][c];
      Tl[c0 + c] = t;
    }
    __syncthreads();
  }
  const int Ne = 8 * Cs, nr = Ne >= 1024 ? 1 : 1024 / Ne;     // border sums over the frames: nr interleaved frame ranges per element
  if (nr == 1) {
    for (int e = tid; e < Ne; e += 1024) {
      double s = 0;
#pragma unroll 4
      for (int b = 0; b < B; ++b) s += part[(long long)b * Ne + e];
      E[e] = s;
    }
  } else {
    const int r = tid / Ne, e = tid - r * Ne;
    if (r < nr) {
      double s = 0;
#pragma unroll 4
      for (int b = r; b < B; b += nr) s += part[(long long)b * Ne + e];
      S[r * Ne + e] = s;                                      // (S is free until the taps are formed below)
    }
    __syncthreads();
    if (tid < Ne) {
      double s = 0;
      for (int k = 0; k < nr; ++k) s += S[k * Ne + tid];
      E[tid] = s;
    }
  }
  __syncthreads();
  for (int i = tid; i < 9 * Cout; i += 1024) {
    const int tap = i / Cout, co = i - tap * Cout, dy = tap / 3 - 1, dx = tap % 3 - 1;
    double s = Tl[co];
    if (dy) s -= E[(dy > 0 ? 1 : 0) * Cs + co];
    if (dx) s -= E[(dx > 0 ? 3 : 2) * Cs + co];
    if (dy && dx) s += E[(4 + (dy > 0 ? 2 : 0) + (dx > 0 ? 1 : 0)) * Cs + co];
    S[tap * 256 + co] = s;
  }
  __syncthreads();
  if (db_b)
    for (int co = tid; co < Cout; co += 1024) db_b[co] += (float)Tl[co];
  // db_a[ca] = sum_{co,tap} w[co][ca][tap] * S[tap][co]: Ca lanes x (1024 / Ca) interleaved co ranges, summed in a fixed order
  double* acc = E;                                      // E is dead now (S holds what the products need); 1024 doubles fit its 2048
  const int np = 1024 / Ca, ca = tid % Ca, pt = tid / Ca;
  if (pt < np) {
    double s = 0;
    for (int co = pt; co < Cout; co += np) {
      const float* wr = w + ((long long)co * Ca + ca) * 9;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) s += (double)wr[tap] * S[tap * 256 + co];
    }
    acc[pt * Ca + ca] = s;
  }
  __syncthreads();
  if (pt == 0) {
    double s = 0;
    for (int r = 0; r < np; ++r) s += acc[r * Ca + ca];
    db_a[ca] += (float)s;
  }
}

extern "C" int64_t egne_pair_bias_bwd_workspace_bytes(int B, int Cp) { return (int64_t)B * 8 * Cp * sizeof(double); }

template <typename T>
static int pair_bias_impl(const T* g, int64_t gs, int go, int Cp, int B, int H, int W, const void* act_ws, const float* w, int Cout,
                          int Ca, float* db_b, float* db_a, void* ws, void* stream) {
  EGNE_REQUIRE(slice_ok(g, gs, go, Cp) && B > 0 && H > 0 && W > 0, "pair_bias_bwd: bad gradient slice");
  EGNE_REQUIRE(Cp <= 256 && Cout > 0 && Cout <= Cp && Ca > 0 && Ca <= 256, "pair_bias_bwd: at most 256 channels on either side of the 3x3");
  EGNE_REQUIRE(act_ws && w && db_a && ws && ((uintptr_t)ws & 7) == 0, "pair_bias_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int nchunk = chunks_for((long long)B * H * W, Cp, 1);
  EGNE_REQUIRE(vec_ok<T>(gs, go, Cp), "pair_bias_bwd: the slice must be 16-byte vectors (8 bf16 channels)");
  hipLaunchKernelGGL(pair_border_k<T>, dim3(B, 2), dim3(256), 0, st, g, (long long)gs, go, Cp, H, W, (double*)ws);
  hipLaunchKernelGGL(pair_bias_final_k, dim3(1), dim3(1024), 0, st, (const double*)act_ws, nchunk, (const double*)ws, B, Cp, Cout, Ca,
                     w, db_b, db_a);
  return egne::check_launch("egne_pair_bias_bwd");
}
// g: the 3x3's masked output gradient AFTER egne_act_bwd_bias(..., dbias = NULL, ws = act_ws) (same npix = B*H*W, same Cp);
// w: the 3x3's fp32 OIHW weights [Cout][Ca][3][3]; db_b (may be NULL) += T, db_a += the 1x1's bias gradient.
extern "C" int egne_pair_bias_bwd(const float* g, int64_t gs, int go, int Cp, int B, int H, int W, const void* act_ws, const float* w,
                                  int Cout, int Ca, float* db_b, float* db_a, void* ws, void* stream) {
  return pair_bias_impl(g, gs, go, Cp, B, H, W, act_ws, w, Cout, Ca, db_b, db_a, ws, stream);
}
extern "C" int egne_pair_bias_bwd_bf16(const void* g, int64_t gs, int go, int Cp, int B, int H, int W, const void* act_ws, const float* w,
                                       int Cout, int Ca, float* db_b, float* db_a, void* ws, void* stream) {
  return pair_bias_impl((const egne_bf16*)g, gs, go, Cp, B, H, W, act_ws, w, Cout, Ca, db_b, db_a, ws, stream);
}

extern "C" int64_t egne_norm_bwd_workspace_bytes(int B, int HW, int Cp, int per_sample) {
  const int Bn = per_sample ? B : 1;
  const long long npix = per_sample ? HW : (long long)B * HW;
  return (int64_t)Bn * chunks_for(npix, Cp, Bn) * Cp * 2 * sizeof(double);
}

template <typename T>
static int norm_bwd_impl(const T* x, int64_t xs, int xo, const float* scale, const float* shift,
                         const float* gamma, const T* gy, int64_t gs, int go, int act_in, int Cp, int B, int HW,
                         int per_sample, T* gx, int64_t gxs, int gxo, float* sums, float* dgamma, float* dbeta,
                         int C, void* ws, void* stream, int poolW, int accumulate) {
  EGNE_REQUIRE(slice_ok(x, xs, xo, Cp) && slice_ok(gy, gs, go, Cp) && slice_ok(gx, gxs, gxo, Cp), "norm_bwd: bad slices");
  EGNE_REQUIRE(vec_ok<T>(xs, xo, Cp) && vec_ok<T>(gs, go, Cp) && vec_ok<T>(gxs, gxo, Cp), "norm_bwd: slices must be 16-byte vectors (8 bf16 channels)");
  EGNE_REQUIRE(scale && shift && sums && ws && ((uintptr_t)ws & 15) == 0 && B > 0 && HW > 0, "norm_bwd: null pointer (ws must be 16-byte aligned)");
  EGNE_REQUIRE((dgamma == nullptr) == (dbeta == nullptr) && (!dgamma || !per_sample), "norm_bwd: dgamma/dbeta only for batch statistics");
  const int Bn = per_sample ? B : 1;
  const long long npix = per_sample ? HW : (long long)B * HW;
  const int nchunk = chunks_for(npix, Cp, Bn);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(norm_bwd_partial<T>, dim3(nchunk, (Cp + 31) / 32, Bn), dim3(256), 0, st, x, (long long)xs, xo, scale, shift,
                     gy, (long long)gs, go, act_in, Cp, npix, nchunk, per_sample, (double*)ws, poolW);
  hipLaunchKernelGGL(norm_bwd_final, dim3((Cp + 31) / 32, Bn), dim3(1024), 0, st, (const double*)ws, Cp, Bn, nchunk, sums,
                     dgamma, dbeta, C);
  EGNE_REQUIRE(npix * (Cp / egne_vt<T>::N) < (1ll << 32) && Bn <= 65535, "norm_bwd: more than 2^32 vectors per statistics group");
  const long long gxa = grid_for((long long)Bn * npix * (Cp / egne_vt<T>::N));
  hipLaunchKernelGGL(norm_bwd_apply<T>, dim3((unsigned)((gxa + Bn - 1) / Bn), Bn), dim3(256), 0, st, x, (long long)xs, xo,
                     scale, shift, gamma, gy, (long long)gs, go, act_in, Cp, npix, Bn, per_sample, sums, gx, (long long)gxs,
                     gxo, poolW, accumulate);
  return egne::check_launch("egne_norm_bwd");
}

extern "C" int egne_norm_bwd(const float* x, int64_t xs, int xo, const float* scale, const float* shift,
                             const float* gamma, const float* gy, int64_t gs, int go, int act_in, int Cp, int B, int HW,
                             int per_sample, float* gx, int64_t gxs, int gxo, float* sums, float* dgamma, float* dbeta,
                             int C, void* ws, void* stream) {
  return norm_bwd_impl(x, xs, xo, scale, shift, gamma, gy, gs, go, act_in, Cp, B, HW, per_sample, gx, gxs, gxo, sums, dgamma, dbeta,
                       C, ws, stream, 0, 1);
}
extern "C" int egne_norm_bwd_bf16(const void* x, int64_t xs, int xo, const float* scale, const float* shift,
                                  const float* gamma, const void* gy, int64_t gs, int go, int act_in, int Cp, int B, int HW,
                                  int per_sample, void* gx, int64_t gxs, int gxo, float* sums, float* dgamma, float* dbeta,
                                  int C, void* ws, void* stream) {
  return norm_bwd_impl((const egne_bf16*)x, xs, xo, scale, shift, gamma, (const egne_bf16*)gy, gs, go, act_in, Cp, B, HW, per_sample,
                       (egne_bf16*)gx, gxs, gxo, sums, dgamma, dbeta, C, ws, stream, 0, 1);
}

// The same with gx STORED instead of accumulated (first writer of a gradient slice: engine.Plan.first_touch).
extern "C" int egne_norm_bwd_store(const float* x, int64_t xs, int xo, const float* scale, const float* shift,
                                   const float* gamma, const float* gy, int64_t gs, int go, int act_in, int Cp, int B, int HW,
                                   int per_sample, float* gx, int64_t gxs, int gxo, float* sums, float* dgamma, float* dbeta,
                                   int C, void* ws, void* stream) {
  return norm_bwd_impl(x, xs, xo, scale, shift, gamma, gy, gs, go, act_in, Cp, B, HW, per_sample, gx, gxs, gxo, sums, dgamma, dbeta,
                       C, ws, stream, 0, 0);
}
extern "C" int egne_norm_bwd_store_bf16(const void* x, int64_t xs, int xo, const float* scale, const float* shift,
                                        const float* gamma, const void* gy, int64_t gs, int go, int act_in, int Cp, int B, int HW,
                                        int per_sample, void* gx, int64_t gxs, int gxo, float* sums, float* dgamma, float* dbeta,
                                        int C, void* ws, void* stream) {
  return norm_bwd_impl((const egne_bf16*)x, xs, xo, scale, shift, gamma, (const egne_bf16*)gy, gs, go, act_in, Cp, B, HW, per_sample,
                       (egne_bf16*)gx, gxs, gxo, sums, dgamma, dbeta, C, ws, stream, 0, 0);
}

// Backward of egne_norm_act_pool2 (zp = avg_pool2d(act(x*scale + shift), 2), per-sample statistics): the InstanceNorm backward
// with gy[n][y][x] = gzp[n][y/2][x/2] / 4 read straight from the pooled gradient.  H and W even.
extern "C" int egne_norm_pool2_bwd(const float* x, int64_t xs, int xo, const float* scale, const float* shift,
                                   const float* gzp, int64_t gs, int go, int act_in, int Cp, int B, int H, int W,
                                   float* gx, int64_t gxs, int gxo, int accumulate, float* sums, void* ws, void* stream) {
  EGNE_REQUIRE(H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "norm_pool2_bwd: even map sizes only (got %dx%d)", H, W);
  return norm_bwd_impl(x, xs, xo, scale, shift, (const float*)nullptr, gzp, gs, go, act_in, Cp, B, H * W, 1, gx, gxs, gxo, sums,
                       (float*)nullptr, (float*)nullptr, 0, ws, stream, W, accumulate);
}
extern "C" int egne_norm_pool2_bwd_bf16(const void* x, int64_t xs, int xo, const float* scale, const float* shift,
                                        const void* gzp, int64_t gs, int go, int act_in, int Cp, int B, int H, int W,
                                        void* gx, int64_t gxs, int gxo, int accumulate, float* sums, void* ws, void* stream) {
  EGNE_REQUIRE(H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "norm_pool2_bwd: even map sizes only (got %dx%d)", H, W);
  return norm_bwd_impl((const egne_bf16*)x, xs, xo, scale, shift, (const float*)nullptr, (const egne_bf16*)gzp, gs, go, act_in, Cp, B,
                       H * W, 1, (egne_bf16*)gx, gxs, gxo, sums, (float*)nullptr, (float*)nullptr, 0, ws, stream, W, accumulate);
}

// Fused backward of a tensor that was InstanceNorm-ed once for up to two consumers (see norm_fuse_partial): statistics of
// G = a1 + act_q'(xh) up(gq) / 4, then gz = act'(x) (g + rstd (G - mean G - xh mean(G xh))) in place over g, + the bias sums of the
// layer whose output x is (dbias += sums, or the chunk sums in ws_bias for egne_pair_bias_bwd).  B samples of H x W pixels.
// ws_norm: egne_norm_bwd_workspace_bytes(B, H*W, Cp, 1); sums: [B][Cp][2] floats; ws_bias: egne_act_bwd_bias_workspace_bytes(B*H*W, Cp).
template <typename T>
static int act_norm_bwd_impl(T* g, int64_t gs, int go, const T* x, int64_t xs, int xo, int act, const float* scale, const float* shift,
                             const T* a1, int64_t a1s, int a1o, const T* gq, int64_t gqs, int gqo, int act_q, int Cp, int B, int H, int W,
                             float* sums, void* ws_norm, float* dbias, int C, void* ws_bias, void* stream, int acc_samples,
                             int per_sample = 1, const float* gamma = nullptr, float* dgamma = nullptr, float* dbeta = nullptr, int Cn = 0) {
  EGNE_REQUIRE(slice_ok(g, gs, go, Cp) && slice_ok(x, xs, xo, Cp) && scale && shift && sums && ws_norm && ws_bias && (a1 || gq) && acc_samples >= 0 && acc_samples <= B,
               "act_norm_bwd: bad arguments");
  EGNE_REQUIRE(vec_ok<T>(gs, go, Cp) && vec_ok<T>(xs, xo, Cp) && (!a1 || vec_ok<T>(a1s, a1o, Cp)) && (!gq || vec_ok<T>(gqs, gqo, Cp)),
               "act_norm_bwd: slices must be 16-byte vectors (8 bf16 channels)");
  EGNE_REQUIRE(B > 0 && H > 0 && W > 0 && (!gq || (H % 2 == 0 && W % 2 == 0)) && (long long)B * H * W < (1ll << 32) && B <= 65535 &&
               ((uintptr_t)ws_norm & 15) == 0, "act_norm_bwd: shape (even maps for a pooled addend)");
  const long long HW = (long long)H * W, npix = (long long)B * HW;
  NormAddends<T> A{a1, (long long)a1s, a1o, gq, (long long)gqs, gqo, W, act_q};
  hipStream_t st = (hipStream_t)stream;
  const int nchunk = chunks_for(HW, Cp, B);
  hipLaunchKernelGGL(norm_fuse_partial<T>, dim3(nchunk, (Cp + 31) / 32, B), dim3(256), 0, st, x, (long long)xs, xo, scale, shift, A, Cp, HW, nchunk, (double*)ws_norm, per_sample);
  // batch statistics: the B x nchunk rows are one sample's worth of chunks for the second stage (which also adds dgamma / dbeta)
  if (per_sample) hipLaunchKernelGGL(norm_bwd_final, dim3((Cp + 31) / 32, B), dim3(1024), 0, st, (const double*)ws_norm, Cp, B, nchunk, sums, (float*)nullptr, (float*)nullptr, 0);
  else hipLaunchKernelGGL(norm_bwd_final, dim3((Cp + 31) / 32, 1), dim3(1024), 0, st, (const double*)ws_norm, Cp, 1, B * nchunk, sums, dgamma, dbeta, Cn);
  // ws_bias holds chunks_for(B H W) rows (egne_act_bwd_bias_workspace_bytes: what egne_pair_bias_bwd / the reduction below sum over): nps
  // chunks per sample fill the first B * nps of them, the rest stay zero (the caller's zero-filled allocation is never written there)
  const int nchb = chunks_for(npix, Cp, 1);
  EGNE_REQUIRE(B <= nchb, "act_norm_bwd: %d samples for %d partial-sum rows", B, nchb);
  const int nps = nchb / B;
  hipLaunchKernelGGL(act_norm_bwd_partial<T>, dim3(nps, (Cp + 31) / 32, B), dim3(256), 0, st, g, (long long)gs, go, x, (long long)xs, xo, act, scale, shift,
                     (const float*)sums, A, Cp, HW, nps, (double*)ws_bias, per_sample, gamma, per_sample ? 1.f / (float)HW : 1.f / ((float)HW * (float)B), acc_samples);
  if (dbias) hipLaunchKernelGGL(reduce_chunks_k, dim3((C + 31) / 32), dim3(1024), 0, st, (const double*)ws_bias, Cp, C, nchb, dbias, 1);
  return egne::check_launch("egne_act_norm_bwd");
}
extern "C" int egne_act_norm_bwd(float* g, int64_t gs, int go, const float* x, int64_t xs, int xo, int act, const float* scale, const float* shift,
                                 const float* a1, int64_t a1s, int a1o, const float* gq, int64_t gqs, int gqo, int act_q, int Cp, int B, int H, int W,
                                 float* sums, void* ws_norm, float* dbias, int C, void* ws_bias, int acc_samples, void* stream) {
  return act_norm_bwd_impl(g, gs, go, x, xs, xo, act, scale, shift, a1, a1s, a1o, gq, gqs, gqo, act_q, Cp, B, H, W, sums, ws_norm, dbias, C, ws_bias, stream, acc_samples);
}
extern "C" int egne_act_norm_bwd_bf16(void* g, int64_t gs, int go, const void* x, int64_t xs, int xo, int act, const float* scale, const float* shift,
                                      const void* a1, int64_t a1s, int a1o, const void* gq, int64_t gqs, int gqo, int act_q, int Cp, int B, int H, int W,
                                      float* sums, void* ws_norm, float* dbias, int C, void* ws_bias, int acc_samples, void* stream) {
  return act_norm_bwd_impl((egne_bf16*)g, gs, go, (const egne_bf16*)x, xs, xo, act, scale, shift, (const egne_bf16*)a1, a1s, a1o, (const egne_bf16*)gq, gqs, gqo,
                           act_q, Cp, B, H, W, sums, ws_norm, dbias, C, ws_bias, stream, acc_samples);
}

// BatchNorm backward (training-mode batch statistics over B samples: utils.py:1049) with the masking pass of the layer in front of it
// (round 5): gz = act'(x) rstd gamma (gy - mean gy - xh mean(gy xh)) written over gx, x = that layer's activated output = the
// BatchNorm's input; dgamma += sum gy xh, dbeta += sum gy; the layer's bias sums as egne_act_bwd_bias leaves them (ws_bias chunk
// sums; dbias += their total when given).  Replaces egne_norm_bwd + egne_act_bwd_bias.
template <typename T>
static int bn_act_bwd_impl(const T* x, int64_t xs, int xo, int act, const float* scale, const float* shift, const float* gamma, const T* gy, int64_t gys, int gyo,
                           int Cp, int B, int H, int W, T* gx, int64_t gxs, int gxo, float* sums, void* ws_norm, float* dgamma, float* dbeta, int Cn,
                           float* dbias, int C, void* ws_bias, void* stream) {
  EGNE_REQUIRE(gy && gx && (dgamma == nullptr) == (dbeta == nullptr), "bn_act_bwd: bad arguments");
  return act_norm_bwd_impl(gx, gxs, gxo, x, xs, xo, act, scale, shift, gy, gys, gyo, (const T*)nullptr, 0, 0, EGNE_ACT_NONE, Cp, B, H, W, sums, ws_norm, dbias, C,
                           ws_bias, stream, 0, 0, gamma, dgamma, dbeta, Cn);
}
extern "C" int egne_bn_act_bwd(const float* x, int64_t xs, int xo, int act, const float* scale, const float* shift, const float* gamma, const float* gy, int64_t gys,
                               int gyo, int Cp, int B, int H, int W, float* gx, int64_t gxs, int gxo, float* sums, void* ws_norm, float* dgamma, float* dbeta,
                               int Cn, float* dbias, int C, void* ws_bias, void* stream) {
  return bn_act_bwd_impl(x, xs, xo, act, scale, shift, gamma, gy, gys, gyo, Cp, B, H, W, gx, gxs, gxo, sums, ws_norm, dgamma, dbeta, Cn, dbias, C, ws_bias, stream);
}
extern "C" int egne_bn_act_bwd_bf16(const void* x, int64_t xs, int xo, int act, const float* scale, const float* shift, const float* gamma, const void* gy, int64_t gys,
                                    int gyo, int Cp, int B, int H, int W, void* gx, int64_t gxs, int gxo, float* sums, void* ws_norm, float* dgamma, float* dbeta,
                                    int Cn, float* dbias, int C, void* ws_bias, void* stream) {
  return bn_act_bwd_impl((const egne_bf16*)x, xs, xo, act, scale, shift, gamma, (const egne_bf16*)gy, gys, gyo, Cp, B, H, W, (egne_bf16*)gx, gxs, gxo, sums, ws_norm,
                         dgamma, dbeta, Cn, dbias, C, ws_bias, stream);
}

template <typename T>
static int avgpool2_bwd_impl(const T* gy, int64_t gs, int go, T* gx, int64_t xs, int xo, int B, int H, int W, int Cp, void* stream) {
  EGNE_REQUIRE(slice_ok(gy, gs, go, Cp) && slice_ok(gx, xs, xo, Cp) && B > 0 && H >= 2 && W >= 2, "avgpool2_bwd: bad arguments");
  hipLaunchKernelGGL(avgpool2_bwd_k<T>, dim3(grid_for((long long)B * (H / 2) * (W / 2) * (Cp / 4))), dim3(256), 0,
                     (hipStream_t)stream, gy, (long long)gs, go, gx, (long long)xs, xo, B, H, W, Cp);
  return egne::check_launch("egne_avgpool2_bwd");
}
extern "C" int egne_avgpool2_bwd(const float* gy, int64_t gs, int go, float* gx, int64_t xs, int xo, int B, int H, int W,
                                 int Cp, void* stream) {
  return avgpool2_bwd_impl(gy, gs, go, gx, xs, xo, B, H, W, Cp, stream);
}
extern "C" int egne_avgpool2_bwd_bf16(const void* gy, int64_t gs, int go, void* gx, int64_t xs, int xo, int B, int H, int W,
                                      int Cp, void* stream) {
  return avgpool2_bwd_impl((const egne_bf16*)gy, gs, go, (egne_bf16*)gx, xs, xo, B, H, W, Cp, stream);
}

template <typename T, bool ACC = true>
static int upsample2x_bwd_impl(const T* gy, int64_t gs, int go, T* gx, int64_t xs, int xo, int B, int H, int W, int Cp, void* stream) {
  EGNE_REQUIRE(slice_ok(gy, gs, go, Cp) && slice_ok(gx, xs, xo, Cp) && B > 0 && H > 0 && W > 0, "upsample2x_bwd: bad arguments");
  EGNE_REQUIRE(vec_ok<T>(gs, go, Cp) && vec_ok<T>(xs, xo, Cp) && H <= 65535 && B <= 65535, "upsample2x_bwd: slices must be 16-byte vectors; grid limits");
  hipLaunchKernelGGL((upsample2x_bwd_k<T, ACC>), dim3((unsigned)((W * (Cp / egne_vt<T>::N) + 255) / 256), (unsigned)H, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, gy, (long long)gs, go, gx, (long long)xs, xo, B, H, W, Cp);
  return egne::check_launch("egne_upsample2x_bwd");
}
// the same, STORED instead of accumulated (the first writer of a gradient slice: no zero pass, no read of gx)
extern "C" int egne_upsample2x_bwd_store(const float* gy, int64_t gs, int go, float* gx, int64_t xs, int xo, int B, int H, int W,
                                         int Cp, void* stream) {
  return upsample2x_bwd_impl<float, false>(gy, gs, go, gx, xs, xo, B, H, W, Cp, stream);
}
extern "C" int egne_upsample2x_bwd_store_bf16(const void* gy, int64_t gs, int go, void* gx, int64_t xs, int xo, int B, int H, int W,
                                              int Cp, void* stream) {
  return upsample2x_bwd_impl<egne_bf16, false>((const egne_bf16*)gy, gs, go, (egne_bf16*)gx, xs, xo, B, H, W, Cp, stream);
}
extern "C" int egne_upsample2x_bwd(const float* gy, int64_t gs, int go, float* gx, int64_t xs, int xo, int B, int H, int W,
                                   int Cp, void* stream) {
  return upsample2x_bwd_impl(gy, gs, go, gx, xs, xo, B, H, W, Cp, stream);
}
extern "C" int egne_upsample2x_bwd_bf16(const void* gy, int64_t gs, int go, void* gx, int64_t xs, int xo, int B, int H, int W,
                                        int Cp, void* stream) {
  return upsample2x_bwd_impl((const egne_bf16*)gy, gs, go, (egne_bf16*)gx, xs, xo, B, H, W, Cp, stream);
}

template <typename T>
static int upsample2x_nearest_bwd_impl(const T* gy, int64_t gs, int go, T* gx, int64_t xs, int xo, int B, int H, int W, int Cp, void* stream) {
  EGNE_REQUIRE(slice_ok(gy, gs, go, Cp) && slice_ok(gx, xs, xo, Cp) && B > 0 && H > 0 && W > 0, "upsample2x_nearest_bwd: bad arguments");
  hipLaunchKernelGGL(upsample2x_nearest_bwd_k<T>, dim3(grid_for((long long)B * H * W * (Cp / 4))), dim3(256), 0, (hipStream_t)stream, gy,
                     (long long)gs, go, gx, (long long)xs, xo, B, H, W, Cp);
  return egne::check_launch("egne_upsample2x_nearest_bwd");
}
extern "C" int egne_upsample2x_nearest_bwd(const float* gy, int64_t gs, int go, float* gx, int64_t xs, int xo, int B, int H, int W, int Cp, void* stream) {
  return upsample2x_nearest_bwd_impl(gy, gs, go, gx, xs, xo, B, H, W, Cp, stream);
}
extern "C" int egne_upsample2x_nearest_bwd_bf16(const void* gy, int64_t gs, int go, void* gx, int64_t xs, int xo, int B, int H, int W, int Cp, void* stream) {
  return upsample2x_nearest_bwd_impl((const egne_bf16*)gy, gs, go, (egne_bf16*)gx, xs, xo, B, H, W, Cp, stream);
}

template <typename T>
static int head_act_bwd_impl(T* g, const T* y, int B, int ld, void* stream) {
  EGNE_REQUIRE(g && y && B > 0 && ld >= 10, "head_act_bwd: bad arguments");
  hipLaunchKernelGGL(head_act_bwd_k<T>, dim3((B * 10 + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, y, B, ld);
  return egne::check_launch("egne_ellipse_head_act_bwd");
}
extern "C" int egne_ellipse_head_act_bwd(float* g, const float* y, int B, int ld, void* stream) { return head_act_bwd_impl(g, y, B, ld, stream); }
extern "C" int egne_ellipse_head_act_bwd_bf16(void* g, const void* y, int B, int ld, void* stream) {
  return head_act_bwd_impl((egne_bf16*)g, (const egne_bf16*)y, B, ld, stream);
}

template <typename T>
static int selu_bwd_impl(T* g, const T* y, int64_t n, void* stream) {
  EGNE_REQUIRE(g && y && n > 0, "selu_bwd: bad arguments");
  hipLaunchKernelGGL(selu_bwd_k<T>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, y, (long long)n);
  return egne::check_launch("egne_selu_bwd");
}
extern "C" int egne_selu_bwd(float* g, const float* y, int64_t n, void* stream) { return selu_bwd_impl(g, y, n, stream); }
extern "C" int egne_selu_bwd_bf16(void* g, const void* y, int64_t n, void* stream) { return selu_bwd_impl((egne_bf16*)g, (const egne_bf16*)y, n, stream); }

template <typename T>
static int spatial_mean_bwd_impl(const T* g, int gld, T* gx, int64_t xs, int xo, int C, int B, int HW, void* stream) {
  EGNE_REQUIRE(g && gx && C > 0 && xo + C <= xs && gld >= C && B > 0 && HW > 0, "spatial_mean_bwd: bad arguments");
  hipLaunchKernelGGL(spatial_mean_bwd_k<T>, dim3(B), dim3(256), 0, (hipStream_t)stream, g, gld, gx, (long long)xs, xo, C, HW);
  return egne::check_launch("egne_spatial_mean_bwd");
}
extern "C" int egne_spatial_mean_bwd(const float* g, int gld, float* gx, int64_t xs, int xo, int C, int B, int HW, void* stream) {
  return spatial_mean_bwd_impl(g, gld, gx, xs, xo, C, B, HW, stream);
}
extern "C" int egne_spatial_mean_bwd_bf16(const void* g, int gld, void* gx, int64_t xs, int xo, int C, int B, int HW, void* stream) {
  return spatial_mean_bwd_impl((const egne_bf16*)g, gld, (egne_bf16*)gx, xs, xo, C, B, HW, stream);
}

template <typename T>
static int conf_loss_bwd_impl(const T* pred, int ld, const int64_t* gt, int B, int C, int flag, const float* gscale, T* gpred, int gld, void* stream) {
  EGNE_REQUIRE(pred && gpred && gscale && B > 0 && C > 0 && ld >= C && gld >= C && (flag || gt), "conf_loss_bwd: bad arguments");
  hipLaunchKernelGGL(conf_loss_bwd_k<T>, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, pred, ld,
                     (const long long*)gt, B, C, flag, gscale, gpred, gld);
  return egne::check_launch("egne_conf_loss_bwd");
}
extern "C" int egne_conf_loss_bwd(const float* pred, int ld, const int64_t* gt, int B, int C, int flag, const float* gscale,
                                  float* gpred, int gld, void* stream) {
  return conf_loss_bwd_impl(pred, ld, gt, B, C, flag, gscale, gpred, gld, stream);
}
extern "C" int egne_conf_loss_bwd_bf16(const void* pred, int ld, const int64_t* gt, int B, int C, int flag, const float* gscale,
                                       void* gpred, int gld, void* stream) {
  return conf_loss_bwd_impl((const egne_bf16*)pred, ld, gt, B, C, flag, gscale, (egne_bf16*)gpred, gld, stream);
}

// all-pairs 1x1 form: shapes it takes, chunk length and pairs per wave
static bool w1_supported(const egne_conv_desc& d, int* nco_, int* nkc_, int* ch_) {
  static const bool off = [] { const char* e = getenv("EGNE_WGRAD_1X1"); return e && e[0] == '0'; }();
  if (off || d.kh != 1 || d.kw != 1 || d.stride != 1 || d.pad_h != 0 || d.pad_w != 0 || d.ngroups != 1 || d.H != d.Ho || d.W != d.Wo) return false;
  int nkc = 0;
  for (int s = 0; s < d.nseg; ++s) nkc += (d.seg[s].Cp + 31) / 32;
  const int nco = d.CoutP / 32;
  // (two or three blocks: the tile-per-workgroup kernel already runs near the HBM rate and keeps all four waves busy -- measured
  //  921 vs 1151 us for 64 -> 32 at 240x320; from six blocks on the restaging dominates: 12 blocks at 120x160 1596 -> 1224 us)
  if (nco + nkc > W1_MAXT || nco * nkc > 32 || nco * nkc < 6 || (long long)d.B * d.Ho * d.Wo < 4096) return false;
  *nco_ = nco; *nkc_ = nkc;
  *ch_ = (nco + nkc) * 64 * W1LD * 4 <= 78 * 1024 ? 64 : 32;      // two workgroups per CU where the tiles allow
  return true;
}

static int w1_splits(const egne_conv_desc& d, int ch) {
  const long long M = (long long)d.B * d.Ho * d.Wo;
  long long ns = (M + ch * 4 - 1) / (ch * 4);      // at least four chunks per workgroup
  if (ns > 512) ns = 512;
  return (int)(ns < 1 ? 1 : ns);
}

extern "C" int egne_conv2d_wgrad_splits(const egne_conv_desc* dp) {
  if (!dp) return 0;
  const egne_conv_desc& d = *dp;
  if (d.dtype == 1 && egne::wgrad3x3_bf16_supported(d, d.out_pix_stride)) return egne::wgrad3x3_bf16_splits(d);
  if (d.dtype == 1 && egne::wgrad1x1_bf16_supported(d, d.out_pix_stride)) return egne::wgrad1x1_bf16_splits(d);
  if (d.dtype == 0 && egne::wgrad_halo_supported(d, d.out_pix_stride)) return egne::wgrad_halo_splits(d);
  { int a, b2, ch; if (w1_supported(d, &a, &b2, &ch)) return w1_splits(d, ch); }
  int per_tap = 0;
  for (int s = 0; s < d.nseg; ++s) per_tap += (d.seg[s].Cp + 31) / 32;
  long long tiles = (long long)(d.CoutP / 32) * per_tap * d.kh * d.kw * d.ngroups;
  // bf16 tensors, one 8-channel slice and 64 outputs: the wide folded form takes two tap groups x both output blocks per workgroup --
  // 7 column tiles instead of 98 for a 7x7, so more pixel splits to fill the chip
  if (d.dtype == 1 && d.ngroups == 1 && d.nseg == 1 && d.seg[0].Cp == 8 && d.CoutP == 64 && d.kh * d.kw >= 4) tiles = 2 * (((d.kh * d.kw + 3) / 4 + 1) / 2);
  const long long M = (long long)d.B * d.Ho * d.Wo;
  long long ns = 4096 / (tiles > 0 ? tiles : 1);
  if (ns < 1) ns = 1;
  const long long mx = (M + 1023) / 1024;
  if (ns > mx) ns = mx;
  if (ns > 512) ns = 512;
  return (int)ns;
}

extern "C" int64_t egne_conv2d_wgrad_workspace_bytes(const egne_conv_desc* dp) {
  if (!dp) return 0;
  const egne_conv_desc& d = *dp;
  return (int64_t)egne_conv2d_wgrad_splits(dp) * d.ngroups * d.kh * d.kw * d.CoutP * d.Ktot * sizeof(float);
}

// gz: gradient w.r.t. the pre-activation output (Cout_store channels).  gw[g]: OIHW gradient tensors,
// accumulated into.  kinv as in egne_pack_conv_weight.  ws: egne_conv2d_wgrad_workspace_bytes.
template <typename TS>
static int wgrad_impl(const egne_conv_desc* dp, const TS* gz, int64_t gzs, int gzo, const uint32_t* gz_dyn, int Cout, int Cin,
                      const int32_t* kinv, float* const* gw, void* ws, void* stream) {
  constexpr bool BF = !std::is_same<TS, float>::value;
  EGNE_REQUIRE(dp && gz && kinv && gw && ws, "wgrad: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.nseg >= 1 && d.nseg <= EGNE_MAXSEG && d.ngroups >= 1 && d.ngroups <= EGNE_MAXGROUP, "wgrad: bad descriptor");
  EGNE_REQUIRE(slice_ok(gz, gzs, gzo, d.Cout_store), "wgrad: bad gz slice");
  int ktot = 0, per_tap = 0;
  for (int s = 0; s < d.nseg; ++s) {
    const egne_seg& g = d.seg[s];
    EGNE_REQUIRE(g.ptr && g.Cp % 8 == 0 && g.ch_off % 4 == 0 && g.pix_stride % 4 == 0 && ((uintptr_t)g.ptr & 15) == 0, "wgrad: seg %d", s);
    ktot += g.Cp; per_tap += (g.Cp + 31) / 32;
  }
  EGNE_REQUIRE(ktot == d.Ktot && d.CoutP % 32 == 0 && Cout <= d.CoutP && Cin <= d.Ktot, "wgrad: inconsistent sizes");
  // (every argument is checked BEFORE the first launch: the forms below that rely on a zero-filled workspace leave it dirty between
  //  their partial-sum kernel and the reduction, so no validation may return in between)
  for (int g = 0; g < d.ngroups; ++g) EGNE_REQUIRE(gw[g], "wgrad: null gradient tensor %d", g);
  const int T = d.kh * d.kw, nsplit = egne_conv2d_wgrad_splits(dp);
  hipStream_t st = (hipStream_t)stream;
  bool fast3x3 = false;
  int clean = 0;
  if constexpr (BF) {
    if (egne::wgrad3x3_bf16_supported(d, d.out_pix_stride)) {         // 3x3 "same" convolutions over one slice: bf16 MFMA (wgrad_bf16.hip)
      // the gradient buffer mirrors the output buffer (same pixel stride): the split count above assumed it
      EGNE_REQUIRE(gzs == d.out_pix_stride, "wgrad: gz stride %lld differs from the output stride %lld", (long long)gzs, (long long)d.out_pix_stride);
      const int rc = egne::wgrad3x3_bf16_launch(d, gz, (long long)gzs, gzo, (float*)ws, st);     // writes every partial it owns
      if (rc != EGNE_OK) return rc;
      fast3x3 = true;
    } else if (egne::wgrad1x1_bf16_supported(d, d.out_pix_stride)) {     // 1x1 over raw slices: every block pair in one workgroup, bf16 MFMA
      EGNE_REQUIRE(gzs == d.out_pix_stride, "wgrad: gz stride %lld differs from the output stride %lld", (long long)gzs, (long long)d.out_pix_stride);
      const int rc = egne::wgrad1x1_bf16_launch(d, gz, (long long)gzs, gzo, (float*)ws, st);
      if (rc != EGNE_OK) return rc;
      fast3x3 = true;
      clean = 1;
    }
  }
  if (fast3x3) {
  } else if (!BF && egne::wgrad_halo_supported(d, d.out_pix_stride)) {
    // the gradient buffer mirrors the output buffer (same pixel stride): the split count above assumed it
    EGNE_REQUIRE(gzs == d.out_pix_stride, "wgrad: gz stride %lld differs from the output stride %lld", (long long)gzs, (long long)d.out_pix_stride);
    // split-f16 products when the caller supplies max|gz| on the device AND x has a pre-scale (normalised on load or the
    // forward launch's device word); exact fp32 otherwise
    const bool f16 = gz_dyn && (d.seg[0].scale || d.dyn_scale);
    int rc = EGNE_OK;
    if constexpr (!BF)
      rc = f16 ? egne::wgrad_halo_f16_launch(d, gz, (long long)gzs, gzo, (const unsigned*)gz_dyn, (float*)ws, st)
               : egne::wgrad_halo_launch(d, gz, (long long)gzs, gzo, (float*)ws, st);   // writes every partial it owns
    if (rc != EGNE_OK) return rc;
  } else if (int nco = 0, nkc = 0, ch = 0; w1_supported(d, &nco, &nkc, &ch)) {
    clean = 1;          // (ws arrives zero-filled: the caller's first fill, then the reduction of the previous call)
    W1Tab tab{};
    int j = nco, kofs = 0;
    for (int s = 0; s < d.nseg; ++s) {
      for (int c0 = 0; c0 < d.seg[s].Cp; c0 += 32, ++j) { tab.seg[j] = (short)s; tab.c0[j] = (short)c0; tab.kofs[j] = (short)kofs; }
      kofs += d.seg[s].Cp;
    }
    const int ppw = (nco * nkc + 3) / 4;
    const size_t lds = (size_t)(nco + nkc) * ch * W1LD * sizeof(float);
    auto go = [&](auto kern) -> int {
      const bool raised = egne::raise_lds((const void*)kern, 120 * 1024);
      if (!raised) return egne::fail(EGNE_ERR_LAUNCH, "wgrad 1x1: cannot raise the dynamic LDS limit");
      hipLaunchKernelGGL(kern, dim3(nsplit), dim3(256), lds, st, d, gz, (long long)gzs, gzo, nsplit, nco, nkc, tab, (float*)ws);
      return EGNE_OK;
    };
    int rc;
    if (ch == 64) rc = ppw <= 1 ? go(conv1x1_wgrad_allpairs_kernel<1, 64, TS>) : ppw <= 2 ? go(conv1x1_wgrad_allpairs_kernel<2, 64, TS>)
                     : ppw <= 3 ? go(conv1x1_wgrad_allpairs_kernel<3, 64, TS>) : ppw <= 4 ? go(conv1x1_wgrad_allpairs_kernel<4, 64, TS>)
                     : go(conv1x1_wgrad_allpairs_kernel<8, 64, TS>);
    else rc = ppw <= 2 ? go(conv1x1_wgrad_allpairs_kernel<2, 32, TS>) : ppw <= 4 ? go(conv1x1_wgrad_allpairs_kernel<4, 32, TS>)
              : go(conv1x1_wgrad_allpairs_kernel<8, 32, TS>);
    if (rc != EGNE_OK) return rc;
  } else {
    clean = 1;
    dim3 grid(nsplit, d.CoutP / 32, per_tap * T * d.ngroups);
    static const bool bfm = [] { const char* e = getenv("EGNE_IGEMM_BF16_MFMA"); return !e || e[0] != '0'; }();
    static const bool fold_on = [] { const char* e = getenv("EGNE_IGEMM_FOLD"); return !e || e[0] != '0'; }();
    if constexpr (sizeof(TS) == 2) {
      if (bfm && fold_on && d.nseg == 1 && d.seg[0].Cp == 8 && d.Ktot == 8 && d.ngroups == 1 && T >= 4 && d.CoutP == 64 && gzs % 8 == 0 && gzo % 8 == 0 &&
          ((uintptr_t)gz & 15) == 0 && !d.seg[0].scale && d.seg[0].act_in == EGNE_ACT_NONE && d.seg[0].ch_off % 8 == 0 && d.seg[0].pix_stride % 8 == 0) {
        dim3 gridf(nsplit, 1, ((T + 3) / 4 + 1) / 2);         // two tap groups and both output blocks per workgroup
        hipLaunchKernelGGL(conv_wgrad_wide_kernel<true>, gridf, dim3(256), 0, st, d, (const egne_bf16*)gz, (long long)gzs, gzo, nsplit, (float*)ws);
      } else if (bfm && fold_on && d.nseg == 1 && d.seg[0].Cp == 8 && d.Ktot == 8 && d.ngroups == 1 && T >= 4) {
        dim3 gridf(nsplit, d.CoutP / 32, (T + 3) / 4);
        hipLaunchKernelGGL((conv_wgrad_kernel<TS, true, true>), gridf, dim3(256), 0, st, d, gz, (long long)gzs, gzo, nsplit, (float*)ws);
      } else if (bfm && d.ngroups == 1 && d.CoutP >= 128 && gzs % 8 == 0 && gzo % 8 == 0 && ((uintptr_t)gz & 15) == 0 && [&] {
                   for (int s2 = 0; s2 < d.nseg; ++s2) {
                     const egne_seg& q = d.seg[s2];
                     if (q.scale || q.act_in != EGNE_ACT_NONE || q.ch_off % 8 || q.pix_stride % 8 || q.Cp % 8) return false;
                   }
                   return true;
                 }()) {
        // wide form: four output blocks per workgroup (the strided / reflect-padded convolutions of the StyleEncoder and their kin)
        dim3 gridw(nsplit, (d.CoutP + 127) / 128, per_tap * T);
        hipLaunchKernelGGL(conv_wgrad_wide_kernel<false>, gridw, dim3(256), 0, st, d, (const egne_bf16*)gz, (long long)gzs, gzo, nsplit, (float*)ws);
      } else if (bfm) hipLaunchKernelGGL((conv_wgrad_kernel<TS, true>), grid, dim3(256), 0, st, d, gz, (long long)gzs, gzo, nsplit, (float*)ws);
      else hipLaunchKernelGGL((conv_wgrad_kernel<TS, false>), grid, dim3(256), 0, st, d, gz, (long long)gzs, gzo, nsplit, (float*)ws);
    } else {
      hipLaunchKernelGGL((conv_wgrad_kernel<TS, false>), grid, dim3(256), 0, st, d, gz, (long long)gzs, gzo, nsplit, (float*)ws);
    }
  }
  for (int g = 0; g < d.ngroups; ++g) {
    const long long total = (long long)T * d.CoutP * d.Ktot;
    hipLaunchKernelGGL(wgrad_reduce_k, dim3(grid_for(total * 8)), dim3(256), 0, st, (float*)ws, nsplit, d.ngroups, g, T, Cout,
                       Cin, kinv, d.CoutP, d.Ktot, gw[g], clean);
  }
  const int rc = egne::check_launch("egne_conv2d_wgrad");
  // a launch of this call failed: the zero-filled-workspace contract of the `clean` forms cannot be trusted any more -- restore it
  // here, so that the next (good) call does not add this call's stray partial sums to its gradients
  if (rc != EGNE_OK && clean) (void)hipMemsetAsync(ws, 0, (size_t)egne_conv2d_wgrad_workspace_bytes(dp), st);
  return rc;
}

extern "C" int egne_conv2d_wgrad(const egne_conv_desc* dp, const float* gz, int64_t gzs, int gzo, int Cout, int Cin,
                                 const int32_t* kinv, float* const* gw, void* ws, void* stream) {
  // descriptor dtype 1: the input slices AND gz are bf16 tensors (fp32 products, or bf16 MFMA for 3x3 "same" convolutions)
  if (dp && dp->dtype == 1) return wgrad_impl(dp, (const egne_bf16*)gz, gzs, gzo, (const uint32_t*)nullptr, Cout, Cin, kinv, gw, ws, stream);
  return wgrad_impl(dp, gz, gzs, gzo, (const uint32_t*)nullptr, Cout, Cin, kinv, gw, ws, stream);
}

extern "C" int egne_conv2d_wgrad_f16(const egne_conv_desc* dp, const float* gz, int64_t gzs, int gzo, const uint32_t* gz_absmax_bits,
                                     int Cout, int Cin, const int32_t* kinv, float* const* gw, void* ws, void* stream) {
  EGNE_REQUIRE(dp && dp->dtype == 0, "conv2d_wgrad_f16: fp32 tensors only (bf16 plans call egne_conv2d_wgrad)");
  return wgrad_impl(dp, gz, gzs, gzo, gz_absmax_bits, Cout, Cin, kinv, gw, ws, stream);
}

extern "C" int egne_pack_conv_weight_dgrad(const float* w_oihw, int Cout, int Cin, int kh, int kw, int ci0, int Cpiece,
                                           int CoutPp, int Ktotp, int frag, float* out, void* stream) {
  EGNE_REQUIRE(w_oihw && out, "pack_dgrad: null pointer");
  EGNE_REQUIRE(ci0 >= 0 && Cpiece > 0 && ci0 + Cpiece <= Cin && CoutPp >= Cpiece && CoutPp % 32 == 0 && Ktotp >= Cout && Ktotp % 8 == 0,
               "pack_dgrad: bad sizes");
  const long long total = (long long)kh * kw * CoutPp * Ktotp;
  hipLaunchKernelGGL(pack_weight_dgrad_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, kh * kw,
                     ci0, Cpiece, CoutPp, Ktotp, frag, out);
  return egne::check_launch("egne_pack_conv_weight_dgrad");
}
