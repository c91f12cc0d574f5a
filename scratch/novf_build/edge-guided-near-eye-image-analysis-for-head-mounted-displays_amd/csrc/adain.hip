// AdaIN fusion path of ESF-Net (models/RITnet_v2.py:289-308, calc_mean_std :251-259) and the
// dataset-confusion loss (loss.py:139-157, models/RITnet_v2.py:343-350).  Small HBM-bound kernels.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// channel softmax over the 3 logits of every pixel (nn.Softmax(dim=1) of RITnet_v2.py:290-294)
template <typename T>
__global__ void softmax3_k(const T* __restrict__ x, long long xs, int xo, T* __restrict__ y, long long ys, int yo,
                           int Cp_out, long long npix) {
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const T* s = x + p * xs + xo;
    const float a = ld1(s), b = ld1(s + 1), c = ld1(s + 2);
    const float m = fmaxf(a, fmaxf(b, c));
    const float ea = expf(a - m), eb = expf(b - m), ec = expf(c - m);
    const float inv = 1.f / (ea + eb + ec);
    T* d = y + p * ys + yo;
    st1(d, ea * inv); st1(d + 1, eb * inv); st1(d + 2, ec * inv);
    for (int k = 3; k < Cp_out; ++k) st1(d + k, 0.f);
  }
}

// x' = (x - mean) / sqrt(var_unbiased + eps) * gamma + beta per (n, c); one block per (n, 32 channels).
// gamma[n][c] / beta[n][c] are rows of the MLP output (adain_params[:,0] / [:,1]).
template <typename T>
__global__ __launch_bounds__(256) void adain_k(const T* __restrict__ x, long long xs, int xo, int C,
                                               const T* __restrict__ gamma, const T* __restrict__ beta,
                                               long long gb_stride, int gb_off, T* __restrict__ y, long long ys,
                                               int yo, int HW, float eps) {
  const int n = blockIdx.y, cg = blockIdx.x;
  const int c = cg * 32 + (threadIdx.x & 31), row = threadIdx.x >> 5;  // 32 channels x 8 pixel rows
  const bool ok = c < C;
  const T* src = x + (long long)n * HW * xs + xo + c;
  double s = 0, q = 0;
  if (ok)
    for (int p = row; p < HW; p += 8) { const float v = ld1(src + (long long)p * xs); s += v; q += (double)v * v; }
  __shared__ double sh[2][8][32];
  sh[0][row][threadIdx.x & 31] = s; sh[1][row][threadIdx.x & 31] = q;
  __syncthreads();
  double ts = 0, tq = 0;
  for (int r = 0; r < 8; ++r) { ts += sh[0][r][threadIdx.x & 31]; tq += sh[1][r][threadIdx.x & 31]; }
  const double mean = ts / HW;
  double var = (tq - ts * mean) / (HW - 1);   // torch.var default: unbiased (RITnet_v2.py:256)
  if (var < 0) var = 0;
  const float fm = (float)mean, fs = sqrtf((float)var + eps);
  if (ok) {
    const float g = ld1(gamma + (long long)n * gb_stride + gb_off + c), b = ld1(beta + (long long)n * gb_stride + gb_off + c);
    T* dst = y + (long long)n * HW * ys + yo + c;
    for (int p = row; p < HW; p += 8) st1(dst + (long long)p * ys, (ld1(src + (long long)p * xs) - fm) / fs * g + b);
  }
}

// conf_Loss: flag=1 -> mean |softmax(x) - 1/C|; flag=0 -> cross entropy with gt.  Single block.
template <typename T>
__global__ void conf_loss_k(const T* __restrict__ x, int ld, const long long* __restrict__ gt, int B, int C, int flag,
                            float weight, float* __restrict__ terms) {
  __shared__ float sh[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const T* r = x + (long long)b * ld;
    float m = -INFINITY;
    for (int k = 0; k < C; ++k) m = fmaxf(m, ld1(r + k));
    float se = 0.f;
    for (int k = 0; k < C; ++k) se += expf(ld1(r + k) - m);
    if (flag) {
      for (int k = 0; k < C; ++k) acc += fabsf(expf(ld1(r + k) - m) / se - 1.0f / C);
    } else {
      acc += (m - ld1(r + gt[b])) + logf(se);
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) { if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const float conf = flag ? sh[0] / (float)(B * C) : sh[0] / (float)B;
    terms[7] = conf;
    // RITnet_v2.py:345-350: toggle -> loss += alpha*conf, else loss = conf
    terms[0] = flag ? terms[0] + weight * conf : conf;
  }
}


// ---- backward of the AdaIN fusion path -------------------------------------------------------------
// softmax: gx[c] += y[c] * (gy[c] - sum_k gy[k] y[k])
template <typename T>
__global__ void softmax3_bwd_k(const T* __restrict__ y, long long ys, int yo, const T* __restrict__ gy, long long gs,
                               int go, T* __restrict__ gx, long long xs, int xo, long long npix) {
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const T* ap = y + p * ys + yo;
    const T* gq = gy + p * gs + go;
    const float a[3] = {ld1(ap), ld1(ap + 1), ld1(ap + 2)}, g[3] = {ld1(gq), ld1(gq + 1), ld1(gq + 2)};
    const float dot = g[0] * a[0] + g[1] * a[1] + g[2] * a[2];
    T* d = gx + p * xs + xo;
    for (int k = 0; k < 3; ++k) st1(d + k, ld1(d + k) + a[k] * (g[k] - dot));
  }
}

// y = gamma * xhat + beta, xhat = (x - mean) / sqrt(var_unbiased + eps):
//   ggamma += sum gy xhat, gbeta += sum gy, gx += gamma/std * (gy - mean(gy) - xhat * sum(gy xhat) / (HW-1))
template <typename T>
__global__ __launch_bounds__(256) void adain_bwd_k(const T* __restrict__ x, long long xs, int xo, int C,
                                                   const T* __restrict__ gamma, long long gb_stride, int gb_off,
                                                   const T* __restrict__ gy, long long gys, int gyo,
                                                   T* __restrict__ gx, long long gxs, int gxo,
                                                   T* __restrict__ ggamma, T* __restrict__ gbeta,
                                                   long long gg_stride, int gg_off, int HW, float eps) {
  const int n = blockIdx.y, cg = blockIdx.x;
  const int cl = threadIdx.x & 31, c = cg * 32 + cl, row = threadIdx.x >> 5;
  const bool ok = c < C;
  const T* src = x + (long long)n * HW * xs + xo + c;
  const T* gsrc = gy + (long long)n * HW * gys + gyo + c;
  __shared__ double sh[2][8][32];
  double s = 0, q = 0;
  if (ok)
    for (int p = row; p < HW; p += 8) { const float v = ld1(src + (long long)p * xs); s += v; q += (double)v * v; }
  sh[0][row][cl] = s; sh[1][row][cl] = q;
  __syncthreads();
  double ts = 0, tq = 0;
  for (int r = 0; r < 8; ++r) { ts += sh[0][r][cl]; tq += sh[1][r][cl]; }
  __syncthreads();
  const double mean = ts / HW;
  double var = (tq - ts * mean) / (HW - 1);
  if (var < 0) var = 0;
  const float fm = (float)mean, fs = sqrtf((float)var + eps);
  double sg = 0, sgx = 0;
  if (ok)
    for (int p = row; p < HW; p += 8) {
      const float g = ld1(gsrc + (long long)p * gys);
      sg += g; sgx += (double)g * ((ld1(src + (long long)p * xs) - fm) / fs);
    }
  sh[0][row][cl] = sg; sh[1][row][cl] = sgx;
  __syncthreads();
  double tg = 0, tgx = 0;
  for (int r = 0; r < 8; ++r) { tg += sh[0][r][cl]; tgx += sh[1][r][cl]; }
  if (!ok) return;
  if (row == 0) {
    T* q1 = ggamma + (long long)n * gg_stride + gg_off + c;
    T* q2 = gbeta + (long long)n * gg_stride + gg_off + c;
    st1(q1, ld1(q1) + (float)tgx);
    st1(q2, ld1(q2) + (float)tg);
  }
  const float gm = ld1(gamma + (long long)n * gb_stride + gb_off + c) / fs;
  const float mg = (float)(tg / HW), kx = (float)(tgx / (HW - 1));
  T* dst = gx + (long long)n * HW * gxs + gxo + c;
  for (int p = row; p < HW; p += 8) {
    const float xh = (ld1(src + (long long)p * xs) - fm) / fs;
    st1(dst + (long long)p * gxs, ld1(dst + (long long)p * gxs) + gm * (ld1(gsrc + (long long)p * gys) - mg - xh * kx));
  }
}

// Backward of ReflectionPad2d(P): gx[b,iy,ix,:] += sum of gpad over every padded position that reflects onto
// (iy,ix).  gpad is either dense [B,H+2P,W+2P,Cp] (phase=0) or the phase-packed output of the stride-2
// transposed conv: [B,(H+2P)/2,(W+2P)/2,4*Cp] with channel block (py&1)*2+(px&1) (phase=1).
template <typename T>
__global__ void reflect_pad_bwd_k(const T* __restrict__ gp, long long gs, int go, int phase, int Cp,
                                  T* __restrict__ gx, long long xs, int xo, int B, int H, int W, int P) {
  const int c4 = Cp >> 2;
  const long long total = (long long)B * H * W * c4;
  const int Hp = H + 2 * P, Wp = W + 2 * P;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4) * 4;
    long long q = i / c4;
    const int ix = (int)(q % W); q /= W;
    const int iy = (int)(q % H);
    const int b = (int)(q / H);
    int pys[3], pxs[3], ny = 0, nx = 0;
    pys[ny++] = iy + P;
    if (iy >= 1 && iy <= P) pys[ny++] = P - iy;
    if (iy <= H - 2 && iy >= H - 1 - P) pys[ny++] = P + 2 * (H - 1) - iy;
    pxs[nx++] = ix + P;
    if (ix >= 1 && ix <= P) pxs[nx++] = P - ix;
    if (ix <= W - 2 && ix >= W - 1 - P) pxs[nx++] = P + 2 * (W - 1) - ix;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < ny; ++a)
      for (int e = 0; e < nx; ++e) {
        const int py = pys[a], px = pxs[e];
        const T* s;
        if (phase) s = gp + (((long long)b * (Hp >> 1) + (py >> 1)) * (Wp >> 1) + (px >> 1)) * gs + go + ((py & 1) * 2 + (px & 1)) * Cp + c;
        else       s = gp + (((long long)b * Hp + py) * Wp + px) * gs + go + c;
        acc += ld4(s);
      }
    T* d = gx + (((long long)b * H + iy) * W + ix) * xs + xo + c;
    st4(d, ld4(d) + acc);
  }
}

}  // namespace

template <typename T>
static int softmax3_impl(const T* x, int64_t xs, int xo, T* y, int64_t ys, int yo, int Cp_out, int64_t npix, void* stream) {
  EGNE_REQUIRE(x && y && xo + 3 <= xs && yo + Cp_out <= ys && Cp_out >= 3 && npix > 0, "softmax3: bad arguments");
  long long g = (npix + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(softmax3_k<T>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, (long long)xs, xo, y,
                     (long long)ys, yo, Cp_out, (long long)npix);
  return egne::check_launch("egne_softmax3");
}
extern "C" int egne_softmax3(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo, int Cp_out, int64_t npix,
                             void* stream) {
  return softmax3_impl(x, xs, xo, y, ys, yo, Cp_out, npix, stream);
}
extern "C" int egne_softmax3_bf16(const void* x, int64_t xs, int xo, void* y, int64_t ys, int yo, int Cp_out, int64_t npix,
                                  void* stream) {
  return softmax3_impl((const egne_bf16*)x, xs, xo, (egne_bf16*)y, ys, yo, Cp_out, npix, stream);
}

template <typename T>
static int adain_impl(const T* x, int64_t xs, int xo, int C, const T* gamma, const T* beta, int64_t gb_stride, int gb_off, T* y,
                      int64_t ys, int yo, int B, int HW, float eps, void* stream) {
  EGNE_REQUIRE(x && y && gamma && beta && C > 0 && xo + C <= xs && yo + C <= ys && B > 0 && HW > 1, "adain: bad arguments");
  hipLaunchKernelGGL(adain_k<T>, dim3((C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, x, (long long)xs, xo, C, gamma, beta,
                     (long long)gb_stride, gb_off, y, (long long)ys, yo, HW, eps);
  return egne::check_launch("egne_adain");
}
extern "C" int egne_adain(const float* x, int64_t xs, int xo, int C, const float* gamma, const float* beta,
                          int64_t gb_stride, int gb_off, float* y, int64_t ys, int yo, int B, int HW, float eps,
                          void* stream) {
  return adain_impl(x, xs, xo, C, gamma, beta, gb_stride, gb_off, y, ys, yo, B, HW, eps, stream);
}
/* every tensor bf16, the MLP rows (gamma / beta) included */
extern "C" int egne_adain_bf16(const void* x, int64_t xs, int xo, int C, const void* gamma, const void* beta,
                               int64_t gb_stride, int gb_off, void* y, int64_t ys, int yo, int B, int HW, float eps,
                               void* stream) {
  return adain_impl((const egne_bf16*)x, xs, xo, C, (const egne_bf16*)gamma, (const egne_bf16*)beta, gb_stride, gb_off,
                    (egne_bf16*)y, ys, yo, B, HW, eps, stream);
}

template <typename T>
static int conf_loss_impl(const T* pred, int ld, const int64_t* gt, int B, int C, int flag, float weight, float* terms, void* stream) {
  EGNE_REQUIRE(pred && terms && B > 0 && C > 0 && ld >= C && (flag || gt), "conf_loss: bad arguments");
  hipLaunchKernelGGL(conf_loss_k<T>, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, ld, (const long long*)gt, B, C, flag,
                     weight, terms);
  return egne::check_launch("egne_conf_loss");
}
extern "C" int egne_conf_loss(const float* pred, int ld, const int64_t* gt, int B, int C, int flag, float weight,
                              float* terms, void* stream) {
  return conf_loss_impl(pred, ld, gt, B, C, flag, weight, terms, stream);
}
extern "C" int egne_conf_loss_bf16(const void* pred, int ld, const int64_t* gt, int B, int C, int flag, float weight,
                                   float* terms, void* stream) {
  return conf_loss_impl((const egne_bf16*)pred, ld, gt, B, C, flag, weight, terms, stream);
}

template <typename T>
static int softmax3_bwd_impl(const T* y, int64_t ys, int yo, const T* gy, int64_t gs, int go, T* gx, int64_t xs, int xo, int64_t npix,
                             void* stream) {
  EGNE_REQUIRE(y && gy && gx && yo + 3 <= ys && go + 3 <= gs && xo + 3 <= xs && npix > 0, "softmax3_bwd: bad arguments");
  long long g = (npix + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(softmax3_bwd_k<T>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, y, (long long)ys, yo, gy,
                     (long long)gs, go, gx, (long long)xs, xo, (long long)npix);
  return egne::check_launch("egne_softmax3_bwd");
}
extern "C" int egne_softmax3_bwd(const float* y, int64_t ys, int yo, const float* gy, int64_t gs, int go, float* gx,
                                 int64_t xs, int xo, int64_t npix, void* stream) {
  return softmax3_bwd_impl(y, ys, yo, gy, gs, go, gx, xs, xo, npix, stream);
}
extern "C" int egne_softmax3_bwd_bf16(const void* y, int64_t ys, int yo, const void* gy, int64_t gs, int go, void* gx,
                                      int64_t xs, int xo, int64_t npix, void* stream) {
  return softmax3_bwd_impl((const egne_bf16*)y, ys, yo, (const egne_bf16*)gy, gs, go, (egne_bf16*)gx, xs, xo, npix, stream);
}

template <typename T>
static int adain_bwd_impl(const T* x, int64_t xs, int xo, int C, const T* gamma, int64_t gb_stride, int gb_off,
                          const T* gy, int64_t gys, int gyo, T* gx, int64_t gxs, int gxo, T* ggamma,
                          T* gbeta, int64_t gg_stride, int gg_off, int B, int HW, float eps, void* stream) {
  EGNE_REQUIRE(x && gamma && gy && gx && ggamma && gbeta && C > 0 && xo + C <= xs && gyo + C <= gys && gxo + C <= gxs &&
               B > 0 && HW > 1, "adain_bwd: bad arguments");
  hipLaunchKernelGGL(adain_bwd_k<T>, dim3((C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, x, (long long)xs, xo, C, gamma,
                     (long long)gb_stride, gb_off, gy, (long long)gys, gyo, gx, (long long)gxs, gxo, ggamma, gbeta,
                     (long long)gg_stride, gg_off, HW, eps);
  return egne::check_launch("egne_adain_bwd");
}
extern "C" int egne_adain_bwd(const float* x, int64_t xs, int xo, int C, const float* gamma, int64_t gb_stride, int gb_off,
                              const float* gy, int64_t gys, int gyo, float* gx, int64_t gxs, int gxo, float* ggamma,
                              float* gbeta, int64_t gg_stride, int gg_off, int B, int HW, float eps, void* stream) {
  return adain_bwd_impl(x, xs, xo, C, gamma, gb_stride, gb_off, gy, gys, gyo, gx, gxs, gxo, ggamma, gbeta, gg_stride, gg_off, B, HW, eps, stream);
}
extern "C" int egne_adain_bwd_bf16(const void* x, int64_t xs, int xo, int C, const void* gamma, int64_t gb_stride, int gb_off,
                                   const void* gy, int64_t gys, int gyo, void* gx, int64_t gxs, int gxo, void* ggamma,
                                   void* gbeta, int64_t gg_stride, int gg_off, int B, int HW, float eps, void* stream) {
  return adain_bwd_impl((const egne_bf16*)x, xs, xo, C, (const egne_bf16*)gamma, gb_stride, gb_off, (const egne_bf16*)gy, gys, gyo,
                        (egne_bf16*)gx, gxs, gxo, (egne_bf16*)ggamma, (egne_bf16*)gbeta, gg_stride, gg_off, B, HW, eps, stream);
}

template <typename T>
static int reflect_pad_bwd_impl(const T* gpad, int64_t gs, int go, int phase, int Cp, T* gx, int64_t xs, int xo,
                                int B, int H, int W, int P, void* stream) {
  EGNE_REQUIRE(gpad && gx && Cp > 0 && Cp % 4 == 0 && gs % 4 == 0 && go % 4 == 0 && xs % 4 == 0 && xo % 4 == 0 &&
               ((uintptr_t)gpad & 15) == 0 && ((uintptr_t)gx & 15) == 0, "reflect_pad_bwd: alignment");
  EGNE_REQUIRE(B > 0 && P >= 0 && H > P && W > P && xo + Cp <= xs && go + (phase ? 4 : 1) * Cp <= gs, "reflect_pad_bwd: shape");
  EGNE_REQUIRE(!phase || ((H + 2 * P) % 2 == 0 && (W + 2 * P) % 2 == 0), "reflect_pad_bwd: phase layout needs even padded sizes");
  const long long total = (long long)B * H * W * (Cp / 4);
  long long g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(reflect_pad_bwd_k<T>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, gpad, (long long)gs, go, phase, Cp,
                     gx, (long long)xs, xo, B, H, W, P);
  return egne::check_launch("egne_reflect_pad_bwd");
}
extern "C" int egne_reflect_pad_bwd(const float* gpad, int64_t gs, int go, int phase, int Cp, float* gx, int64_t xs, int xo,
                                    int B, int H, int W, int P, void* stream) {
  return reflect_pad_bwd_impl(gpad, gs, go, phase, Cp, gx, xs, xo, B, H, W, P, stream);
}
extern "C" int egne_reflect_pad_bwd_bf16(const void* gpad, int64_t gs, int go, int phase, int Cp, void* gx, int64_t xs, int xo,
                                         int B, int H, int W, int P, void* stream) {
  return reflect_pad_bwd_impl((const egne_bf16*)gpad, gs, go, phase, Cp, (egne_bf16*)gx, xs, xo, B, H, W, P, stream);
}
