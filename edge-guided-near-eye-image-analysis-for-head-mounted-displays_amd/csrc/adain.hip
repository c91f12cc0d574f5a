// AdaIN fusion path of ESF-Net (models/RITnet_v2.py:289-308, calc_mean_std :251-259) and the
// dataset-confusion loss (loss.py:139-157, models/RITnet_v2.py:343-350).  Small HBM-bound kernels.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// channel softmax over the 3 logits of every pixel (nn.Softmax(dim=1) of RITnet_v2.py:290-294)
__global__ void softmax3_k(const float* __restrict__ x, long long xs, int xo, float* __restrict__ y, long long ys, int yo,
                           int Cp_out, long long npix) {
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const float* s = x + p * xs + xo;
    const float a = s[0], b = s[1], c = s[2];
    const float m = fmaxf(a, fmaxf(b, c));
    const float ea = expf(a - m), eb = expf(b - m), ec = expf(c - m);
    const float inv = 1.f / (ea + eb + ec);
    float* d = y + p * ys + yo;
    d[0] = ea * inv; d[1] = eb * inv; d[2] = ec * inv;
    for (int k = 3; k < Cp_out; ++k) d[k] = 0.f;
  }
}

// x' = (x - mean) / sqrt(var_unbiased + eps) * gamma + beta per (n, c); one block per (n, 32 channels).
// gamma[n][c] / beta[n][c] are rows of the MLP output (adain_params[:,0] / [:,1]).
__global__ __launch_bounds__(256) void adain_k(const float* __restrict__ x, long long xs, int xo, int C,
                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                               long long gb_stride, int gb_off, float* __restrict__ y, long long ys,
                                               int yo, int HW, float eps) {
  const int n = blockIdx.y, cg = blockIdx.x;
  const int c = cg * 32 + (threadIdx.x & 31), row = threadIdx.x >> 5;  // 32 channels x 8 pixel rows
  const bool ok = c < C;
  const float* src = x + (long long)n * HW * xs + xo + c;
  double s = 0, q = 0;
  if (ok)
    for (int p = row; p < HW; p += 8) { const float v = src[(long long)p * xs]; s += v; q += (double)v * v; }
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
    const float g = gamma[(long long)n * gb_stride + gb_off + c], b = beta[(long long)n * gb_stride + gb_off + c];
    float* dst = y + (long long)n * HW * ys + yo + c;
    for (int p = row; p < HW; p += 8) dst[(long long)p * ys] = (src[(long long)p * xs] - fm) / fs * g + b;
  }
}

// conf_Loss: flag=1 -> mean |softmax(x) - 1/C|; flag=0 -> cross entropy with gt.  Single block.
__global__ void conf_loss_k(const float* __restrict__ x, int ld, const long long* __restrict__ gt, int B, int C, int flag,
                            float weight, float* __restrict__ terms) {
  __shared__ float sh[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float* r = x + (long long)b * ld;
    float m = -INFINITY;
    for (int k = 0; k < C; ++k) m = fmaxf(m, r[k]);
    float se = 0.f;
    for (int k = 0; k < C; ++k) se += expf(r[k] - m);
    if (flag) {
      for (int k = 0; k < C; ++k) acc += fabsf(expf(r[k] - m) / se - 1.0f / C);
    } else {
      acc += (m - r[gt[b]]) + logf(se);
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

}  // namespace

extern "C" int egne_softmax3(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo, int Cp_out, int64_t npix,
                             void* stream) {
  EGNE_REQUIRE(x && y && xo + 3 <= xs && yo + Cp_out <= ys && Cp_out >= 3 && npix > 0, "softmax3: bad arguments");
  long long g = (npix + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(softmax3_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, (long long)xs, xo, y,
                     (long long)ys, yo, Cp_out, (long long)npix);
  return egne::check_launch("egne_softmax3");
}

extern "C" int egne_adain(const float* x, int64_t xs, int xo, int C, const float* gamma, const float* beta,
                          int64_t gb_stride, int gb_off, float* y, int64_t ys, int yo, int B, int HW, float eps,
                          void* stream) {
  EGNE_REQUIRE(x && y && gamma && beta && C > 0 && xo + C <= xs && yo + C <= ys && B > 0 && HW > 1, "adain: bad arguments");
  hipLaunchKernelGGL(adain_k, dim3((C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, x, (long long)xs, xo, C, gamma, beta,
                     (long long)gb_stride, gb_off, y, (long long)ys, yo, HW, eps);
  return egne::check_launch("egne_adain");
}

extern "C" int egne_conf_loss(const float* pred, int ld, const int64_t* gt, int B, int C, int flag, float weight,
                              float* terms, void* stream) {
  EGNE_REQUIRE(pred && terms && B > 0 && C > 0 && ld >= C && (flag || gt), "conf_loss: bad arguments");
  hipLaunchKernelGGL(conf_loss_k, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, ld, (const long long*)gt, B, C, flag,
                     weight, terms);
  return egne::check_launch("egne_conf_loss");
}
