// Loss head of ESF-Net (models/RITnet_v2.py:372-432 get_allLoss; loss.py:16-137) without host syncs.
//
// Pass 1 streams the logits once: the channel softmax is computed ONCE per pixel (the reference
// recomputes it in SurfaceLoss, GDiceLoss and wCE), all per-sample partial sums needed by the three
// segmentation terms and both soft-argmax centres of mass are formed per block, and the argmax mask /
// NCHW logits are emitted on the way.  Pass 2 combines the per-block partials per sample (the
// class-presence test replaces np.unique(target.cpu()) of loss.py:98,127) and reduces over the batch.
// Both passes are deterministic (no atomics).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PIX_PER_BLOCK = 2048;
constexpr int NPART = 24;  // floats per (sample, block) partial record
// record layout: [0..2] sum p_c*dist_c, [3..5] count(t==c), [6..8] sum p_c*[t==c], [9..11] sum p_c,
//                [12] sum -log p_t, [13] sum spatWts, [14..17] pupil (max, se, sex, sey), [18..21] iris

struct Lse { float m, s, sx, sy; };

__device__ __forceinline__ void lse_add(Lse& a, float v, float x, float y) {
  if (v > a.m) {
    const float sc = expf(a.m - v);
    a.s = a.s * sc + 1.f; a.sx = a.sx * sc + x; a.sy = a.sy * sc + y; a.m = v;
  } else {
    const float e = expf(v - a.m);
    a.s += e; a.sx += e * x; a.sy += e * y;
  }
}
__device__ __forceinline__ void lse_merge(Lse& a, const Lse& b) {
  const float m = fmaxf(a.m, b.m);
  if (m == -INFINITY) return;
  const float ea = expf(a.m - m), eb = expf(b.m - m);
  a.s = a.s * ea + b.s * eb; a.sx = a.sx * ea + b.sx * eb; a.sy = a.sy * ea + b.sy * eb; a.m = m;
}

// torch.linspace(-1, 1, n)[i] in float32 (ATen RangeFactories: symmetric two-sided formula)
__device__ __forceinline__ float lin11(int i, int n) {
  const float step = 2.0f / (float)(n - 1);
  return i < n / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

template <typename T>
__global__ __launch_bounds__(256) void loss_partial_k(const egne_loss_desc d, int nblk) {
  const int b = blockIdx.y, blk = blockIdx.x;
  const int HW = d.H * d.W;
  const int p0 = blk * PIX_PER_BLOCK;
  const int p1 = min(p0 + PIX_PER_BLOCK, HW);
  float acc[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) acc[i] = 0.f;
  Lse pup = {-INFINITY, 0.f, 0.f, 0.f}, iri = {-INFINITY, 0.f, 0.f, 0.f};
  const long long base = (long long)b * HW;
  for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    const T* lp = (const T*)d.logits + (base + p) * d.pix_stride + d.ch_off;
    const float l0 = ld1(lp), l1 = ld1(lp + 1), l2 = ld1(lp + 2);
    const int t = (int)d.target[base + p];
    const float mx = fmaxf(l0, fmaxf(l1, l2));
    const float e0 = expf(l0 - mx), e1 = expf(l1 - mx), e2 = expf(l2 - mx);
    const float se = e0 + e1 + e2, inv = 1.f / se;
    const float pr[3] = {e0 * inv, e1 * inv, e2 * inv};
    const float lt = t == 0 ? l0 : (t == 1 ? l1 : l2);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      acc[c] += pr[c] * d.distMap[((long long)b * 3 + c) * HW + p];
      acc[3 + c] += (t == c) ? 1.f : 0.f;
      acc[6 + c] += (t == c) ? pr[c] : 0.f;
      acc[9 + c] += pr[c];
    }
    acc[12] += (mx - lt) + logf(se);
    acc[13] += d.spatWts[base + p];
    const int y = p / d.W, x = p - y * d.W;
    const float gx = d.grid_x ? d.grid_x[x] : lin11(x, d.W), gy = d.grid_y ? d.grid_y[y] : lin11(y, d.H);
    lse_add(pup, 4.f * l2, gx, gy);
    lse_add(iri, -4.f * l0, gx, gy);
    if (d.mask) d.mask[base + p] = (l1 > l0) ? ((l2 > l1) ? 2 : 1) : ((l2 > l0) ? 2 : 0);
    if (d.op_nchw) {
      d.op_nchw[((long long)b * 3 + 0) * HW + p] = l0;
      d.op_nchw[((long long)b * 3 + 1) * HW + p] = l1;
      d.op_nchw[((long long)b * 3 + 2) * HW + p] = l2;
    }
  }
  // wave reduction, then across the 4 waves through LDS
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
    for (int i = 0; i < 14; ++i) acc[i] += __shfl_xor(acc[i], o);
    Lse t1 = {__shfl_xor(pup.m, o), __shfl_xor(pup.s, o), __shfl_xor(pup.sx, o), __shfl_xor(pup.sy, o)};
    lse_merge(pup, t1);
    Lse t2 = {__shfl_xor(iri.m, o), __shfl_xor(iri.s, o), __shfl_xor(iri.sx, o), __shfl_xor(iri.sy, o)};
    lse_merge(iri, t2);
  }
  __shared__ float sh[4][NPART];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 14; ++i) sh[wave][i] = acc[i];
    sh[wave][14] = pup.m; sh[wave][15] = pup.s; sh[wave][16] = pup.sx; sh[wave][17] = pup.sy;
    sh[wave][18] = iri.m; sh[wave][19] = iri.s; sh[wave][20] = iri.sx; sh[wave][21] = iri.sy;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* out = d.partials + ((long long)b * nblk + blk) * NPART;
    for (int i = 0; i < 14; ++i) out[i] = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
    Lse a = {sh[0][14], sh[0][15], sh[0][16], sh[0][17]}, c = {sh[0][18], sh[0][19], sh[0][20], sh[0][21]};
    for (int w = 1; w < 4; ++w) {
      Lse t1 = {sh[w][14], sh[w][15], sh[w][16], sh[w][17]};
      lse_merge(a, t1);
      Lse t2 = {sh[w][18], sh[w][19], sh[w][20], sh[w][21]};
      lse_merge(c, t2);
    }
    out[14] = a.m; out[15] = a.s; out[16] = a.sx; out[17] = a.sy;
    out[18] = c.m; out[19] = c.s; out[20] = c.sx; out[21] = c.sy;
    out[22] = 0.f; out[23] = 0.f;
  }
}

// one block; thread i handles samples i, i+blockDim, ...; then a block reduction over the batch.
__global__ __launch_bounds__(256) void loss_final_k(const egne_loss_desc d, int nblk) {
  const int HW = d.H * d.W;
  const float fHW = (float)HW;
  // batch accumulators: [0] sum seg_i (valid), [1] n_mask, [2] sum pup l1, [3] sum iri l1 * mask,
  // [4] sum pt (mask absent), [5] sum ellipse (mask present), [6] bad flag
  float bt[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < d.B; b += blockDim.x) {
    float a[14];
    for (int i = 0; i < 14; ++i) a[i] = 0.f;
    Lse pup = {-INFINITY, 0.f, 0.f, 0.f}, iri = {-INFINITY, 0.f, 0.f, 0.f};
    for (int k = 0; k < nblk; ++k) {
      const float* r = d.partials + ((long long)b * nblk + k) * NPART;
      for (int i = 0; i < 14; ++i) a[i] += r[i];
      Lse t1 = {r[14], r[15], r[16], r[17]};
      lse_merge(pup, t1);
      Lse t2 = {r[18], r[19], r[20], r[21]};
      lse_merge(iri, t2);
    }
    const float mp = 1.f - d.cond[b * 4 + 1];  // mask present
    // centres of mass (loss.py:40-42)
    const float cpx = pup.sx / pup.s, cpy = pup.sy / pup.s;
    const float cix = iri.sx / iri.s, ciy = iri.sy / iri.s;
    // normPts (utils.py:627-634)
    const float gpx = 2.f * (d.pupil_center[b * 2 + 0] / (float)d.W) - 1.f;
    const float gpy = 2.f * (d.pupil_center[b * 2 + 1] / (float)d.H) - 1.f;
    bt[2] += fabsf(cpx - gpx) + fabsf(cpy - gpy);
    bt[3] += mp * (fabsf(cix - d.elNorm[b * 10 + 0]) + fabsf(ciy - d.elNorm[b * 10 + 1]));
    bt[1] += mp;
    d.pred_c[b * 4 + 0] = cix; d.pred_c[b * 4 + 1] = ciy;  // iris first (provisional, see below)
    d.pred_c[b * 4 + 2] = cpx; d.pred_c[b * 4 + 3] = cpy;
    float* cf = d.coef ? d.coef + b * 32 : nullptr;   // per-sample state for egne_loss_bwd
    if (cf) {
      cf[0] = mp; cf[1] = cf[2] = cf[3] = 0.f; cf[4] = 0.f; cf[5] = 1.f; cf[6] = 0.f; cf[7] = a[13] / fHW;
      cf[8] = cpx; cf[9] = cpy; cf[10] = cix; cf[11] = ciy;
      cf[12] = pup.m; cf[13] = pup.s; cf[14] = iri.m; cf[15] = iri.s; cf[16] = gpx; cf[17] = gpy;
    }
    if (mp == 1.f) {
      // SurfaceLoss (loss.py:86-92)
      const float l_sl = ((a[0] / fHW + a[1] / fHW) + a[2] / fHW) / 3.f;
      // GDiceLoss (loss.py:94-121)
      float A = 0.f, Bq = 0.f;
      int absent = 0;
      for (int c = 0; c < 3; ++c) {
        float w = 0.f;
        if (a[3 + c] > 0.f) w = 1.f / fmaxf(a[3 + c] * a[3 + c], 1e-5f); else ++absent;
        A += w * a[6 + c];
        Bq += w * (a[9 + c] + a[3 + c]);
      }
      const float l_gd = 1.f - fmaxf(2.f * A / Bq, 1e-5f);
      if (cf) {
        for (int c = 0; c < 3; ++c) cf[1 + c] = a[3 + c] > 0.f ? 1.f / fmaxf(a[3 + c] * a[3 + c], 1e-5f) : 0.f;
        cf[4] = A; cf[5] = Bq; cf[6] = (2.f * A / Bq > 1e-5f) ? 1.f : 0.f;
      }
      // wCE (loss.py:123-137): mean(spatWts) * CE_mean; ignore_index is the absent class, which by
      // construction labels no pixel, so CE_mean is the plain mean
      const float l_ce = (a[13] / fHW) * (a[12] / fHW);
      if (absent > 1) bt[6] += 1.f;  // reference raises (rmIdx.item() on 2 elements)
      bt[0] += d.alpha * l_sl + (1.f - d.alpha) * l_gd + l_ce;
      float e = 0.f;
      for (int j = 0; j < 10; ++j) e += fabsf(d.elOut[b * 10 + j] - d.elNorm[b * 10 + j]);
      bt[5] += e / 10.f;
    } else {
      bt[4] += (fabsf(d.elOut[b * 10 + 5] - gpx) + fabsf(d.elOut[b * 10 + 6] - gpy)) / 2.f;
    }
  }
  __shared__ float sh[256][7];
  for (int i = 0; i < 7; ++i) sh[threadIdx.x][i] = bt[i];
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (threadIdx.x < s)
      for (int i = 0; i < 7; ++i) sh[threadIdx.x][i] += sh[threadIdx.x + s][i];
    __syncthreads();
  }
  const float nmask = sh[0][1];
  const float fB = (float)d.B;
  if (threadIdx.x == 0) {
    const float l_pup = sh[0][2] / (2.f * fB);
    const float l_iri = nmask > 0.f ? sh[0][3] / (2.f * nmask) : 0.f;
    const float l_seg2pt = 0.5f * l_pup + 0.5f * l_iri;
    const float l_seg = nmask > 0.f ? sh[0][0] / nmask : 0.f;
    const float nabs = fB - nmask;
    const float l_pt = nabs > 0.f ? sh[0][4] / nabs : 0.f;
    const float l_ell = nmask > 0.f ? sh[0][5] / nmask : 0.f;
    d.out_terms[0] = l_seg2pt + 20.f * l_seg + 10.f * (l_pt + l_ell);
    d.out_terms[1] = l_seg2pt; d.out_terms[2] = l_seg; d.out_terms[3] = l_pt; d.out_terms[4] = l_ell;
    d.out_terms[5] = nmask; d.out_terms[6] = sh[0][6]; d.out_terms[7] = 0.f;
  }
  // elPred (RITnet_v2.py:334-335); iris centre falls back to elOut[:,5:7] when the batch has no mask
  for (int b = threadIdx.x; b < d.B; b += blockDim.x) {
    if (!(nmask > 0.f)) {
      d.pred_c[b * 4 + 0] = d.elOut[b * 10 + 5];
      d.pred_c[b * 4 + 1] = d.elOut[b * 10 + 6];
    }
    float* ep = d.elPred + b * 10;
    ep[0] = d.pred_c[b * 4 + 0]; ep[1] = d.pred_c[b * 4 + 1];
    ep[2] = d.elOut[b * 10 + 2]; ep[3] = d.elOut[b * 10 + 3]; ep[4] = d.elOut[b * 10 + 4];
    ep[5] = d.pred_c[b * 4 + 2]; ep[6] = d.pred_c[b * 4 + 3];
    ep[7] = d.elOut[b * 10 + 7]; ep[8] = d.elOut[b * 10 + 8]; ep[9] = d.elOut[b * 10 + 9];
  }
}

inline int loss_nblk(int H, int W) { return (H * W + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK; }

}  // namespace

extern "C" int64_t egne_loss_workspace_floats(int B, int H, int W) {
  return (int64_t)B * loss_nblk(H, W) * NPART;
}

extern "C" int egne_loss_fwd(const egne_loss_desc* dp, void* stream) {
  EGNE_REQUIRE(dp, "loss: null descriptor");
  const egne_loss_desc& d = *dp;
  EGNE_REQUIRE(d.B > 0 && d.H > 1 && d.W > 1, "loss: bad shape");
  EGNE_REQUIRE(d.logits && d.ch_off + 3 <= d.pix_stride, "loss: bad logits slice");
  EGNE_REQUIRE(d.target && d.spatWts && d.distMap && d.cond && d.pupil_center && d.elNorm && d.elOut, "loss: null input");
  EGNE_REQUIRE(d.partials && d.out_terms && d.pred_c && d.elPred, "loss: null output/workspace");
  const int nblk = loss_nblk(d.H, d.W);
  hipStream_t st = (hipStream_t)stream;
  EGNE_REQUIRE(d.dtype == 0 || d.dtype == 1, "loss: dtype %d", d.dtype);
  if (d.dtype == 1) hipLaunchKernelGGL(loss_partial_k<egne_bf16>, dim3(nblk, d.B), dim3(256), 0, st, d, nblk);
  else hipLaunchKernelGGL(loss_partial_k<float>, dim3(nblk, d.B), dim3(256), 0, st, d, nblk);
  hipLaunchKernelGGL(loss_final_k, dim3(1), dim3(256), 0, st, d, nblk);
  return egne::check_launch("egne_loss_fwd");
}

// ---- loss of the DeepVOG comparator (models/deepvog_pytorch.py:148-167 get_allLoss), forward only -------------------------------
// Two output channels; the target is (label == 2).  l_seg = 10 * cross_entropy(softmax(op), target) averaged per frame, then over the
// frames whose mask is present (cond[:,1] == 0); plus the mean L1 distance of the soft-argmax centre of channel 1 (temperature 4) to
// the normalised pupil centre.  Same two passes as above: per-block partials, one combining block.
namespace {

constexpr int DV_NPART = 8;   // (max, se, sex, sey) of channel 1 * 4, [4] sum CE, [5..7] unused

__global__ __launch_bounds__(256) void deepvog_loss_partial_k(const float* __restrict__ logits, long long ps, int ch_off,
                                                             const long long* __restrict__ target, int H, int W, int nblk,
                                                             float* __restrict__ partials, float* __restrict__ op_nchw,
                                                             long long* __restrict__ mask) {
  const int b = blockIdx.y, blk = blockIdx.x, hw = H * W;
  const int p0 = blk * PIX_PER_BLOCK, p1 = min(hw, p0 + PIX_PER_BLOCK);
  Lse a{-INFINITY, 0.f, 0.f, 0.f};
  float ce = 0.f;
  for (int p = p0 + threadIdx.x; p < p1; p += 256) {
    const long long gp = (long long)b * hw + p;
    const float* q = logits + gp * ps + ch_off;
    const float v0 = q[0], v1 = q[1];
    op_nchw[((long long)b * 2) * hw + p] = v0;
    op_nchw[((long long)b * 2 + 1) * hw + p] = v1;
    mask[gp] = v1 > v0 ? 1 : 0;                                   // torch.max returns the first maximum
    const float m = fmaxf(v0, v1), e0 = expf(v0 - m), e1 = expf(v1 - m), inv = 1.f / (e0 + e1);
    const float s0 = e0 * inv, s1 = e1 * inv;                     // softmax over the two channels
    const float sm = fmaxf(s0, s1), lse = sm + logf(expf(s0 - sm) + expf(s1 - sm));
    ce += lse - (target[gp] == 2 ? s1 : s0);                      // F.cross_entropy applied to the PROBABILITIES (:160)
    const int y = p / W, x = p - y * W;
    lse_add(a, 4.f * v1, lin11(x, W), lin11(y, H));
  }
  __shared__ Lse sh[256];
  __shared__ float shc[256];
  sh[threadIdx.x] = a; shc[threadIdx.x] = ce;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { lse_merge(sh[threadIdx.x], sh[threadIdx.x + s]); shc[threadIdx.x] += shc[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* r = partials + ((long long)b * nblk + blk) * DV_NPART;
    r[0] = sh[0].m; r[1] = sh[0].s; r[2] = sh[0].sx; r[3] = sh[0].sy; r[4] = shc[0];
  }
}

// one block; thread b handles frame b (strided); out_terms[0] = loss, [1] = l_seg, [2] = mean seg2pt term
__global__ __launch_bounds__(256) void deepvog_loss_final_k(const float* __restrict__ partials, int nblk, int B, int H, int W,
                                                           const float* __restrict__ pupil_center, const float* __restrict__ cond,
                                                           float* __restrict__ out_terms, float* __restrict__ pred_c) {
  __shared__ double sseg[256], sok[256], spt[256];
  double seg = 0.0, ok = 0.0, pt = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) {
    Lse a{-INFINITY, 0.f, 0.f, 0.f};
    double ce = 0.0;
    for (int k = 0; k < nblk; ++k) {
      const float* r = partials + ((long long)b * nblk + k) * DV_NPART;
      Lse t{r[0], r[1], r[2], r[3]};
      lse_merge(a, t);
      ce += r[4];
    }
    const float cx = a.sx / a.s, cy = a.sy / a.s;
    pred_c[b * 2] = cx; pred_c[b * 2 + 1] = cy;
    const float gx = 2.f * (pupil_center[b * 2] / (float)W) - 1.f, gy = 2.f * (pupil_center[b * 2 + 1] / (float)H) - 1.f;   // utils.normPts
    pt += fabsf(cx - gx) + fabsf(cy - gy);
    const float w = 1.f - cond[b * 4 + 1];
    seg += 10.0 * (ce / ((double)H * W)) * w;
    ok += w;
  }
  sseg[threadIdx.x] = seg; sok[threadIdx.x] = ok; spt[threadIdx.x] = pt;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sseg[threadIdx.x] += sseg[threadIdx.x + s]; sok[threadIdx.x] += sok[threadIdx.x + s]; spt[threadIdx.x] += spt[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double lseg = sok[0] != 0.0 ? sseg[0] / sok[0] : 0.0, lpt = spt[0] / (2.0 * B);
    out_terms[0] = (float)(lseg + lpt); out_terms[1] = (float)lseg; out_terms[2] = (float)lpt;
  }
}

// Backward of the DeepVOG loss w.r.t. the two logits of every pixel (gradient of `gscale` * loss), from the forward's pred_c and per-block
// partials (the soft-argmax normaliser is rebuilt from them).  g = d l_seg + d l_pt:
//   l_seg: w_b = 10 * ok_b / (sum ok * H * W);  q = softmax(p) with p = softmax(op);  dp_c = w_b (q_c - [c == t]);  dop_c = p_c (dp_c - sum_k dp_k p_k)
//   l_pt : pred = sum_i s_i (x_i, y_i), s = softmax_i(4 op_1);  dop_1,i += 4 s_i ((x_i - px) sgn_x + (y_i - py) sgn_y) / (2 B)
__global__ __launch_bounds__(256) void deepvog_loss_bwd_k(const float* __restrict__ logits, long long ps, int ch_off,
                                                         const long long* __restrict__ target, const float* __restrict__ pupil_center,
                                                         const float* __restrict__ cond, const float* __restrict__ partials,
                                                         const float* __restrict__ pred_c, const float* __restrict__ gscale, int B, int H,
                                                         int W, int nblk, float* __restrict__ g, long long gs, int go) {
  const int b = blockIdx.y, hw = H * W;
  __shared__ float sh[4];
  if (threadIdx.x == 0) {
    Lse a{-INFINITY, 0.f, 0.f, 0.f};
    for (int k = 0; k < nblk; ++k) {
      const float* r = partials + ((long long)b * nblk + k) * DV_NPART;
      Lse t{r[0], r[1], r[2], r[3]};
      lse_merge(a, t);
    }
    float ok = 0.f;
    for (int i = 0; i < B; ++i) ok += 1.f - cond[i * 4 + 1];
    sh[0] = a.m; sh[1] = a.s;
    sh[2] = ok != 0.f ? 10.f * (1.f - cond[b * 4 + 1]) / (ok * (float)hw) : 0.f;
  }
  __syncthreads();
  const float m = sh[0], S = sh[1], wseg = sh[2] * gscale[0];
  const float px = pred_c[b * 2], py = pred_c[b * 2 + 1];
  const float tx = 2.f * (pupil_center[b * 2] / (float)W) - 1.f, ty = 2.f * (pupil_center[b * 2 + 1] / (float)H) - 1.f;
  const float sx = (px > tx) - (px < tx), sy = (py > ty) - (py < ty);
  const float wpt = gscale[0] * 4.f / (2.f * (float)B);
  for (int p = blockIdx.x * 256 + threadIdx.x; p < hw; p += gridDim.x * 256) {
    const long long gp = (long long)b * hw + p;
    const float* q = logits + gp * ps + ch_off;
    const float v0 = q[0], v1 = q[1];
    const float mm = fmaxf(v0, v1), e0 = expf(v0 - mm), e1 = expf(v1 - mm), inv = 1.f / (e0 + e1);
    const float p0 = e0 * inv, p1 = e1 * inv;
    const float pm = fmaxf(p0, p1), f0 = expf(p0 - pm), f1 = expf(p1 - pm), fi = 1.f / (f0 + f1);
    const int t = target[gp] == 2;
    const float d0 = wseg * (f0 * fi - (t ? 0.f : 1.f)), d1 = wseg * (f1 * fi - (t ? 1.f : 0.f));
    const float dot = d0 * p0 + d1 * p1;
    const int y = p / W, x = p - y * W;
    const float s = expf(4.f * v1 - m) / S;
    float* o = g + gp * gs + go;
    o[0] = p0 * (d0 - dot);
    o[1] = p1 * (d1 - dot) + wpt * s * ((lin11(x, W) - px) * sx + (lin11(y, H) - py) * sy);
  }
}

}  // namespace

extern "C" int egne_deepvog_loss_bwd(const float* logits, int64_t pix_stride, int ch_off, const int64_t* target, const float* pupil_center,
                                     const float* cond, int B, int H, int W, const float* partials, const float* pred_c, const float* gscale,
                                     float* g_logits, int64_t gs, int go, void* stream) {
  EGNE_REQUIRE(logits && target && pupil_center && cond && partials && pred_c && gscale && g_logits, "deepvog_loss_bwd: null pointer");
  EGNE_REQUIRE(B > 0 && H > 1 && W > 1 && ch_off + 2 <= pix_stride && go + 2 <= gs, "deepvog_loss_bwd: bad shape");
  hipLaunchKernelGGL(deepvog_loss_bwd_k, dim3(64, B), dim3(256), 0, (hipStream_t)stream, logits, (long long)pix_stride, ch_off,
                     (const long long*)target, pupil_center, cond, partials, pred_c, gscale, B, H, W, loss_nblk(H, W), g_logits, (long long)gs, go);
  return egne::check_launch("egne_deepvog_loss_bwd");
}

extern "C" int64_t egne_deepvog_loss_workspace_floats(int B, int H, int W) { return (int64_t)B * loss_nblk(H, W) * DV_NPART; }

extern "C" int egne_deepvog_loss_fwd(const float* logits, int64_t pix_stride, int ch_off, const int64_t* target, const float* pupil_center,
                                     const float* cond, int B, int H, int W, float* partials, float* out_terms, float* pred_c,
                                     float* op_nchw, int64_t* mask, void* stream) {
  EGNE_REQUIRE(logits && target && pupil_center && cond && partials && out_terms && pred_c && op_nchw && mask, "deepvog_loss: null pointer");
  EGNE_REQUIRE(B > 0 && H > 1 && W > 1 && ch_off + 2 <= pix_stride, "deepvog_loss: bad shape");
  const int nblk = loss_nblk(H, W);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(deepvog_loss_partial_k, dim3(nblk, B), dim3(256), 0, st, logits, (long long)pix_stride, ch_off, (const long long*)target,
                     H, W, nblk, partials, op_nchw, (long long*)mask);
  hipLaunchKernelGGL(deepvog_loss_final_k, dim3(1), dim3(256), 0, st, partials, nblk, B, H, W, pupil_center, cond, out_terms, pred_c);
  return egne::check_launch("egne_deepvog_loss_fwd");
}
