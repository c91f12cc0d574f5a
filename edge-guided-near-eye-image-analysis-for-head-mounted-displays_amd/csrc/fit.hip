// Ellipse fit of evaluate.py on the device: utils.py:450-486 (coordinate hill-climb over a, b, angle)
// with utils.py:176-204 (IoU of the class mask with a rasterised ellipse on the [-1,1] mesh) and the
// float64 conic algebra of helperfunctions.py:13-63,102-129.
//
// One workgroup per (frame, class).  The class mask is bit-packed into LDS once; every IoU evaluation
// is then a popcount pass over those words, so the <=281 sequential evaluations of the search never
// leave the CU (the reference builds each map on the host, copies it and calls .item() three times).
//
// Numerics follow the reference bit for bit: float32 mesh supplied by the host (torch.linspace, the
// same call create_meshgrid makes), float32 map arithmetic with one rounding per operation (built with
// -ffp-contract=off and written with the _rn intrinsics), float64 conic normalisation, 3.14159.
#include "common.h"

namespace {

constexpr int NT = 512;
constexpr double PI_REF = 3.14159;
constexpr double EPS_B = 1e-40;  // helperfunctions.py:10

struct M3 { double v[3][3]; };

__device__ M3 mul(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = a.v[i][0] * b.v[0][j];
      s = s + a.v[i][1] * b.v[1][j];
      s = s + a.v[i][2] * b.v[2][j];
      r.v[i][j] = s;
    }
  return r;
}
__device__ M3 tr(const M3& a) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.v[i][j] = a.v[j][i];
  return r;
}
__device__ M3 rot(double t) {
  const double c = cos(t), s = sin(t);
  M3 r = {{{c, -s, 0.0}, {s, c, 0.0}, {0.0, 0.0, 1.0}}};
  return r;
}
__device__ M3 trans(double x, double y) {
  M3 r = {{{1.0, 0.0, x}, {0.0, 1.0, y}, {0.0, 0.0, 1.0}}};
  return r;
}

// pixel ellipse (cx,cy,a,b,theta) -> parameters on the normalised mesh (helperfunctions.py:25-33,
// :124-129 with H = [[2/W,0,-1],[0,2/H,-1],[0,0,1]], :50-63)
__device__ void normalise(const double* el, int Hh, int Ww, double* out) {
  const M3 Hr = rot(-el[4]), Ht = trans(-el[0], -el[1]);
  M3 Q = {{{1.0 / (el[2] * el[2]), 0, 0}, {0, 1.0 / (el[3] * el[3]), 0}, {0, 0, -1.0}}};
  M3 mat = mul(mul(mul(mul(tr(Ht), tr(Hr)), Q), Hr), Ht);
  // inverse of the normalising homography, analytically
  M3 Hi = {{{Ww / 2.0, 0, Ww / 2.0}, {0, Hh / 2.0, Hh / 2.0}, {0, 0, 1.0}}};
  M3 mt = mul(mul(tr(Hi), mat), Hi);
  const double a = mt.v[0][0], b = 2 * mt.v[0][1], c = mt.v[1][1], dd = 2 * mt.v[0][2], e = 2 * mt.v[1][2];
  double theta;
  if (fabs(b) <= EPS_B && a <= c) theta = 0.0;
  else if (fabs(b) <= EPS_B && a > c) theta = 3.141592653589793 / 2;
  else theta = 0.5 * atan2(b, a - c);
  const double den = b * b - 4 * a * c;
  const double tx = (2 * c * dd - b * e) / den, ty = (2 * a * e - b * dd) / den;
  const M3 R = rot(theta), T = trans(tx, ty);
  M3 mn = mul(mul(mul(mul(tr(R), tr(T)), mt), T), R);
  out[0] = tx; out[1] = ty;
  out[2] = sqrt(1.0 / mn.v[0][0]); out[3] = sqrt(1.0 / mn.v[1][1]);
  out[4] = theta;
}

struct Shared {
  float prm[6];          // cx, cy, a, b, cos, sin on the mesh (float32)
  unsigned red[2][NT / 64];
  float score;
  double now[3], d[3], rt;
  int flag, nseg;
};

__global__ __launch_bounds__(NT) void ellipse_fit_k(const long long* __restrict__ mask, int nframes, const int* __restrict__ frame_of,
                                                    const int* __restrict__ cls, int H, int W,
                                                    const float* __restrict__ xs, const float* __restrict__ ys,
                                                    const double* __restrict__ init, double* __restrict__ out,
                                                    int* __restrict__ evals) {
  extern __shared__ unsigned bits[];  // [H][wpr] packed mask, then xs[W], ys[H]
  __shared__ Shared sh;
  const int e = blockIdx.x, tid = threadIdx.x;
  const int wpr = (W + 31) >> 5, nwords = H * wpr;
  float* lxs = (float*)(bits + nwords);
  float* lys = lxs + W;
  const int fr = frame_of[e];
  if (fr < 0 || fr >= nframes) {   // a fit that names a frame the mask tensor does not hold: report NaN, read nothing
    if (tid < 5) out[e * 5 + tid] = __longlong_as_double(0x7ff8000000000000ll);
    if (tid == 0 && evals) evals[e] = 0;
    return;
  }
  const long long* m = mask + (long long)fr * H * W;
  const int k = cls[e];
  unsigned cnt = 0;
  for (int w = tid; w < nwords; w += NT) {
    const int y = w / wpr, x0 = (w - y * wpr) << 5;
    unsigned word = 0;
    for (int j = 0; j < 32; ++j) {
      const int x = x0 + j;
      if (x < W && m[(long long)y * W + x] == k) word |= 1u << j;
    }
    bits[w] = word;
    cnt += __popc(word);
  }
  for (int i = tid; i < W; i += NT) lxs[i] = xs[i];
  for (int i = tid; i < H; i += NT) lys[i] = ys[i];
  // block sum of cnt
  for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((tid & 63) == 0) sh.red[0][tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) {
    unsigned s = 0;
    for (int i = 0; i < NT / 64; ++i) s += sh.red[0][i];
    sh.nseg = (int)s;
  }
  __syncthreads();

  const double cx = init[e * 5 + 0], cy = init[e * 5 + 1];

  // IoU of the packed mask with the ellipse (cx, cy, now[0], now[1], now[2] degrees): all threads call it
  auto evaluate = [&]() -> float {
    if (tid == 0) {
      double el[5] = {cx, cy, sh.now[0], sh.now[1], sh.now[2] / 180. * PI_REF};
      double nm[5];
      normalise(el, H, W, nm);
      sh.prm[0] = (float)nm[0]; sh.prm[1] = (float)nm[1]; sh.prm[2] = (float)nm[2]; sh.prm[3] = (float)nm[3];
      sh.prm[4] = (float)cos(nm[4]); sh.prm[5] = (float)sin(nm[4]);
    }
    __syncthreads();
    const float ecx = sh.prm[0], ecy = sh.prm[1], ea = sh.prm[2], eb = sh.prm[3], ct = sh.prm[4], st = sh.prm[5];
    // Only pixels inside the ellipse count (ne, ni), and the ellipse lies within max(a, b) of its centre: scan the
    // bounding box (0.1 % + 2 pixels of slack, orders of magnitude above float32 round-off) instead of the frame.
    // Degenerate parameters (NaN / huge axes) fall back to the full frame, where the map is evaluated as before.
    int y_lo = 0, y_hi = H - 1, w_lo = 0, w_hi = wpr - 1;
    const float rr = fmaxf(ea, eb) * 1.001f;
    if (rr < 4.f && fabsf(ecx) < 4.f && fabsf(ecy) < 4.f) {
      const float sx = 0.5f * (float)(W - 1), sy = 0.5f * (float)(H - 1);
      const int xl = (int)floorf((ecx - rr + 1.f) * sx) - 2, xh = (int)ceilf((ecx + rr + 1.f) * sx) + 2;
      const int yl = (int)floorf((ecy - rr + 1.f) * sy) - 2, yh = (int)ceilf((ecy + rr + 1.f) * sy) + 2;
      y_lo = max(yl, 0); y_hi = min(yh, H - 1);
      w_lo = max(xl, 0) >> 5; w_hi = min(xh, W - 1) >> 5;
    }
    const int bw = w_hi - w_lo + 1, nbox = (y_hi >= y_lo && bw > 0) ? (y_hi - y_lo + 1) * bw : 0;
    unsigned ne = 0, ni = 0;
    for (int q = tid; q < nbox; q += NT) {
      const int yq = q / bw, y = y_lo + yq, wx = w_lo + (q - yq * bw), w = y * wpr + wx, x0 = wx << 5;
      const float dy = __fsub_rn(lys[y], ecy);
      const float dyst = __fmul_rn(dy, st), dyct = __fmul_rn(dy, ct);
      unsigned word = 0;
      for (int j = 0; j < 32; ++j) {
        const int x = x0 + j;
        if (x < W) {
          const float dx = __fsub_rn(lxs[x], ecx);
          const float X = __fadd_rn(__fmul_rn(dx, ct), dyst);
          const float Y = __fadd_rn(__fmul_rn(-dx, st), dyct);
          const float u = __fdiv_rn(X, ea), v = __fdiv_rn(Y, eb);
          const float wt = __fsub_rn(__fadd_rn(__fmul_rn(u, u), __fmul_rn(v, v)), 1.0f);
          if (wt <= 0.f) word |= 1u << j;
        }
      }
      ne += __popc(word);
      ni += __popc(word & bits[w]);
    }
    for (int o = 32; o >= 1; o >>= 1) { ne += __shfl_xor(ne, o); ni += __shfl_xor(ni, o); }
    if ((tid & 63) == 0) { sh.red[0][tid >> 6] = ne; sh.red[1][tid >> 6] = ni; }
    __syncthreads();
    if (tid == 0) {
      unsigned a = 0, b = 0;
      for (int i = 0; i < NT / 64; ++i) { a += sh.red[0][i]; b += sh.red[1][i]; }
      const float fi = (float)b;
      sh.score = __fdiv_rn(fi, __fsub_rn(__fadd_rn((float)sh.nseg, (float)a), fi));
    }
    __syncthreads();
    return sh.score;
  };

  if (tid == 0) {
    sh.now[0] = init[e * 5 + 2]; sh.now[1] = init[e * 5 + 3]; sh.now[2] = init[e * 5 + 4] * 180. / PI_REF;
    sh.d[0] = sh.d[1] = sh.d[2] = 1.0;
  }
  __syncthreads();
  int nev = 1;
  float s0 = evaluate();
  if (tid == 0) sh.rt = (double)s0;
  __syncthreads();
  for (int sweep = 0; sweep < 40; ++sweep) {
    if (tid == 0) sh.flag = 0;
    __syncthreads();
    for (int j = 0; j < 3; ++j) {
      if (tid == 0) sh.now[j] -= sh.d[j];
      __syncthreads();
      float sc = evaluate(); ++nev;
      bool better = (double)sc > sh.rt;  // uniform: sh.rt only changes between sweeps
      if (better) { if (tid == 0) sh.flag = 1; __syncthreads(); continue; }
      if (tid == 0) sh.now[j] += 2. * sh.d[j];
      __syncthreads();
      sc = evaluate(); ++nev;
      better = (double)sc > sh.rt;
      if (better) { if (tid == 0) sh.flag = 1; __syncthreads(); continue; }
      if (tid == 0) { sh.now[j] -= sh.d[j]; sh.d[j] *= 0.8; }
      __syncthreads();
    }
    const float sc = evaluate(); ++nev;
    __syncthreads();
    const int flag = sh.flag;
    __syncthreads();
    if (tid == 0 && (double)sc > sh.rt) sh.rt = (double)sc;
    __syncthreads();
    if (!flag) break;
  }
  if (tid == 0) {
    out[e * 5 + 0] = cx; out[e * 5 + 1] = cy; out[e * 5 + 2] = sh.now[0]; out[e * 5 + 3] = sh.now[1];
    out[e * 5 + 4] = sh.now[2] / 180.0 * PI_REF;
    if (evals) evals[e] = nev;
  }
}

// evaluate.py:135-151: the regressed ellipses (normalised [-1,1] coordinates, float32) -> pixel ellipses that seed the
// search: my_ellipse(p).transform(H)[0][:-1] with H = [[W/2,0,W/2],[0,H/2,H/2],[0,0,1]] (helperfunctions.py:25-33,
// :50-63,:124-129) in float64.  Fit 2f = iris (elPred[f,0:5], class 1), fit 2f+1 = pupil (elPred[f,5:10], class 2).
__global__ void ellipse_init_k(const float* __restrict__ elPred, int nframes, int H, int W, double* __restrict__ init,
                               int* __restrict__ frame_of, int* __restrict__ cls) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * nframes) return;
  const int f = e >> 1, k = e & 1;
  double el[5];
  for (int i = 0; i < 5; ++i) el[i] = (double)elPred[f * 10 + 5 * k + i];
  const M3 Hr = rot(-el[4]), Ht = trans(-el[0], -el[1]);
  M3 Q = {{{1.0 / (el[2] * el[2]), 0, 0}, {0, 1.0 / (el[3] * el[3]), 0}, {0, 0, -1.0}}};
  const M3 mat = mul(mul(mul(mul(tr(Ht), tr(Hr)), Q), Hr), Ht);
  const M3 Hi = {{{2.0 / W, 0, -1.0}, {0, 2.0 / H, -1.0}, {0, 0, 1.0}}};       // inverse of the un-normalising homography
  const M3 mt = mul(mul(tr(Hi), mat), Hi);
  const double a = mt.v[0][0], b = 2 * mt.v[0][1], c = mt.v[1][1], dd = 2 * mt.v[0][2], ee = 2 * mt.v[1][2];
  double theta;
  if (fabs(b) <= EPS_B && a <= c) theta = 0.0;
  else if (fabs(b) <= EPS_B && a > c) theta = 3.141592653589793 / 2;
  else theta = 0.5 * atan2(b, a - c);
  const double den = b * b - 4 * a * c;
  const double tx = (2 * c * dd - b * ee) / den, ty = (2 * a * ee - b * dd) / den;
  const M3 R = rot(theta), T = trans(tx, ty);
  const M3 mn = mul(mul(mul(mul(tr(R), tr(T)), mt), T), R);
  init[e * 5 + 0] = tx; init[e * 5 + 1] = ty;
  init[e * 5 + 2] = sqrt(1.0 / mn.v[0][0]); init[e * 5 + 3] = sqrt(1.0 / mn.v[1][1]);
  init[e * 5 + 4] = theta;
  frame_of[e] = f;
  cls[e] = 1 + k;
}

}  // namespace

extern "C" int egne_ellipse_init_from_pred(const float* elPred, int nframes, int H, int W, double* init, int32_t* frame_of,
                                           int32_t* cls, void* stream) {
  EGNE_REQUIRE(elPred && init && frame_of && cls && nframes > 0 && H > 1 && W > 1, "ellipse_init_from_pred: bad arguments");
  hipLaunchKernelGGL(ellipse_init_k, dim3((2 * nframes + 63) / 64), dim3(64), 0, (hipStream_t)stream, elPred, nframes, H, W, init,
                     frame_of, cls);
  return egne::check_launch("egne_ellipse_init_from_pred");
}

extern "C" int egne_ellipse_fit(const int64_t* mask, int nframes, const int32_t* frame_of, const int32_t* cls, int n, int H, int W,
                                const float* xs, const float* ys, const double* init, double* out, int32_t* evals,
                                void* stream) {
  EGNE_REQUIRE(mask && frame_of && cls && xs && ys && init && out, "ellipse_fit: null pointer");
  EGNE_REQUIRE(n > 0 && nframes > 0 && H > 1 && W > 1, "ellipse_fit: bad shape");
  const size_t lds = ((size_t)H * ((W + 31) / 32) + W + H) * 4;
  EGNE_REQUIRE(lds <= 120 * 1024, "ellipse_fit: %dx%d mask does not fit LDS", H, W);
  hipLaunchKernelGGL(ellipse_fit_k, dim3(n), dim3(NT), lds, (hipStream_t)stream, (const long long*)mask, nframes, frame_of, cls, H,
                     W, xs, ys, init, out, evals);
  return egne::check_launch("egne_ellipse_fit");
}
