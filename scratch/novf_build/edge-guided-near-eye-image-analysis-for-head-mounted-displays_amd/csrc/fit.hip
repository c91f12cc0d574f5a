// Ellipse fit of evaluate.py on the device: utils.py:450-486 (coordinate hill-climb over a, b, angle)
// with utils.py:176-204 (IoU of the class mask with a rasterised ellipse on the [-1,1] mesh) and the
// float64 conic algebra of helperfunctions.py:13-63,102-129.
//
// One wave per (frame, class).  The class mask is bit-packed into LDS once; every IoU evaluation is then a pass
// over the rows of the ellipse's bounding box (exact end points of the inside interval + a popcount), so the
// <=281 sequential evaluations of the search never leave the CU (the reference builds each map on the host,
// copies it and calls .item() three times).
//
// Numerics follow the reference bit for bit: float32 mesh supplied by the host (torch.linspace, the
// same call create_meshgrid makes), float32 map arithmetic with one rounding per operation (built with
// -ffp-contract=off and written with the _rn intrinsics), float64 conic normalisation, 3.14159.
#include "common.h"

namespace {

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

// ------------------------------------------------------------------------------------------------------------------------
// A PAIR of waves per (frame, class), two searches per workgroup.
//
// Why not a workgroup per search (rounds 1-2: 512 threads, the mask scanned word by word): the 128 searches of a 64-frame batch
// then sit on 128 CUs for 3.2 ms with ~12 KB of LDS each, and while they do, the network's persistent kernels on the other
// stream (one 100-160 KB workgroup per CU) can only be placed on the remaining CUs -- the fit stage cost the step 3.4 ms although it
// is 1.4 % of its work.  Here a batch's searches occupy 32 CUs for about a millisecond.
//
// What makes one wave enough: an evaluation no longer tests every pixel of the ellipse's bounding box.  On a row the inside set
// of the reference's float32 predicate  ((dx ct + dy st)/a)^2 + ((-dx st + dy ct)/b)^2 - 1 <= 0  is an interval; lane = row solves
// the row's quadratic for approximate end points and then walks each end with the EXACT predicate (same operations, one rounding
// each) until pixel il is inside and il - 1 is not (likewise ir): the count is ir - il + 1 and the overlap a popcount of the
// row's mask bits under the interval.  Rows whose interval is short (tangent rows, where round-off could matter over more than a
// pixel) or empty are tested pixel by pixel around it; walks that do not settle within a few steps and degenerate ellipses fall
// back to testing every pixel of the row -- so the result is the reference's bit for bit (tests/golden/fit_cases.npz,
// evaluate_real_frames.npz).
//
// The search itself is a chain of up to 281 dependent evaluations (utils.py:450-486), ~8 us each: its length is the latency of a
// one- or two-frame call.  Two things shorten the chain without changing a single result.  (1) The two candidates of a coordinate
// step, now[j] - d and (now[j] - d) + 2d, are known before either is scored: the waves of a pair score one each at the same time
// and swap the scores through LDS; the step then takes the reference's decisions in the reference's order (the evaluation counter
// counts what the reference would have evaluated).  (2) The score a sweep ends with is the score of a parameter vector that has
// usually been scored already -- nothing accepted: the vector the sweep started from; one coordinate accepted: that candidate --
// so it is looked up by exact (bitwise) comparison of the three parameters and only evaluated on a miss (the reference's
// subtract / add-twice / subtract sequence can move a rejected coordinate by a rounding, which is then a miss).
// s_barrier waits on the surviving waves only, so a pair that finishes leaves the other pair of its workgroup running.
// ------------------------------------------------------------------------------------------------------------------------
// FIT_PAIRS searches per workgroup, ROWW waves per candidate.  A whole batch: two searches of two waves (64 compute units for its 128
// searches; eight per workgroup were measured too: the lock step of 16 waves at every barrier stretches the launch from 1.85 to
// 3.0 ms and the step gains nothing).  Few searches (a one- or two-frame call, where the chain IS the latency): a workgroup of eight
// waves per search, four per candidate, so that the ~140 rows of an iris ellipse are one pass of 256 lanes instead of three of 64;
// the partial counts of the waves are integers, so their sum does not depend on the split.

struct Ell { float cx, cy, a, b, ct, st; };

__device__ __forceinline__ bool inside_px(const Ell& e, float xv, float dyst, float dyct) {
  const float dx = __fsub_rn(xv, e.cx);
  const float X = __fadd_rn(__fmul_rn(dx, e.ct), dyst);
  const float Y = __fadd_rn(__fmul_rn(-dx, e.st), dyct);
  const float u = __fdiv_rn(X, e.a), v = __fdiv_rn(Y, e.b);
  const float wt = __fsub_rn(__fadd_rn(__fmul_rn(u, u), __fmul_rn(v, v)), 1.0f);
  return wt <= 0.f;
}

// bits of row `rowbits` (wpr words) in pixel range [x0, x1] (inclusive, 0 <= x0 <= x1 < W): popcount
__device__ __forceinline__ unsigned row_pop(const unsigned* rowbits, int x0, int x1) {
  unsigned n = 0;
  for (int w = x0 >> 5; w <= (x1 >> 5); ++w) {
    unsigned m = 0xffffffffu;
    if (w == (x0 >> 5)) m &= 0xffffffffu << (x0 & 31);
    if (w == (x1 >> 5)) m &= 0xffffffffu >> (31 - (x1 & 31));
    n += __popc(rowbits[w] & m);
  }
  return n;
}

// LOCAL: the waves of a search meet through LDS flags of their own instead of the workgroup barrier, so that MANY searches can share a
// workgroup without running in lock step (eight searches per workgroup: 16 of the chip's 256 compute units host the batch's 128
// searches instead of 64 -- a compute unit that hosts a search wave cannot take a workgroup of the network's persistent kernels,
// whose two 256-register waves per SIMD need the whole register file, and that workgroup's share of the tiles then waits).
template <int FIT_PAIRS, int ROWW, bool LOCAL = false>
__global__ __launch_bounds__(128 * FIT_PAIRS * ROWW) void ellipse_fit_k(const long long* __restrict__ mask, int nframes, const int* __restrict__ frame_of,
                                                                       const int* __restrict__ cls, int n, int H, int W,
                                                                       const float* __restrict__ xs, const float* __restrict__ ys,
                                                                       const double* __restrict__ init, double* __restrict__ out,
                                                                       int* __restrict__ evals) {
  // NW = 2 * ROWW waves per search: wave w scores candidate sub = w / ROWW over the rows y = y_lo + part * 64 + lane (+ 64 * ROWW ...)
  constexpr int NW = 2 * ROWW, SWAP = 2 * NW * 2;     // swap area per search: [2 parities][NW waves][ne, ni]
  extern __shared__ unsigned fit_lds[];  // xs[W], ys[H], swap areas, then per search [H][wpr] packed mask
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pair = wave / NW, wv = wave % NW, sub = wv / ROWW, part = wv % ROWW;
  const int wpr = (W + 31) >> 5, nwords = H * wpr;
  float* lxs = (float*)fit_lds;
  float* lys = lxs + W;
  volatile unsigned* swap = fit_lds + W + H + pair * SWAP;
  volatile unsigned* flags = fit_lds + W + H + FIT_PAIRS * SWAP + pair * NW;       // LOCAL: exchange count of every wave of the search
  unsigned* bits = fit_lds + W + H + FIT_PAIRS * (SWAP + NW) + pair * nwords;
  for (int i = threadIdx.x; i < W; i += blockDim.x) lxs[i] = xs[i];
  for (int i = threadIdx.x; i < H; i += blockDim.x) lys[i] = ys[i];
  if (threadIdx.x < FIT_PAIRS * NW) fit_lds[W + H + FIT_PAIRS * SWAP + threadIdx.x] = 0u;
  __syncthreads();
  unsigned seq = 0;
  const int e = blockIdx.x * FIT_PAIRS + pair;
  if (e >= n) return;                     // (whole searches: all waves of a search take every barrier below together)
  const int fr = frame_of[e];
  if (fr < 0 || fr >= nframes) {   // a fit that names a frame the mask tensor does not hold: report NaN, read nothing
    if (wv == 0 && lane < 5) out[e * 5 + lane] = __longlong_as_double(0x7ff8000000000000ll);
    if (wv == 0 && lane == 0 && evals) evals[e] = 0;
    return;
  }
  const long long* m = mask + (long long)fr * H * W;
  const int k = cls[e];
  unsigned cnt = 0;            // (wave-uniform: ballots); the waves of the search pack the rows in turn
  for (int y = wv; y < H; y += NW)
    for (int x0 = 0; x0 < W; x0 += 64) {          // one coalesced 512-byte load per step, the class test of 64 pixels as one ballot
      const int x = x0 + lane;
      const unsigned long long bal = __ballot(x < W && m[(long long)y * W + x] == k);
      cnt += (unsigned)__popcll(bal);
      if (lane == 0) bits[y * wpr + (x0 >> 5)] = (unsigned)bal;
      if (lane == 1 && (x0 >> 5) + 1 < wpr) bits[y * wpr + (x0 >> 5) + 1] = (unsigned)(bal >> 32);
    }
  int parity = 0;
  // every wave leaves its two words, all meet at the barrier and read the sums per candidate (double buffered: one barrier per exchange)
  auto exchange = [&](unsigned m0, unsigned m1, unsigned* s0, unsigned* s1) {
    volatile unsigned* sw = swap + parity * NW * 2;
    if (lane == 0) { sw[wv * 2] = m0; sw[wv * 2 + 1] = m1; }
    if constexpr (LOCAL) {
      // release: this wave's words (and, the first time, its rows of the packed mask) before its count; then wait for the counts of
      // all waves of THIS search.  The areas are double buffered and a wave can run at most one exchange ahead of its partners.
      ++seq;
      __threadfence_block();
      if (lane == 0) flags[wv] = seq;
      if (lane < NW) { while (flags[lane] < seq) __builtin_amdgcn_s_sleep(2); }
      __threadfence_block();
    } else {
      __syncthreads();
    }
    s0[0] = s0[1] = s1[0] = s1[1] = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) { s0[i / ROWW] += sw[i * 2]; s1[i / ROWW] += sw[i * 2 + 1]; }
    parity ^= 1;
  };
  unsigned ca[2], cb[2];
  exchange(cnt, 0u, ca, cb);              // (the barrier also orders the mask words of all waves before the first evaluation)
  const int nseg = (int)(ca[0] + ca[1]);

  const double cx = init[e * 5 + 0], cy = init[e * 5 + 1];
  double now[3] = {init[e * 5 + 2], init[e * 5 + 3], init[e * 5 + 4] * 180. / PI_REF}, d[3] = {1.0, 1.0, 1.0};

  // IoU of the packed mask with the ellipse (cx, cy, q[0], q[1], q[2] degrees); every lane computes the same parameters
  auto evaluate = [&](const double* q, unsigned& ne_out, unsigned& ni_out) {
    double el[5] = {cx, cy, q[0], q[1], q[2] / 180. * PI_REF};
    double nm[5];
    normalise(el, H, W, nm);
    const Ell E = {(float)nm[0], (float)nm[1], (float)nm[2], (float)nm[3], (float)cos(nm[4]), (float)sin(nm[4])};
    // Only pixels inside the ellipse count (ne, ni), and the ellipse lies within max(a, b) of its centre: rows of the
    // bounding box (0.1 % + 2 pixels of slack, orders of magnitude above float32 round-off) instead of the frame.
    // Degenerate parameters (NaN / huge axes) fall back to the full frame, every pixel tested.
    int y_lo = 0, y_hi = H - 1, x_lo = 0, x_hi = W - 1;
    const float rr = fmaxf(E.a, E.b) * 1.001f;
    const bool tame = rr < 4.f && fabsf(E.cx) < 4.f && fabsf(E.cy) < 4.f && fminf(E.a, E.b) > 1e-3f;
    const float sx = 0.5f * (float)(W - 1), sy = 0.5f * (float)(H - 1);
    if (tame) {
      const int xl = (int)floorf((E.cx - rr + 1.f) * sx) - 2, xh = (int)ceilf((E.cx + rr + 1.f) * sx) + 2;
      const int yl = (int)floorf((E.cy - rr + 1.f) * sy) - 2, yh = (int)ceilf((E.cy + rr + 1.f) * sy) + 2;
      y_lo = max(yl, 0); y_hi = min(yh, H - 1);
      x_lo = max(xl, 0); x_hi = min(xh, W - 1);
    }
    unsigned ne = 0, ni = 0;
    if (y_hi >= y_lo && x_hi >= x_lo) {
      // the row quadratic A dx^2 + Bq dx + C <= 0 (approximate arithmetic: it only seeds the exact walk)
      const float ia = 1.f / (E.a * E.a), ib = 1.f / (E.b * E.b);
      const float A = E.ct * E.ct * ia + E.st * E.st * ib, Bc = 2.f * E.st * E.ct * (ia - ib), Cc = E.st * E.st * ia + E.ct * E.ct * ib;
      for (int y = y_lo + part * 64 + lane; y <= y_hi; y += 64 * ROWW) {
        const float dy = __fsub_rn(lys[y], E.cy);
        const float dyst = __fmul_rn(dy, E.st), dyct = __fmul_rn(dy, E.ct);
        const unsigned* rowbits = bits + y * wpr;
        // pixels [t0, t1] of the row tested one by one; an interval [il, ir] counted as a whole
        int t0 = 0, t1 = -1, il = 0, ir = -1;
        if (!tame) {
          t0 = x_lo; t1 = x_hi;
        } else {
          const float Bq = Bc * dy, C = Cc * dy * dy - 1.f, disc = Bq * Bq - 4.f * A * C;
          const int xv = (int)floorf((-Bq / (2.f * A) + E.cx + 1.f) * sx);        // pixel next to the row's closest approach
          bool whole = false;
          if (!(disc > 0.f)) {
            // the row misses the ellipse (or grazes it): round-off can only matter next to the closest approach
            t0 = max(xv - 4, 0); t1 = min(xv + 5, W - 1);
            whole = t1 >= t0 && (inside_px(E, lxs[t0], dyst, dyct) || inside_px(E, lxs[t1], dyst, dyct));
          } else {
            const float sq = sqrtf(disc), dl = (-Bq - sq) / (2.f * A), dr = (-Bq + sq) / (2.f * A);
            il = (int)ceilf((dl + E.cx + 1.f) * sx);
            ir = (int)floorf((dr + E.cx + 1.f) * sx);
            if (ir - il < 12) {
              // short interval (tangent rows): its pixels and four more on either side, one by one; the outermost must be outside
              t0 = max(il - 4, 0); t1 = min(ir + 4, W - 1);
              whole = t1 >= t0 && ((t0 > 0 && inside_px(E, lxs[t0], dyst, dyct)) || (t1 < W - 1 && inside_px(E, lxs[t1], dyst, dyct)));
              il = 0; ir = -1;
            } else {
              // walk each end with the exact predicate: il inside and il - 1 outside (or il = 0), ir inside and ir + 1 outside (or ir = W - 1)
              il = min(max(il, 0), W - 1); ir = min(max(ir, 0), W - 1);
              int steps = 0;
              while (steps < 8 && il > 0 && inside_px(E, lxs[il - 1], dyst, dyct)) { --il; ++steps; }
              while (steps < 8 && il < W - 1 && !inside_px(E, lxs[il], dyst, dyct)) { ++il; ++steps; }
              steps = 0;
              while (steps < 8 && ir < W - 1 && inside_px(E, lxs[ir + 1], dyst, dyct)) { ++ir; ++steps; }
              while (steps < 8 && ir > 0 && !inside_px(E, lxs[ir], dyst, dyct)) { --ir; ++steps; }
              const bool settled = il <= ir && inside_px(E, lxs[il], dyst, dyct) && (il == 0 || !inside_px(E, lxs[il - 1], dyst, dyct)) &&
                                   inside_px(E, lxs[ir], dyst, dyct) && (ir == W - 1 || !inside_px(E, lxs[ir + 1], dyst, dyct));
              if (!settled) { whole = true; il = 0; ir = -1; }
            }
          }
          if (whole) { t0 = 0; t1 = W - 1; }          // something unexpected: every pixel of the row, as the reference does
        }
        for (int x = t0; x <= t1; ++x)
          if (inside_px(E, lxs[x], dyst, dyct)) { ++ne; ni += (rowbits[x >> 5] >> (x & 31)) & 1u; }
        if (ir >= il) {
          ne += (unsigned)(ir - il + 1);
          ni += row_pop(rowbits, il, ir);
        }
      }
    }
    for (int o = 32; o >= 1; o >>= 1) { ne += __shfl_xor(ne, o); ni += __shfl_xor(ni, o); }
    ne_out = ne; ni_out = ni;
  };
  auto iou = [&](unsigned ne, unsigned ni) -> float {
    const float fi = (float)ni;
    return __fdiv_rn(fi, __fsub_rn(__fadd_rn((float)nseg, (float)ne), fi));
  };
  // one round: candidate qa on the waves with sub = 0, qb on the others (each over its share of the rows), both scores to every wave
  auto score2 = [&](const double* qa, const double* qb, float& sa, float& sb) {
    unsigned ne, ni, nes[2], nis[2];
    evaluate(sub ? qb : qa, ne, ni);
    exchange(ne, ni, nes, nis);
    sa = iou(nes[0], nis[0]); sb = iou(nes[1], nis[1]);
  };

  auto same = [](const double* a, const double* b) {
    return __double_as_longlong(a[0]) == __double_as_longlong(b[0]) && __double_as_longlong(a[1]) == __double_as_longlong(b[1]) &&
           __double_as_longlong(a[2]) == __double_as_longlong(b[2]);
  };
  int nev = 1;
  float base_sc, unused;
  score2(now, now, base_sc, unused);           // score of `now` at the start of the sweep
  double base_q[3] = {now[0], now[1], now[2]};
  double rt = (double)base_sc;
  for (int sweep = 0; sweep < 40; ++sweep) {
    int flag = 0;
    float acc_sc = 0.f;                          // the last accepted candidate and its score
    double acc_q[3] = {0., 0., 0.};
    bool have_acc = false;
    for (int j = 0; j < 3; ++j) {
      const double lo = now[j] - d[j], hi = lo + 2. * d[j];          // the reference's two candidates, in its arithmetic
      double qa[3] = {now[0], now[1], now[2]}, qb[3] = {now[0], now[1], now[2]};
      qa[j] = lo; qb[j] = hi;
      float sc_lo, sc_hi;
      score2(qa, qb, sc_lo, sc_hi);
      ++nev;
      if ((double)sc_lo > rt) {                  // (rt only changes between sweeps)
        now[j] = lo; flag = 1; have_acc = true; acc_sc = sc_lo;
        acc_q[0] = now[0]; acc_q[1] = now[1]; acc_q[2] = now[2];
        continue;
      }
      ++nev;
      if ((double)sc_hi > rt) {
        now[j] = hi; flag = 1; have_acc = true; acc_sc = sc_hi;
        acc_q[0] = now[0]; acc_q[1] = now[1]; acc_q[2] = now[2];
        continue;
      }
      now[j] = hi - d[j]; d[j] *= 0.8;
    }
    ++nev;
    float sc;
    if (same(now, base_q)) sc = base_sc;
    else if (have_acc && same(now, acc_q)) sc = acc_sc;
    else score2(now, now, sc, unused);
    if ((double)sc > rt) rt = (double)sc;
    base_sc = sc; base_q[0] = now[0]; base_q[1] = now[1]; base_q[2] = now[2];
    if (!flag) break;
  }
  if (wv == 0 && lane == 0) {
    out[e * 5 + 0] = cx; out[e * 5 + 1] = cy; out[e * 5 + 2] = now[0]; out[e * 5 + 3] = now[1];
    out[e * 5 + 4] = now[2] / 180.0 * PI_REF;
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
  const size_t per_search = ((size_t)H * ((W + 31) / 32) + 32) * 4, fixed = (size_t)(W + H) * 4;
  EGNE_REQUIRE(2 * per_search + fixed <= 64 * 1024, "ellipse_fit: %dx%d masks do not fit LDS", H, W);
  hipStream_t st = (hipStream_t)stream;
  const long long* mk = (const long long*)mask;
  static const int dense = [] { const char* e = getenv("EGNE_FIT_DENSE"); return e ? atoi(e) : 4; }();      // searches per workgroup of a large batch (2: round 3's form; 4 measured best: 32.3 vs 32.5 ms per step with the fit stage, 8 spills at 128 registers)
  if (n >= 64 && dense == 8 && 8 * per_search + fixed <= 150 * 1024) {
    // a large batch: EIGHT searches of two waves per workgroup, pair-local synchronisation (see LOCAL above)
    const size_t lds = fixed + 8 * ((size_t)H * ((W + 31) / 32) + 8 + 2) * 4;
    static bool once = hipFuncSetAttribute((const void*)ellipse_fit_k<8, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
    if (!once) return egne::fail(EGNE_ERR_LAUNCH, "ellipse_fit: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL((ellipse_fit_k<8, 1, true>), dim3((unsigned)((n + 7) / 8)), dim3(1024), lds, st, mk, nframes, frame_of, cls, n, H, W, xs, ys, init, out, evals);
  } else if (n >= 64 && dense == 4 && fixed + 4 * ((size_t)H * ((W + 31) / 32) + 8 + 2) * 4 <= 64 * 1024) {
    // (masks where two searches fit the default 64 KB of dynamic LDS but four do not -- 320x480, 384x512 -- take the two-search form below)
    const size_t lds = fixed + 4 * ((size_t)H * ((W + 31) / 32) + 8 + 2) * 4;
    hipLaunchKernelGGL((ellipse_fit_k<4, 1, true>), dim3((unsigned)((n + 3) / 4)), dim3(512), lds, st, mk, nframes, frame_of, cls, n, H, W, xs, ys, init, out, evals);
  } else if (n >= 16) {           // a batch: two searches of two waves per workgroup
    const size_t lds = fixed + 2 * ((size_t)H * ((W + 31) / 32) + 8 + 2) * 4;
    hipLaunchKernelGGL((ellipse_fit_k<2, 1>), dim3((unsigned)((n + 1) / 2)), dim3(256), lds, st, mk, nframes, frame_of, cls, n, H, W, xs, ys, init, out, evals);
  } else {                 // one or two frames: a workgroup of eight waves per search (the rows of an evaluation in one pass)
    const size_t lds = fixed + ((size_t)H * ((W + 31) / 32) + 32 + 8) * 4;
    hipLaunchKernelGGL((ellipse_fit_k<1, 4>), dim3((unsigned)n), dim3(512), lds, st, mk, nframes, frame_of, cls, n, H, W, xs, ys, init, out, evals);
  }
  return egne::check_launch("egne_ellipse_fit");
}
