// Which convolution entry point serves a layer: the C side of engine.Plan._conv_impl / _conv_bf16 (round 6).
//
// The reference has no dispatch of its own -- F.conv2d picks a backend kernel inside ATen (models/RITnet_v2.py:57-62,85-87,
// bdcn_new.py:49-55, vgg16_c.py:65-88 all end there).  A maintainer who binds include/egne_hip.h directly
// has 15 forward entry points to choose from; until this round the choice lived in Python only (engine.py).  egne_conv2d_auto_kind
// answers it from a layer description: the same predicates, in the same order, with the DEFAULT thresholds of engine.py (its
// environment switches are experiment knobs of the Python planner, not part of this contract).  The planner checks itself against this
// function for every convolution it plans when EGNE_CHECK_DISPATCH=1 (tests/test_gpu_nets.py: the B = 64 and B = 2 plans of both
// networks, the training plans in both storages).
//
// What it does NOT decide: fusing a 1x1 into the 3x3 that consumes it (egne_conv1x1_3x3_fused_f16_fwd, egne_conv3x3c4_3x3_fused_f16_fwd,
// egne_conv1x1_pool_f16x3_fwd) is a decision about PAIRS of layers, taken by the plan builder (esf_engine.py) before either layer reaches
// this function.
#include "common.h"
#include <cstring>

namespace {

inline int pad8i(int c) { return (c + 7) / 8 * 8; }
inline int pad32i(int c) { return (c + 31) / 32 * 32; }
inline int cdivi(int a, int b) { return (a + b - 1) / b; }

// engine.py defaults
constexpr int HALO_MIN_W = 30, HALO_MAX_COUTP = 128, HALO_F16_MIN_W = 30, HALO_F16_MIN_W_NARROW = 30, HALO_F16_MAX_COUTP = 256;
constexpr int LATTICE_MIN_W = 20, RS_MIN_W = 60, RW_MIN_W = 120, RW_MAX_COUTP = 32, RW_MAX_CP = 512;
constexpr int RS_MIN_W_F16 = 30, RW_MIN_W_F16 = 30;      // plain-f16 plans (egne_conv_query.f16_products = 1): engine.RS_MIN_W_F16 / RW_MIN_W_F16
constexpr int BIG_MIN_COUT = 256, BIG_MIN_CIN = 64, BIG_CUS = 256;
constexpr long long MS1X1_MIN_PIX = 30000, S1X1_MIN_PIX = 100000;

int set_kind(egne_conv_choice* c, int kind, const char* name) {
  c->kind = kind;
  std::strncpy(c->name, name, sizeof(c->name) - 1);
  c->name[sizeof(c->name) - 1] = 0;
  return EGNE_OK;
}

}  // namespace

extern "C" int egne_conv2d_auto_kind(const egne_conv_query* qp, egne_conv_choice* out) {
  EGNE_REQUIRE(qp && out, "conv2d_auto_kind: null pointer");
  const egne_conv_query& q = *qp;
  EGNE_REQUIRE(q.nseg >= 1 && q.nseg <= EGNE_MAXSEG && q.ngroups >= 1 && q.ngroups <= EGNE_MAXGROUP && q.kh >= 1 && q.kw >= 1 && q.stride >= 1,
               "conv2d_auto_kind: bad geometry");
  std::memset(out, 0, sizeof(*out));
  const int B = q.B, H = q.H, W = q.W;
  // (padding counts TAPS, as in egne_conv_desc: a dilated "same" 3x3 has pad 1 and reaches dil pixels)
  const int Ho = (H + 2 * q.pad_h * q.dil[0] - q.dil[0] * (q.kh - 1) - 1) / q.stride + 1, Wo = (W + 2 * q.pad_w * q.dil[0] - q.dil[0] * (q.kw - 1) - 1) / q.stride + 1;
  int Cin = 0, Ktot = 0;
  bool raw = true;
  for (int i = 0; i < q.nseg; ++i) { Cin += q.seg_C[i]; Ktot += q.seg_Cp[i]; raw = raw && !q.seg_affine[i]; }
  const int Cout = q.Cout, CoutP = pad32i(Cout), Cout_store = q.Cout_store > 0 ? q.Cout_store : pad8i(Cout);
  const int G = q.ngroups, d0 = q.dil[0];
  const bool k3 = q.kh == 3 && q.kw == 3, k1 = q.kh == 1 && q.kw == 1, same1 = q.pad_h == 1 && q.pad_w == 1;
  const int Cp0 = q.seg_Cp[0];
  const long long st0 = q.seg_pix_stride[0];
  const int cstore = Cout_store < q.dst_Cp ? Cout_store : q.dst_Cp;

  // ---------------------------------------------------------------------------------- plans with bf16 activation storage (_conv_bf16)
  if (q.dtype == 1) {
    const bool one = q.nseg == 1 && q.stride == 1 && q.pad_mode == 0 && d0 == 1 && k3 && same1 && G == 1;
    const bool smallcin = one && Cin <= 4 && pad8i(Cout) <= 64 && raw && !q.has_residual && !q.is_dgrad && !q.has_post;
    const bool fast3 = one && !smallcin && !q.has_post && Cp0 % 8 == 0 && q.seg_ch_off[0] % 8 == 0 && st0 % 8 == 0 && CoutP <= 256 && cstore % 4 == 0 &&
                       cstore >= 8 && (long long)H * W * (st0 > q.dst_pix_stride ? st0 : q.dst_pix_stride) < (1ll << 30) &&
                       (!q.has_residual || (long long)H * W * q.res_pix_stride < (1ll << 30));
    bool fast1 = k1 && q.stride == 1 && q.pad_h == 0 && q.pad_w == 0 && !q.has_post && G == 1 && (long long)B * H * W >= 4096 && cstore % 8 == 0 &&
                 q.dst_ch_off % 8 == 0 && q.dst_pix_stride % 8 == 0 && (!q.has_residual || (q.res_ch_off % 8 == 0 && q.res_pix_stride % 8 == 0));
    for (int i = 0; i < q.nseg && fast1; ++i)
      fast1 = !q.seg_affine[i] && q.seg_Cp[i] % 8 == 0 && q.seg_ch_off[i] % 8 == 0 && q.seg_pix_stride[i] % 8 == 0;
    if (fast1) {
      egne_conv_desc dq{};
      dq.nseg = q.nseg; dq.CoutP = CoutP; dq.Ktot = Ktot;
      for (int i = 0; i < q.nseg; ++i) dq.seg[i].Cp = q.seg_Cp[i];
      fast1 = egne_conv1x1_bf16_pack_elems(&dq) > 0;
    }
    if (smallcin) return set_kind(out, EGNE_KIND_SMALLCIN, "conv3x3_smallcin");
    if (fast3) return set_kind(out, EGNE_KIND_BF16_3X3, "conv_bf16:3x3");
    if (fast1) return set_kind(out, EGNE_KIND_BF16_1X1, "conv_bf16:1x1");
    if (q.narrow_bf16_ok) return set_kind(out, EGNE_KIND_BF16_NARROW, "conv_bf16:narrow");     // (egne_conv_narrow_bf16_supported on the finished descriptor)
    return set_kind(out, EGNE_KIND_IGEMM, "conv_igemm");
  }

  // ---------------------------------------------------------------------------------- fp32 tensors (_conv_impl)
  // <= 4 output channels over a narrow raw slice: exact fp32 on the vector ALU
  if (!q.train && !q.dyn_scales && k3 && q.stride == 1 && G == 1 && same1 && q.pad_mode == 0 && d0 == 1 && Cout <= 4 && q.nseg == 1 && !q.seg_planar &&
      raw && !q.seg_presplit && Cp0 >= 32 && Cp0 <= 64 && !q.has_residual && !q.want_stats && !q.want_scores && !q.want_pool &&
      (q.act == EGNE_ACT_NONE || q.act == EGNE_ACT_RELU || q.act == EGNE_ACT_LEAKY) && (long long)H * W * st0 < (1ll << 29))
    return set_kind(out, EGNE_KIND_NARROW_F32, "conv3x3_narrow");

  bool halo = k3 && q.stride == 1 && G == 1 && same1 && q.pad_mode == 0 && q.nseg == 1 && d0 <= 2 && W >= HALO_MIN_W && CoutP <= HALO_MAX_COUTP &&
              (long long)H * W * st0 < (1ll << 31);
  bool smallcin = k3 && q.stride == 1 && G == 1 && same1 && q.pad_mode == 0 && q.nseg == 1 && d0 == 1 && Cin <= 4 && pad8i(Cout) <= 64 && raw &&
                  !q.has_residual && !q.is_dgrad;
  const bool c4h = smallcin && (q.split || q.split_c4) && pad8i(Cout) > 32;          // C4H_MODE "wide"
  bool split = q.split && q.stride == 1 && q.pad_mode == 0 && q.nseg == 1 && Cp0 >= 32;
  bool shalo = split && k3 && G == 1 && same1 && d0 <= 2 && W >= (CoutP > 64 ? HALO_F16_MIN_W : HALO_F16_MIN_W_NARROW) && CoutP <= HALO_F16_MAX_COUTP &&
               !q.has_residual && (long long)H * W * st0 < (1ll << 31);
  int minlat = 1 << 30;
  for (int g = 0; g < G; ++g) { const int v = W / (q.dil[g] > 0 ? q.dil[g] : 1); if (v < minlat) minlat = v; }
  bool lattice = split && G == 3 && k3 && same1 && (CoutP == 32 || CoutP == 64) && q.has_residual && raw && minlat >= LATTICE_MIN_W &&
                 (long long)H * W * st0 < (1ll << 31);
  // 1x1 over raw slices: streaming split-f16 kernel (its weight image and, with an up-sampled addend, the waves' patches must fit 80 KB of LDS)
  long long k16 = 0;
  for (int i = 0; i < q.nseg; ++i) k16 += (q.seg_Cp[i] + 15) / 16;
  const int nb = CoutP == 32 ? 1 : 2;
  bool s1x1 = !q.has_residual && !q.dyn_scales && q.split1 && k1 && q.stride == 1 && G == 1 && q.pad_h == 0 && q.pad_w == 0 && !q.has_post && raw &&
              (CoutP == 32 || CoutP % 64 == 0) && k16 * nb * 2048 + (q.up_add ? 4ll * 18 * (32 * nb + 8) * 4 : 0) <= 80 * 1024 &&
              (long long)B * H * W >= S1X1_MIN_PIX;
  bool ms1x1 = !s1x1 && q.split1 && k1 && q.stride == 1 && G == 1 && q.pad_h == 0 && q.pad_w == 0 && !q.has_residual && !q.has_post && raw && Cout > 32 &&
               (long long)B * H * W >= MS1X1_MIN_PIX;
  bool big = split && G == 1 && raw && Cp0 % 32 == 0 && !q.has_residual && !q.has_post && Cout % 128 == 0 && Cout >= BIG_MIN_COUT && Cin >= BIG_MIN_CIN &&
             (long long)B * Ho * Wo >= 256 * 128;
  bool msdil = split && G == 3 && k3 && same1 && q.dil[0] == 4 && q.dil[1] == 8 && q.dil[2] == 12 && CoutP == 32 && Cp0 == 32 && q.has_residual && raw &&
               q.act == EGNE_ACT_RELU && !q.has_post && (long long)H * W * st0 < (1ll << 29) && (long long)H * W * q.dst_pix_stride < (1ll << 29);
  if (msdil) lattice = false;
  if (q.dyn_scales) s1x1 = ms1x1 = big = lattice = msdil = false;
  const int sfrag_coutp = CoutP <= 64 ? CoutP : (CoutP + 63) / 64 * 64;
  const int split_coutp = (Cout > 64 && G == 1) ? (Cout + 127) / 128 * 128 : CoutP;
  const long long mx_stride = st0 > q.dst_pix_stride ? st0 : q.dst_pix_stride;
  bool rs = split && !lattice && !msdil && k3 && G == 1 && same1 && q.stride == 1 && q.pad_mode == 0 && d0 == 1 && q.nseg == 1 && W >= (q.f16_products == 1 ? RS_MIN_W_F16 : RS_MIN_W) &&
            (long long)H * W * mx_stride < (1ll << 29) && (!q.has_residual || (long long)H * W * q.res_pix_stride < (1ll << 29));
  const bool rw_wide = rs && Cp0 > 64 && Cp0 <= RW_MAX_CP && sfrag_coutp <= RW_MAX_COUTP && !q.want_stats && W >= (q.f16_products == 1 ? RW_MIN_W_F16 : RW_MIN_W) && cstore % 8 == 0 &&
                       q.dst_pix_stride % 4 == 0 && q.dst_ch_off % 4 == 0 && (!q.has_residual || (q.res_pix_stride % 4 == 0 && q.res_ch_off % 4 == 0));
  rs = rw_wide || (rs && Cp0 >= 8 && Cp0 <= 64 && (sfrag_coutp == 32 || sfrag_coutp == 64 || sfrag_coutp == 128) && !(Cp0 <= 32 && sfrag_coutp == 128));
  if (rs && !rw_wide && Cp0 > 32 && Cp0 <= 64 && Cp0 % 32 > 0 && Cp0 % 32 <= 16) rs = false;          // a short tail chunk: the halo kernel skips its zero half
  shalo = shalo || rs;
  if (lattice || msdil) shalo = true;
  if (split && !shalo && halo && !raw && CoutP <= 32 && W >= HALO_F16_MIN_W) split = false;          // narrow fused-affine layers: the fp32 halo kernel
  if (smallcin) split = shalo = false;
  // small problems (one or two frames at the deep levels): 64-wide tiles + split-K on the flat kernel
  long long small_ws = -1;
  if (split && !q.train && G == 1 && !(lattice || msdil)) {
    egne_conv_desc dq{};
    dq.B = B; dq.Ho = Ho; dq.Wo = Wo; dq.kh = q.kh; dq.kw = q.kw; dq.ngroups = 1; dq.CoutP = split_coutp;
    dq.seg[0].Cp = Cp0;
    small_ws = egne_conv2d_f16x3_small_workspace_floats(&dq);
  }
  bool small = small_ws >= 0;
  if (small && q.want_stats && shalo && q.dst_Cp == cstore) {
    const bool tall = (long long)cdivi(H, 32) * cdivi(W, 8) < (long long)cdivi(W, 32) * cdivi(H, 8);
    if (rs || (d0 == 1 && !tall)) small = false;          // the halo / role-split kernel writes the consumer's InstanceNorm sums from its epilogue
  }
  if (small) shalo = rs = big = false;
  if (smallcin || split) halo = false;
  if (big) {
    smallcin = shalo = halo = lattice = s1x1 = false;
    const int bn = Cout % 256 == 0 ? 256 : 128, ny = cdivi(Cout, bn);
    auto wgs = [&](long long nbf) { return (nbf * Ho * Wo + 255) / 256 * ny; };
    const long long full = wgs(B) / BIG_CUS;
    const double frac = (double)wgs(B) / BIG_CUS - (double)full;
    if (full >= 1 && frac > 0.04 && frac < 0.65 && !q.f16_storage) {      // (f16 tensors: the flat kernel behind a ragged round reads and writes fp32)
      long long b1 = B;
      while (b1 > 1 && wgs(b1) > full * BIG_CUS) --b1;
      if ((double)wgs(b1) >= 0.9 * (double)(full * BIG_CUS)) out->tail_frames = (int)(B - b1);
    }
    return set_kind(out, EGNE_KIND_F16X3_BIG, "conv_f16x3:big");
  }
  if (s1x1) return set_kind(out, EGNE_KIND_F16X3_STREAM1X1, "conv_f16x3:stream1x1");
  if (ms1x1) return set_kind(out, EGNE_KIND_F16X3_GEMM1X1, "conv_f16x3:gemm1x1");
  if (msdil) return set_kind(out, EGNE_KIND_F16X3_MSDIL, "conv_f16x3:msdil");
  if (lattice) return set_kind(out, EGNE_KIND_F16X3_LATTICE, "conv_f16x3:lattice");
  if (shalo && rs) {
    const bool pooled = q.want_pool && Cp0 > 32 && sfrag_coutp >= 64 && !q.has_post &&
                        (q.act == EGNE_ACT_NONE || q.act == EGNE_ACT_RELU || q.act == EGNE_ACT_LEAKY) && q.pool_Cp >= cstore;
    const bool fuse_stats = q.want_stats && q.dst_Cp == cstore;
    const bool rw = rw_wide || (!fuse_stats && cstore % 8 == 0 && q.dst_pix_stride % 4 == 0 && q.dst_ch_off % 4 == 0 &&
                                (!q.has_residual || (q.res_pix_stride % 4 == 0 && q.res_ch_off % 4 == 0)));
    out->fused_pool = pooled;
    out->fused_stats = fuse_stats;
    return rw ? set_kind(out, EGNE_KIND_F16X3_RW, "conv_f16x3:rw") : set_kind(out, EGNE_KIND_F16X3_RS, "conv_f16x3:rs");
  }
  if (shalo) {
    out->fused_stats = q.want_stats && d0 == 1 && q.dst_Cp == cstore;
    out->fused_pool = q.want_pool && !q.want_stats && d0 == 1 && !q.has_post && !q.has_residual &&
                      (q.act == EGNE_ACT_NONE || q.act == EGNE_ACT_RELU || q.act == EGNE_ACT_LEAKY) && q.pool_Cp >= cstore && sfrag_coutp % 64 == 0 &&
                      (long long)((H + 1) / 2) * ((W + 1) / 2) * q.pool_pix_stride < (1ll << 29);
    return set_kind(out, EGNE_KIND_F16X3_HALO, "conv_f16x3:halo");
  }
  if (split && small) { out->small_ws_floats = small_ws; return set_kind(out, EGNE_KIND_F16X3_SMALL, "conv_f16x3:small"); }
  if (split) return set_kind(out, EGNE_KIND_F16X3_FLAT, "conv_f16x3:flat");
  if (smallcin && c4h) return set_kind(out, EGNE_KIND_F16X3_FIRST, "conv_f16x3:first");
  if (smallcin) return set_kind(out, EGNE_KIND_SMALLCIN, "conv3x3_smallcin");
  if (halo) return set_kind(out, EGNE_KIND_HALO_F32, "conv3x3_halo");
  return set_kind(out, EGNE_KIND_IGEMM, "conv_igemm");
}
