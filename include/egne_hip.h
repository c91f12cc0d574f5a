/*
 * egne_hip.h -- C-ABI of the MI355X (gfx950) hot path of edge-guided near-eye segmentation.
 *
 * The reference (zhaoyuhsin/Edge-Guided-Near-Eye-Image-Analysis-for-Head-Mounted-Displays) has no
 * FFI layer of its own: its hot path is a chain of PyTorch ATen calls made from Python
 * (SURVEY.md section 2.3).  Each entry point below therefore names the reference call site(s)
 * whose ATen op(s) it replaces (file:line under /root/reference).  The Python host in
 * edge-guided-near-eye-image-analysis-for-head-mounted-displays_amd/ binds these with ctypes
 * (_lib.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative egne_status otherwise; egne_last_error()
 *     returns a message for the calling thread;
 *   - all pointers are DEVICE pointers owned by the caller, nothing is allocated or freed here;
 *   - activations are fp32 NHWC; a tensor is addressed as (base pointer, floats per pixel,
 *     first channel, channel count) so that torch.cat never has to materialise: producers write
 *     into channel slices of a shared buffer and consumers read up to EGNE_MAXSEG slices;
 *   - channel counts of slices are padded to a multiple of 8 with zero weights / zero data;
 *   - `stream` is a hipStream_t passed as void*; kernels are asynchronous on it.
 */
#ifndef EGNE_HIP_H
#define EGNE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGNE_MAXSEG 8
#define EGNE_MAXGROUP 3

typedef enum {
  EGNE_OK = 0,
  EGNE_ERR_ARG = -1,     /* shape / alignment / range check failed on the host side */
  EGNE_ERR_LAUNCH = -2,  /* hipLaunchKernel or a runtime call failed */
  EGNE_ERR_NODEVICE = -3
} egne_status;

typedef enum { EGNE_ACT_NONE = 0, EGNE_ACT_RELU = 1, EGNE_ACT_LEAKY = 2 } egne_act;

/* One input slice of a convolution (a piece of a would-be torch.cat). */
typedef struct {
  const float* ptr;     /* NHWC base of the buffer that holds the slice */
  int64_t pix_stride;   /* floats per pixel of that buffer */
  int32_t ch_off;       /* first channel of the slice (multiple of 4) */
  int32_t Cp;           /* padded channel count of the slice (multiple of 8) */
  const float* scale;   /* optional per-(n,c) affine applied while loading: x*scale+shift  */
  const float* shift;   /*   ([B][Cp], used to fuse InstanceNorm / BatchNorm into the consumer) */
  int32_t act_in;       /* egne_act applied after the affine (LeakyReLU of Transition_down) */
  int32_t presplit;     /* 1: the slice is held in SPLIT-PAIR storage (see egne_conv_desc.out_split), written with the scale that is
                         * passed as this launch's a_scale; honoured by egne_msblock_dil(_scores)_f16_fwd only, 0 everywhere else.
                         * 2: the slice is held as F16 (out_split = 2 of its producer; ptr at halfs, pix_stride / ch_off in halfs, multiples
                         * of 8), written with this launch's a_scale: egne_conv3x3_rw_f16_fwd with f16_products = 1 only */
} egne_seg;

/*
 * Implicit-GEMM convolution, fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32.
 * Replaces F.conv2d at: vgg16_c.py:66-78 (3x3, dilation 1/2), bdcn_new.py:50-53 (3x3 + three
 * dilated 3x3 of MSBlock, fused as `ngroups`=3 with the 4-way sum of :54 as `residual`),
 * utils.py:1047-1048 (convBlock), models/RITnet_v2.py:57-62,41 (down block 3x3 / concat-free
 * 1x1 / Transition_down), :85-87 (up block), utils.py:1016-1019 (regressionModule convs),
 * utils.py:1020-1021 (its Linear layers, as 1x1 convs), models/RITnet_v2.py:91-121 (StyleEncoder
 * with reflect padding, MLP).
 */
typedef struct {
  int32_t B, H, W;            /* input batch / height / width */
  int32_t Ho, Wo;             /* output height / width */
  int32_t kh, kw, stride;
  int32_t pad_h, pad_w;       /* padding for dilation 1; group g pads by pad*dil[g] */
  int32_t pad_mode;           /* 0 zero, 1 reflect */
  int32_t ngroups;            /* 1, or 3 for the fused MSBlock dilated branch */
  int32_t dil[EGNE_MAXGROUP];
  int32_t nseg;
  egne_seg seg[EGNE_MAXSEG];
  int32_t Ktot;               /* sum of seg[i].Cp */
  int32_t CoutP;              /* rows of the packed weight, multiple of 32 */
  const float* w;             /* packed [group][tap][CoutP][Ktot], zero padded */
  const float* bias;          /* [group][CoutP], zero padded (may be NULL) */
  int32_t act;                /* egne_act applied to (acc + bias) of every group */
  const float* post_scale;    /* optional per-channel affine after act (eval BatchNorm) [CoutP] */
  const float* post_shift;
  const float* residual;      /* optional tensor added last (same pixels as the output) */
  int64_t res_pix_stride;
  int32_t res_ch_off;
  float* out;
  int64_t out_pix_stride;
  int32_t out_ch_off;
  int32_t Cout_store;         /* channels written (logical Cout rounded up to the slice's padding) */
  /* Optional (egne_conv3x3_halo_f16_fwd, egne_conv1x1_3x3_fused_f16_fwd; round 5: egne_conv3x3_bf16_fwd): per-(frame, chunk, channel) partial sums of the
   * STORED output values, [B][stats_nchunk][Cout_store][2] doubles (sum, sum of squares), for a following
   * egne_norm_stats_finish -- the InstanceNorm statistics of the consumer (models/RITnet_v2.py:40,57) without another
   * pass over the tensor.  A chunk is one wave's rows of one tile: stats_nchunk = ceil(W/32) * ceil(H/8) * 4. */
  double* stats_ws;
  int32_t stats_nchunk;
  /* Optional (egne_conv3x3_rs_f16_fwd, shapes whose consumer waves own two tile rows): second output = the 2x2 / stride 2 /
   * ceil-mode max pooling of the stored result (vgg16_c.py:70 pool1 behind conv1_2), [B][ceil(H/2)][ceil(W/2)] pixels of
   * pool_pix_stride floats, Cout_store channels from pool_ch_off; needs a monotonic activation and no post affine. */
  float* pool_out;
  int64_t pool_pix_stride;
  int32_t pool_ch_off;
  /* Optional (egne_conv3x3_halo_f16_fwd, egne_conv3x3_rs_f16_fwd, egne_conv3x3_rw_f16_fwd, egne_conv2d_f16x3_fwd): the
   * activation pre-scale of the split taken ON THE DEVICE from this word = bit pattern of max |x| of the input (written by
   * egne_absmax on the same stream): a_scale = the power of two that puts it in [1024, 2048); the a_scale argument is then
   * ignored (pass 1).  Training plans use it: their weights move every step, so a host-side calibration would cost a
   * synchronisation per launch and step.  Zero / denormal maxima leave a_scale = 1. */
  const uint32_t* dyn_scale;
  /* Optional (egne_conv2d_fwd, the exact-fp32 implicit GEMM): atomic max of the bit patterns of |stored value| into this word
   * (cleared by the caller) -- the dyn_scale word of the split-f16 launch that reads this output next, without another pass
   * over the tensor. */
  uint32_t* absmax_out;
  /* Storage type of the ACTIVATION tensors of this call: 0 = fp32 (everything above as declared), 1 = bf16 -- seg[i].ptr, out,
   * residual and pool_out (and the gz argument of the weight-gradient entry points) then point at bf16 elements; strides and
   * offsets stay in ELEMENTS, slices must start on multiples of 4 elements.  Weights, bias, affine tables and statistics stay
   * fp32, accumulation is fp32 (training plans with bf16 activation storage: BASELINE.json configs[2..4]).  Honoured by
   * egne_conv2d_fwd, egne_conv3x3_smallcin_fwd, egne_conv2d_wgrad, egne_conv3x3_bf16_fwd; every other convolution entry
   * point requires 0. */
  int32_t dtype;
  /* Optional (egne_conv3x3_rw_f16_fwd): SPLIT-PAIR storage of the output.  The split-f16 kernels multiply every fp32 element by a
   * power of two s and split it into hi = f16(x s), lo = f16(x s - hi) while staging it; a tensor whose ONLY consumer stages each
   * element many times (bdcn_new.py:50-54: `o` of an MSBlock feeds three dilated 3x3 convolutions, 13.5 stagings per element)
   * is written in that form once, by its producer: per pixel and 32-channel block 128 bytes = [hi x 32 | lo x 32] halves instead
   * of 32 floats (same bytes, same strides / offsets in units of 4 bytes; out_ch_off and Cout_store whole blocks); position p of
   * either plane holds channel 16 * ((p >> 2) & 1) + 4 * (p >> 3) + (p & 3) of the block (the order the producer's lanes end
   * with, so that a lane stores 16 contiguous bytes per plane) and the consumer's weights are packed in that order.  The consumer
   * (seg.presplit = 1) copies 16-byte pieces straight into its LDS operand image and recovers x = (hi + lo) / s where it needs
   * the value itself (the 4-way sum of bdcn_new.py:54), exact to 2^-22 |x|.
   * out_split = 2 (egne_conv3x3_rw_f16_fwd and egne_conv3x3_smallcin_f16_fwd with f16_products = 1; round 6): F16 storage -- the output
   * (and egne_conv3x3_rw_f16_fwd's pooled second output) is written as f16(v s), one half per element in CHANNEL order; `out` /
   * `pool_out` then point at halfs, strides and offsets count halfs (slices on multiples of 8).  A plain-f16 consumer rounds its operand to
   * exactly this value while staging it (f16(x a_scale), a_scale = s), so the stored tensor carries everything the consumer would have
   * kept at half the bytes: conv1_1 / conv1_2 / pool1 of the frozen edge network next to a bf16-storage training plan
   * (vgg16_c.py:66-70 under utils.py:646).  Consumers read it through seg.presplit = 2 with a_scale = s. */
  int32_t out_split;
  float out_split_scale;      /* s > 0, a power of two */
  /* Optional (every split-f16 entry point): sticky overflow word.  The kernel sets bit 0 when a value it stores is not finite --
   * the trace of an operand that left the f16 range under the launch's calibrated pre-scale (or of non-finite input).  The
   * caller clears it, reads it behind the launches it covers and answers by re-calibrating (engine.Plan.check_overflow). */
  uint32_t* ovf_flag;
  /* Optional (egne_conv2d_f16x3_big_fwd, egne_msblock_dil(_scores)_f16_fwd on split-pair input, egne_conv3x3_rw_f16_fwd): products per
   * multiply of the split-f16 arithmetic.  0 / 3: a b ~= hi hi + hi lo + lo hi (22-bit significand, the inference plans and every
   * fp32-storage plan).  1: hi hi only -- plain f16 operands (11-bit significand), fp32 accumulation: the frozen edge network next
   * to a training plan with BF16 activation storage (BASELINE.json configs[2..4]), whose input the edge map is rounded to bf16
   * (8-bit significand) anyway.  Entry points that do not know the field compute all three products. */
  int32_t f16_products;
  /* Optional (egne_conv3x3_bf16_fwd used as a DATA GRADIENT, round 5): the launch is the last writer of a gradient slice and applies the
   * activation mask of the layer whose OUTPUT the slice is the gradient of -- stored value = v * act'(y), y = that layer's activated
   * output (bf16, same pixels and channels; mask_act = its egne_act) -- and leaves the channel sums of what it stored for that layer's
   * bias gradient: mask_sums [egne_conv3x3_bf16_sum_rows()][Cout_store] floats, one row per consumer wave (zero-initialised by the caller:
   * a wave writes the 32 channels of its output block only), added by egne_group_sums_reduce in a fixed order.  Replaces an
   * egne_act_bwd_bias pass (read g, read y, write g) over the slice. */
  const void* mask_y; int64_t mask_pix_stride; int32_t mask_ch_off; int32_t mask_act;
  float* mask_sums;
} egne_conv_desc;

int egne_conv2d_fwd(const egne_conv_desc* d, void* stream);
int egne_conv3x3_bf16_sum_rows(void);      /* rows of egne_conv_desc.mask_sums a launch of egne_conv3x3_bf16_fwd writes */

/*
 * Fast path for the 3x3 / stride 1 / "same" / dilation<=2 / single-slice convolutions (most of
 * vgg16_c.py:66-78, utils.py:1047-1048 and the 3x3s of models/RITnet_v2.py:57-62,85-87): the input halo
 * tile is staged once per 32 channels in LDS and all 9 taps run from it.  Same descriptor as
 * egne_conv2d_fwd, except that `w` must be the FRAGMENT-order pack of egne_pack_conv_weight_frag
 * ([tap][Ktot/8][CoutP/32][lane][4], each wave's B fragment is one coalesced 1-KiB load).
 */
int egne_conv3x3_halo_supported(const egne_conv_desc* d);
int egne_conv3x3_halo_fwd(const egne_conv_desc* d, void* stream);
int egne_pack_conv_weight_frag(const float* w_oihw, int Cout, int Cin, int kh, int kw,
                               const int32_t* kinv, int CoutP, int Ktot, float* w_packed, void* stream);

/*
 * Split-precision convolution for the FROZEN edge extractor (vgg16_c.py:66-78, bdcn_new.py:50): fp32 tensors,
 * every operand split into two f16 halves (22-bit significand), three v_mfma_f32_32x32x16_f16 per product,
 * fp32 accumulation -- 5.3x the matrix rate of the exact-fp32 MFMA.  Same descriptor as egne_conv2d_fwd
 * (`w` unused; CoutP = row count of the f16 pack, Ktot = slice width rounded up to 32).  One input slice
 * (fused affine allowed), stride 1, zero padding; ngroups = 3 fuses the MSBlock dilated branch.  a_scale / w_scale: exact power-of-two pre-scales that keep
 * the low halves in the f16 normal range (w_scale is baked into the pack).
 */
int egne_pack_conv_weight_f16x2(const float* w_oihw, int Cout, int Cin, int kh, int kw, int CoutP, int Ktot,
                                float wscale, void* whi, void* wlo, void* stream);
int egne_conv2d_f16x3_fwd(const egne_conv_desc* d, const void* whi, const void* wlo, float a_scale,
                          float w_scale, void* stream);
/* The same convolution for SMALL problems (one or two frames at the 30x40 / 15x20 levels: evaluate.py:235-249 feeds one eye per call):
 * few output tiles and a long K loop would leave most compute units idle, so the launch uses 64 x 64 (or 128 x 32) tiles and splits the K
 * range over gridDim.z workgroups; their partial sums go through `ws` ([Z][B*Ho*Wo][CoutP] floats) and a second launch applies bias /
 * activation / post affine / residual.  egne_conv2d_f16x3_small_workspace_floats: floats of workspace this descriptor needs (0: small tiles,
 * no split), or -1 if the problem is not small -- egne_conv2d_f16x3_small_fwd then runs exactly egne_conv2d_f16x3_fwd.  ngroups = 1 only. */
int64_t egne_conv2d_f16x3_small_workspace_floats(const egne_conv_desc* d);
int egne_conv2d_f16x3_small_fwd(const egne_conv_desc* d, const void* whi, const void* wlo, float a_scale, float w_scale,
                                float* ws, int64_t ws_floats, void* stream);

/* Deep variant of egne_conv2d_f16x3_fwd for the wide trunk layers (vgg16_c.py:70-78): 256 x 256 (or 256 x 128) tile, 8 waves,
 * two LDS stages, one barrier per K step; weights as ready-made LDS images staged by LDS-DMA
 * ([Cout tile][step = chunk*taps + tap][BN rows][128 B], hi | lo granules, swizzled; BN = 256 if Cout % 256 == 0 else 128),
 * fp32 activations converted while staging.  One input slice without fused affine, Cp % 32 == 0, stride 1, zero padding. */
int egne_pack_conv_weight_f16img(const float* w_oihw, int Cout, int Cin, int kh, int kw, int BN, int Ktot, float wscale,
                                 void* wimg, void* stream);
int egne_conv2d_f16x3_big_fwd(const egne_conv_desc* d, const void* wimg, float a_scale, float w_scale, void* stream);
/* The deep variant on PLAIN f16 operands (d->f16_products = 1: the frozen edge network next to a bf16-storage training plan; vgg16_c.py:70-78
 * under utils.py:646): same tile, same accumulation order (bit-identical to egne_conv2d_f16x3_big_fwd with f16_products = 1), but 64-byte LDS
 * rows (hi halves only) in a FOUR-stage ring: activations written two K steps ahead, weight images requested three ahead, the next step's
 * fragments read during the current one.  Needs an even number of K steps (Cp / 32 * kh * kw: egne_conv2d_f16_big1_supported) and weight
 * images without lo granules ([Cout tile][step][BN rows][64 B], egne_pack_conv_weight_f16img1: half the bytes of the f16img pack). */
int egne_pack_conv_weight_f16img1(const float* w_oihw, int Cout, int Cin, int kh, int kw, int BN, int Ktot, float wscale,
                                  void* wimg, void* stream);
int egne_conv2d_f16_big1_supported(const egne_conv_desc* d);
int egne_conv2d_f16_big1_fwd(const egne_conv_desc* d, const void* wimg, float a_scale, float w_scale, void* stream);

/* Split-f16 variant of the LDS-halo 3x3 kernel (narrow full-resolution layers: Cout 32 / 64, dilation <= 2, one
 * input slice, fused affine allowed).  Weights: hi / lo f16 in MFMA-fragment order
 * [tap][Ktot/16][CoutP/32][lane][8], Ktot = slice width rounded up to 32. */
int egne_pack_conv_weight_f16frag(const float* w_oihw, int Cout, int Cin, int kh, int kw, int CoutP, int Ktot,
                                  float wscale, void* fhi, void* flo, void* stream);
int egne_conv3x3_halo_f16_fwd(const egne_conv_desc* d, const void* fhi, const void* flo, float a_scale,
                              float w_scale, void* stream);

/* The dilated branch of a BDCN MSBlock in ONE launch (bdcn_new.py:51-54): out = o + sum_g relu(conv3x3_{dil g}(o) + b_g),
 * 32 -> 32 channels, dilations {4, 8, 12}.  Same descriptor as the grouped form of egne_conv2d_f16x3_fwd (ngroups = 3,
 * one raw 32-channel slice, CoutP = Ktot = 32, bias [3][32], act = ReLU, residual = the input slice); weights: three
 * consecutive egne_pack_conv_weight_f16frag packs.  A workgroup keeps the 4-way sum of its 8 x 32 tile in registers, so o
 * is read once and out written once (round 1: three launches accumulating through HBM). */
int egne_msblock_dil_f16_fwd(const egne_conv_desc* d, const void* fhi, const void* flo, float a_scale, float w_scale,
                             void* stream);
/* ... with the stage's score heads in the epilogue (bdcn_new.py:118-166: conv*_down 32 -> 21 summed over the stage's blocks,
 * then score_dsn* / score_dsn*_1 21 -> 1; linear, so per block and head one 32-vector): s0[p] (+)= score_w[0] . out[p],
 * s1[p] (+)= score_w[1] . out[p]; the first block of a stage (accumulate = 0) also adds the constants score_c[2].  With
 * d->out == NULL the 32-channel block output is not stored at all (nothing else reads it). */
int egne_msblock_dil_scores_f16_fwd(const egne_conv_desc* d, const void* fhi, const void* flo, float a_scale, float w_scale,
                                    const float* score_w, const float* score_c, float* s0, float* s1, int accumulate,
                                    void* stream);

/* Transition_down of an inference plan in one pass (models/RITnet_v2.py:32-44: avg_pool2d(conv1x1(leaky(IN(cat(out, x)))), 2) --
 * pooling and 1x1 are linear, the 2x2 average moves in front): the 1x1's operand is the window average of
 * act_in(x * scale + shift).  d: kh = kw = 1, H / W = input size, Ho / Wo = H / 2, W / 2, every slice with scale / shift
 * [B][Cp]; weights as for egne_conv1x1_f16x3_fwd. */
int egne_conv1x1_pool2_f16x3_fwd(const egne_conv_desc* d, const void* fhi, const void* flo, float a_scale, float w_scale,
                                 void* stream);

/* Role-split variant of egne_conv3x3_halo_f16_fwd for narrow inputs (one slice of <= 64 channels: vgg16_c.py:66-69 conv1_2 /
 * conv2_1, bdcn_new.py:50 stage-1 MSBlock convs, models/RITnet_v2.py:57 down-block conv1, utils.py:1047-1048 decoder convBlock):
 * 4 producer waves stage the halo (gather, fused affine, fp32 -> hi / lo) while 4 consumer waves do nothing but LDS reads and
 * MFMAs.  Same descriptor and weight pack; dilation 1, Ktot 32 or 64, CoutP 32 / 64 / 128; stats_ws allowed. */
int egne_conv3x3_rs_f16_fwd(const egne_conv_desc* d, const void* fhi, const void* flo, float a_scale, float w_scale,
                            void* stream);

/* The same convolution with the workgroup's weights RESIDENT in LDS (one 32-channel output block per workgroup, the input staged
 * once per block): the consumer waves issue no vector loads, so their stores are never waited for (loads and stores retire through
 * one in-order counter).  Same descriptor and pack; Cout_store % 8 == 0, 16-byte aligned slices, no statistics; optional pooled
 * second output (pool_out). */
int egne_conv3x3_rw_f16_fwd(const egne_conv_desc* d, const void* fhi, const void* flo, float a_scale, float w_scale,
                            void* stream);

/* Streaming 1x1 convolution over a concatenation of raw NHWC slices (models/RITnet_v2.py:59,61,84,86 conv21 / conv31 /
 * conv11, :38 Transition_down in eval plans) on the split-f16 path: no staging, every lane loads its MFMA operand
 * straight from HBM, weight fragments stay in LDS.  Same descriptor as egne_conv2d_fwd (`w` unused; no fused affine,
 * residual or post affine; CoutP 32 or a multiple of 64).  Weights: hi / lo f16 fragments [G][CoutP/32][lane][8] where
 * G = sum over slices of ceil(Cp/16) and kmap[g*16 + h*8 + j] (device int32) names the logical input channel in K
 * slot (half h, j) of group g (-1 = padding); the kernel's slot order is channel 16g + (j<4 ? 4h+j : 8+4h+j-4).
 * Up-sampled addend: with d->residual set and Ho * 2 == H, Wo * 2 == W (instead of Ho == H, Wo == W) the residual is a
 * HALF-resolution tensor [B][Ho][Wo] (res_pix_stride / res_ch_off, 16-byte aligned channel vectors) whose bilinear x2
 * up-sampling (F.interpolate, scale 2, align_corners False) is added to the activation-free result:
 * conv11(cat(up(x), skip)) = up(W_up x) + W_skip skip of an up block (models/RITnet_v2.py:84-86) without the up-sampled
 * tensor. */
int egne_pack_conv1x1_weight_f16(const float* w_oihw, int Cout, int Cin, const int32_t* kmap, int G, int CoutP,
                                 float wscale, void* fhi, void* flo, void* stream);
int egne_conv1x1_f16x3_fwd(const egne_conv_desc* d, const void* fhi, const void* flo, float a_scale,
                           float w_scale, void* stream);

/* A 1x1 convolution over raw slices FUSED with the 3x3 / pad 1 convolution that is its only consumer
 * (models/RITnet_v2.py:59-62 conv22(conv21(.)), conv32(conv31(.)); :84-87 conv12(conv11(.)), conv22(conv21(.))): the 1x1 is
 * evaluated on each tile's halo and kept in LDS as hi / lo halves, the 3x3 runs from there -- its input is never stored.
 * d1 = the 1x1 (slices as for egne_conv1x1_f16x3_fwd; CoutP 32 or 64; no activation; output fields ignored),
 * d2 = the 3x3 (Ktot = d1->CoutP, CoutP 32 or 64, bias / act / post affine / residual / output as for
 * egne_conv3x3_halo_f16_fwd; seg[] ignored).  w1hi / w1lo: egne_pack_conv1x1_weight_f16, f2hi / f2lo:
 * egne_pack_conv_weight_f16frag.  a1 pre-scales the slices, a2 the 1x1 result (powers of two). */
int egne_conv1x1_3x3_fused_f16_fwd(const egne_conv_desc* d1, const egne_conv_desc* d2, const void* w1hi, const void* w1lo,
                                   float a1, float w1_scale, const void* f2hi, const void* f2lo, float a2, float w2_scale,
                                   void* stream);

/* The same with a 3x3 / pad 1 convolution on <= 4 input channels in front (utils.py:1047-1048 convBlock: conv1 -> LeakyReLU ->
 * conv2 -> LeakyReLU [-> eval BatchNorm = d2's post affine]): d1 = that convolution (one slice with >= 4 padded channels,
 * CoutP = 32; its activation is applied before the 3x3), c4hi / c4lo = egne_pack_conv3x3_c4_weight_f16 (CoutP 32).
 * A one-channel input may be given in place as [B][H][W] floats (= NCHW with C = 1): seg[0].pix_stride = 1, ch_off = 0. */
int egne_conv3x3c4_3x3_fused_f16_fwd(const egne_conv_desc* d1, const egne_conv_desc* d2, const void* c4hi, const void* c4lo,
                                     float a1, float w1_scale, const void* f2hi, const void* f2lo, float a2, float w2_scale,
                                     void* stream);

/* LDS-staged variant of the split-f16 1x1 convolution over raw slices, for the layers whose K or Cout exceed what the
 * streaming kernel keeps in LDS (decoder conv11 / conv21 at 30x40 and 60x80, models/RITnet_v2.py:84,86; dense block 3):
 * 128x128 (CoutP % 128 == 0) or 256x64 tiles.  d->Ktot = sum of the slice widths rounded up to 32 each; weights: hi / lo
 * f16 [CoutP][Ktot] with kmap[k] (device int32) naming the logical input channel of K column k (-1 = padding). */
int egne_pack_conv1x1_weight_f16x2_map(const float* w_oihw, int Cout, int Cin, const int32_t* kmap, int CoutP, int Ktot,
                                       float wscale, void* whi, void* wlo, void* stream);
int egne_conv1x1_ms_f16x3_fwd(const egne_conv_desc* d, const void* whi, const void* wlo, float a_scale,
                              float w_scale, void* stream);

/* First layers (vgg16_c.py:66 conv1_1 on 3 channels, utils.py:1047 convBlock conv1 on 1-2 channels):
 * 3x3 / stride 1 / pad 1, logical Cin <= 4, Cout <= 64.  The 9 taps are folded into K (one 40-wide K step,
 * exact fp32 MFMA), so the layer is a pure store stream.  w40: [32 or 64][40] fp32, column tap*4 + c. */
int egne_conv3x3_smallcin_fwd(const egne_conv_desc* d, const float* w40, void* stream);

/* Split-f16, streaming form of egne_conv3x3_smallcin_fwd for frozen / inference plans: the 9 taps folded into K = 48 (three
 * K=16 steps), nothing staged -- a lane's operand for one step is the 4-channel vectors of two taps of its pixel (two 16-byte
 * buffer loads), weight fragments in registers, 16-byte stores.  Weights: hi / lo f16 [3][CoutP/32][lane][8], K slot
 * (step s, half h, j) = tap 4s + 2h + (j>>2), channel j&3; d->CoutP = 32 or 64. */
int egne_pack_conv3x3_c4_weight_f16(const float* w_oihw, int Cout, int Cin, int CoutP, float wscale, void* fhi, void* flo,
                                    void* stream);
int egne_conv3x3_smallcin_f16_fwd(const egne_conv_desc* d, const void* fhi, const void* flo, float a_scale,
                                  float w_scale, void* stream);

/* OIHW (torch layout) -> packed [tap][CoutP][Ktot].  kinv[k] (device int32, k < Ktot) names the
 * input channel stored at padded K position k, or -1 for a padding column; rows Cout..CoutP-1 are
 * zero.  Used at load_state_dict / after optimizer steps. */
int egne_pack_conv_weight(const float* w_oihw, int Cout, int Cin, int kh, int kw,
                          const int32_t* kinv, int CoutP, int Ktot, float* w_packed, void* stream);

/* max |x| over the channel slice [ch_off, ch_off+Cp) of npix NHWC pixels, written as the float's bit pattern with an
 * integer atomic max into *out_bits (zero it first; a NaN input gives a pattern above +inf).  Used once per plan to
 * choose the power-of-two pre-scale of the split-f16 kernels (a_scale): |x| * a_scale must stay below the f16 range,
 * which the fixed scale of round 1 did not guarantee for arbitrary checkpoints. */
int egne_absmax(const float* x, int64_t pix_stride, int ch_off, int Cp, int64_t npix, void* out_bits, void* stream);
/* The same over a slice held as F16 (egne_conv_desc.out_split = 2; pix_stride / ch_off in halfs, multiples of 8): the word receives the
 * bit pattern of max |x| as a FLOAT, so the host derives scales from either measurement alike. */
int egne_absmax_f16(const void* x, int64_t pix_stride, int ch_off, int Cp, int64_t npix, void* out_bits, void* stream);

/* Per-(n,c) mean / inverse std over H*W of an NHWC slice -> scale = rstd, shift = -mean*rstd
 * ([B][Cp]).  F.instance_norm at models/RITnet_v2.py:40,57 (eps 1e-5, biased variance).  With
 * per_sample=0 the statistics run over (B,H,W): training-mode BatchNorm2d of utils.py:1049 and
 * the result is [1][Cp]; mean/var (biased) are also written when the pointers are non-NULL.
 * Two deterministic stages with fp64 partial sums in `ws` (egne_norm_stats_workspace_bytes; 16-byte aligned, as for egne_norm_bwd). */
int64_t egne_norm_stats_workspace_bytes(int B, int HW, int Cp, int per_sample);
int egne_norm_stats(const float* x, int64_t pix_stride, int ch_off, int Cp, int B, int HW,
                    int per_sample, float eps, float* scale, float* shift,
                    float* mean_out, float* var_out, void* ws, void* stream);

/* Second stage of egne_norm_stats on partial sums that a convolution wrote from its epilogue (egne_conv_desc.stats_ws):
 * ws [B][nchunk][Cp][2] doubles -> scale = rstd, shift = -mean*rstd ([B][Cp]); fixed summation order (deterministic). */
int egne_norm_stats_finish(const void* ws, int Cp, int B, int nchunk, int HW, float eps, float* scale, float* shift,
                           void* stream);
/* The same with mean / biased variance written too ([B][Cp]): batch statistics of a training-mode BatchNorm (utils.py:1049) from a
 * convolution's epilogue (round 5) -- the B samples of the batch as ONE sample of B * nchunk chunks and B * HW pixels. */
int egne_norm_stats_finish_moments(const void* ws, int Cp, int B, int nchunk, int HW, float eps, float* scale, float* shift,
                                   float* mean_out, float* var_out, void* stream);

/* y = x*scale[c] + shift[c] in place over an NHWC slice (training-mode BatchNorm apply). */
int egne_affine_inplace(float* x, int64_t pix_stride, int ch_off, int Cp, int64_t npix,
                        const float* scale, const float* shift, void* stream);

/* y = x*scale[c] + shift[c], out of place (training-mode BatchNorm keeps its input for backward). */
int egne_affine(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo, int Cp, int64_t npix,
                const float* scale, const float* shift, void* stream);

/* nn.AvgPool2d(2) (models/RITnet_v2.py:36,43; utils.py:1017) on an NHWC slice. */
int egne_avgpool2(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo,
                  int B, int H, int W, int Cp, void* stream);

/* avgpool2(act(x*scale[n][c] + shift[n][c])): Transition_down (models/RITnet_v2.py:32-44) with the
 * AvgPool2d commuted in front of its 1x1 conv (both linear; rounding-level difference only). */
int egne_norm_act_pool2(const float* x, int64_t xs, int xo, const float* scale, const float* shift, int act,
                        float* y, int64_t ys, int yo, int B, int H, int W, int Cp, void* stream);

/* nn.MaxPool2d(2, stride, ceil_mode=True) (vgg16_c.py:15,20,27,34) on an NHWC slice. */
int egne_maxpool2(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo,
                  int B, int H, int W, int Ho, int Wo, int stride, int Cp, void* stream);
/* The same pooling over a slice held as F16 (egne_conv_desc.out_split = 2; strides / offsets in halfs, multiples of 8): the maximum commutes with
 * the storage scale, so the output is held under the input's scale (pool3 / pool4 of vgg16_c.py:76-82 in a plain-f16 plan). */
int egne_maxpool2_f16(const void* x, int64_t xs, int xo, void* y, int64_t ys, int yo, int B, int H, int W, int Ho, int Wo, int stride,
                      int Cp, void* stream);

/* F.interpolate(bilinear, scale_factor=2, align_corners=False) (models/RITnet_v2.py:80-83). */
int egne_upsample2x(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo,
                    int B, int H, int W, int Cp, void* stream);

/* NCHW [B,C,H,W] -> NHWC slice (zero fills channels C..Cp-1) and back. */
int egne_nchw_to_nhwc(const float* x, int B, int C, int H, int W, float* y, int64_t ys, int yo,
                      int Cp, void* stream);
int egne_nhwc_to_nchw(const float* x, int64_t xs, int xo, int B, int C, int H, int W, float* y,
                      void* stream);

/*
 * BDCN side outputs of one stage (bdcn_new.py:118-164): the 1x1 `convK_j_down` (32->21) of every
 * MSBlock output of the stage are summed and the two 1x1 score heads (21->1) applied; writes the two
 * low-resolution score maps s (score_dsnK) and s1 (score_dsnK_1), each [B,h,w].
 * ms is a HOST array of nblk device pointers to NHWC 32-channel tensors; wd [nblk][21][32], bd [nblk][21]; ws/ws1 [21], bs/bs1 scalars
 * passed by pointer (device).
 */
int egne_bdcn_stage_scores(const float* const* ms, int nblk, int64_t ms_pix_stride, int64_t npix,
                           const float* wd, const float* bd, const float* ws, const float* bs,
                           const float* ws1, const float* bs1, float* s, float* s1, void* stream);

/*
 * BDCN tail (bdcn_new.py:127-191): ConvTranspose2d upsampling (k=2*stride, weights from the
 * checkpoint) + crop of stages 2..5, the two deep-supervision cascades, the 1x1 fuse over the 10
 * maps and the sigmoids.  s[k]/s1[k] are the stage score maps [B,h_k,w_k]; up[k] the k_k x k_k
 * upsampling kernels (up[0] unused).  out[0..10] are [B,1,H,W] maps (any may be NULL to skip it;
 * calc_edge only needs out[10], the fused edge map).
 */
typedef struct {
  int32_t B, H, W;
  const float* s[5];
  const float* s1[5];
  int32_t h[5], w[5];
  int32_t stride[5], crop[5];
  const float* up[5];
  const float* fuse_w;   /* [10] */
  const float* fuse_b;   /* [1]  */
  float* out[11];
  int32_t edge_thres;    /* utils.py:653-655: out[10] >= 0.1 -> 1 */
} egne_bdcn_tail_desc;
int egne_bdcn_tail(const egne_bdcn_tail_desc* d, void* stream);

/*
 * Loss head (models/RITnet_v2.py:372-432, loss.py:16-137) in two launches and no host sync.
 * Pass 1 reads the logits once (NHWC slice, 3 classes) and writes per-block partial sums;
 * pass 2 (one block per sample + one finishing block) reduces them to
 *   out_terms[0..7] = total, l_seg2pt, l_seg, l_pt, l_ellipse, n_mask_present, bad_sample_flag, 0
 *   pred_c[B][2][2] = (iris, pupil) soft-argmax centres (iris = elOut[:,5:7] when no mask in batch)
 *   mask[B][H][W]   = argmax over the 3 classes (int64, first max on ties; utils.py:65-81), optional
 *   op_nchw         = logits transposed to [B,3,H,W], optional
 */
typedef struct {
  int32_t B, H, W;
  const float* logits; int64_t pix_stride; int32_t ch_off;
  const int64_t* target;       /* [B,H,W] */
  const float* spatWts;        /* [B,H,W] */
  const float* distMap;        /* [B,3,H,W] */
  const float* cond;           /* [B,4], 0 = annotation present */
  const float* pupil_center;   /* [B,2] pixels */
  const float* elNorm;         /* [B,2,5] */
  const float* elOut;          /* [B,10] */
  float alpha;
  const float* grid_x;         /* optional [W] / [H] mesh axes (torch.linspace(-1,1,.)); NULL = closed form */
  const float* grid_y;
  float* partials;             /* workspace, egne_loss_workspace_floats(B,H,W) floats */
  float* out_terms;            /* [8] */
  float* pred_c;               /* [B,2,2] */
  float* elPred;               /* [B,10] = (c_iris, elOut[2:5], c_pupil, elOut[7:10]) */
  int64_t* mask;               /* optional */
  float* op_nchw;              /* optional */
  float* coef;                 /* optional [B][32]: per-sample state kept for egne_loss_bwd */
  int32_t dtype;               /* storage of `logits`: 0 fp32, 1 bf16 (pix_stride / ch_off in elements); everything else stays fp32 */
  /* egne_loss_bwd only, all optional (round 5): gradients a caller back-propagates through the OUTPUTS next to the loss
   * (models/RITnet_v2.py:334-354 returns op, elPred, elOut with grad): added to what the loss terms themselves give. */
  const float* g_op_nchw;      /* [B,3,H,W] upstream gradient w.r.t. the logits */
  const float* g_pred_c;       /* [B,2,2] upstream gradient w.r.t. pred_c (iris, pupil soft-argmax centres; the iris row is ignored when
                                  no sample of the batch has a mask: pred_c's iris row is then a copy of elOut[:,5:7], RITnet_v2.py:404) */
  const float* g_elOut_up;     /* [B,10] upstream gradient w.r.t. elOut */
} egne_loss_desc;
int64_t egne_loss_workspace_floats(int B, int H, int W);
int egne_loss_fwd(const egne_loss_desc* d, void* stream);

/* Loss of the DeepVOG comparator (models/deepvog_pytorch.py:148-167 get_allLoss), forward only: 10 * cross_entropy(softmax(op), label == 2)
 * averaged per frame and over the frames with cond[:,1] == 0, plus the mean L1 distance between the soft-argmax centre of channel 1
 * (loss.py:16-46, temperature 4) and utils.normPts(pupil_center).  logits: two channels of an NHWC fp32 slice.  Also writes the NCHW logits
 * [B,2,H,W], the argmax mask [B,H,W] int64 and pred_c [B,2]; out_terms[0] = loss, [1] = segmentation term, [2] = centre term. */
int64_t egne_deepvog_loss_workspace_floats(int B, int H, int W);
int egne_deepvog_loss_fwd(const float* logits, int64_t pix_stride, int ch_off, const int64_t* target, const float* pupil_center,
                          const float* cond, int B, int H, int W, float* partials, float* out_terms, float* pred_c,
                          float* op_nchw, int64_t* mask, void* stream);
/* Its backward (loss.backward() of train.py:286 for this comparator): g_logits[b][y][x][go .. go+2) = gscale[0] * d loss / d logits, from the
 * forward's `partials` and `pred_c` (the launch stores, it does not accumulate). */
int egne_deepvog_loss_bwd(const float* logits, int64_t pix_stride, int ch_off, const int64_t* target, const float* pupil_center,
                          const float* cond, int B, int H, int W, const float* partials, const float* pred_c, const float* gscale /* device */,
                          float* g_logits, int64_t gs, int go, void* stream);
/* y = act(x * scale[c] + shift[c]) over an NHWC fp32 slice: BatchNorm (scale / shift from its statistics) followed by its activation,
 * the order of models/deepvog_pytorch.py:36-41 (conv -> bn -> relu). */
int egne_affine_act(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo, int Cp, int64_t npix, const float* scale,
                    const float* shift, int act, void* stream);

/* Nearest-neighbour x2 up-sampling of an NHWC slice and its transpose (gx += the four copies): F.interpolate(scale_factor=2,
 * mode='nearest') of the comparator model models/RITnet_v1.py:89 (H, W = INPUT size).  bf16 twins: *_bf16. */
int egne_upsample2x_nearest(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo, int B, int H, int W, int Cp, void* stream);
int egne_upsample2x_nearest_bwd(const float* gy, int64_t gs, int go, float* gx, int64_t xs, int xo, int B, int H, int W, int Cp, void* stream);
int egne_upsample2x_nearest_bf16(const void* x, int64_t xs, int xo, void* y, int64_t ys, int yo, int B, int H, int W, int Cp, void* stream);
int egne_upsample2x_nearest_bwd_bf16(const void* gy, int64_t gs, int go, void* gx, int64_t xs, int xo, int B, int H, int W, int Cp, void* stream);

/* regressionModule output activations (utils.py:1023-1036): tanh / sigmoid / identity split of the
 * 10 raw outputs, in place over [B,10] (row stride `ld`). */
int egne_ellipse_head_act(float* x, int B, int ld, void* stream);
/* torch.selu in place over n floats (utils.py:1021). */
int egne_selu_inplace(float* x, int64_t n, void* stream);
/* latent = mean over H*W of an NHWC slice -> [B][C] (models/RITnet_v2.py:282). */
int egne_spatial_mean(const float* x, int64_t pix_stride, int ch_off, int C, int B, int HW,
                      float* out, void* stream);

/* AdaIN fusion path (models/RITnet_v2.py:289-308).  egne_softmax3: nn.Softmax(dim=1) over the 3 logits
 * of every pixel into an NHWC slice of Cp_out channels (zero padded) that feeds the StyleEncoder.
 * egne_adain: calc_mean_std (:251-259, UNBIASED variance + eps) and
 * x' = (x-mean)/std * gamma[n][c] + beta[n][c]; gamma/beta are rows of the MLP output
 * (element [n*gb_stride + gb_off + c]). */
int egne_softmax3(const float* x, int64_t xs, int xo, float* y, int64_t ys, int yo, int Cp_out,
                  int64_t npix, void* stream);
int egne_adain(const float* x, int64_t xs, int xo, int C, const float* gamma, const float* beta,
               int64_t gb_stride, int gb_off, float* y, int64_t ys, int yo, int B, int HW, float eps,
               void* stream);

/* conf_Loss (loss.py:139-157) on pred [B,C] (row stride ld).  flag=1: conf = mean|softmax - 1/C| and
 * terms[0] += weight*conf (RITnet_v2.py:345-347); flag=0: conf = cross-entropy(pred, gt) and
 * terms[0] = conf (:348-350).  terms[7] = conf.  Runs after egne_loss_fwd on the same stream. */
int egne_conf_loss(const float* pred, int ld, const int64_t* gt, int B, int C, int flag, float weight,
                   float* terms, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of ESF-Net (loss.backward() of train.py:286; BDCN is frozen and has no backward).
 * Data gradients of convolutions are forward convolutions with the weight pack of
 * egne_pack_conv_weight_dgrad (flipped taps, in/out channels swapped, restricted to one input slice),
 * accumulated into the slice's gradient through `residual`.
 * ------------------------------------------------------------------------------------------------ */

/* d total_loss * gscale / d logits (written to an NHWC slice, 3 channels) and / d elOut [B,10].
 * Needs the descriptor of the matching egne_loss_fwd call with `coef`, `grid_x`, `grid_y` set. */
int egne_loss_bwd(const egne_loss_desc* d, const float* gscale /* device, 1 float */, float* g_logits,
                  int64_t gs, int go, float* g_elOut, void* stream);

/* In place g <- g * act'(y) over an NHWC slice (y = forward output of the conv) and the bias gradient
 * dbias[c] (+)= sum over pixels of the masked g, c < C. */
int64_t egne_act_bwd_bias_workspace_bytes(int64_t npix, int Cp);
/* egne_act_bwd_bias_absmax: the same, plus the atomic max of the bit patterns of |gz| into *absmax_bits (the dyn_scale word
 * of the split-f16 data-gradient launches that read gz next). */
int egne_act_bwd_bias_absmax(float* g, int64_t gs, int go, const float* y, int64_t ys, int yo, int act, int Cp,
                             int64_t npix, float* dbias, int C, int accumulate, void* ws, uint32_t* absmax_bits, void* stream);
int egne_act_bwd_bias(float* g, int64_t gs, int go, const float* y, int64_t ys, int yo, int act, int Cp,
                      int64_t npix, float* dbias, int C, int accumulate, void* ws, void* stream);
/* Zero a list of device buffers in one launch (a backward plan's gradient twins before each pass; the reference's autograd
 * allocates fresh gradients instead).  table: device array of nseg x {uint64 address, uint64 bytes, uint64 first_block}, addresses
 * 16-byte aligned, bytes a multiple of 16, first_block = the running sum of ceil(bytes / 65536); nblocks = that sum over all. */
int egne_zero_many(const void* table, int nseg, int64_t nblocks, void* stream);
/* Bias gradient of an activation-free 1x1 that feeds a 3x3 (stride 1, zero pad 1) -- the 'a' layer of the reference's dense-block
 * pairs (RITnet_v2.py:145-156 conv1x1 -> conv3x3) -- without a pass over the 3x3's data gradient g_tmp: sum_q g_tmp[q][ca] =
 * sum_{co,tap} w[co][ca][tap] * S_tap[co], S_tap = the 3x3's per-channel gradient total minus the border row / column the tap
 * cannot reach.  g: the 3x3's masked output gradient after egne_act_bwd_bias(..., dbias = NULL, ws = act_ws) over the same
 * B*H*W pixels and Cp channels (its chunk sums are read from act_ws); w: fp32 OIHW [Cout][Ca][3][3]; db_b (may be NULL) += the
 * 3x3's bias gradient, db_a += the 1x1's.  ws: egne_pair_bias_bwd_workspace_bytes(B, Cp) bytes, 8-byte aligned.  Cp, Ca <= 256. */
int64_t egne_pair_bias_bwd_workspace_bytes(int B, int Cp);
int egne_pair_bias_bwd(const float* g, int64_t gs, int go, int Cp, int B, int H, int W, const void* act_ws, const float* w,
                       int Cout, int Ca, float* db_b, float* db_a, void* ws, void* stream);

/* Backward of the normalisation that the forward fused into a conv's load (InstanceNorm, per_sample=1)
 * or of training-mode BatchNorm (per_sample=0, gamma/dgamma/dbeta given): xh = x*scale+shift,
 * g = gy*act_in'(xh), gx += scale*gamma*(g - mean(g) - xh*mean(g*xh)).  sums: [Bn][Cp][2] scratch. */
int64_t egne_norm_bwd_workspace_bytes(int B, int HW, int Cp, int per_sample);
int egne_norm_bwd(const float* x, int64_t xs, int xo, const float* scale, const float* shift,
                  const float* gamma, const float* gy, int64_t gs, int go, int act_in, int Cp, int B, int HW,
                  int per_sample, float* gx, int64_t gxs, int gxo, float* sums, float* dgamma, float* dbeta,
                  int C, void* ws, void* stream);

/* Backward of egne_norm_act_pool2 (training-mode Transition_down with the pooling in front of the 1x1, models/RITnet_v2.py:32-44):
 * egne_norm_bwd (per-sample statistics) whose gy is a quarter of the pooled cell's gradient gzp [B][H/2][W/2]; H, W even;
 * accumulate = 0 stores gx instead of adding to it; workspace as egne_norm_bwd_workspace_bytes(B, H*W, Cp, 1). */
int egne_norm_pool2_bwd(const float* x, int64_t xs, int xo, const float* scale, const float* shift,
                        const float* gzp, int64_t gs, int go, int act_in, int Cp, int B, int H, int W,
                        float* gx, int64_t gxs, int gxo, int accumulate, float* sums, void* ws, void* stream);
/* egne_norm_bwd_store: egne_norm_bwd with gx stored (=) instead of accumulated (+=): the first writer of a gradient slice. */
int egne_norm_bwd_store(const float* x, int64_t xs, int xo, const float* scale, const float* shift,
                        const float* gamma, const float* gy, int64_t gs, int go, int act_in, int Cp, int B, int HW,
                        int per_sample, float* gx, int64_t gxs, int gxo, float* sums, float* dgamma, float* dbeta,
                        int C, void* ws, void* stream);
int egne_avgpool2_bwd(const float* gy, int64_t gs, int go, float* gx, int64_t xs, int xo,
                      int B, int H, int W, int Cp, void* stream);       /* gx += ; H,W = input size */
int egne_upsample2x_bwd(const float* gy, int64_t gs, int go, float* gx, int64_t xs, int xo,
                        int B, int H, int W, int Cp, void* stream);     /* gx += ; H,W = input size */
int egne_ellipse_head_act_bwd(float* g, const float* y, int B, int ld, void* stream);
int egne_selu_bwd(float* g, const float* y, int64_t n, void* stream);
/* Backward of the AdaIN fusion path (models/RITnet_v2.py:289-308); every output is ACCUMULATED (+=).
 * egne_softmax3_bwd: gx[c] += y[c]*(gy[c] - sum_k gy[k]y[k]) over the 3 logits of a pixel.
 * egne_adain_bwd: gradients of x' = gamma*(x-mean)/sqrt(var_unbiased+eps) + beta w.r.t. x (gx), gamma and beta
 *   (rows of the MLP output gradient: element [n*gg_stride + gg_off + c]).
 * egne_reflect_pad_bwd: backward of nn.ReflectionPad2d(P) in front of a Conv2dBlock (utils.py:1099-1100,1144):
 *   gx += fold of the padded-input gradient; phase=1 reads the phase-packed [B,(H+2P)/2,(W+2P)/2,4*Cp] output of the
 *   stride-2 transposed convolution (channel block (py&1)*2+(px&1)). */
int egne_softmax3_bwd(const float* y, int64_t ys, int yo, const float* gy, int64_t gs, int go, float* gx,
                      int64_t xs, int xo, int64_t npix, void* stream);
int egne_adain_bwd(const float* x, int64_t xs, int xo, int C, const float* gamma, int64_t gb_stride, int gb_off,
                   const float* gy, int64_t gys, int gyo, float* gx, int64_t gxs, int gxo, float* ggamma,
                   float* gbeta, int64_t gg_stride, int gg_off, int B, int HW, float eps, void* stream);
int egne_reflect_pad_bwd(const float* gpad, int64_t gs, int go, int phase, int Cp, float* gx, int64_t xs, int xo,
                         int B, int H, int W, int P, void* stream);
int egne_spatial_mean_bwd(const float* g, int gld, float* gx, int64_t xs, int xo, int C, int B, int HW,
                          void* stream);
int egne_conf_loss_bwd(const float* pred, int ld, const int64_t* gt, int B, int C, int flag,
                       const float* gscale /* device, 1 float */, float* gpred, int gld, void* stream);

/* Fused backward of a tensor x that was InstanceNorm-ed once (scale / shift per sample, egne_norm_stats) for up to two consumers
 * (round 5; models/RITnet_v2.py:57: conv1 behind IN(x); :40-44: Transition_down behind avg_pool2d(leaky(IN(.)))): the
 * normalisation's backward is linear in its upstream gradient G = a1 + act_q'(xh) up(gq) / 4 (a1: full-resolution addend, gq: the
 * gradient of the 2x2-pooled tensor; either may be NULL), and its result joins the gradient of x where the layer that PRODUCED x
 * masks it: g <- act'(x) (g + rstd (G - mean G - xh mean(G xh))) in place, with that layer's bias sums (dbias += ..., and the chunk
 * sums in ws_bias as egne_act_bwd_bias leaves them for egne_pair_bias_bwd).  Replaces egne_norm_bwd + egne_norm_pool2_bwd +
 * egne_act_bwd_bias: five passes over full-resolution tensors less.  B samples of H x W pixels (even for a pooled addend).
 * sums: [B][Cp][2] floats (scratch); ws_norm: egne_norm_bwd_workspace_bytes(B, H*W, Cp, 1); ws_bias:
 * egne_act_bwd_bias_workspace_bytes(B*H*W, Cp).  acc_samples: g of the first acc_samples samples is read (accumulated onto), the
 * rest of g is written only (no earlier writer; B: plain read-modify-write of all of g). */
int egne_act_norm_bwd(float* g, int64_t gs, int go, const float* x, int64_t xs, int xo, int act, const float* scale, const float* shift,
                      const float* a1, int64_t a1s, int a1o, const float* gq, int64_t gqs, int gqo, int act_q, int Cp, int B, int H, int W,
                      float* sums, void* ws_norm, float* dbias /* may be NULL */, int C, void* ws_bias, int acc_samples, void* stream);
int egne_act_norm_bwd_bf16(void* g, int64_t gs, int go, const void* x, int64_t xs, int xo, int act, const float* scale, const float* shift,
                           const void* a1, int64_t a1s, int a1o, const void* gq, int64_t gqs, int gqo, int act_q, int Cp, int B, int H, int W,
                           float* sums, void* ws_norm, float* dbias, int C, void* ws_bias, int acc_samples, void* stream);

/* Training-mode BatchNorm backward (batch statistics over B samples, utils.py:1049) together with the masking pass of the layer in
 * front of it (round 5): x = that layer's activated output = the BatchNorm's input, scale / shift = rstd / -mean rstd ([Cp], one row:
 * egne_norm_stats with per_sample = 0), gamma [Cp], gy = gradient of the BatchNorm's output;
 *   gx <- act'(x) rstd gamma (gy - mean gy - xh mean(gy xh))      (stored: the BatchNorm is the only reader of x)
 *   dgamma += sum gy xh, dbeta += sum gy (first Cn channels), dbias += sum gx (first C channels; may be NULL), chunk sums of gx in
 * ws_bias as egne_act_bwd_bias leaves them.  Replaces egne_norm_bwd + egne_act_bwd_bias: one statistics pass (reads gy, x) and one
 * apply pass (reads gy, x, writes gx) instead of those two and a read-modify-write of gx.  sums: [Cp][2] floats (scratch); ws_norm:
 * egne_norm_bwd_workspace_bytes(B, H*W, Cp, 1); ws_bias: egne_act_bwd_bias_workspace_bytes(B*H*W, Cp). */
int egne_bn_act_bwd(const float* x, int64_t xs, int xo, int act, const float* scale, const float* shift, const float* gamma, const float* gy,
                    int64_t gys, int gyo, int Cp, int B, int H, int W, float* gx, int64_t gxs, int gxo, float* sums, void* ws_norm,
                    float* dgamma, float* dbeta, int Cn, float* dbias /* may be NULL */, int C, void* ws_bias, void* stream);
int egne_bn_act_bwd_bf16(const void* x, int64_t xs, int xo, int act, const float* scale, const float* shift, const float* gamma, const void* gy,
                         int64_t gys, int gyo, int Cp, int B, int H, int W, void* gx, int64_t gxs, int gxo, float* sums, void* ws_norm,
                         float* dgamma, float* dbeta, int Cn, float* dbias, int C, void* ws_bias, void* stream);

/* Weight gradient of the convolution described by `d` (same descriptor as the forward call):
 * gw[g][co][ci][kh][kw] += sum_pixels gz[pixel][co] * input[pixel + tap][ci], gz = gradient w.r.t. the
 * pre-activation output.  gw is a HOST array of ngroups device pointers (OIHW, torch layout).
 * ws: egne_conv2d_wgrad_workspace_bytes(d) bytes of split partial sums, ZERO-FILLED by the caller before its first use and kept for
 * calls with the same descriptor shape: the forms that do not write every partial (1x1 over slices, the generic implicit GEMM)
 * rely on the zeros, and their reduction clears what it reads, so the workspace is zero-filled again when the call returns (round 4:
 * this replaced a fill in front of every such launch).  Every argument is validated before the first launch (a rejected call queues
 * nothing and leaves ws as it was); a call whose launches fail clears ws itself before it returns the error (round 5). */
int egne_conv2d_wgrad_splits(const egne_conv_desc* d);
int64_t egne_conv2d_wgrad_workspace_bytes(const egne_conv_desc* d);
int egne_conv2d_wgrad(const egne_conv_desc* d, const float* gz, int64_t gzs, int gzo, int Cout, int Cin,
                      const int32_t* kinv, float* const* gw, void* ws, void* stream);
/* egne_conv2d_wgrad_f16: the same with split-f16 products (3 x v_mfma_f32_32x32x16_f16, fp32 accumulation) for the 3x3 / stride 1 /
 * pad 1 single-slice shapes when gz_absmax_bits (device word: bit pattern of max|gz|, written by egne_act_bwd_bias_absmax) is given
 * and the input carries a pre-scale (normalised on load, or d->dyn_scale from the forward launch); exact fp32 otherwise. */
int egne_conv2d_wgrad_f16(const egne_conv_desc* d, const float* gz, int64_t gzs, int gzo, const uint32_t* gz_absmax_bits,
                          int Cout, int Cin, const int32_t* kinv, float* const* gw, void* ws, void* stream);
int egne_pack_conv_weight_dgrad(const float* w_oihw, int Cout, int Cin, int kh, int kw, int ci0, int Cpiece,
                                int CoutPp, int Ktotp, int frag, float* out, void* stream);

/*
 * Ellipse fit of evaluate.py (utils.py:450-486 search_proper_parameter_iou_for_our_data with
 * calc_ell_iou utils.py:176-204 and the conic algebra of helperfunctions.py:13-63,102-129):
 * one workgroup per (frame, class) runs the whole coordinate search on the device.
 * mask [frames][H][W] int64 class maps, frame_of[n] / cls[n] the frame and class id of each fit,
 * xs[W] / ys[H] the float32 mesh axes exactly as torch.linspace(-1,1,.) produces them (utils.py:27-60
 * create_meshgrid; passed in because ATen's vectorised linspace is not a closed formula),
 * init [n][5] (cx,cy,a,b,theta) pixels, out [n][5] doubles, evals[n] IoU evaluation count (optional).
 * nframes = number of class maps in `mask`; a fit whose frame_of is outside [0, nframes) reads nothing and
 * reports NaN.  Each IoU evaluation scans the ellipse's bounding box only (pixels outside it cannot be inside
 * the ellipse, so the counts -- and therefore the search -- are unchanged).  The two candidates of a coordinate step are scored at the
 * same time by different waves and a sweep's closing score is looked up when its parameters were scored before; decisions are taken in the
 * reference's order and evals[] counts the evaluations the reference's sequential search performs (csrc/fit.hip).
 *
 * egne_ellipse_init_from_pred: the seeds of that search straight from the network's regression output on the
 * device (evaluate.py:135-151 does this per frame on the host): elPred [nframes][10] float32 normalised
 * ellipses -> init [2*nframes][5] float64 pixel ellipses via my_ellipse(p).transform(H)
 * (helperfunctions.py:124-129), fit 2f = iris / class 1 from elPred[f][0:5], fit 2f+1 = pupil / class 2 from
 * elPred[f][5:10]; also fills frame_of / cls [2*nframes].
 */
int egne_ellipse_fit(const int64_t* mask, int nframes, const int32_t* frame_of, const int32_t* cls, int n,
                     int H, int W, const float* xs, const float* ys, const double* init, double* out,
                     int32_t* evals, void* stream);
int egne_ellipse_init_from_pred(const float* elPred, int nframes, int H, int W, double* init,
                                int32_t* frame_of, int32_t* cls, void* stream);

/* Device-side batch preparation (the reference does this per sample on the host in its Dataset).
 * egne_dist_maps: out[b][c] = helperfunctions.one_hot2dist(label[b] == c) (helperfunctions.py:356-371, called from
 *   CurriculumLib.py:131-136): signed exact Euclidean distance transform normalised by the image diagonal, 0 for an absent
 *   class; label int64 [B,H,W], out float32 [B,ncls,H,W] (the float32 cast of the reference's float64 result, bit-identical).
 * egne_zscore: (img - img.mean()) / img.std() per image (CurriculumLib.py:139), statistics in double. */
int64_t egne_dist_maps_workspace_bytes(int B, int H, int W, int ncls);
int egne_dist_maps(const int64_t* label, int B, int H, int W, int ncls, float* out, void* ws, void* stream);
int egne_zscore(const float* x, float* y, int B, int n, void* stream);
/* egne_spatial_weights: 1 + 20 * cv2.dilate(cv2.Canny(label, 0, 1) / 255, (3, 3)) per frame (CurriculumLib.py:128-129); label int64
 *   [B,H,W] (class indices), out float32 [B,H,W].  PARITY UNPINNED: neither OpenCV nor a fixture of this function exists in the
 *   build container; the kernel is bit-identical to the restatement of OpenCV's published algorithm in oracle/dataprep.py. */
int egne_spatial_weights(const int64_t* label, int B, int H, int W, float* out, void* stream);
/* egne_augment: the NumPy branches of data_augment.augment (data_augment.py:12-130; called per sample from CurriculumLib.py:114-120)
 *   over a batch: choice[b] = 0 flip left-right (image and label), 2 gamma = lut[b][pixel] (256-entry uint8 table per frame, built on
 *   the host as the reference builds it), 3 exposure pixel + param[b], 4 noise pixel + param[b] * noise[b][y][x] (standard-normal
 *   draws, double), >= 7 copy; double arithmetic, clip to [0,255], truncation.  The cv2 branches (1 blur, 5 lines, 6 rotate) are not
 *   implemented: the host wrapper rejects them.  img / out_img uint8 [B,H,W], label / out_label int64 [B,H,W]; not in place. */
int egne_augment(const uint8_t* img, const int64_t* label, const int32_t* choice, const double* param, const uint8_t* lut,
                 const double* noise, uint8_t* out_img, int64_t* out_label, int B, int H, int W, void* stream);

/*
 * ---- bf16 activation storage (training plans; BASELINE.json configs[2..4], reference loop train.py:262-287, --prec args.py:17-28) ----
 * Twins of the entry points above for plans that keep activations and activation gradients in HBM as bf16 (NHWC, strides and
 * offsets in ELEMENTS, slices on multiples of 4 elements): same arguments, every activation / gradient pointer addresses bf16
 * elements, all arithmetic and every statistic / table / parameter gradient stays fp32 (fp64 partial sums as in the fp32 forms),
 * stores round to nearest even.  Descriptor-based convolutions (egne_conv2d_fwd, egne_conv3x3_smallcin_fwd, egne_conv2d_wgrad)
 * and the loss head take the storage type from egne_conv_desc.dtype / egne_loss_desc.dtype instead of a twin.
 */
int egne_norm_stats_bf16(const void* x, int64_t pix_stride, int ch_off, int Cp, int B, int HW, int per_sample, float eps,
                         float* scale, float* shift, float* mean_out, float* var_out, void* ws, void* stream);
int egne_affine_bf16(const void* x, int64_t xs, int xo, void* y, int64_t ys, int yo, int Cp, int64_t npix, const float* scale,
                     const float* shift, void* stream);
int egne_avgpool2_bf16(const void* x, int64_t xs, int xo, void* y, int64_t ys, int yo, int B, int H, int W, int Cp, void* stream);
int egne_norm_act_pool2_bf16(const void* x, int64_t xs, int xo, const float* scale, const float* shift, int act, void* y,
                             int64_t ys, int yo, int B, int H, int W, int Cp, void* stream);
int egne_upsample2x_bf16(const void* x, int64_t xs, int xo, void* y, int64_t ys, int yo, int B, int H, int W, int Cp, void* stream);
int egne_nchw_to_nhwc_bf16(const float* x, int B, int C, int H, int W, void* y, int64_t ys, int yo, int Cp, void* stream); /* fp32 in */
int egne_ellipse_head_act_bf16(void* x, int B, int ld, void* stream);
int egne_selu_inplace_bf16(void* x, int64_t n, void* stream);
int egne_spatial_mean_bf16(const void* x, int64_t pix_stride, int ch_off, int C, int B, int HW, void* out, void* stream); /* bf16 out */
int egne_softmax3_bf16(const void* x, int64_t xs, int xo, void* y, int64_t ys, int yo, int Cp_out, int64_t npix, void* stream);
int egne_adain_bf16(const void* x, int64_t xs, int xo, int C, const void* gamma, const void* beta, int64_t gb_stride, int gb_off,
                    void* y, int64_t ys, int yo, int B, int HW, float eps, void* stream);
int egne_conf_loss_bf16(const void* pred, int ld, const int64_t* gt, int B, int C, int flag, float weight, float* terms, void* stream);
int egne_loss_bwd_bf16(const egne_loss_desc* d, const float* gscale, void* g_logits, int64_t gs, int go, float* g_elOut, void* stream);
int egne_act_bwd_bias_bf16(void* g, int64_t gs, int go, const void* y, int64_t ys, int yo, int act, int Cp, int64_t npix,
                           float* dbias, int C, int accumulate, void* ws, void* stream);
int egne_pair_bias_bwd_bf16(const void* g, int64_t gs, int go, int Cp, int B, int H, int W, const void* act_ws, const float* w,
                            int Cout, int Ca, float* db_b, float* db_a, void* ws, void* stream);
int egne_norm_bwd_bf16(const void* x, int64_t xs, int xo, const float* scale, const float* shift, const float* gamma,
                       const void* gy, int64_t gs, int go, int act_in, int Cp, int B, int HW, int per_sample, void* gx,
                       int64_t gxs, int gxo, float* sums, float* dgamma, float* dbeta, int C, void* ws, void* stream);
int egne_norm_bwd_store_bf16(const void* x, int64_t xs, int xo, const float* scale, const float* shift, const float* gamma,
                             const void* gy, int64_t gs, int go, int act_in, int Cp, int B, int HW, int per_sample, void* gx,
                             int64_t gxs, int gxo, float* sums, float* dgamma, float* dbeta, int C, void* ws, void* stream);
int egne_norm_pool2_bwd_bf16(const void* x, int64_t xs, int xo, const float* scale, const float* shift, const void* gzp,
                             int64_t gs, int go, int act_in, int Cp, int B, int H, int W, void* gx, int64_t gxs, int gxo,
                             int accumulate, float* sums, void* ws, void* stream);
int egne_avgpool2_bwd_bf16(const void* gy, int64_t gs, int go, void* gx, int64_t xs, int xo, int B, int H, int W, int Cp, void* stream);
int egne_upsample2x_bwd_bf16(const void* gy, int64_t gs, int go, void* gx, int64_t xs, int xo, int B, int H, int W, int Cp, void* stream);
/* egne_upsample2x_bwd with gx STORED instead of accumulated onto (round 5: the first writer of a gradient slice) */
int egne_upsample2x_bwd_store(const float* gy, int64_t gs, int go, float* gx, int64_t xs, int xo, int B, int H, int W, int Cp, void* stream);
int egne_upsample2x_bwd_store_bf16(const void* gy, int64_t gs, int go, void* gx, int64_t xs, int xo, int B, int H, int W, int Cp, void* stream);
int egne_ellipse_head_act_bwd_bf16(void* g, const void* y, int B, int ld, void* stream);
int egne_selu_bwd_bf16(void* g, const void* y, int64_t n, void* stream);
int egne_softmax3_bwd_bf16(const void* y, int64_t ys, int yo, const void* gy, int64_t gs, int go, void* gx, int64_t xs, int xo,
                           int64_t npix, void* stream);
int egne_adain_bwd_bf16(const void* x, int64_t xs, int xo, int C, const void* gamma, int64_t gb_stride, int gb_off, const void* gy,
                        int64_t gys, int gyo, void* gx, int64_t gxs, int gxo, void* ggamma, void* gbeta, int64_t gg_stride,
                        int gg_off, int B, int HW, float eps, void* stream);
int egne_reflect_pad_bwd_bf16(const void* gpad, int64_t gs, int go, int phase, int Cp, void* gx, int64_t xs, int xo, int B, int H,
                              int W, int P, void* stream);
int egne_spatial_mean_bwd_bf16(const void* g, int gld, void* gx, int64_t xs, int xo, int C, int B, int HW, void* stream);
int egne_conf_loss_bwd_bf16(const void* pred, int ld, const int64_t* gt, int B, int C, int flag, const float* gscale, void* gpred,
                            int gld, void* stream);

/*
 * 3x3 / stride 1 / pad 1 / dilation 1 convolution over ONE bf16 input slice on v_mfma_f32_16x16x32_bf16 (fp32 accumulate): the
 * 3x3 convolutions of models/RITnet_v2.py:57-62,85-87 and utils.py:1047-1048 and their data gradients (a 3x3 over gz with
 * flipped / transposed weights) in training plans with bf16 storage (egne_conv_desc.dtype must be 1).  Descriptor as for
 * egne_conv3x3_rw_f16_fwd: optional fused per-(n,c) affine + activation on load, bias, activation, residual (bf16, accumulated
 * onto in fp32), Ktot = slice width rounded up to 32, CoutP a multiple of 32 (<= 256), Cout_store a multiple of 8.  wfrag: the
 * weights rounded to bf16 in MFMA-fragment order [tap][Ktot/16][CoutP/32][lane][8] (egne_pack_conv_weight_bf16frag; fp32 master
 * weights stay with the caller).
 */
int egne_pack_conv_weight_bf16frag(const float* w_oihw, int Cout, int Cin, int kh, int kw, int CoutP, int Ktot, void* wfrag,
                                   void* stream);
/* ... and the fragments of its DATA GRADIENT w.r.t. input channels [c0, c0 + cn) of the forward weight w_oihw [Cout][Cin][kh][kw]: an
 * ordinary convolution over gz with W'[ci][co][j][i] = w[co][c0 + ci][kh-1-j][kw-1-i] (CoutP >= cn rows, Ktot >= Cout columns). */
int egne_pack_conv_weight_bf16frag_dgrad(const float* w_oihw, int Cout, int Cin, int kh, int kw, int c0, int cn, int CoutP, int Ktot,
                                         void* wfrag, void* stream);
int egne_conv3x3_bf16_fwd(const egne_conv_desc* d, const void* wfrag, void* stream);

/*
 * k x k (k = 3, 5, 7) / stride 1 / zero-padded convolution with a NARROW output (Cout_store <= 8) over ONE bf16 slice of 32 or 64
 * channels, LDS-resident halo, v_mfma_f32_16x16x32_bf16 (egne_conv_desc.dtype must be 1; d.w = the fp32 flat pack
 * [tap][CoutP][Ktot] of egne_pack_conv_weight, rounded to bf16 inside).  Replaces egne_conv2d_fwd for the data gradient of the
 * StyleEncoder's reflect-padded 7x7 (models/RITnet_v2.py:95 under train.py:285-286: 64 channels back to the 3 softmax channels
 * over 49 taps), which the implicit GEMM ran at 22 TFLOP/s.  egne_conv_narrow_bf16_supported tells whether a descriptor qualifies.
 */
int egne_conv_narrow_bf16_supported(const egne_conv_desc* d);
int egne_conv_narrow_bf16_fwd(const egne_conv_desc* d, void* stream);

/*
 * 3x3 / stride 1 / pad 1 convolution with a NARROW output (<= 4 channels; Cout_store <= 8, stored channels past the fourth are
 * written as zeros: the padding of an 8-channel slice) over ONE raw fp32 slice of 4..64 channels, EXACT fp32 on
 * the vector ALU (one rounding per fused multiply-add, accumulation tap-major then channel; bias, optional ReLU / LeakyReLU, post affine): the
 * logits layer of ESF-Net (models/RITnet_v2.py:249 `final` / utils.py:1047 convBlock conv2: 32 -> 3 classes at 240x320), which the
 * matrix kernels ran as a 32-wide output block.  d.w = [9 taps][CP][4 outputs] fp32 from egne_pack_conv3x3_narrow_weight (CP = the
 * slice's channels rounded up to 32; 64-byte aligned; read through the scalar cache; d.Ktot / d.CoutP are ignored).  Post affine
 * allowed; no residual / statistics / fused input affine; egne_conv3x3_narrow_supported tells whether a descriptor qualifies.
 * Replaces egne_conv3x3_rw_f16_fwd for that layer in inference plans (egne_conv2d_fwd semantics).
 */
int egne_pack_conv3x3_narrow_weight(const float* w_oihw, int Cout, int Cin, float* out, void* stream);
int egne_conv3x3_narrow_supported(const egne_conv_desc* d);
int egne_conv3x3_narrow_fwd(const egne_conv_desc* d, void* stream);

/*
 * 1x1 convolution over up to EGNE_MAXSEG RAW bf16 slices on v_mfma_f32_32x32x16_bf16 (fp32 accumulate, bf16 output): conv21 / conv31
 * of the dense blocks, Transition_down behind its pooling, conv11 / conv21 of the up blocks (models/RITnet_v2.py:38-41,59-61,85-86)
 * and their merged data gradients in training plans with bf16 storage (dtype 1).  Streaming: a lane's operand of a 16-channel
 * k-step is one 16-byte load of 8 channels of its pixel, the weights (bf16 fragments [k-step][CoutP/32][lane][8], every slice
 * padded to whole k-steps; egne_pack_conv1x1_bf16 builds them from the fp32 pack [CoutP][Ktot] of the same descriptor) stay in
 * LDS.  Bias, activation and an accumulated bf16 residual as for egne_conv2d_fwd; no fused affine, no post affine.
 * egne_conv1x1_bf16_pack_elems: bf16 elements of the fragment pack (-1: too many k-steps).
 */
int64_t egne_conv1x1_bf16_pack_elems(const egne_conv_desc* d);
int egne_pack_conv1x1_bf16(const egne_conv_desc* d, const float* wflat, const int32_t* seginfo, void* wfrag, void* stream);
/* Several 1x1 convolutions over the SAME bf16 input slices in one launch (round 5): the per-member data gradients of a 1x1 over a
 * would-be torch.cat (models/RITnet_v2.py:59-61,85-86; replaces one egne_conv1x1_bf16_fwd per member, each re-reading gz).
 * A destination may be the LAST writer of its gradient slice: then it applies the activation mask of the layer whose output the
 * slice is the gradient of (mask_y, act: gz = g * act'(y), the pass egne_act_bwd_bias would make); ONE destination per launch (at most
 * 128 channels) may also leave the channel sums of its stored values for that layer's bias gradient (sums: one row [C] per wave,
 * egne_conv1x1_bf16_multi_waves rows, added by egne_group_sums_reduce in a fixed order). */
#define EGNE_MAXDST 6
typedef struct {
  void* out; int64_t out_pix_stride; int32_t out_ch_off;
  int32_t C;                    /* channels stored (multiple of 8) */
  int32_t CoutP;                /* rows of this destination's weight pack (multiple of 32) */
  const void* wfrag;            /* egne_pack_conv1x1_bf16 fragments for (the input slices, CoutP) */
  const void* residual; int64_t res_pix_stride; int32_t res_ch_off;      /* optional accumulated tensor (bf16) */
  const void* mask_y; int64_t mask_pix_stride; int32_t mask_ch_off;      /* optional: activated output whose sign masks the result */
  int32_t act;                  /* egne_act of that layer */
  float* sums;                  /* optional [egne_conv1x1_bf16_multi_waves][C] */
  int64_t res_pixels;           /* 0: the residual covers every pixel; k > 0: only the first k pixels accumulate (the rest of the slice has
                                 * no earlier writer: the decoder's skip gradients reach the image half of the encoder's batch only) */
} egne_dst;
int egne_conv1x1_bf16_multi_supported(const egne_conv_desc* d, int ndst, const egne_dst* dsts);
int egne_conv1x1_bf16_multi_fwd(const egne_conv_desc* d, int ndst, const egne_dst* dsts, void* stream);
int64_t egne_conv1x1_bf16_multi_waves(const egne_conv_desc* d, int ndst, const egne_dst* dsts);   /* rows of egne_dst.sums the launch writes */
int egne_group_sums_reduce(const float* sums, int64_t nrows, int ld, int C, float* out /* [C], may be NULL */, double* total /* [C], may be NULL */,
                           int accumulate, void* stream);
int egne_conv1x1_bf16_fwd(const egne_conv_desc* d, const void* wfrag, void* stream);

/* ----------------------------------------------------------------------------------------------------------------------------
 * Which forward entry point serves a convolution (round 6, csrc/dispatch.hip).  The reference leaves the choice of a kernel to ATen
 * (F.conv2d at models/RITnet_v2.py:57-62,85-87, bdcn_new.py:49-55, vgg16_c.py:65-88); here the engine's planner makes it
 * (engine.Plan._conv_impl / _conv_bf16) and this function restates its predicates, in its order, with its default thresholds, for
 * callers that bind this header directly: describe the layer, get the entry point (egne_conv_kind + the `kind` string the planner
 * records in Plan.meta) and what its epilogue takes along (statistics, pooled output, frames handed to the flat kernel behind the
 * deep trunk kernel, workspace of the small-problem form).  The planner checks every convolution it plans against this function
 * under EGNE_CHECK_DISPATCH=1.  Pair fusions (1x1 into the 3x3 that consumes it, Transition_down with its pooling) are decided
 * by the plan builder before a layer gets here. */
typedef enum {
  EGNE_KIND_IGEMM = 0,            /* egne_conv2d_fwd (exact fp32 implicit GEMM; bf16 tensors too) */
  EGNE_KIND_HALO_F32 = 1,         /* egne_conv3x3_halo_fwd */
  EGNE_KIND_SMALLCIN = 2,         /* egne_conv3x3_smallcin_fwd */
  EGNE_KIND_NARROW_F32 = 3,       /* egne_conv3x3_narrow_fwd */
  EGNE_KIND_F16X3_FLAT = 4,       /* egne_conv2d_f16x3_fwd */
  EGNE_KIND_F16X3_SMALL = 5,      /* egne_conv2d_f16x3_small_fwd */
  EGNE_KIND_F16X3_BIG = 6,        /* egne_conv2d_f16x3_big_fwd (+ egne_conv2d_f16x3_fwd for tail_frames); with f16_products = 1 and egne_conv2d_f16_big1_supported(): egne_conv2d_f16_big1_fwd */
  EGNE_KIND_F16X3_HALO = 7,       /* egne_conv3x3_halo_f16_fwd */
  EGNE_KIND_F16X3_RS = 8,         /* egne_conv3x3_rs_f16_fwd */
  EGNE_KIND_F16X3_RW = 9,         /* egne_conv3x3_rw_f16_fwd */
  EGNE_KIND_F16X3_LATTICE = 10,   /* three egne_conv3x3_halo_f16_fwd launches on dilation lattices */
  EGNE_KIND_F16X3_MSDIL = 11,     /* egne_msblock_dil(_scores)_f16_fwd */
  EGNE_KIND_F16X3_STREAM1X1 = 12, /* egne_conv1x1_f16x3_fwd */
  EGNE_KIND_F16X3_GEMM1X1 = 13,   /* egne_conv1x1_ms_f16x3_fwd */
  EGNE_KIND_F16X3_FIRST = 14,     /* egne_conv3x3_smallcin_f16_fwd */
  EGNE_KIND_BF16_3X3 = 15,        /* egne_conv3x3_bf16_fwd */
  EGNE_KIND_BF16_1X1 = 16,        /* egne_conv1x1_bf16_fwd */
  EGNE_KIND_BF16_NARROW = 17      /* egne_conv_narrow_bf16_fwd */
} egne_conv_kind;

typedef struct {
  int32_t dtype;                  /* storage of the activation tensors: 0 fp32, 1 bf16 (egne_conv_desc.dtype) */
  int32_t B, H, W;                /* frames and INPUT map */
  int32_t kh, kw, stride, pad_h, pad_w, pad_mode, ngroups;     /* padding in TAPS, as in egne_conv_desc (a dilated "same" 3x3: 1) */
  int32_t dil[EGNE_MAXGROUP];
  int32_t nseg;                   /* input slices in concat order */
  int32_t seg_C[EGNE_MAXSEG];     /* logical channels */
  int32_t seg_Cp[EGNE_MAXSEG];    /* stored channels (padded to 8) */
  int32_t seg_ch_off[EGNE_MAXSEG];
  int64_t seg_pix_stride[EGNE_MAXSEG];
  int32_t seg_affine[EGNE_MAXSEG];/* 1: per-(n, c) affine (+ activation) applied on load (egne_seg.scale) */
  int32_t seg_planar;             /* the one slice is an NCHW tensor read in place (first layer) */
  int32_t seg_presplit;           /* the one slice is in split-pair storage (egne_seg.presplit) */
  int32_t Cout;                   /* logical output channels of ONE group */
  int32_t Cout_store;             /* channels the layer stores (0: Cout rounded up to 8) */
  int32_t dst_Cp, dst_ch_off; int64_t dst_pix_stride;
  int32_t act, has_post, has_residual; int64_t res_pix_stride; int32_t res_ch_off;
  int32_t split;                  /* k x k products may be split-f16 (22-bit significand; ConvLayer.split) */
  int32_t split1;                 /* the same for a 1x1 over raw slices (ConvLayer.split1) */
  int32_t split_c4;               /* ... for a first layer on <= 4 channels (ConvLayer.split_c4) */
  int32_t train;                  /* the plan records a backward pass */
  int32_t dyn_scales;             /* split-f16 pre-scales are taken on the device (training plans with fp32 storage) */
  int32_t is_dgrad;               /* the layer is a data gradient (derived weights) */
  int32_t want_stats;             /* the consumer needs InstanceNorm statistics of the output */
  int32_t want_pool; int32_t pool_Cp; int64_t pool_pix_stride;   /* a 2x2 ceil-mode max pooling of the result is wanted */
  int32_t want_scores;            /* MSBlock: the block's share of the stage's score maps instead of its output */
  int32_t up_add;                 /* a half-resolution addend is up-sampled into the result (streaming 1x1) */
  int32_t narrow_bf16_ok;         /* dtype 1: egne_conv_narrow_bf16_supported said yes on the finished descriptor */
  int32_t f16_products;           /* egne_conv_desc.f16_products of the plan's split-f16 launches (1: plain f16 operands -- the role-split / streamed-weights
                                   * 3x3 forms then take maps from 30 pixels of width on) */
  int32_t f16_storage;            /* the input slice or the destination is held as f16 (egne_seg.presplit = 2 / egne_conv_desc.out_split = 2) */
} egne_conv_query;

typedef struct {
  int32_t kind;                   /* egne_conv_kind */
  char name[32];                  /* the planner's name for it ("conv_f16x3:rw", "conv_bf16:3x3", ...) */
  int32_t tail_frames;            /* EGNE_KIND_F16X3_BIG: trailing frames that go to egne_conv2d_f16x3_fwd (whole rounds of 256 workgroups for the rest) */
  int32_t fused_stats;            /* the launch writes the statistics partial sums (egne_conv_desc.stats_ws) */
  int32_t fused_pool;             /* ... and the pooled output (egne_conv_desc.pool_out) */
  int64_t small_ws_floats;        /* EGNE_KIND_F16X3_SMALL: workspace of egne_conv2d_f16x3_small_fwd */
} egne_conv_choice;

int egne_conv2d_auto_kind(const egne_conv_query* q, egne_conv_choice* out);

const char* egne_last_error(void);
int egne_version(void);
int egne_sizeof(int which);   /* 0 egne_conv_desc, 1 egne_loss_desc, 2 egne_bdcn_tail_desc, 3 egne_dst, 4 egne_conv_query, 5 egne_conv_choice */

#ifdef __cplusplus
}
#endif
#endif
