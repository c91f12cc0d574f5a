"""ctypes binding of the C-ABI in ``include/egne_hip.h`` (``csrc/libegne_hip.so``).

There is no fallback: :func:`lib` raises ``RuntimeError`` when the shared library has not been
built (``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C .../csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (EGNE_LIB: another build of the same sources, for A/B timing of a kernel change inside one gpurun call -- never a different backend)
LIB_PATH = os.environ.get("EGNE_LIB") or os.path.join(_HERE, "csrc", "libegne_hip.so")

MAXSEG, MAXGROUP = 8, 3
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2

c_fp = C.c_void_p  # device pointers travel as integers


class Seg(C.Structure):
    _fields_ = [("ptr", c_fp), ("pix_stride", C.c_int64), ("ch_off", C.c_int32), ("Cp", C.c_int32),
                ("scale", c_fp), ("shift", c_fp), ("act_in", C.c_int32), ("presplit", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32),
                ("pad_h", C.c_int32), ("pad_w", C.c_int32), ("pad_mode", C.c_int32),
                ("ngroups", C.c_int32), ("dil", C.c_int32 * MAXGROUP),
                ("nseg", C.c_int32), ("seg", Seg * MAXSEG),
                ("Ktot", C.c_int32), ("CoutP", C.c_int32),
                ("w", c_fp), ("bias", c_fp), ("act", C.c_int32),
                ("post_scale", c_fp), ("post_shift", c_fp),
                ("residual", c_fp), ("res_pix_stride", C.c_int64), ("res_ch_off", C.c_int32),
                ("out", c_fp), ("out_pix_stride", C.c_int64), ("out_ch_off", C.c_int32),
                ("Cout_store", C.c_int32),
                ("stats_ws", c_fp), ("stats_nchunk", C.c_int32),
                ("pool_out", c_fp), ("pool_pix_stride", C.c_int64), ("pool_ch_off", C.c_int32),
                ("dyn_scale", C.c_void_p), ("absmax_out", C.c_void_p), ("dtype", C.c_int32),
                ("out_split", C.c_int32), ("out_split_scale", C.c_float), ("ovf_flag", C.c_void_p), ("f16_products", C.c_int32),
                ("mask_y", C.c_void_p), ("mask_pix_stride", C.c_int64), ("mask_ch_off", C.c_int32), ("mask_act", C.c_int32), ("mask_sums", C.c_void_p)]


class Dst(C.Structure):
    """egne_dst: one destination of egne_conv1x1_bf16_multi_fwd."""
    _fields_ = [("out", C.c_void_p), ("out_pix_stride", C.c_int64), ("out_ch_off", C.c_int32), ("C", C.c_int32), ("CoutP", C.c_int32),
                ("wfrag", C.c_void_p), ("residual", C.c_void_p), ("res_pix_stride", C.c_int64), ("res_ch_off", C.c_int32),
                ("mask_y", C.c_void_p), ("mask_pix_stride", C.c_int64), ("mask_ch_off", C.c_int32), ("act", C.c_int32),
                ("sums", C.c_void_p), ("res_pixels", C.c_int64)]


MAXDST = 6


class ConvQuery(C.Structure):
    """egne_conv_query: a layer described to egne_conv2d_auto_kind (csrc/dispatch.hip)."""
    _fields_ = [("dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad_h", C.c_int32), ("pad_w", C.c_int32), ("pad_mode", C.c_int32),
                ("ngroups", C.c_int32), ("dil", C.c_int32 * MAXGROUP), ("nseg", C.c_int32),
                ("seg_C", C.c_int32 * MAXSEG), ("seg_Cp", C.c_int32 * MAXSEG), ("seg_ch_off", C.c_int32 * MAXSEG), ("seg_pix_stride", C.c_int64 * MAXSEG),
                ("seg_affine", C.c_int32 * MAXSEG), ("seg_planar", C.c_int32), ("seg_presplit", C.c_int32),
                ("Cout", C.c_int32), ("Cout_store", C.c_int32), ("dst_Cp", C.c_int32), ("dst_ch_off", C.c_int32), ("dst_pix_stride", C.c_int64),
                ("act", C.c_int32), ("has_post", C.c_int32), ("has_residual", C.c_int32), ("res_pix_stride", C.c_int64), ("res_ch_off", C.c_int32),
                ("split", C.c_int32), ("split1", C.c_int32), ("split_c4", C.c_int32), ("train", C.c_int32), ("dyn_scales", C.c_int32),
                ("is_dgrad", C.c_int32), ("want_stats", C.c_int32), ("want_pool", C.c_int32), ("pool_Cp", C.c_int32), ("pool_pix_stride", C.c_int64),
                ("want_scores", C.c_int32), ("up_add", C.c_int32), ("narrow_bf16_ok", C.c_int32), ("f16_products", C.c_int32), ("f16_storage", C.c_int32)]


class ConvChoice(C.Structure):
    """egne_conv_choice: the entry point egne_conv2d_auto_kind picked and what its epilogue takes along."""
    _fields_ = [("kind", C.c_int32), ("name", C.c_char * 32), ("tail_frames", C.c_int32), ("fused_stats", C.c_int32), ("fused_pool", C.c_int32),
                ("small_ws_floats", C.c_int64)]


class BdcnTailDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("s", c_fp * 5), ("s1", c_fp * 5), ("h", C.c_int32 * 5), ("w", C.c_int32 * 5),
                ("stride", C.c_int32 * 5), ("crop", C.c_int32 * 5), ("up", c_fp * 5),
                ("fuse_w", c_fp), ("fuse_b", c_fp), ("out", c_fp * 11), ("edge_thres", C.c_int32)]


class LossDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("logits", c_fp), ("pix_stride", C.c_int64), ("ch_off", C.c_int32),
                ("target", c_fp), ("spatWts", c_fp), ("distMap", c_fp), ("cond", c_fp),
                ("pupil_center", c_fp), ("elNorm", c_fp), ("elOut", c_fp), ("alpha", C.c_float),
                ("grid_x", c_fp), ("grid_y", c_fp),
                ("partials", c_fp), ("out_terms", c_fp), ("pred_c", c_fp), ("elPred", c_fp),
                ("mask", c_fp), ("op_nchw", c_fp), ("coef", c_fp), ("dtype", C.c_int32),
                ("g_op_nchw", c_fp), ("g_pred_c", c_fp), ("g_elOut_up", c_fp)]


# name -> (restype, argtypes); every symbol include/egne_hip.h declares
i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
SIGNATURES = {
    "egne_conv2d_fwd": (i32, [C.POINTER(ConvDesc), vp]),
    "egne_pack_conv_weight": (i32, [vp, i32, i32, i32, i32, vp, i32, i32, vp, vp]),
    "egne_pack_conv_weight_frag": (i32, [vp, i32, i32, i32, i32, vp, i32, i32, vp, vp]),
    "egne_conv3x3_halo_supported": (i32, [C.POINTER(ConvDesc)]),
    "egne_conv3x3_halo_fwd": (i32, [C.POINTER(ConvDesc), vp]),
    "egne_conv3x3_smallcin_fwd": (i32, [C.POINTER(ConvDesc), vp, vp]),
    "egne_pack_conv_weight_f16x2": (i32, [vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp]),
    "egne_conv2d_f16x3_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, f32, f32, vp]),
    "egne_conv2d_f16x3_small_workspace_floats": (i64, [C.POINTER(ConvDesc)]),
    "egne_conv2d_f16x3_small_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, f32, f32, vp, i64, vp]),
    "egne_pack_conv_weight_f16frag": (i32, [vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp]),
    "egne_conv3x3_halo_f16_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, f32, f32, vp]),
    "egne_conv1x1_3x3_fused_f16_fwd": (i32, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), vp, vp, f32, f32, vp, vp, f32, f32, vp]),
    "egne_conv3x3c4_3x3_fused_f16_fwd": (i32, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), vp, vp, f32, f32, vp, vp, f32, f32, vp]),
    "egne_msblock_dil_scores_f16_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, f32, f32, vp, vp, vp, vp, i32, vp]),
    "egne_msblock_dil_f16_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, f32, f32, vp]),
    "egne_conv3x3_rs_f16_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, f32, f32, vp]),
    "egne_conv3x3_rw_f16_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, f32, f32, vp]),
    "egne_conv1x1_pool2_f16x3_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, f32, f32, vp]),
    "egne_conv1x1_f16x3_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, f32, f32, vp]),
    "egne_dist_maps_workspace_bytes": (i64, [i32, i32, i32, i32]),
    "egne_dist_maps": (i32, [vp, i32, i32, i32, i32, vp, vp, vp]),
    "egne_zscore": (i32, [vp, vp, i32, i32, vp]),
    "egne_spatial_weights": (i32, [vp, i32, i32, i32, vp, vp]),
    "egne_deepvog_loss_workspace_floats": (i64, [i32, i32, i32]),
    "egne_deepvog_loss_fwd": (i32, [vp, i64, i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "egne_deepvog_loss_bwd": (i32, [vp, i64, i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, i64, i32, vp]),
    "egne_affine_act": (i32, [vp, i64, i32, vp, i64, i32, i32, i64, vp, vp, i32, vp]),
    "egne_augment": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "egne_conv3x3_smallcin_f16_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, f32, f32, vp]),
    "egne_pack_conv3x3_c4_weight_f16": (i32, [vp, i32, i32, i32, f32, vp, vp, vp]),
    "egne_conv1x1_ms_f16x3_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, f32, f32, vp]),
    "egne_pack_conv1x1_weight_f16x2_map": (i32, [vp, i32, i32, vp, i32, i32, f32, vp, vp, vp]),
    "egne_conv2d_f16x3_big_fwd": (i32, [C.POINTER(ConvDesc), vp, f32, f32, vp]),
    "egne_pack_conv_weight_f16img": (i32, [vp, i32, i32, i32, i32, i32, i32, f32, vp, vp]),
    "egne_conv2d_f16_big1_fwd": (i32, [C.POINTER(ConvDesc), vp, f32, f32, vp]),
    "egne_conv2d_f16_big1_supported": (i32, [C.POINTER(ConvDesc)]),
    "egne_pack_conv_weight_f16img1": (i32, [vp, i32, i32, i32, i32, i32, i32, f32, vp, vp]),
    "egne_pack_conv1x1_weight_f16": (i32, [vp, i32, i32, vp, i32, i32, f32, vp, vp, vp]),
    "egne_absmax": (i32, [vp, i64, i32, i32, i64, vp, vp]),
    "egne_absmax_f16": (i32, [vp, i64, i32, i32, i64, vp, vp]),
    "egne_norm_stats_workspace_bytes": (i64, [i32, i32, i32, i32]),
    "egne_norm_stats": (i32, [vp, i64, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp]),
    "egne_norm_stats_finish": (i32, [vp, i32, i32, i32, i32, f32, vp, vp, vp]),
    "egne_norm_stats_finish_moments": (i32, [vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp]),
    "egne_affine_inplace": (i32, [vp, i64, i32, i32, i64, vp, vp, vp]),
    "egne_affine": (i32, [vp, i64, i32, vp, i64, i32, i32, i64, vp, vp, vp]),
    "egne_avgpool2": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, vp]),
    "egne_norm_act_pool2": (i32, [vp, i64, i32, vp, vp, i32, vp, i64, i32, i32, i32, i32, i32, vp]),
    "egne_maxpool2": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "egne_maxpool2_f16": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "egne_upsample2x": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, vp]),
    "egne_upsample2x_nearest": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, vp]),
    "egne_upsample2x_nearest_bwd": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, vp]),
    "egne_nchw_to_nhwc": (i32, [vp, i32, i32, i32, i32, vp, i64, i32, i32, vp]),
    "egne_nhwc_to_nchw": (i32, [vp, i64, i32, i32, i32, i32, i32, vp, vp]),
    "egne_bdcn_stage_scores": (i32, [C.POINTER(vp), i32, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "egne_bdcn_tail": (i32, [C.POINTER(BdcnTailDesc), vp]),
    "egne_loss_workspace_floats": (i64, [i32, i32, i32]),
    "egne_loss_fwd": (i32, [C.POINTER(LossDesc), vp]),
    "egne_ellipse_head_act": (i32, [vp, i32, i32, vp]),
    "egne_selu_inplace": (i32, [vp, i64, vp]),
    "egne_spatial_mean": (i32, [vp, i64, i32, i32, i32, i32, vp, vp]),
    "egne_softmax3": (i32, [vp, i64, i32, vp, i64, i32, i32, i64, vp]),
    "egne_adain": (i32, [vp, i64, i32, i32, vp, vp, i64, i32, vp, i64, i32, i32, i32, f32, vp]),
    "egne_conf_loss": (i32, [vp, i32, vp, i32, i32, i32, f32, vp, vp]),
    "egne_loss_bwd": (i32, [C.POINTER(LossDesc), vp, vp, i64, i32, vp, vp]),
    "egne_act_bwd_bias_workspace_bytes": (i64, [i64, i32]),
    "egne_act_bwd_bias": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i64, vp, i32, i32, vp, vp]),
    "egne_act_bwd_bias_absmax": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i64, vp, i32, i32, vp, vp, vp]),
    "egne_act_norm_bwd": (i32, [vp, i64, i32, vp, i64, i32, i32, vp, vp, vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, i32, vp]),
    "egne_bn_act_bwd": (i32, [vp, i64, i32, i32, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp, i64, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp]),
    "egne_zero_many": (i32, [vp, i32, i64, vp]),
    "egne_pair_bias_bwd_workspace_bytes": (i64, [i32, i32]),
    "egne_pair_bias_bwd": (i32, [vp, i64, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp]),
    "egne_norm_bwd_workspace_bytes": (i64, [i32, i32, i32, i32]),
    "egne_norm_pool2_bwd": (i32, [vp, i64, i32, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp, i64, i32, i32, vp, vp, vp]),
    "egne_norm_bwd_store": (i32, [vp, i64, i32, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp, i64, i32, vp, vp, vp,
                                  i32, vp, vp]),
    "egne_norm_bwd": (i32, [vp, i64, i32, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp, i64, i32, vp, vp, vp,
                            i32, vp, vp]),
    "egne_avgpool2_bwd": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, vp]),
    "egne_upsample2x_bwd": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, vp]),
    "egne_upsample2x_bwd_store": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, vp]),
    "egne_ellipse_head_act_bwd": (i32, [vp, vp, i32, i32, vp]),
    "egne_selu_bwd": (i32, [vp, vp, i64, vp]),
    "egne_softmax3_bwd": (i32, [vp, i64, i32, vp, i64, i32, vp, i64, i32, i64, vp]),
    "egne_adain_bwd": (i32, [vp, i64, i32, i32, vp, i64, i32, vp, i64, i32, vp, i64, i32, vp, vp, i64, i32, i32, i32, f32, vp]),
    "egne_reflect_pad_bwd": (i32, [vp, i64, i32, i32, i32, vp, i64, i32, i32, i32, i32, i32, vp]),
    "egne_spatial_mean_bwd": (i32, [vp, i32, vp, i64, i32, i32, i32, i32, vp]),
    "egne_conf_loss_bwd": (i32, [vp, i32, vp, i32, i32, i32, vp, vp, i32, vp]),
    "egne_conv2d_wgrad_splits": (i32, [C.POINTER(ConvDesc)]),
    "egne_conv2d_wgrad_workspace_bytes": (i64, [C.POINTER(ConvDesc)]),
    "egne_conv2d_wgrad": (i32, [C.POINTER(ConvDesc), vp, i64, i32, i32, i32, vp, C.POINTER(vp), vp, vp]),
    "egne_conv2d_wgrad_f16": (i32, [C.POINTER(ConvDesc), vp, i64, i32, vp, i32, i32, vp, C.POINTER(vp), vp, vp]),
    "egne_pack_conv_weight_dgrad": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "egne_ellipse_fit": (i32, [vp, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "egne_ellipse_init_from_pred": (i32, [vp, i32, i32, i32, vp, vp, vp, vp]),
    "egne_last_error": (C.c_char_p, []),
    "egne_version": (i32, []),
    "egne_sizeof": (i32, [i32]),
    "egne_conv2d_auto_kind": (i32, [C.POINTER(ConvQuery), C.POINTER(ConvChoice)]),
}

# Entry points with a bf16-storage twin (same arguments, activation pointers are bf16; suffix _bf16): training plans that keep
# activations and activation gradients as bf16 in HBM (engine.Plan(dtype=torch.bfloat16)).  Descriptor-based convolutions carry
# the storage type in egne_conv_desc.dtype instead.
BF16_TWINS = ["egne_upsample2x_nearest", "egne_upsample2x_nearest_bwd", "egne_norm_stats", "egne_affine", "egne_avgpool2", "egne_norm_act_pool2", "egne_upsample2x", "egne_nchw_to_nhwc",
              "egne_ellipse_head_act", "egne_selu_inplace", "egne_spatial_mean", "egne_softmax3", "egne_adain", "egne_conf_loss",
              "egne_loss_bwd", "egne_act_bwd_bias", "egne_act_norm_bwd", "egne_bn_act_bwd", "egne_pair_bias_bwd", "egne_norm_pool2_bwd", "egne_norm_bwd_store", "egne_norm_bwd",
              "egne_avgpool2_bwd", "egne_upsample2x_bwd", "egne_upsample2x_bwd_store", "egne_ellipse_head_act_bwd", "egne_selu_bwd", "egne_softmax3_bwd",
              "egne_adain_bwd", "egne_reflect_pad_bwd", "egne_spatial_mean_bwd", "egne_conf_loss_bwd"]
for _n in BF16_TWINS:
    SIGNATURES[_n + "_bf16"] = SIGNATURES[_n]
SIGNATURES.update({
    "egne_pack_conv_weight_bf16frag": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "egne_pack_conv_weight_bf16frag_dgrad": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "egne_conv3x3_bf16_fwd": (i32, [C.POINTER(ConvDesc), vp, vp]),
    "egne_conv3x3_bf16_sum_rows": (i32, []),
    "egne_conv_narrow_bf16_supported": (i32, [C.POINTER(ConvDesc)]),
    "egne_conv_narrow_bf16_fwd": (i32, [C.POINTER(ConvDesc), vp]),
    "egne_pack_conv3x3_narrow_weight": (i32, [vp, i32, i32, vp, vp]),
    "egne_conv3x3_narrow_supported": (i32, [C.POINTER(ConvDesc)]),
    "egne_conv3x3_narrow_fwd": (i32, [C.POINTER(ConvDesc), vp]),
    "egne_conv1x1_bf16_pack_elems": (i64, [C.POINTER(ConvDesc)]),
    "egne_pack_conv1x1_bf16": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp]),
    "egne_conv1x1_bf16_fwd": (i32, [C.POINTER(ConvDesc), vp, vp]),
    "egne_conv1x1_bf16_multi_supported": (i32, [C.POINTER(ConvDesc), i32, C.POINTER(Dst)]),
    "egne_conv1x1_bf16_multi_fwd": (i32, [C.POINTER(ConvDesc), i32, C.POINTER(Dst), vp]),
    "egne_conv1x1_bf16_multi_waves": (i64, [C.POINTER(ConvDesc), i32, C.POINTER(Dst)]),
    "egne_group_sums_reduce": (i32, [vp, i64, i32, i32, vp, vp, i32, vp]),
})

_lib = None


def lib():
    """Load the HIP library (once).  Raises RuntimeError if it is missing -- by design."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "HIP extension not built: %s is missing (run __graft_entry__.build()). "
                "This package has no CPU fallback." % LIB_PATH)
        # torch first: its wheel bundles its own libamdhip64 / libhsa-runtime64, and the extension (linked against
        # libamdhip64.so.7) must bind to THAT copy.  Loaded before torch it would pull /opt/rocm's runtime in, and two HIP
        # runtimes in one process do not see each other's device context ("no ROCm-capable device is detected").
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().egne_last_error()
        raise RuntimeError("%s failed (%d): %s" % (what or "egne call", status, (msg or b"").decode()))


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
