"""ctypes binding of libvnqa_hip.so (the C ABI declared in include/vnqa_hip.h).

There is NO fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VNQA_LIB", os.path.join(_HERE, "lib", "libvnqa_hip.so"))
LIB_PATH_F16 = os.path.join(_HERE, "lib", "libvnqa_hip_f16.so")

# The library's 16-bit storage format is a BUILD property (csrc/vnqa_common.h): libvnqa_hip.so stores bf16,
# libvnqa_hip_f16.so IEEE fp16 (same MFMA rate, 8x finer rounding, +-65504 range).  One format per process: it is fixed by
# the first model / stem built (precision 'bf16' | 'fp16' | 'fp16h'; 'fp32' works with either) or by VNQA_HALF=bf16|f16; a process
# that loads the library before anybody asked gets f16, the format of the default precision 'fp16h' (round 6; bf16 until round 5).
_half = os.environ.get("VNQA_HALF")          # None until somebody needs a 16-bit format


def set_half(fmt):
    """Select the process's 16-bit storage format ('bf16' | 'f16').  Raises if the other library is already in use."""
    global _half
    assert fmt in ("bf16", "f16"), fmt
    if _half is not None and _half != fmt and _lib is not None:
        raise VnqaError("this process already runs the %s build of the HIP library; precision '%s' needs the other one — "
                        "one 16-bit storage format per process (use a separate process, or VNQA_HALF)" % (_half, fmt))
    _half = fmt


def half_dtype():
    """torch dtype of the library's 16-bit storage format."""
    return torch.float16 if _half == "f16" else torch.bfloat16


def is_half(dt):
    return dt in (torch.bfloat16, torch.float16)

BF16, F32 = 0, 1
ABI_VERSION = 432          # include/vnqa_hip.h: VNQA_ABI_VERSION (checked against vnqa_version() of the loaded library)
TILE_AUTO, TILE_256x256, TILE_256x128, TILE_256x64, TILE_128x128, TILE_128x64, TILE_STEM_256x256 = range(7)
TILE_256x256_W16 = 13      # include/vnqa_hip.h: VNQA_TILE_256x256_W16
TILE_I5_256x256, TILE_STEM_I5_256x256 = 18, 19     # hand-pipelined main loop (PIPE 5)
TILE_PS_224x256, TILE_STEM_PS_224x256 = 20, 21     # patch-stationary 3x3 conv (csrc/conv_ps.hip)
TILE_P4_256x256, TILE_P4_256x128, TILE_P4_256x64, TILE_256x128_W24 = 7, 8, 9, 10

_vp, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, _i32) for n in ("dtype", "n_img", "h", "w", "c_in", "c_out", "c_y", "taps",
                                    "x_halo", "y_halo", "relu", "pool2", "tile", "wt_tiled", "depth", "flags")]


CONV_ZERO_HALO = 1      # vnqa_conv_desc.flags
CONV_XCD_SPLIT_N = 2
CONV_X_WRAP2 = 4
CONV_DUAL_OUT = 0x20000  # y = [h16(v) | h16(v - h16(v))], 2 c_out channels (patch-stationary tiles)
CONV_DUAL_HI2 = 0x40000  # with CONV_DUAL_OUT: y = [hi | lo | hi], 3 c_out channels (a three-product consumer's operand)
CONV_F32_EPILOGUE = 0x80000      # one plain output, pool / affine in fp32, rounded once
CONV_RELU_FLOOR = 0x200000       # stem 256x256 tile: post_shift (no post_scale) = per-channel ReLU floor (mean-shifted output, one rounding)
CONV_FIRST_MID_SHIFT = 0x100000  # vnqa_conv_first_c64_fwd: b1 = [bias | shift of the first conv's stored output]
LAYOUT_MAX_BATCH = 256     # vnqa_frame_layout: VNQA_LAYOUT_MAX_BATCH
GEMM_X_WRAP2 = 0x400
WGRAD_FUSED_REDUCE = 0x100     # option bit of vnqa_conv2d_wgrad's dtype argument
WGRAD_X_PAIR = 0x200           # x is a [hi | lo] tensor (2 c_in physical channels): contract its hi half
WGRAD_X_TRIPLE = 0x400         # x is a [hi | lo | hi] tensor (3 c_in physical channels): contract its first segment
WGRAD_EIGHT_WAVES = 0x800      # the first 16-bit form of the weight-gradient kernel (cross-check of the default 4-wave ring form)
GEMM_OUT_F32 = 0x200           # option bit of vnqa_gemm_nt's dtype argument: 16-bit operands, fp32 output


def conv_reserve_flags(reserve_cus):
    """VNQA_CONV_RESERVE_CUS(n): bits 8..15 of vnqa_conv_desc.flags — the persistent conv kernels leave n CUs (multiple of 8,
    <= 224) to the other streams.  A per-call argument: the library keeps no process-wide setting."""
    n = max(0, min(224, int(reserve_cus)))
    return (n // 8) << 8


class ConvEpilogue(ctypes.Structure):
    """include/vnqa_hip.h: vnqa_conv_epilogue"""
    _fields_ = [("kind", _i32), ("n_frames", _i32), ("min_frame_images", _i32), ("film_ld", _i32), ("film_c", _i32),
                ("frame_of", _vp), ("frame_off", _vp), ("partial", _vp), ("mean", _vp), ("var", _vp),
                ("gamma", _vp), ("beta", _vp), ("res", _vp), ("y2", _vp)]


EPI_NONE, EPI_BNSTATS, EPI_FILM_RES, EPI_ADD_MASK, EPI_SPLIT_OUT = 0, 1, 2, 3, 4

_MAC_PTRS = ("control memory pq ctxw know pre mask_c wc w_ca b_ca wm bm w1 w_ra b_ra wr wmm bw "
             "cq qv p_c cnew mem v t u p_r read concat d_cnew d_concat d_control d_memory d_cq "
             "ds_r d_read ds_c d_c du dv dqv d_mem d_t g_wc g_wca g_wm g_bm g_w1 g_wra g_wr g_wmm g_bw ones workspace").split()


class MacCore(ctypes.Structure):
    """include/vnqa_hip.h: vnqa_mac_core"""
    _fields_ = ([(n, _i32) for n in ("n", "d", "lq", "s", "ld", "dtype")] + [(n, _vp) for n in _MAC_PTRS]
                + [("defer_wgrad", _i32)])


_MAC_WGRAD_PTRS = ("d_concat read memory v d_t d_mem d_cq control dv cnew dqv cq "
                   "g_wc g_wca g_wm g_bm g_w1 g_wra g_wr g_wmm g_bw workspace").split()


class View5(ctypes.Structure):
    """include/vnqa_hip.h: vnqa_view5 (element strides of an (n, d, h, w, c)-indexed tensor)"""
    _fields_ = [(n, _i64) for n in ("base", "sn", "sd", "sh", "sw", "sc")] + [(n, _i32) for n in ("d", "h", "w")]


class SgemmProblem(ctypes.Structure):
    """include/vnqa_hip.h: vnqa_sgemm_problem"""
    _fields_ = ([(n, _vp) for n in ("a", "b", "c", "bias", "addend", "out2", "out2_col", "out2_mul")] +
                [(n, _i64) for n in ("a_rs", "a_cs", "b_rs", "b_cs")] +
                [(n, _i32) for n in ("ldc", "m", "n", "k", "relu", "accumulate")])


class MacWgrad(ctypes.Structure):
    """include/vnqa_hip.h: vnqa_mac_wgrad"""
    _fields_ = [("rows", _i32), ("d", _i32)] + [(n, _vp) for n in _MAC_WGRAD_PTRS]

_SIGNATURES = {
    "vnqa_version": (ctypes.c_int, []),
    "vnqa_last_error": (ctypes.c_char_p, []),
    "vnqa_stream_create_reserved": (ctypes.c_int, [_i32, ctypes.POINTER(ctypes.c_void_p)]),
    "vnqa_stream_create_masked": (ctypes.c_int, [_vp, _i32, ctypes.POINTER(ctypes.c_void_p)]),
    "vnqa_conv2d_igemm_fwd": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vnqa_split3_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _i64, _vp, _vp]),
    "vnqa_conv2d_c64_fwd": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vnqa_layernorm_fwd": (ctypes.c_int, [_vp] * 7 + [_i32, _i32, _f32, _vp]),
    "vnqa_layernorm_bwd": (ctypes.c_int, [_vp] * 9 + [_i32, _i32, _i32, _vp]),
    "vnqa_scatter_add_rows": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "vnqa_hop_fwd": (ctypes.c_int, [_vp] * 8 + [_i32, _i32, _i32, _vp]),
    "vnqa_hop_bwd": (ctypes.c_int, [_vp] * 10 + [_i32, _i32, _i32, _vp]),
    "vnqa_frame_max_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 7 + [_vp]),
    "vnqa_frame_max_bwd": (ctypes.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 5 + [_f32, _i32, _vp]),
    "vnqa_conv2d_wreg_supported": (ctypes.c_int, [ctypes.POINTER(ConvDesc)]),
    "vnqa_conv_ps_supported": (ctypes.c_int, [ctypes.POINTER(ConvDesc)]),
    "vnqa_conv2d_wreg_fwd": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vnqa_conv_first_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_pack_conv_weight": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "vnqa_conv_weight_tiled_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "vnqa_pack_conv_weight_tiled": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "vnqa_unpack_conv_wgrad": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "vnqa_feat_to_nhwc": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_nchw_to_nhwc": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_nhwc_to_nchw": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_frame_bn_stats": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_frame_bn_apply": (ctypes.c_int, [_vp] * 7 + [_i32] * 5 + [_vp]),
    "vnqa_second_order_round": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "vnqa_frame_bn_stats_split": (ctypes.c_int, [_vp] * 5 + [_i32] * 4 + [_vp]),
    "vnqa_frame_bn_apply_split": (ctypes.c_int, [_vp] * 8 + [_i32] * 4 + [_vp]),
    "vnqa_frame_bn_bwd": (ctypes.c_int, [_vp] * 10 + [_i32] * 7 + [_vp]),
    "vnqa_film_relu_res_fwd": (ctypes.c_int, [_vp] * 5 + [_i32] * 5 + [_vp]),
    "vnqa_film_relu_res_bwd": (ctypes.c_int, [_vp] * 7 + [_i32] * 5 + [_vp]),
    "vnqa_film_relu_res_fwd_ld": (ctypes.c_int, [_vp] * 5 + [_i32] * 7 + [_vp]),
    "vnqa_film_relu_res_bwd_ld": (ctypes.c_int, [_vp] * 7 + [_i32] * 8 + [_vp]),
    "vnqa_conv2d_bnstats_workspace": (_i64, [ctypes.POINTER(ConvDesc), _i32]),
    "vnqa_conv2d_igemm_fused_fwd": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.POINTER(ConvEpilogue), _vp, _vp]),
    "vnqa_relu_bwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "vnqa_temporal_attn_fwd": (ctypes.c_int, [_vp] * 7 + [_i32] * 3 + [_vp]),
    "vnqa_temporal_attn_bwd": (ctypes.c_int, [_vp] * 8 + [_i32] * 3 + [_vp]),
    "vnqa_temporal_attn_packed_fwd": (ctypes.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "vnqa_temporal_attn_packed_bwd": (ctypes.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "vnqa_unpack_conv_wgrad_scaled": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _f32, _vp]),
    "vnqa_unpack_fc_wgrad_scaled": (ctypes.c_int, [_vp] + [_i32] * 5 + [_vp, _f32, _vp]),
    "vnqa_unpack_fc_wgrad_dev": (ctypes.c_int, [_vp] + [_i32] * 5 + [_vp, _f32, _vp, _vp]),
    "vnqa_unpack_conv_wgrad_dev": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _f32, _vp, _vp]),
    "vnqa_sgemm_workspace": (_i64, [_i32, _i32, _i32]),
    "vnqa_sgemm": (ctypes.c_int, [_vp] * 7 + [_i64] * 4 + [_i32] * 6 + [_vp, _vp, _vp]),
    "vnqa_mac_core_workspace": (_i64, [_i32, _i32]),
    "vnqa_mac_core_fwd": (ctypes.c_int, [_vp, _vp]),
    "vnqa_mac_core_bwd": (ctypes.c_int, [_vp, _vp]),
    "vnqa_c3d_stats_blocks": (_i32, [_i64]),
    "vnqa_c3d_stats_ncdhw": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i64, _i32, _vp]),
    "vnqa_c3d_stats_rows": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "vnqa_bn_finalize": (ctypes.c_int, [_vp, _i32, _i32, ctypes.c_double, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "vnqa_bn_rows_apply": (ctypes.c_int, [_vp, _i32, _vp, _i32, ctypes.POINTER(View5), _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "vnqa_bn_rows_bwd": (ctypes.c_int, [_vp, _i32, ctypes.POINTER(View5), _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _f32, _i64, _i32, _vp]),
    "vnqa_pool444_blocks": (_i32, [_i32] * 5),
    "vnqa_pool444_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 5 + [_vp]),
    "vnqa_pool444_bwd": (ctypes.c_int, [_vp, _vp, _vp] + [_i32] * 5 + [_vp]),
    "vnqa_c3d_conv1_supported": (ctypes.c_int, [_i32] * 4),
    "vnqa_c3d_conv1_fwd_blocks": (_i32, [_i32] * 3),
    "vnqa_c3d_conv1_bwd_blocks": (_i32, [_i32] * 3),
    "vnqa_c3d_conv1_fwd": (ctypes.c_int, [_vp] * 10 + [_i32] * 4 + [_vp]),
    "vnqa_c3d_conv1_bwd": (ctypes.c_int, [_vp] * 9 + [_f32] + [_vp] * 4 + [_i32] * 4 + [_vp]),
    "vnqa_sgemm2": (ctypes.c_int, [_vp] * 4 + [_i64] * 4 + [_i32] * 4 + [_vp] * 6),
    "vnqa_sgemm_batch": (ctypes.c_int, [_vp, _i32, _vp]),
    "vnqa_mac_chain_fwd": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp]),
    "vnqa_mac_chain_bwd": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "vnqa_mac_core_wgrad_workspace": (_i64, [_i32, _i32]),
    "vnqa_mac_core_wgrad": (ctypes.c_int, [_vp, _vp]),
    "vnqa_colsum": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_gather_rows": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "vnqa_embed_proj_fwd": (ctypes.c_int, [_vp] * 8 + [_i32] * 5 + [_vp]),
    "vnqa_token_dsum": (ctypes.c_int, [_vp] * 3 + [_i32] * 3 + [_vp]),
    "vnqa_lstm_fold_dxg": (ctypes.c_int, [_vp] * 3 + [_i32] * 5 + [_vp]),
    "vnqa_lstm_wgrad_operands": (ctypes.c_int, [_vp] * 5 + [_i32] * 4 + [_vp]),
    "vnqa_ce_loss": (ctypes.c_int, [_vp] * 6 + [_i32] * 3 + [_vp]),
    "vnqa_bn_running_update": (ctypes.c_int, [_vp] * 5 + [_i32] * 4 + [_f32, _vp]),
    "vnqa_lstm_seq_fwd": (ctypes.c_int, [_vp] * 9 + [_i32] * 5 + [_vp]),
    "vnqa_lstm_seq_bwd": (ctypes.c_int, [_vp] * 10 + [_i32] * 4 + [_vp]),
    "vnqa_lstm_wide_fwd": (ctypes.c_int, [_vp] * 8 + [_i32] * 4 + [_vp]),
    "vnqa_lstm_wide_bwd": (ctypes.c_int, [_vp] * 8 + [_i32] * 4 + [_vp]),
    "vnqa_lstm_wide_bidir_fwd": (ctypes.c_int, [_vp] * 11 + [_i32] * 3 + [_vp]),
    "vnqa_lstm_wide_bidir_bwd": (ctypes.c_int, [_vp] * 13 + [_i32] * 3 + [_vp]),
    "vnqa_conv2d_igemm_fwd_ex": (ctypes.c_int, [_vp] * 9),
    "vnqa_conv2d_ring_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 7 + [_vp]),
    "vnqa_ring_edge_conv_fwd": (ctypes.c_int, [_vp, _vp, _vp] + [_i32] * 7 + [_vp]),
    "vnqa_conv2d_ring_edge_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 7 + [_vp]),
    "vnqa_ring_im2col": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_ring_edge_gather": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_ring_edge_gather_all": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_ring_assemble": (ctypes.c_int, [_vp] * 5 + [_i32] * 5 + [_vp]),
    "vnqa_zero_halo": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_frame_layout": (ctypes.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "vnqa_fc_dx": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_pack_fc_weight": (ctypes.c_int, [_vp] + [_i32] * 7 + [_vp, _vp, _vp]),
    "vnqa_unpack_fc_wgrad": (ctypes.c_int, [_vp] + [_i32] * 5 + [_vp, _vp]),
    "vnqa_clip_to_nhwc4": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_conv2d_border_edge_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_ring_assemble_corners": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_clip_to_nhwc4_shifted": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_clip_u8_to_nhwc4": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_conv_first_c64_fwd": (ctypes.c_int, [_vp] * 10),
    "vnqa_conv_first_c64_fwd_sched": (ctypes.c_int, [_vp] * 11),
    "vnqa_mac_read_fwd": (ctypes.c_int, [_vp] * 7 + [_i32] * 5 + [_vp]),
    "vnqa_mac_read_bwd": (ctypes.c_int, [_vp] * 7 + [_i32] * 5 + [_vp]),
    "vnqa_mac_read_fwd_scaled": (ctypes.c_int, [_vp] * 10 + [_i32] * 5 + [_vp]),
    "vnqa_mac_read_bwd_fused": (ctypes.c_int, [_vp] * 13 + [_i32] * 5 + [_vp]),
    "vnqa_mac_read_accum": (ctypes.c_int, [_vp] * 7 + [_i32] * 6 + [_vp]),
    "vnqa_l2norm_blocks": (_i32, [_i64]),
    "vnqa_l2norm_partial": (ctypes.c_int, [_vp, _i64, _vp, _vp]),
    "vnqa_clip_adam": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i32, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _vp]),
    "vnqa_conv3d_wgrad_workspace": (_i64, [_i32] * 6),
    "vnqa_conv3d_wgrad": (ctypes.c_int, [_vp] * 5 + [_i32] * 7 + [_vp]),
    "vnqa_gemm_nt_workspace": (_i64, [_i32, _i32, _i32, _i32]),
    "vnqa_gemm_nt": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_gemm_nt_grouped": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_gemm_tn_workspace": (_i64, [_i32, _i32, _i32, _i32]),
    "vnqa_gemm_tn": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vnqa_conv2d_wgrad_workspace": (_i64, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "vnqa_conv2d_wgrad": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
}

_lib = None


class VnqaError(RuntimeError):
    pass


def lib():
    """Load (once) and return the CDLL; raises if the HIP library has not been built."""
    global _lib, _half
    if _lib is None:
        if _half is None:          # nobody asked for a format yet: the one the default precision ('fp16h') stores in
            _half = "f16"
        path = LIB_PATH_F16 if (_half == "f16" and "VNQA_LIB" not in os.environ) else LIB_PATH
        if "VNQA_LIB" not in os.environ and os.path.exists("/opt/rocm/bin/hipcc") \
                and os.environ.get("VNQA_NO_REBUILD", "0") != "1":
            # not a fallback: (re)build the HIP library itself when it is missing OR older than its sources (the digest
            # check is cheap; build() takes a file lock, so torchrun ranks do not compile over each other)
            from .build import build as _build
            try:
                _build(verbose=False, variant=_half)
            except (OSError, RuntimeError) as e:      # read-only / NFS install, lock failure, compiler error
                if not os.path.exists(path):
                    raise VnqaError("cannot build %s (%s) — run `python -m videonavqa_amd.build` where the tree is "
                                    "writable" % (path, e))
                import warnings
                warnings.warn("HIP library may be stale: the implicit rebuild failed (%s); run `python -m "
                              "videonavqa_amd.build`" % e)
        if not os.path.exists(path):
            raise VnqaError(
                "%s not found — build it with `python -m videonavqa_amd.build` "
                "(there is no CPU/PyTorch fallback for the HIP path)" % path)
        cdll = ctypes.CDLL(path)
        cdll.vnqa_version.restype = ctypes.c_int
        got = cdll.vnqa_version()
        if got != ABI_VERSION:
            raise VnqaError("%s reports ABI version %d, this binding needs %d — stale library: rebuild it with `python -m "
                            "videonavqa_amd.build`" % (path, got, ABI_VERSION))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(cdll, name)  # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = cdll
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc, what):
    if rc != 0:
        raise VnqaError("%s failed (%d): %s" % (what, rc, lib().vnqa_last_error().decode()))


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "vnqa kernels need contiguous device tensors"
    return ctypes.c_void_p(t.data_ptr())


def to_device_async(t, device):
    """Host tensor -> device without blocking the launch thread: stage through pinned memory (the caching host
    allocator keeps the staging block alive until the copy has run).  A pageable-memory `.to(device)` makes the
    host wait until the current stream has drained up to the copy."""
    device = torch.device(device)
    if device.type != "cuda" or t.is_cuda:
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def vptr(t):
    """Device pointer of a (possibly strided) tensor view: the caller passes the strides to the kernel itself."""
    assert t.is_cuda
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def reserved_stream(reserve_cus, device=None):
    """A torch stream whose kernels never run on `reserve_cus` of the chip's CUs (vnqa_stream_create_reserved; a multiple of 32 on
    MI355X).  The persistent conv kernels' grids are sized to match per call (FrozenStem.reserve_cus -> conv_reserve_flags).  For the frozen stem when the trunk on the other stream is
    a latency-bound chain of small kernels (MACNetwork) or a collective must start at once (N > 1)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        out = ctypes.c_void_p()
        check(lib().vnqa_stream_create_reserved(int(reserve_cus), ctypes.byref(out)), "vnqa_stream_create_reserved")
        return torch.cuda.ExternalStream(out.value, device=dev)


def dtype_id(dt):
    if dt in (torch.bfloat16, torch.float16):
        if _lib is None:
            set_half("f16" if dt == torch.float16 else "bf16")
        if dt != half_dtype():
            raise VnqaError("%s tensor handed to the %s build of the HIP library (one 16-bit storage format per process)"
                            % (dt, _half))
        return BF16          # "the library's 16-bit format"
    if dt == torch.float32:
        return F32
    raise VnqaError("unsupported dtype %s" % dt)


def round_up(x, m):
    return (x + m - 1) // m * m
