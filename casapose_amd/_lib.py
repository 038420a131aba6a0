"""ctypes binding of libcasapose_hip.so (the C ABI declared in include/casapose_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails,
`CasaposeHipError` is raised.  Build the library with `python __graft_entry__.py` (or
`make -C casapose_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# PyTorch-ROCm ships its own libamdhip64; import it FIRST so that our library's dependency on
# the same SONAME binds to the runtime that owns torch's device context and streams (two HIP
# runtimes in one process do not share devices: launches fail with "no ROCm-capable device").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# CASAPOSE_HIP_LIB overrides the library path (kernel-variant experiments only)
LIB_PATH = os.environ.get("CASAPOSE_HIP_LIB") or os.path.join(_HERE, "libcasapose_hip.so")


class CasaposeHipError(RuntimeError):
    pass


class ConvSource(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("channels", C.c_int),
        ("ld", C.c_int),
        ("mode", C.c_int),
        ("sel", C.c_void_p),
        ("pre_scale", C.c_void_p),
        ("pre_shift", C.c_void_p),
    ]


class ConvDesc(C.Structure):
    """cp_conv_desc of include/casapose_hip.h, field for field (tests/test_capi_symbols.py parses the header and compares).  `struct_size` is
    filled by the constructor; load() refuses a library whose sizeof(cp_conv_desc) differs from this declaration."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(ConvDesc)

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("batch", C.c_int),
        ("in_h", C.c_int),
        ("in_w", C.c_int),
        ("out_h", C.c_int),
        ("out_w", C.c_int),
        ("cout", C.c_int),
        ("kh", C.c_int),
        ("kw", C.c_int),
        ("stride", C.c_int),
        ("dilation", C.c_int),
        ("pad", C.c_int),
        ("num_sources", C.c_int),
        ("src", ConvSource * 2),
        ("weights", C.c_void_p),
        ("weights_halo", C.c_void_p),
        ("tap_label", C.c_void_p),
        ("row_scale", C.c_void_p),
        ("residual", C.c_void_p),
        ("residual_ld", C.c_int),
        ("scale", C.c_void_p),
        ("shift", C.c_void_p),
        ("epi_label", C.c_void_p),
        ("act", C.c_int),
        ("out_raw", C.c_void_p),
        ("out_raw_ld", C.c_int),
        ("out_act", C.c_void_p),
        ("out_act_ld", C.c_int),
        ("tile_hint", C.c_int),
        ("head_weights", C.c_void_p),
        ("head_out", C.c_void_p),
        ("head_cout", C.c_int),
        ("head_out_ld", C.c_int),
        ("group_rows", C.c_int),
        ("group_weight_stride", C.c_int),
        ("head_label_out", C.c_void_p),
        ("head_label_classes", C.c_int),
        ("head_prefix", C.c_void_p),
        ("head_prefix_n", C.c_int),
        ("head_prefix_ld", C.c_int),
    ]


ABI_VERSION = 302  # CP_ABI_VERSION of include/casapose_hip.h
PLANES_F16X2 = 0x12  # CP_PLANES_F16X2: the fp16 two-way split (three products, fp32-level accuracy)

SRC_DIRECT, SRC_NEAREST_SEL, SRC_BILINEAR_X2, SRC_ZERO_INSERT_X2 = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_LEAKY01 = 0, 1, 2
TILE_AUTO, TILE_128x128, TILE_64x128, TILE_128x64, TILE_128x32, TILE_64x64, TILE_256x32, TILE_HALO, TILE_STEM = range(9)
# host-side selectors (never passed to the library): the bf16-pipe kernel of csrc/conv_hsplit.hip with 3 planes (exact fp32 split) / 1 plane (bf16)
TILE_SPLIT3, TILE_BF16, TILE_F16X2 = 100, 101, 102

# every symbol include/casapose_hip.h declares: (name, restype, argtypes)
_vp, _i, _ll, _f = C.c_void_p, C.c_int, C.c_longlong, C.c_float
SYMBOLS = [
    ("cp_last_error", C.c_char_p, []),
    ("cp_version", _i, []),
    ("cp_conv_desc_size", C.c_size_t, []),
    ("cp_conv_source_size", C.c_size_t, []),
    ("cp_device_count", _i, []),
    ("cp_set_persistent_blocks", _i, [_i]),
    ("cp_get_persistent_blocks", _i, []),
    ("cp_mfma_probe_workspace_bytes", C.c_size_t, []),
    ("cp_mfma_probe", _i, [_i, _i, _vp, C.POINTER(C.c_double), _vp]),
    ("cp_conv_ktot", _i, [_i, _i, _i, C.POINTER(_i)]),
    ("cp_conv_pack_weights_host", _i, [_vp, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), _vp]),
    ("cp_conv_halo_weight_floats", _i, [_i, _i, C.POINTER(_i)]),
    ("cp_conv_pack_weights_halo_host", _i, [_vp, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), _vp]),
    ("cp_conv_pack_weights_stem_host", _i, [_vp, _i, _i, _vp]),
    ("cp_conv_pack_head_weights_host", _i, [_vp, _i, _vp]),
    ("cp_conv2d_fwd_f32", _i, [C.POINTER(ConvDesc), _vp]),
    ("cp_conv_selected_tile", _i, [C.POINTER(ConvDesc)]),
    ("cp_conv_split_applicable", _i, [C.POINTER(ConvDesc)]),
    ("cp_conv_split_weight_floats", _i, [_i, _i, C.POINTER(_i)]),
    ("cp_conv_split_weight_bytes", C.c_size_t, [_i, _i, C.POINTER(_i), _i]),
    ("cp_conv_pack_weights_split_host", _i, [_vp, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), _vp]),
    ("cp_conv_split_weights_f32", _i, [_vp, _ll, _i, _vp, _vp]),
    ("cp_conv_pack_head_split_host", _i, [_vp, _i, _vp]),
    ("cp_conv2d_fwd_split", _i, [C.POINTER(ConvDesc), _vp, _vp, _i, _vp]),
    ("cp_conv_split_weights_scaled_f32", _i, [_vp, _ll, _i, C.c_float, _vp, _vp]),
    ("cp_conv2d_fwd_split_scaled", _i, [C.POINTER(ConvDesc), _vp, _vp, _i, C.c_float, C.c_float, _vp]),
    ("cp_conv_bf16_deep_applicable", _i, [C.POINTER(ConvDesc)]),
    ("cp_conv2d_fwd_bf16_deep", _i, [C.POINTER(ConvDesc), _vp, _vp]),
    ("cp_pad_channels_3to4", _i, [_vp, _vp, _ll, _vp]),
    ("cp_maxpool3x3s2_f32", _i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    ("cp_upsample_bilinear_x2_f32", _i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    ("cp_guided_upsample_x2_f32", _i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    ("cp_mask_to_labels_f32", _i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    ("cp_argmax_labels", _i, [_vp, _i, _i, _ll, _vp, _vp]),
    ("cp_label_pyramid", _i, [_vp, _i, _i, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _vp]),
    ("cp_ls_vote_f32", _i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    ("cp_ls_vote_w_f32", _i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    ("cp_ls_vote_workspace_bytes", C.c_size_t, [_i, _i, _i]),
    ("cp_ccl_filter_labels", _i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    ("cp_ccl_workspace_bytes", C.c_size_t, [_i, _i, _i, _i]),
    ("cp_ransac_vote_f32", _i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _f, _f, _i, _i, _i, _vp, _vp, _vp, _vp]),
    ("cp_ransac_vote_seeded_f32", _i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, C.c_ulonglong, _i, _f, _f, _i, _i, _i, _vp, _vp, _vp, _vp]),
    ("cp_ransac_workspace_bytes", C.c_size_t, [_i, _i, _i, _i, _i, _i]),
    ("cp_guided_match_mask", _i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    ("cp_guided_bilinear_upsample_x2_f32", _i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    ("cp_guided_bilinear_upsample_x2_bwd_f32", _i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    ("cp_wino_tiles", _i, [_i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    ("cp_wino_gemm_f32", _i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    ("cp_wino_gemm_split_f32", _i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    ("cp_wino_gemm_split_planes_f32", _i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    ("cp_wino_split_weights_bytes", C.c_size_t, [_i, _i, _i]),
    ("cp_wino_split_weights_f32", _i, [_vp, _i, _i, _i, _vp, _vp]),
    ("cp_f16x2_weight_scale", C.c_float, [C.c_float]),
    ("cp_f16x2_monitor_set", _i, [_vp]),
    ("cp_f16x2_monitor_get", _vp, []),
    ("cp_f16x2_range_check", _i, [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]),
    ("cp_amax_f32", _i, [_vp, _ll, _ll, _ll, _vp, _vp]),
    ("cp_wino_split_weights_scaled_f32", _i, [_vp, _i, _i, _i, _i, C.c_float, _vp, _vp]),
    ("cp_wino_gemm_split_scaled_f32", _i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_float, _vp]),
    ("cp_wino_pack_weights_host", _i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    ("cp_wino_transform_weights_f32", _i, [_vp, _ll, _ll, _ll, _ll, _i, _i, _i, _i, _i, _vp, _vp]),
    ("cp_wino_dy_transform_f32", _i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    ("cp_wino_weight_grad_f32", _i, [_vp, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _vp, _i, _vp]),
    ("cp_wino_wgrad_split_applicable", _i, [_i, _i, _i, _i]),
    ("cp_wino_wgrad_split_f32", _i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    ("cp_wino_wgrad_split_scaled_f32", _i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp]),
    ("cp_wino_input_transform_f32", _i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    ("cp_wino_output_transform_f32", _i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp]),
    ("cp_conv_stem_split_weight_floats", _i, []),
    ("cp_conv_pack_weights_stem_split_host", _i, [_vp, _i, _i, _vp]),
    ("cp_conv2d_fwd_stem_split", _i, [C.POINTER(ConvDesc), _vp, _i, _vp]),
    ("cp_conv2d_fwd_stem_split_scaled", _i, [C.POINTER(ConvDesc), _vp, _i, C.c_float, _vp]),
    ("cp_wino_output_input_applicable", _i, [_i, _i, _i, _i, _i]),
    ("cp_wino_output_input_transform_f32", _i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp]),
    ("cp_wino_output_transform_stats_f32", _i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp]),
    ("cp_wino_input_transform_pre_f32", _i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _vp]),
    # ---- training path ----
    ("cp_conv2d_wgrad_f32", _i, [C.POINTER(ConvDesc), _vp, _i, _vp, _i, _vp]),
    ("cp_conv_wgrad_split_applicable", _i, [C.POINTER(ConvDesc)]),
    ("cp_head1x1_fwd_f32", _i, [_vp, _i, _ll, _vp, _i, _vp, _i, _vp]),
    ("cp_head1x1_dgrad_f32", _i, [_vp, _i, _i, _ll, _vp, _i, _vp, _i, _i, _vp]),
    ("cp_head1x1_wgrad_f32", _i, [_vp, _i, _vp, _i, _ll, _i, _vp, _i, _vp]),
    ("cp_head1x1_fwd_affine_f32", _i, [_vp, _i, _ll, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp]),
    ("cp_head1x1_fwd_affine_record_f32", _i, [_vp, _i, _ll, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp]),
    ("cp_head1x1_wgrad_affine_f32", _i, [_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _ll, _i, _vp, _i, _vp]),
    ("cp_head1x1_bn_bwd_reduce_f32", _i, [_vp, _i, _vp, _i, _i, _ll, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    ("cp_head1x1_bn_bwd_apply_f32", _i, [_vp, _i, _vp, _i, _i, _ll, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, C.c_double, _vp, _vp, _i, _vp]),
    ("cp_conv2d_wgrad_split", _i, [C.POINTER(ConvDesc), _vp, _i, _vp, _i, _i, _vp]),
    ("cp_bn_stats_f32", _i, [_vp, _ll, _i, _i, _vp, _vp]),
    ("cp_bn_finalize_f32", _i, [_vp, C.c_double, _i, _i, _i, _vp, _vp, _f, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("cp_bn_param_grads_f32", _i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    ("cp_affine_act_f32", _i, [_vp, _ll, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    ("cp_bn_act_bwd_reduce_f32", _i, [_vp, _i, _vp, _i, _ll, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    ("cp_bn_act_bwd_apply_f32", _i, [_vp, _i, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, C.c_double, _vp, _vp, _i, _i, _vp]),
    ("cp_maxpool3x3s2_bwd_f32", _i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    ("cp_maxpool3x3s2_idx_f32", _i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    ("cp_maxpool3x3s2_bwd_idx_f32", _i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    ("cp_upsample_bilinear_x2_bwd_f32", _i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    ("cp_guided_upsample_x2_bwd_f32", _i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    ("cp_gather_f32", _i, [_vp, _vp, _ll, _vp, _vp]),
    ("cp_scatter_f32", _i, [_vp, _vp, _ll, _vp, _i, _vp]),
    ("cp_axpby_f32", _i, [_vp, _f, _vp, _f, _ll, _vp, _vp]),
    ("cp_adam_step_f32", _i, [_vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _i, _f, _vp]),
    ("cp_pose_loss_workspace_bytes", C.c_size_t, [_i, _i, _i]),
    ("cp_ls_vote_bwd_f32", _i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    ("cp_kp_stats_f32", _i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    ("cp_kp_reproj_loss_f32", _i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    ("cp_pose_loss_sep_workspace_bytes", C.c_size_t, [_i, _i, _i, _i]),
    ("cp_pose_loss_sep_f32", _i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _i, _i, _vp, _vp]),
    ("cp_vector_field_f32", _i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    ("cp_smooth_l1_f32", _i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _ll, _vp, _vp, _vp]),
    ("cp_proxy_voting_f32", _i, [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("cp_pose_loss_f32", _i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _i, _i, _vp, _vp, _vp]),
]

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library (once) and bind every declared symbol.  Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CasaposeHipError(
            "libcasapose_hip.so not found at %s -- build it with `python __graft_entry__.py` "
            "(there is no CPU fallback for the product path)" % LIB_PATH
        )
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    # ABI check: the descriptor grew between rounds; a library built from another revision of the header must not be handed this struct
    if lib.cp_version() != ABI_VERSION or lib.cp_conv_desc_size() != C.sizeof(ConvDesc) or lib.cp_conv_source_size() != C.sizeof(ConvSource):
        raise CasaposeHipError("%s has ABI %d with sizeof(cp_conv_desc) = %d, sizeof(cp_conv_source) = %d; this binding is ABI %d with %d / %d -- "
                               "rebuild the library (python __graft_entry__.py)" % (LIB_PATH, lib.cp_version(), lib.cp_conv_desc_size(),
                                                                                  lib.cp_conv_source_size(), ABI_VERSION, C.sizeof(ConvDesc), C.sizeof(ConvSource)))
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().cp_last_error()
        raise CasaposeHipError("%s failed (%d): %s" % (what or "casapose_hip call", status, (msg or b"").decode()))
