"""ctypes binding of libssd_gfx950.so (C ABI in include/ssd_gfx950.h).

The shared library is the only compute path: if it cannot be loaded the import
fails loudly -- there is no CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libssd_gfx950.so")

SSD_ABI_VERSION = 1


class ConvGeom(C.Structure):
    """struct ssd_conv_geom"""
    _fields_ = [(n, C.c_int32) for n in ("N", "H", "W", "Ci", "Ho", "Wo", "Co", "R", "S", "stride", "pad", "dil")]


class ImageDesc(C.Structure):
    """struct ssd_image_desc"""
    _fields_ = [("src_offset", C.c_int64)] + [(n, C.c_int32) for n in (
        "src_h", "src_w", "canvas_h", "canvas_w", "place_top", "place_left", "crop_top", "crop_left", "crop_h", "crop_w",
        "flip", "reserved")]


class WeightJob(C.Structure):
    """struct ssd_weight_job"""
    _fields_ = [("w0", C.c_void_p), ("w1", C.c_void_p), ("out_fwd", C.c_void_p), ("out_bwd", C.c_void_p)] + [(n, C.c_int32) for n in (
        "co0", "co", "ci", "taps", "co_pad", "kind", "pad0", "pad1")]


class PhotoDesc(C.Structure):
    """struct ssd_photo_desc"""
    _fields_ = [("n_ops", C.c_int32), ("kind", C.c_int32 * 4), ("alpha", C.c_float * 4), ("hue_delta", C.c_int32 * 4)]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_Z = C.c_size_t
_G = C.POINTER(ConvGeom)

# name -> (restype, argtypes); must list every symbol declared in include/ssd_gfx950.h
SIGNATURES = {
    "ssd_abi_version": (_I, []),
    "ssd_status_string": (C.c_char_p, [_I]),
    "ssd_weight_oihw_to_ohwi": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ssd_weight_oihw_to_ihwo": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ssd_conv2d_fwd": (_I, [_P, _P, _P, _P, _I, _G, _I, _P]),
    "ssd_conv2d_dgrad": (_I, [_P, _I, _P, _I, _P, _P, _I, _G, _P]),
    "ssd_conv2d_fwd_bf16": (_I, [_P, _P, _P, _P, _I, _G, _I, _P]),
    "ssd_conv2d_dgrad_bf16": (_I, [_P, _I, _P, _I, _P, _P, _I, _G, _P]),
    "ssd_conv2d_wgrad_bf16": (_I, [_P, _P, _I, _P, _P, _G, _P, _Z, _P]),
    "ssd_conv2d_fwd_bf16_ws": (_I, [_P, _P, _P, _P, _I, _G, _I, _P, _Z, _P]),
    "ssd_conv2d_dgrad_bf16_ws": (_I, [_P, _I, _P, _I, _P, _P, _I, _G, _P, _Z, _P]),
    "ssd_tune_set_igemm_bf16": (_I, [_I]),
    "ssd_weight_split_bf16x3": (_I, [_P, _P, _Z, _P]),
    "ssd_conv2d_fwd_x3": (_I, [_P, _P, _I, _P, _P, _I, _G, _I, _P]),
    "ssd_conv2d_dgrad_x3": (_I, [_P, _I, _P, _I, _P, _P, _I, _G, _P]),
    "ssd_tune_set_igemm_x3": (_I, [_I]),
    "ssd_tune_set_halo": (_I, [_I]),
    "ssd_clock_probe": (_I, [_P, _P]),
    "ssd_tune_set_x3_mfma": (_I, [_I]),
    "ssd_tune_set_x3_big": (_I, [_I]),
    "ssd_tune_set_loss_form": (_I, [_I]),
    "ssd_gemm_planes_x3v2": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ssd_graph_node_counts": (_I, [_P, _P, _P]),
    "ssd_photometric_workspace": (_Z, [_I]),
    "ssd_photometric_u8": (_I, [_P, _P, _P, _P, _P, _I, _P, _Z, _P]),
    "ssd_preprocess_workspace": (_Z, [_P, _I, _I, _I]),
    "ssd_preprocess_u8": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "ssd_map_eval_workspace": (_Z, [_I, _I]),
    "ssd_map_eval": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "ssd_im2col_nchw3": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ssd_conv2d_fwd_accum": (_I, [_P, _P, _P, _P, _I, _G, _I, _P]),
    "ssd_conv2d_fwd_accum_bf16": (_I, [_P, _P, _P, _P, _I, _G, _I, _P]),
    "ssd_channel_affine": (_I, [_P, _P, _P, _P, C.c_size_t, _I, _I, _P]),
    "ssd_conv3x3_bf16": (_I, [_P, _I, _P, _I, _I, _P, _P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ssd_tune_set_conv_bf16": (_I, [_I, _I]),
    "ssd_tune_set_conv_bf16_k64": (_I, [_I]),
    "ssd_tune_set_conv_bf16_mfma": (_I, [_I]),
    "ssd_wino_uses_x3": (_I, [_I, _I]),
    "ssd_tune_set_wino_x3": (_I, [_I]),
    "ssd_gemm_x3_weights_bytes": (_Z, [_I, _I, _I]),
    "ssd_gemm_x3_split_weights": (_I, [_P, _P, _I, _I, _I, _P]),
    "ssd_gemm_planes_x3": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ssd_gemm_planes_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ssd_conv1x1_fwd_x3": (_I, [_P, _P, _P, _P, _I, _G, _I, _P]),
    "ssd_conv1x1_dgrad_x3": (_I, [_P, _I, _P, _P, _P, _I, _G, _P]),
    "ssd_conv1x1_wgrad_x3_workspace": (_Z, [_G, _I]),
    "ssd_conv1x1_wgrad_x3": (_I, [_P, _P, _I, _P, _P, _G, _P, _Z, _P]),
    "ssd_has_experimental": (_I, []),
    "ssd_conv3x3_wgrad_bf16t": (_I, [_P, _P, _I, _P, _P, _G, _P, _Z, _P]),
    "ssd_conv1_first_fwd_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ssd_conv1_first_wgrad_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "ssd_maxpool_fwd_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ssd_maxpool_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ssd_l2norm_fwd_bf16": (_I, [_P, _P, _P, _I, _I, _P]),
    "ssd_l2norm_bwd_bf16_workspace": (_Z, [_I, _I]),
    "ssd_l2norm_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _Z, _P]),
    "ssd_heads_gather_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ssd_cast_f32_bf16": (_I, [_P, _P, _Z, _P]),
    "ssd_cast_bf16_f32": (_I, [_P, _P, _Z, _P]),
    "ssd_conv3x3_halo_fwd_bf16": (_I, [_P, _P, _I, _P, _P, _I, _G, _I, _P]),
    "ssd_conv3x3_halo_dgrad_bf16": (_I, [_P, _I, _P, _I, _P, _P, _I, _G, _P]),
    "ssd_conv2d_wgrad_workspace": (_Z, [_G]),
    "ssd_conv2d_wgrad": (_I, [_P, _P, _I, _P, _P, _G, _P, _Z, _P]),
    "ssd_conv2d_igemm_tile": (_I, [_G, _I, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ssd_conv2d_wgrad_tile": (_I, [_G, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ssd_conv2d_igemm_workspace": (_Z, [_G, _I]),
    "ssd_conv2d_fwd_ws": (_I, [_P, _P, _P, _P, _I, _G, _I, _P, _Z, _P]),
    "ssd_conv2d_dgrad_ws": (_I, [_P, _I, _P, _I, _P, _P, _I, _G, _P, _Z, _P]),
    "ssd_tune_set_igemm_splitk": (_I, [_I]),
    "ssd_wino_weights": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "ssd_conv3x3_wino_workspace": (_Z, [_G, _I, _I]),
    "ssd_conv3x3_wino_fwd": (_I, [_P, _P, _P, _P, _I, _G, _I, _I, _P, _Z, _P]),
    "ssd_conv3x3_wino_fwd_pool": (_I, [_P, _P, _P, _P, _P, _G, _I, _P, _P, _Z, _P]),
    "ssd_conv3x3_wino_fwd_keep": (_I, [_P, _P, _P, _P, _I, _G, _I, _P, _P, _Z, _P]),
    "ssd_weight_job_blocks": (_I, [_P]),
    "ssd_weights_prepare": (_I, [_P, _P, _I, _I, _P]),
    "ssd_wino4_bias_partial_floats": (_Z, [_G, _I]),
    "ssd_wino4_dy_transform": (_I, [_P, _I, _G, _P, _P, _P, _P]),
    "ssd_wino_weights_adj": (_I, [_P, _P, _I, _I, _I, _P]),
    "ssd_conv1_first_wino_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ssd_conv3x3_wino_fwd_from_planes": (_I, [_P, _P, _P, _P, _I, _P, _P, _G, _I, _I, _P, _Z, _P]),
    "ssd_wino4_adj_planes_floats": (_Z, [_G]),
    "ssd_conv3x3_wino_dgrad_adj_gemm": (_I, [_P, _I, _P, _P, _G, _P]),
    "ssd_wino4_adj_output": (_I, [_P, _P, _P, _P, _I, _G, _P]),
    "ssd_wino4_adj_output_to_planes": (_I, [_P, _P, _P, _G, _G, _P, _P, _P]),
    "ssd_wino4_dy_transform_pooled": (_I, [_P, _P, _P, _I, _I, _I, _G, _P, _P, _P, _P]),
    "ssd_conv3x3_wino_wgrad_planes_pooled": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _G, _P, _P, _Z, _P]),
    "ssd_wino4_wgrad_gemm_workspace": (_Z, [_G, _I]),
    "ssd_wino4_wgrad_gemm": (_I, [_P, _P, _I, _P, _P, _P, _G, _P, _Z, _P]),
    "ssd_conv3x3_wino_fwd_keep_bits": (_I, [_P, _P, _P, _P, _I, _G, _I, _P, _P, _P, _Z, _P]),
    "ssd_conv3x3_wino_fwd_pool_bits": (_I, [_P, _P, _P, _P, _P, _G, _I, _P, _P, _P, _Z, _P]),
    "ssd_conv3x3_wino_dgrad_planes_bits": (_I, [_P, _P, _I, _P, _P, _I, _G, _P, _Z, _P]),
    "ssd_conv3x3_wino_wgrad_planes": (_I, [_P, _P, _I, _P, _P, _G, _P, _P, _Z, _P]),
    "ssd_conv3x3_wino_dgrad_planes": (_I, [_P, _P, _I, _P, _P, _I, _G, _P, _Z, _P]),
    "ssd_conv3x3_wino_dgrad": (_I, [_P, _I, _P, _I, _P, _P, _I, _G, _I, _P, _Z, _P]),
    "ssd_conv3x3_wino_wgrad_workspace": (_Z, [_G, _I, _I]),
    "ssd_conv3x3_wino_wgrad": (_I, [_P, _P, _I, _P, _P, _G, _I, _P, _Z, _P]),
    "ssd_tune_set_wino_wgrad_tn": (_I, [_I]),
    "ssd_tune_set_wino_fused": (_I, [_I]),
    "ssd_tune_set_gemm_nt": (_I, [_I]),
    "ssd_tune_set_wino_bias_tail": (_I, [_I]),
    "ssd_tune_set_batched_units": (_I, [_I]),
    "ssd_tune_set_dgrad_parity": (_I, [_I]),
    "ssd_tune_set_wino_full": (_I, [_I]),
    "ssd_conv3x3_wino_uses_full": (_I, [_G, _I]),
    "ssd_conv3x3_wino_dgrad_bits": (_I, [_P, _I, _P, _I, _P, _P, _I, _G, _P, _Z, _P]),
    "ssd_tune_set_wino_xform_blocks": (_I, [_I]),
    "ssd_tune_set_wino_fused_stamps": (_I, [_P]),
    "ssd_tune_set_wino_fused_stagger": (_I, [_I]),
    "ssd_prof_gemm_begin": (_I, []),
    "ssd_prof_gemm_collect": (_I, [_P, _P, _I]),
    "ssd_prof_gemm_collect_kinds": (_I, [_P, _P, _P, _I]),
    "ssd_tune_set_igemm": (_I, [_I, _I]),
    "ssd_tune_set_igemm_lds_pad": (_I, [_I]),
    "ssd_tune_set_igemm_stamps": (_I, [_P]),
    "ssd_tune_set_wgrad": (_I, [_I, _I, _I]),
    "ssd_tune_set_wgrad_patch": (_I, [_I]),
    "ssd_im2col_first": (_I, [_P, _P, _I, _I, _I, _P]),
    "ssd_conv1_first_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ssd_conv1_first_wgrad_workspace": (_Z, [_I, _I, _I]),
    "ssd_conv1_first_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "ssd_maxpool_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ssd_maxpool_bwd_gated": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ssd_maxpool_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ssd_l2norm_fwd": (_I, [_P, _P, _P, _I, _I, _P]),
    "ssd_l2norm_bwd_workspace": (_Z, [_I, _I]),
    "ssd_l2norm_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _Z, _P]),
    "ssd_heads_scatter": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ssd_heads_gather": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ssd_multibox_loss_workspace": (_Z, [_I, _I, _I]),
    "ssd_multibox_loss": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _F, _I, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "ssd_decode_nms_workspace": (_Z, [_I, _I]),
    "ssd_decode_nms": (_I, [_P, _P, _P, _I, _I, _F, _F, _I, _F, _F, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "ssd_decode_nms_batch_workspace": (_Z, [_I, _I, _I]),
    "ssd_decode_nms_batch": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "ssd_sgd_momentum": (_I, [_P, _P, _P, _Z, _F, _F, _F, _P, _I, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load (building first if the .so is absent and hipcc is present)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel carries its own libamdhip64, and the process must end up with ONE HIP runtime.  Loading this
    # library before torch binds it to /opt/rocm's copy instead, and the first kernel launch then fails.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        from . import build as _build          # raises if hipcc is missing
        _build.build()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:                      # pragma: no cover
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}.  The gfx950 HIP extension is the only compute "
                           f"path of this package; build it with `python -m objectdetection_ssd_amd.build`.") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)               # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.ssd_abi_version() != SSD_ABI_VERSION:
        raise RuntimeError("libssd_gfx950.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().ssd_status_string(status).decode()
        raise RuntimeError(f"libssd_gfx950 {what}: {msg} (status {status})")
