"""ctypes binding of librnnpose_hip.so (C ABI declared in include/rnnpose_hip.h).

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 3          # include/rnnpose_hip.h: RNNPOSE_ABI_VERSION of the library this front end was written against
LIB_PATH = os.environ.get("RNNPOSE_LIB") or os.path.join(_PKG, "lib", "librnnpose_hip.so")

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_z = C.c_size_t
_ll = C.c_longlong

class ConvSrc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("c_stride", C.c_int), ("c_offset", C.c_int), ("c_count", C.c_int)]


class ConvDesc(C.Structure):       # rnnpose_conv_desc_t
    _fields_ = [("src", ConvSrc * 4), ("n_src", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int), ("w_packed", C.c_void_p), ("bias", C.c_void_p),
                ("c_out", C.c_int), ("a_scale", C.c_float), ("w_scale", C.c_float), ("epilogue", C.c_int),
                ("dst", C.c_void_p), ("dst_c_stride", C.c_int), ("dst_c_offset", C.c_int),
                ("aux0", C.c_void_p), ("aux0_c_stride", C.c_int), ("aux0_c_offset", C.c_int),
                ("aux1", C.c_void_p), ("aux1_c_stride", C.c_int), ("aux1_c_offset", C.c_int),
                ("dst2", C.c_void_p), ("dst2_c_stride", C.c_int), ("dst2_c_offset", C.c_int), ("gru_c", C.c_int),
                ("tile_stats", C.c_void_p), ("add_map", C.c_void_p), ("add_c_stride", C.c_int), ("add_c_offset", C.c_int),
                ("src0_mean_rstd", C.c_void_p), ("src_hl", C.c_int), ("dst_hl", C.c_int), ("dst2_hl", C.c_int),
                ("dst_split", C.c_void_p), ("dst_split_c_stride", C.c_int), ("dst_split_c_offset", C.c_int), ("src_bounded", C.c_int),
                ("tile", C.c_int), ("ksplit_ws", C.c_void_p), ("ksplit_ws_bytes", C.c_size_t), ("single_product", C.c_int),
                ("tile_stats_records", C.c_int)]


# name -> (restype, argtypes); mirrors include/rnnpose_hip.h one to one
PROTOTYPES = {
    "rnnpose_abi_version": (_i, []),
    "rnnpose_last_error": (C.c_char_p, []),
    "rnnpose_device_info": (_i, [_i, C.c_char_p, _i, C.POINTER(_i)]),
    "rnnpose_corr_pyramid_layout": (_i, [_i, _i, _i, _i, C.POINTER(C.c_int64), C.POINTER(_i), C.POINTER(_i)]),
    "rnnpose_corr_pyramid_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "rnnpose_corr_pyramid_f16x3_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "rnnpose_corr_pyramid_f16x3": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _z, _p, _p]),
    "rnnpose_corr_pyramid_split": (_i, [_p, _p, _i, _i, _i, _i, _i, _f, _p, _p]),
    "rnnpose_corr_supertile": (_i, [_i]),
    "rnnpose_corr_variant": (_i, [_i]),
    "rnnpose_fmap_pyramid_floats": (_z, [_i, _i, _i, _i, _i]),
    "rnnpose_fmap_pyramid_f32": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "rnnpose_corr_alt_lookup_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _i, _p]),
    "rnnpose_corr_lookup_variant": (_i, [_i]),
    "rnnpose_corr_lookup_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "rnnpose_context_prep_f32": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "rnnpose_flow_to_coords_f32": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "rnnpose_convex_upsample_f32": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "rnnpose_induced_flow_f32": (_i, [_p, _p, _p, _i, _i, _i, _f, _i, _p, _p, _p]),
    "rnnpose_induced_coords_lowres_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p]),
    "rnnpose_corr_weight_pairs": (_i, [_i]),
    "rnnpose_corr_weight_f32": (_i, [_p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _p, _p]),
    "rnnpose_lm_workspace_bytes": (_z, [_i, _i, _i]),
    "rnnpose_lm_normal_eq_f64": (_i, [_p, _i, _p, _p, _f, _p, _p, _i, _i, _i, _p, _z, _p, _p, _p]),
    "rnnpose_lm_solve_update_f32": (_i, [_p, _p, _p, _i, _d, _d, _d, _p, _p, _p, _p]),
    "rnnpose_lm_step_f32": (_i, [_p, _i, _p, _p, _f, _p, _p, _i, _i, _i, _i, _d, _d, _d, _p, _z, _p, _p, _p, _p, _p]),
    "rnnpose_lm_step_io_f32": (_i, [_p, _i, _p, _p, _f, _p, _p, _p, _i, _i, _i, _i, _d, _d, _d, _p, _z, _p, _p, _p, _p, _p]),
    "rnnpose_lm_fused_tail": (_i, [_i]),
    "rnnpose_se3_exp_f32": (_i, [_p, _i, _p, _p]),
    "rnnpose_se3_compose_f32": (_i, [_p, _p, _i, _p, _p]),
    "rnnpose_se3_inverse_f32": (_i, [_p, _i, _p, _p]),
    "rnnpose_se3_outer_update_f32": (_i, [_p, _p, _i, _i, _p, _p, _p]),
    "rnnpose_gru_gate_f32": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "rnnpose_gru_update_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _i, _p]),
    "rnnpose_conv_tiles_per_image": (_i, [_i, _i, _i, _i, _i]),
    "rnnpose_conv_tiles_per_image_ex": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "rnnpose_conv_tiles_per_image_desc": (_i, [C.POINTER(ConvDesc)]),
    "rnnpose_conv_products_desc": (_i, [C.POINTER(ConvDesc)]),
    "rnnpose_conv_spatial_tiles": (_i, [_i]),
    "rnnpose_conv_strip": (_i, [_i]),
    "rnnpose_conv_ksplit": (_i, [_i]),
    "rnnpose_conv_ksplit_limits": (_i, [_i, _i]),
    "rnnpose_conv_ksplit_workspace_bytes": (_z, []),
    "rnnpose_conv_packed_halfs": (C.c_longlong, [_i, _i, _i, C.POINTER(_i), _i]),
    "rnnpose_conv_pack_weights_f16x3": (_i, [_p, _i, _i, _i, _i, C.POINTER(_i), _i, _f, _p, _p]),
    "rnnpose_conv2d_nhwc_f16x3": (_i, [C.POINTER(ConvDesc), _p]),
    "rnnpose_f16x3_saturation_check": (_i, [_i]),
    "rnnpose_f16x3_saturation_count": (_i, [C.POINTER(C.c_ulonglong), _i, _p]),
    "rnnpose_f16x3_saturation_peek": (_i, [_p, _p]),
    "rnnpose_stem_packed_halfs": (C.c_longlong, []),
    "rnnpose_stem_workgroups": (_i, [_i]),
    "rnnpose_stem_pack_weights_f16x3": (_i, [_p, _f, _p, _p, _p]),
    "rnnpose_stem_tiles": (_i, [_i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "rnnpose_stem_conv7x7_s2_f16x3": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _f, _f, _p, _p, _p]),
    "rnnpose_corr_lookup_nhwc_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "rnnpose_corr_lookup_nhwc_part_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "rnnpose_nchw_to_nhwc_f32": (_i, [_p, _i, _i, _i, _p, _i, _i, _p]),
    "rnnpose_nhwc_to_nchw_f32": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "rnnpose_flow_prep_f32": (_i, [_p, _i, _i, _i, _i, _p, _p, _i, _i, _p]),
    "rnnpose_flow_conv7x7_relu_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _i, _i, _p]),
    "rnnpose_flow_features_f32": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _p, _i, _i, _p, _i, _i, _i, _i, _f, _p]),
    "rnnpose_flow_features_induced_f32": (_i, [_p, _p, _p, _i, _i, _f, _p, _p, _i, _i, _i, _i, _p, _i, _i, _p, _i, _i, _i, _i, _f, _p]),
    "rnnpose_corr_lookup_induced_nhwc_part_f32": (_i, [_p, _p, _p, _p, _i, _i, _f, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "rnnpose_split_hl_f32": (_i, [_p, _i, _i, _ll, _i, _f, _p, _i, _i, _p]),
    "rnnpose_flow_head_out_f32": (_i, [_p, _i, _i, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "rnnpose_convex_upsample_nhwc_f32": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "rnnpose_mask_upsample_f16x3": (_i, [_p, _i, _i, _p, _i, _p, _f, _f, _p, _i, _i, _i, _p, _p]),
    "rnnpose_mask_upsample_packed_bytes": (_z, []),
    "rnnpose_conv1x1_resident_packed_bytes": (_z, [_i]),
    "rnnpose_conv1x1_resident_pack_f16x3": (_i, [_p, _i, _i, _f, _p, _p]),
    "rnnpose_conv1x1_resident_f16x3": (_i, [_p, _i, _i, _i, _p, _p, _f, _f, _i, _ll, _p, _i, _i, _i, _p]),
    "rnnpose_corr_lookup_convc1_f16x3": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _f, _f, _i, _p, _i, _i, _i, _p]),
    "rnnpose_mask_upsample_pack_f16x3": (_i, [_p, _f, _f, _p, _p]),
    "rnnpose_instnorm_workspace_bytes": (_z, [_i, _i, _i]),
    "rnnpose_instnorm_tiles_nhwc_f32": (_i, [_p, _i, _i, _i, _f, _i, _p, _p, _i, _p, _i, _p, _p, _p]),
    "rnnpose_instnorm_nhwc_f32": (_i, [_p, _i, _i, _i, _f, _i, _p, _p, _z, _p, _p, _p]),
    "rnnpose_nn_search_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "findNearestPointIdxLauncher": (None, [_p, _p, _p, _i, _i, _i, _i, _i]),
    "rnnpose_pointcloud_depth_workspace_bytes": (_z, [_i, _i, _i]),
    "rnnpose_pointcloud_depth_f32": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _p, _z, _p, _p]),
    "rnnpose_mask_bbox_f32": (_i, [_p, _i, _i, _i, _p, _p]),
    "rnnpose_zoom_crop_params_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p, _p]),
    "rnnpose_zoom_crop_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "rnnpose_raster_workspace_bytes": (_z, [_i, _i, _i]),
    "rnnpose_raster_mesh_f32": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _f, _f, _i, _p, _z, _p]),
    "rnnpose_raster_resolve_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _i, _p, _p, _p, _i, _p, _i, _i, _f, _p, _p, _p, _p]),
    "rnnpose_pose_metrics_workspace_bytes": (_z, [_i, _i]),
    "rnnpose_pose_metrics_f64": (_i, [_p, _i, _p, _p, _p, _i, _i, _p, _z, _p, _p]),
}

_lib = None


def load() -> C.CDLL:
    """Load (once) and type the library.  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -m rnnpose_amd.build` (hipcc, gfx950). "
            "rnnpose_amd has no CPU fallback for the refinement hot path.")
    # PyTorch-ROCm ships its own libamdhip64 (same SONAME as the system one).  Load torch's FIRST so that this
    # library binds to the runtime that owns torch's device context and streams; loading ours first would pull
    # in /opt/rocm's copy and leave the process with two HIP runtimes (kernel launches then fail with
    # hipErrorNoDevice on streams created by the other one).
    import torch  # noqa: F401
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        C.CDLL(tl, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.rnnpose_abi_version() != ABI_VERSION:
        raise RuntimeError("librnnpose_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(rc: int, name: str) -> None:
    if rc != 0:
        msg = load().rnnpose_last_error().decode(errors="replace")
        raise RuntimeError(f"{name} failed (code {rc}): {msg}")


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)
