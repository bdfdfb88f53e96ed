"""ctypes binding of libogs_hip.so (C ABI: include/ogs_raster.h, include/ogs_kmeans.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load, ``lib()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OGS_LIB_PATH: another build of the same library (A-B timing of two kernel versions on one box); never a fallback
LIB_PATH = os.environ.get("OGS_LIB_PATH") or os.path.join(_HERE, "lib", "libogs_hip.so")

_f32p = C.POINTER(C.c_float)
_vp = C.c_void_p


class OgsRasterFwdArgs(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("W", C.c_int32), ("H", C.c_int32), ("C", C.c_int32),
        ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("bg", _vp), ("means3D", _vp), ("colors_precomp", _vp), ("shs", _vp), ("opacities", _vp),
        ("scales", _vp), ("rotations", _vp), ("cov3D_precomp", _vp),
        ("viewmatrix", _vp), ("projmatrix", _vp), ("campos", _vp),
        ("out_color", _vp), ("out_depth", _vp), ("out_alpha", _vp), ("radii", _vp),
        ("geom_buffer", _vp), ("geom_tmp", _vp), ("image_buffer", _vp), ("point_list", _vp),
        ("binning_tmp", _vp), ("sorted_rec", _vp), ("quad_list", _vp), ("group_ids", _vp), ("num_groups", C.c_int32),
        ("full_binning", C.c_int32),
    ]


class OgsRasterBwdArgs(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("W", C.c_int32), ("H", C.c_int32), ("C", C.c_int32),
        ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("debug", C.c_int32), ("num_rendered", C.c_int32), ("geom_channels", C.c_int32),
        ("bg", _vp), ("means3D", _vp), ("colors_precomp", _vp), ("shs", _vp), ("opacities", _vp),
        ("scales", _vp), ("rotations", _vp), ("cov3D_precomp", _vp),
        ("viewmatrix", _vp), ("projmatrix", _vp), ("campos", _vp),
        ("radii", _vp), ("out_alpha", _vp), ("dL_dcolor", _vp), ("dL_ddepth", _vp), ("dL_dalpha", _vp),
        ("geom_buffer", _vp), ("image_buffer", _vp), ("point_list", _vp), ("sorted_rec", _vp), ("quad_list", _vp), ("bwd_tmp", _vp),
        ("dL_dmeans2D", _vp), ("dL_dcolors", _vp), ("dL_dopacity", _vp), ("dL_dmeans3D", _vp),
        ("dL_dcov3D", _vp), ("dL_dsh", _vp), ("dL_dscales", _vp), ("dL_drotations", _vp), ("num_groups", C.c_int32), ("dL_dsh_rgb", _vp),
    ]


class OgsAdamTensor(C.Structure):
    _fields_ = [("param", _vp), ("grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("numel", C.c_int64),
                ("lr", C.c_double), ("step", C.c_int64)]


class OgsRowTensor(C.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("width", C.c_int32), ("zero_new", C.c_int32)]


class OgsDensifyArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("grad_accum", _vp), ("denom", _vp), ("scaling", _vp), ("opacity", _vp),
                ("max_grad", C.c_float), ("min_opacity", C.c_float), ("extent", C.c_float), ("percent_dense", C.c_float),
                ("prune_world_size", C.c_int32)]


# name -> (restype, argtypes); also the list the symbol-export test checks against the headers
SIGNATURES = {
    "ogs_version": (C.c_int, []),
    "ogs_last_error": (C.c_char_p, []),
    "ogs_raster_geom_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "ogs_raster_geom_tmp_bytes": (C.c_size_t, [C.c_int32]),
    "ogs_raster_image_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "ogs_raster_image_bytes_grouped": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "ogs_raster_binning_tmp_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "ogs_raster_backward_tmp_bytes": (C.c_size_t, [C.c_int32]),
    "ogs_raster_sorted_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "ogs_raster_quad_list_bytes": (C.c_size_t, [C.c_int64]),
    "ogs_raster_forward_geometry": (C.c_int, [C.POINTER(OgsRasterFwdArgs), _vp, C.POINTER(C.c_int64)]),
    "ogs_raster_forward_render": (C.c_int, [C.POINTER(OgsRasterFwdArgs), C.c_int64, _vp]),
    "ogs_raster_read_num_rendered_async": (C.c_int, [C.POINTER(OgsRasterFwdArgs), _vp, _vp]),
    "ogs_raster_forward_render_deferred": (C.c_int, [C.POINTER(OgsRasterFwdArgs), C.c_int64, _vp]),
    "ogs_raster_backward": (C.c_int, [C.POINTER(OgsRasterBwdArgs), _vp]),
    "ogs_raster_tiny_max_points": (C.c_size_t, []),
    "ogs_raster_forward_tiny": (C.c_int, [C.POINTER(OgsRasterFwdArgs), _vp]),
    "ogs_raster_forward_reblend": (C.c_int, [C.POINTER(OgsRasterFwdArgs), _vp]),
    "ogs_raster_compact_kept": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ogs_mark_visible": (C.c_int, [C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "ogs_sh_grad_from_views": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "ogs_raster_export_binning": (C.c_int, [C.POINTER(OgsRasterFwdArgs), C.c_int64, _vp, _vp, _vp, _vp]),
    "ogs_selftest_wave_fold16": (C.c_int, [_vp, _vp, _vp]),
    "ogs_selftest_tile_order": (C.c_int, [_vp, C.c_int64, _vp, _vp]),
    "ogs_selftest_radix_tmp_bytes": (C.c_size_t, [C.c_int64]),
    "ogs_selftest_radix_sort": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    "ogs_check_async_status": (C.c_int, []),
    "ogs_prof_enable": (C.c_int, [C.c_int]),
    "ogs_prof_filter": (C.c_int, [C.c_char_p]),
    "ogs_prof_collect": (C.c_int, [C.c_char_p, C.c_size_t]),
    "ogs_kmeans_tmp_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "ogs_kmeans_lloyd": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   _vp, C.c_int64, _vp, _vp]),
    "ogs_kmeans_assign": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int32, _vp, C.c_int64, _vp]),
    "ogs_mask_feature_sums": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _vp, _vp]),
    "ogs_mask_feature_sums_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int64, _vp, _vp, _vp]),
    "ogs_mask_cohesion": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int64, _vp, _vp]),
    "ogs_mask_cohesion_backward": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int64, _vp, _vp, _vp]),
    "ogs_separation_loss": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    "ogs_adam_step": (C.c_int, [C.POINTER(OgsAdamTensor), C.c_int32, C.c_double, C.c_double, C.c_double, _vp]),
    "ogs_rows_gather": (C.c_int, [C.POINTER(OgsRowTensor), C.c_int32, _vp, _vp, C.c_int64, _vp]),
    "ogs_densify_tmp_bytes": (C.c_size_t, [C.c_int32]),
    "ogs_densify_plan": (C.c_int, [C.POINTER(OgsDensifyArgs), _vp, C.POINTER(C.c_uint32), _vp]),
    "ogs_densify_map": (C.c_int, [C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "ogs_densify_split_children": (C.c_int, [C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ogs_densify_stats": (C.c_int, [C.c_int32, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ogs_kmeans_accumulate": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "ogs_kmeans_update": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "ogs_kmeans_gather": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, C.c_int32, _vp, _vp]),
}

_lib = None


class OgsError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libogs_hip.so once.  Raises (loudly) when it is missing: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OgsError(f"{LIB_PATH} not found -- build it with `python -m opengaussian_amd.build` "
                           "(the product path has no CPU fallback)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise OgsError(f"{what} failed (code {rc}): {lib().ogs_last_error().decode()}")


def prof_filter(prefix: str):
    """mode 2 of prof_enable brackets the kernels whose name starts with `prefix` (default "blend_")."""
    check(lib().ogs_prof_filter(prefix.encode()), "ogs_prof_filter")


def prof_enable(mode):
    """0/False: off, 1/True: every launch, 2: only the blend kernels (cheap enough for a timed region)."""
    check(lib().ogs_prof_enable(int(mode)), "ogs_prof_enable")


def prof_collect() -> dict:
    """{kernel name: {"calls": n, "total_ms": t}} for launches since prof_enable(True)."""
    import json
    buf = C.create_string_buffer(1 << 16)
    check(lib().ogs_prof_collect(buf, len(buf)), "ogs_prof_collect")
    return json.loads(buf.value.decode())


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()
