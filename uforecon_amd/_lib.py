"""ctypes binding of libufr.so (include/ufr.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# UFR_LIB selects an alternative in-tree build (A/B kernel variants during development)
LIB_PATH = os.environ.get("UFR_LIB") or os.path.join(HERE, "lib", "libufr.so")

ABI_VERSION = 503   # = UFR_ABI_VERSION of include/ufr.h; load() refuses a library built against another header
MAX_VIEWS = 7
NUM_STAGES = 3
TOKEN_DIM = 80
RAY_DIM = 88

fptr = C.c_void_p  # device pointers travel as integers


class LayerWeights(C.Structure):
    _fields_ = [(n, fptr) for n in ("q", "k", "v", "merge", "mlp0", "mlp2", "norm1_w", "norm1_b", "norm2_w", "norm2_b")]


class Mlp3Weights(C.Structure):
    _fields_ = [(n, fptr) for n in ("w0", "b0", "w2", "b2", "w4", "b4")]


class RawWeights(C.Structure):
    _fields_ = [("pre_sim", Mlp3Weights), ("view", LayerWeights), ("ray", LayerWeights),
                ("density", Mlp3Weights), ("radiance", Mlp3Weights), ("view_token", fptr), ("variance", fptr)]


class RawGrads(C.Structure):
    _fields_ = [("p", fptr * 40)]


class FmtLayerWeights(C.Structure):
    _fields_ = [(n, fptr) for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "w1", "b1", "w2", "b2",
                                    "n1w", "n1b", "n2w", "n2b")]


class FrameDesc(C.Structure):
    _fields_ = [("NV", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("source_imgs", fptr), ("depth_info", fptr), ("feat", fptr), ("match", fptr),
                ("vol_feat", fptr * NUM_STAGES), ("vol_weight", fptr * NUM_STAGES),
                ("vol_D", C.c_int32 * NUM_STAGES), ("vol_H", C.c_int32 * NUM_STAGES), ("vol_W", C.c_int32 * NUM_STAGES),
                ("source_poses", C.POINTER(C.c_float)), ("source_cam_pos", C.POINTER(C.c_float)),
                ("ref_cam_pos", C.POINTER(C.c_float)), ("w2c_row2", C.POINTER(C.c_float)),
                ("vol_near", C.c_float), ("vol_far", C.c_float)]


class Frame(C.Structure):
    _fields_ = [("opaque", C.c_uint64 * 160)]


class RenderArgs(C.Structure):
    _fields_ = [("frame", C.POINTER(Frame)), ("packed_weights", fptr), ("raw", C.POINTER(RawWeights)),
                ("ray_idx", fptr), ("ray_d", fptr), ("cam_ray_d", fptr),
                ("ray_o", C.c_float * 3), ("near_z", C.c_float), ("far_z", C.c_float),
                ("U1", fptr), ("U2", fptr), ("RN", C.c_int32), ("SN", C.c_int32), ("PN", C.c_int32),
                ("coarse_only", C.c_int32), ("precision", C.c_int32),
                ("depth", fptr), ("depth_z", fptr), ("rgb", fptr), ("srdf", fptr), ("z_all", fptr),
                ("chunk_rays", C.c_int32), ("n_streams", C.c_int32), ("workspace", fptr), ("workspace_bytes", C.c_size_t)]


# name -> (restype, argtypes); every symbol include/ufr.h declares
i32, sz, vp = C.c_int32, C.c_size_t, C.c_void_p
SIGNATURES = {
    "ufr_version": (C.c_int, []),
    "ufr_last_error": (C.c_char_p, []),
    "ufr_set_matrix_precision": (C.c_int, [C.c_int]),
    "ufr_get_matrix_precision": (C.c_int, []),
    "ufr_status_poll": (C.c_int, [vp, i32, C.POINTER(i32)]),
    "ufr_status_poll_bits": (C.c_int, [vp, i32, i32, C.POINTER(i32)]),
    "ufr_packed_weights_bytes": (sz, []),
    "ufr_weights_pack": (C.c_int, [C.POINTER(RawWeights), vp, vp]),
    "ufr_weights_pack_for": (C.c_int, [C.POINTER(RawWeights), vp, C.c_float, vp]),
    "ufr_weights_fit_frame": (C.c_int, [vp, C.POINTER(Frame), vp]),
    "ufr_packed_scale_table_offset": (sz, []),
    "ufr_packed_scale_table_entries": (C.c_int, []),
    "ufr_pack_plan": (C.c_int, [C.POINTER(i32), C.POINTER(i32)]),
    "ufr_packed_fp32_floats": (sz, []),
    "ufr_packed_f16_halfwords": (sz, []),
    "ufr_packed_bwd_halfwords": (sz, []),
    "ufr_pack_plan_bwd": (C.c_int, [C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "ufr_pack_plan_f16": (C.c_int, [C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "ufr_frame_workspace_bytes": (sz, [C.POINTER(FrameDesc)]),
    "ufr_frame_prepare": (C.c_int, [C.POINTER(FrameDesc), vp, sz, C.POINTER(Frame), vp]),
    "ufr_sample_fixed": (C.c_int, [vp, vp, vp, vp, i32, i32, vp]),
    "ufr_sample_importance_merge": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "ufr_points": (C.c_int, [vp, i32, vp, vp, vp, i32, i32, vp]),
    "ufr_project_gather": (C.c_int, [C.POINTER(Frame), C.POINTER(RawWeights), vp, i32, vp, vp, i32, i32,
                                     vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "ufr_aggregate_workspace_bytes": (sz, [i32, i32, i32]),
    "ufr_aggregate": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp]),
    "ufr_composite": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "ufr_composite_bwd": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp]),
    "ufr_render_loss": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp]),
    "ufr_aggregate_bwd_workspace_bytes": (sz, [i32, i32, i32]),
    "ufr_aggregate_bwd": (C.c_int, [C.POINTER(RawWeights), C.POINTER(RawGrads), vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp,
                                    vp, i32, vp]),
    "ufr_project_gather_bwd_workspace_bytes": (sz, [C.POINTER(Frame)]),
    "ufr_project_gather_bwd": (C.c_int, [C.POINTER(Frame), C.POINTER(RawWeights), C.POINTER(RawGrads), vp, i32, vp, vp,
                                         i32, i32, vp, vp, vp, C.POINTER(vp), C.POINTER(vp), i32, vp, i32, vp]),
    "ufr_sample_importance_pool": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "ufr_view_transform": (C.c_int, [vp, vp, vp, vp, i32, i32, vp, vp, i32, vp]),
    "ufr_ray_transform_workspace_bytes": (sz, [i32]),
    "ufr_ray_transform": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, i32, vp]),
    "ufr_ray_transform_bwd_workspace_bytes": (sz, [i32, i32]),
    "ufr_ray_transform_bwd": (C.c_int, [C.POINTER(RawWeights), C.POINTER(RawGrads), vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, i32, vp]),
    "ufr_view_transform_bwd_workspace_bytes": (sz, [i32, i32]),
    "ufr_view_transform_bwd": (C.c_int, [C.POINTER(RawWeights), C.POINTER(RawGrads), vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, vp]),
    "ufr_view_tape_block_points": (i32, [i32]),
    "ufr_view_transform_tape": (C.c_int, [vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, i32, vp]),
    "ufr_ray_transform_tape": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, i32, vp]),
    "ufr_ray_transform_bwd_stages": (C.c_int, [C.POINTER(RawWeights), C.POINTER(RawGrads), vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, i32, i32, vp]),
    "ufr_view_transform_bwd_stages": (C.c_int, [C.POINTER(RawWeights), C.POINTER(RawGrads), vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, vp]),
    "ufr_render_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "ufr_default_chunk_rays": (i32, []),
    "ufr_render_rays": (C.c_int, [C.POINTER(RenderArgs), vp]),
    "ufr_correlate_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "ufr_frustum_correlate": (C.c_int, [vp, vp, C.POINTER(C.c_float), vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    "ufr_conv3d": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "ufr_conv3d_planes_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "ufr_conv3d_planes": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp,
                                    sz, i32, vp]),
    "ufr_absmax": (C.c_int, [vp, sz, vp, vp]),
    "ufr_conv3d_bwd_data": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ufr_conv3d_bwd_weight": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ufr_conv3d_bwd_weight_heads": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ufr_tsdf_integrate": (C.c_int, [vp, vp, vp, C.POINTER(i32), C.POINTER(C.c_float), C.c_float, C.c_float,
                                     C.POINTER(C.c_float), C.POINTER(C.c_float), vp, vp, i32, i32, C.c_float, i32, vp]),
    "ufr_pixelwise_view_weights": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ufr_deform_conv2d_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "ufr_deform_conv2d": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]),
    "ufr_conv2d": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ufr_upsample_add": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "ufr_deform_conv2d_cl": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ufr_fmt_layer_workspace_bytes": (sz, [i32, i32]),
    "ufr_fmt_layer": (C.c_int, [C.POINTER(FmtLayerWeights), vp, vp, i32, i32, i32, vp, vp, vp]),
    "ufr_profile_enable": (None, [C.c_int]),
    "ufr_profile_read": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(i32), C.c_int]),
}

_lib = None


class UfrError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libufr.so (built by uforecon_amd.build).  Raises if it is absent: there is no CPU path."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64.so.7; it must be mapped first so that libufr.so binds to
        # the SAME HIP runtime instance (loading /opt/rocm's copy first leaves torch and the kernels on
        # two runtimes: "no ROCm-capable device is detected")
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            raise UfrError(f"{LIB_PATH} not found: build it with `python -m uforecon_amd.build` "
                           "(the per-ray path has no non-HIP implementation)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if lib.ufr_version() != ABI_VERSION:   # same symbols, different argument lists: refuse instead of mis-calling
            raise UfrError(f"{LIB_PATH} has ABI version {lib.ufr_version()}, this binding needs {ABI_VERSION}: rebuild it "
                           "(`python -m uforecon_amd.build`)")
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ufr_last_error().decode()
        raise UfrError(f"{what or 'libufr'} failed ({rc}): {msg}")
