"""ctypes binding of the C ABI (include/avr_hip.h) implemented by libavr_hip.so.

The library is built in-tree by ``amrvolumerenderer_amd.build.build()`` (hipcc, gfx950).  There
is no fallback: if the shared object is missing or a HIP device is absent, the compute entry
points raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AVR_HIP_LIBRARY: developer override used by the kernel A/B tools (tools/ab_march.sh) to load
# another build of the SAME library; there is still no fallback of any kind.
LIB_PATH = os.environ.get("AVR_HIP_LIBRARY") or os.path.join(_HERE, "libavr_hip.so")

AVR_OK = 0
AVR_ERR_INVALID_ARGUMENT = -1
AVR_ERR_RUNTIME = -2
AVR_ERR_NO_DEVICE = -3
AVR_ERR_OUT_OF_MEMORY = -4


class AvrError(RuntimeError):
    """std::runtime_error of the reference API (HIP failures included)."""


class AvrNoDevice(AvrError):
    pass


class Box(C.Structure):
    _fields_ = [
        ("min_corner", C.c_double * 3),
        ("max_corner", C.c_double * 3),
        ("dims", C.c_int32 * 3),
        ("level", C.c_int32),
        ("cells", C.c_void_p),
        ("jstride", C.c_int64),
        ("kstride", C.c_int64),
    ]


class ScalarTransform(C.Structure):
    _fields_ = [
        ("log_scale_input", C.c_int32),
        ("normalize_to_unit_range", C.c_int32),
        ("positive_floor", C.c_double),
        ("normalization_min", C.c_double),
        ("inverse_normalization_span", C.c_double),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("eye", C.c_double * 3),
        ("look_at", C.c_double * 3),
        ("up", C.c_double * 3),
        ("fov_y_degrees", C.c_float),
        ("near_plane", C.c_float),
        ("far_plane", C.c_float),
    ]


class ColormapPoint(C.Structure):
    _fields_ = [("value", C.c_float), ("red", C.c_float), ("green", C.c_float),
                ("blue", C.c_float), ("alpha", C.c_float)]


class Speculation(C.Structure):
    """avr_speculation (include/avr_hip.h): device pointers of a speculative frame's flags."""
    _fields_ = [("classified", C.c_void_p), ("visited", C.c_void_p), ("missed", C.c_void_p),
                ("miss_count", C.c_void_p), ("host_miss_flag", C.c_void_p), ("gate", C.c_void_p),
                ("classified_host", C.c_void_p), ("dirty_workgroups", C.c_void_p)]


class PaintParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("scalar_range", C.c_float * 2),
        ("box_transparency", C.c_float),
        ("reference_sample_distance", C.c_float),
        ("bounds_min", C.c_double * 3),
        ("bounds_max", C.c_double * 3),
        ("colormap", C.POINTER(ColormapPoint)),
        ("colormap_count", C.c_int32),
    ]


class FramePlanInfo(C.Structure):
    _fields_ = [
        ("n_ranks", C.c_int32), ("rank", C.c_int32), ("n_runs_total", C.c_int32),
        ("n_local_runs", C.c_int32), ("n_local_boxes", C.c_int32),
        ("n_pixels", C.c_int64), ("piece_begin", C.c_int64), ("piece_end", C.c_int64),
        ("send_floats", C.c_int64), ("recv_floats", C.c_int64),
        ("piece_layout", C.c_int32), ("band_rows", C.c_int32),
    ]


PIECES_CONTIGUOUS, PIECES_ROW_BANDS = 0, 1
DEFAULT_STREAM = C.c_void_p(-1).value    # AVR_DEFAULT_STREAM: the legacy default (null) stream


class RunInfo(C.Structure):
    _fields_ = [("owner", C.c_int32), ("local_run", C.c_int32), ("first_layer", C.c_int32),
                ("n_layers", C.c_int32), ("rect", C.c_int32 * 4)]


class RenderParams(C.Structure):
    """avr_render_params."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("box_transparency", C.c_float),
                ("antialiasing", C.c_int32), ("use_visibility_graph", C.c_int32),
                ("draw_bounds", C.c_int32), ("write_visibility_graph", C.c_int32)]


COMM_ID_BYTES = 128

# name -> (restype, argtypes); every symbol include/avr_hip.h declares.
_vp = C.c_void_p
_i64 = C.c_int64
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
SIGNATURES = {
    "avr_last_error": (C.c_char_p, []),
    "avr_abi_version": (C.c_int, []),
    "avr_context_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "avr_context_destroy": (None, [_vp]),
    "avr_context_set_stream": (C.c_int, [_vp, _vp]),
    "avr_context_set_march_occupancy": (C.c_int, [_vp, C.c_int]),
    "avr_context_set_classify_lds_reserve": (C.c_int, [_vp, C.c_int]),
    "avr_context_set_march_counters": (C.c_int, [_vp, _vp]),
    "avr_context_synchronize": (C.c_int, [_vp]),
    "avr_build_color_table": (C.c_int, [C.c_float, C.c_float, _fp, C.POINTER(ColormapPoint),
                                         C.c_int, _fp]),
    "avr_box_sampling": (C.c_int, [C.POINTER(Box), C.POINTER(PaintParams), _fp, _fp, _fp]),
    "avr_box_depth_hint": (C.c_int, [C.POINTER(Box), C.POINTER(Camera), _fp]),
    "avr_box_footprint": (C.c_int, [C.POINTER(Box), C.POINTER(Camera), C.c_int, C.c_int, _ip, _ip,
                                    _ip]),
    "avr_reference_sample_distance": (C.c_int, [C.POINTER(Box), C.c_int, C.POINTER(C.c_double),
                                                 C.POINTER(C.c_double), _fp]),
    "avr_layer_order": (C.c_int, [_fp, _ip, _ip, C.c_int, _ip, _ip, C.POINTER(C.c_int)]),
    "avr_piece_range": (C.c_int, [_i64, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "avr_paint_box": (C.c_int, [_vp, C.POINTER(Box), C.POINTER(ScalarTransform),
                                 C.POINTER(PaintParams), C.POINTER(Camera), _vp, _vp]),
    "avr_scene_create": (C.c_int, [_vp, C.POINTER(Box), C.c_int, C.POINTER(ScalarTransform),
                                    C.POINTER(_vp)]),
    "avr_scene_set_classification_cache": (C.c_int, [_vp, C.c_int]),
    "avr_scene_invalidate": (C.c_int, [_vp]),
    "avr_scene_destroy": (None, [_vp]),
    "avr_render_runs": (C.c_int, [_vp, _vp, C.POINTER(PaintParams), C.POINTER(Camera), _ip,
                                   C.c_int, _ip, C.c_int, C.c_int, _vp, _vp]),
    "avr_frame_plan_create": (C.c_int, [C.POINTER(Box), _ip, C.c_int, C.c_int, C.c_int, _ip,
                                         C.POINTER(PaintParams), C.POINTER(Camera),
                                         C.POINTER(_vp)]),
    "avr_frame_plan_create_pieces": (C.c_int, [C.POINTER(Box), _ip, C.c_int, C.c_int, C.c_int, _ip,
                                                C.POINTER(PaintParams), C.POINTER(Camera),
                                                C.c_int, C.c_int, C.POINTER(_vp)]),
    "avr_layered_plan_create": (C.c_int, [_fp, _ip, C.c_int, C.c_int, C.c_int, _ip, C.c_int,
                                           C.c_int, C.POINTER(_vp)]),
    "avr_pack_layers": (C.c_int, [_vp, _vp, C.POINTER(_vp), C.c_int, _vp]),
    "avr_frame_plan_destroy": (None, [_vp]),
    "avr_frame_plan_get_info": (C.c_int, [_vp, C.POINTER(FramePlanInfo)]),
    "avr_frame_plan_splits": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "avr_frame_plan_layers": (C.c_int, [_vp, _ip]),
    "avr_frame_plan_runs": (C.c_int, [_vp, C.POINTER(RunInfo)]),
    "avr_frame_plan_tighten": (C.c_int, [_vp, C.POINTER(Box), C.c_int]),
    "avr_frame_plan_send_block": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_i64), _ip, _ip]),
    "avr_frame_plan_recv_block": (C.c_int, [_vp, C.c_int, C.POINTER(_i64), _ip, _ip]),
    "avr_render_plan": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "avr_classify_plan": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "avr_march_plan": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    "avr_classify_plan_chunked": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.POINTER(_vp), C.c_int]),
    "avr_march_plan_chunked": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.POINTER(_vp)]),
    "avr_render_plan_culled": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp]),
    "avr_classify_plan_flagged": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    "avr_classify_plan_positions": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, C.c_int]),
    "avr_march_plan_workgroups": (C.c_int, [_vp, _vp]),
    "avr_march_plan_speculative": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    "avr_fold_plan": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "avr_fold_plan_own": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "avr_fold_plan_image": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "avr_visibility_graph_create": (C.c_int, [C.POINTER(Box), C.POINTER(C.c_int32), C.c_int,
                                               C.c_int, C.POINTER(_vp)]),
    "avr_visibility_graph_destroy": (None, [_vp]),
    "avr_visibility_order": (C.c_int, [_vp, C.POINTER(Camera), C.c_float, C.c_int, C.c_char_p,
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]),
    "avr_tight_bounds": (C.c_int, [C.POINTER(Box), C.c_int, C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double)]),
    "avr_bbox_overlay": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(Camera), C.c_int, C.c_int, C.c_int, _i64, _i64, _vp,
                                    _vp]),
    "avr_bbox_overlay_piece": (C.c_int, [_vp, _vp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(Camera), _vp, _vp]),
    "avr_assemble_rows": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "avr_scene_scalar_stats": (C.c_int, [_vp, _vp, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "avr_scene_transform_from_stats": (C.c_int, [C.POINTER(C.c_double), _i64, C.c_int, C.c_int,
                                                  C.POINTER(ScalarTransform),
                                                  C.POINTER(C.c_double), _fp, _fp]),
    "avr_scene_histogram": (C.c_int, [_vp, _vp, C.POINTER(ScalarTransform), C.c_float, C.c_float,
                                       C.c_int, _vp]),
    "avr_blend_depthsort_f32x5": (C.c_int, [_vp, _vp, _vp, _vp, _i64]),
    "avr_blend_rgba_f32x4": (C.c_int, [_vp, _vp, _vp, _vp, _i64]),
    "avr_blend_rgba_u8x4": (C.c_int, [_vp, _vp, _vp, _vp, _i64]),
    "avr_blend_regions": (C.c_int, [_vp, C.c_int, _vp, _i64, _i64, _vp, _i64, _i64, _vp]),
    "avr_encode_rgba_u8": (C.c_int, [_vp, _vp, _vp, _i64]),
    "avr_decode_rgba_u8": (C.c_int, [_vp, _vp, _vp, _i64]),
    "avr_fold_runs_depthsort": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, _vp, _i64]),
    "avr_downsample_depthsort": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "avr_quantize_rgb8": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "avr_flip_rows": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp]),
    "avr_context_create_with_priority": (C.c_int, [C.c_int, C.c_int, C.POINTER(_vp)]),
    "avr_context_stream": (_vp, [_vp]),
    "avr_comm_unique_id": (C.c_int, [C.c_char_p]),
    "avr_comm_create": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(_vp)]),
    "avr_comm_create_local": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "avr_comm_create_solo": (C.c_int, [C.c_int, C.c_int, C.POINTER(_vp)]),
    "avr_comm_create_solo_rccl": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "avr_comm_destroy": (None, [_vp]),
    "avr_comm_rank": (C.c_int, [_vp]),
    "avr_comm_size": (C.c_int, [_vp]),
    "avr_exchange": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "avr_exchange_peers": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "avr_gather": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int]),
    "avr_renderer_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.POINTER(Box), _ip,
                                       C.c_int, C.POINTER(ScalarTransform), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), _fp, C.POINTER(ColormapPoint),
                                       C.c_int, C.POINTER(_vp)]),
    "avr_renderer_destroy": (None, [_vp]),
    "avr_renderer_set_options": (C.c_int, [_vp, C.c_int, C.c_int]),
    "avr_renderer_set_scalar_range": (C.c_int, [_vp, _fp]),
    "avr_renderer_invalidate": (C.c_int, [_vp]),
    "avr_renderer_set_overlap": (C.c_int, [_vp, C.c_int]),
    "avr_comm_create_shared": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_size_t, C.POINTER(_vp)]),
    "avr_exchange_pieces": (C.c_int, [_vp, _vp, _ip, _i64, C.c_int, _vp, _vp]),
    "avr_renderer_set_host_backpressure": (C.c_int, [_vp, C.c_int]),
    "avr_renderer_set_tighten": (C.c_int, [_vp, C.c_int]),
    "avr_renderer_set_piece_layout": (C.c_int, [_vp, C.c_int, C.c_int]),
    "avr_renderer_set_classify_share": (C.c_int, [_vp, C.c_int]),
    "avr_renderer_corun_state": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                           C.POINTER(C.c_int), C.POINTER(C.c_long)]),
    "avr_renderer_reference_sample_distance": (C.c_int, [_vp, _fp]),
    "avr_renderer_render": (C.c_int, [_vp, C.POINTER(RenderParams), C.POINTER(Camera), _ip, _vp,
                                       _vp, C.c_int, _vp, _vp]),
    "avr_renderer_prepare": (C.c_int, [_vp, C.POINTER(RenderParams), C.POINTER(Camera), _ip]),
    "avr_renderer_synchronize": (C.c_int, [_vp]),
    "avr_renderer_set_frame_chunks": (C.c_int, [_vp, C.c_int]),
    "avr_renderer_set_occlusion_culling": (C.c_int, [_vp, C.c_int]),
    "avr_renderer_set_corun_balance": (C.c_int, [_vp, C.c_int]),
    "avr_renderer_last_frame_chunks": (C.c_int, [_vp]),
    "avr_renderer_set_visibility_speculation": (C.c_int, [_vp, C.c_int]),
    "avr_renderer_speculation_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "avr_renderer_debug_set_speculation_threshold": (C.c_int, [_vp, C.c_float]),
    "avr_renderer_outputs_complete": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "avr_renderer_stream": (_vp, [_vp, C.c_int]),
    "avr_renderer_plan_info": (C.c_int, [_vp, C.POINTER(FramePlanInfo)]),
    "avr_renderer_host_profile": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_long), C.c_int]),
    "avr_renderer_set_timing": (C.c_int, [_vp, C.c_int]),
    "avr_renderer_timings": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "avr_set_frame_timeout_ms": (C.c_int, [C.c_int]),
    "avr_debug_stall_stream": (C.c_int, [_vp, C.c_int]),
    "avr_comm_set_control": (C.c_int, [_vp, _vp, _vp]),
    "avr_comm_control_allgather": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int]),
    "avr_comm_control_rounds": (C.c_long, [_vp]),
    "avr_frame_plan_agree": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "avr_renderer_set_plan_check": (C.c_int, [_vp, C.c_int]),
    "avr_renderer_set_deferred_gather": (C.c_int, [_vp, C.c_int]),
    "avr_frame_plan_piece_ranges": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "avr_gather_run": (C.c_int, [_vp, _vp, _vp]),
    "avr_exchange_peers_gather": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "avr_assemble_rows_own": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "avr_renderer_set_corun_coordination": (C.c_int, [_vp, C.c_int]),
    "avr_renderer_set_corun_history": (C.c_int, [_vp, C.c_int]),
    "avr_renderer_corun_history": (C.c_int, [_vp, C.POINTER(C.c_int16), C.c_int,
                                             C.POINTER(C.c_int)]),
    "avr_renderer_failure": (C.c_char_p, [_vp]),
}

CONTROL_MAX_BYTES = 2048
# int (*avr_control_allgather_fn)(void *user, const void *mine, void *all, int bytes_per_rank)
CONTROL_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)

_lib = None


def lib():
    """The loaded shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AvrError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # One HIP runtime per process: torch (device memory, streams, RCCL for the Python layer)
        # ships its own libamdhip64 and must be loaded first, so that this library binds to the
        # same runtime instead of bringing the system copy in beside it -- with two runtimes in
        # one process the one loaded second reports no device.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def last_error() -> str:
    return lib().avr_last_error().decode("utf-8", "replace")


def check(status: int) -> None:
    """Maps a status code to the exception the reference API raises for it."""
    if status == AVR_OK:
        return
    message = last_error()
    if status == AVR_ERR_INVALID_ARGUMENT:
        raise ValueError(message)          # std::invalid_argument
    if status == AVR_ERR_NO_DEVICE:
        raise AvrNoDevice(message)
    if status == AVR_ERR_OUT_OF_MEMORY:
        raise MemoryError(message)
    raise AvrError(message)                # std::runtime_error
