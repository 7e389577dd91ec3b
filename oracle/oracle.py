"""ctypes binding of the CPU oracle (oracle/avr_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package amrvolumerenderer_amd never does.

Pin status: the over-blends and region logic are pinned by the reference's own fixtures, the
layered compose by properties measured on the reference; the painter (K1), colour table and
the SURVEY 8(f) functions are PARITY UNPINNED (avr_oracle.h, oracle/README.md).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile (gcc, CPU only)."""
    src = os.path.join(_HERE, "avr_oracle.c")
    hdr = os.path.join(_HERE, "avr_oracle.h")
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Box(C.Structure):
    _fields_ = [
        ("min_corner", C.c_double * 3),
        ("max_corner", C.c_double * 3),
        ("dims", C.c_int * 3),
        ("cells", C.c_void_p),
        ("jstride", C.c_int64),
        ("kstride", C.c_int64),
    ]


class Transform(C.Structure):
    _fields_ = [
        ("log_scale_input", C.c_int),
        ("normalize_to_unit_range", C.c_int),
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


class PaintParams(C.Structure):
    _fields_ = [
        ("width", C.c_int),
        ("height", C.c_int),
        ("scalar_range", C.c_float * 2),
        ("box_transparency", C.c_float),
        ("reference_sample_distance", C.c_float),
        ("bounds_min", C.c_double * 3),
        ("bounds_max", C.c_double * 3),
        ("colormap", C.POINTER(ColormapPoint)),
        ("colormap_count", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.orc_build_color_table.argtypes = [C.c_float, C.c_float, fp, C.POINTER(ColormapPoint),
                                            C.c_int, fp]
        L.orc_build_color_table.restype = None
        L.orc_box_sampling.argtypes = [C.POINTER(Box), C.POINTER(PaintParams), fp, fp, fp]
        L.orc_box_sampling.restype = None
        L.orc_paint_box.argtypes = [C.POINTER(Box), C.POINTER(Transform), C.POINTER(PaintParams),
                                    C.POINTER(Camera), fp, C.c_int]
        L.orc_paint_box.restype = C.c_uint64
        L.orc_paint_box_window.argtypes = [C.POINTER(Box), C.POINTER(Transform),
                                           C.POINTER(PaintParams), C.POINTER(Camera), C.c_int,
                                           C.c_int, C.c_int, C.c_int, fp, C.c_int]
        L.orc_paint_box_window.restype = C.c_uint64
        L.orc_box_depth_hint.argtypes = [C.POINTER(Box), C.POINTER(Camera)]
        L.orc_box_depth_hint.restype = C.c_float
        L.orc_reference_sample_distance.argtypes = [C.POINTER(Box), C.c_int,
                                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_reference_sample_distance.restype = C.c_float
        L.orc_apply_scalar_transform.argtypes = [C.c_double, C.POINTER(Transform)]
        L.orc_apply_scalar_transform.restype = C.c_float
        for name in ("orc_blend_depthsort", "orc_blend_rgba_f32", "orc_blend_rgba_u8"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
            getattr(L, name).restype = None
        L.orc_blend_regions.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_int, C.c_int, C.c_void_p]
        L.orc_blend_regions.restype = None
        L.orc_encode_rgba_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_encode_rgba_u8.restype = None
        L.orc_decode_rgba_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_decode_rgba_u8.restype = None
        L.orc_piece_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]
        L.orc_piece_range.restype = None
        L.orc_compose_layered.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_int, C.c_void_p, C.c_void_p]
        L.orc_compose_layered.restype = C.c_int
        L.orc_layer_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p]
        L.orc_layer_order.restype = C.c_int
        L.orc_downsample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_downsample.restype = None
        L.orc_quantize_rgb8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_quantize_rgb8.restype = None
        L.orc_scalar_stats.argtypes = [C.POINTER(Box), C.c_int, C.POINTER(C.c_double)]
        L.orc_scalar_stats.restype = C.c_int64
        L.orc_scene_transform.argtypes = [C.POINTER(C.c_double), C.c_int64, C.c_int, C.c_int,
                                          C.POINTER(Transform), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), fp, fp]
        L.orc_scene_transform.restype = C.c_int
        L.orc_histogram.argtypes = [C.POINTER(Box), C.c_int, C.POINTER(Transform), C.c_float,
                                    C.c_float, C.c_int, C.c_void_p]
        L.orc_histogram.restype = None
        dp = C.POINTER(C.c_double)
        L.orc_tight_bounds.argtypes = [C.POINTER(Box), C.c_int, dp, dp, dp, dp]
        L.orc_tight_bounds.restype = None
        L.orc_bbox_overlay.argtypes = [dp, dp, C.POINTER(Camera), C.c_int, C.c_void_p, C.c_int,
                                       C.c_int]
        L.orc_bbox_overlay.restype = None
        L.orc_set_shim_variant.argtypes = [C.c_int]
        L.orc_set_shim_variant.restype = None
        L.orc_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_fnv1a64.restype = C.c_uint64
        _lib = L
    return _lib


# ---------------------------------------------------------------------------------------------
# numpy-level helpers
# ---------------------------------------------------------------------------------------------

def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def make_box(cells: np.ndarray, min_corner, max_corner) -> Box:
    """cells: float64 array indexed [k, j, i] (x fastest), C-contiguous or a view with unit
    i-stride.  The Box keeps a reference to the array."""
    assert cells.dtype == np.float64 and cells.ndim == 3
    assert cells.strides[2] == 8, "x must be the fastest axis"
    b = Box()
    for c in range(3):
        b.min_corner[c] = float(min_corner[c])
        b.max_corner[c] = float(max_corner[c])
    nz, ny, nx = cells.shape
    b.dims[0], b.dims[1], b.dims[2] = nx, ny, nz
    b.cells = cells.ctypes.data
    b.jstride = cells.strides[1] // 8
    b.kstride = cells.strides[0] // 8
    b._keep = cells
    return b


def make_transform(log_scale=False, normalize=True, positive_floor=0.0, norm_min=0.0,
                   inv_norm_span=1.0) -> Transform:
    t = Transform()
    t.log_scale_input = int(bool(log_scale))
    t.normalize_to_unit_range = int(bool(normalize))
    t.positive_floor = float(positive_floor)
    t.normalization_min = float(norm_min)
    t.inverse_normalization_span = float(inv_norm_span)
    return t


def make_camera(eye, look_at, up=(0.0, 1.0, 0.0), fov_y=45.0, near=0.1, far=1000.0) -> Camera:
    cam = Camera()
    for c in range(3):
        cam.eye[c] = float(eye[c])
        cam.look_at[c] = float(look_at[c])
        cam.up[c] = float(up[c])
    cam.fov_y_degrees = float(fov_y)
    cam.near_plane = float(near)
    cam.far_plane = float(far)
    return cam


def make_params(width, height, scalar_range=(0.0, 1.0), box_transparency=0.0,
                reference_sample_distance=0.0, bounds_min=(-0.05,) * 3, bounds_max=(1.05,) * 3,
                colormap: Optional[Sequence[Sequence[float]]] = None) -> PaintParams:
    p = PaintParams()
    p.width, p.height = int(width), int(height)
    p.scalar_range[0], p.scalar_range[1] = float(scalar_range[0]), float(scalar_range[1])
    p.box_transparency = float(box_transparency)
    p.reference_sample_distance = float(reference_sample_distance)
    for c in range(3):
        p.bounds_min[c] = float(bounds_min[c])
        p.bounds_max[c] = float(bounds_max[c])
    if colormap:
        arr = (ColormapPoint * len(colormap))()
        for i, pt in enumerate(colormap):
            arr[i].value, arr[i].red, arr[i].green, arr[i].blue, arr[i].alpha = map(float, pt)
        p.colormap = C.cast(arr, C.POINTER(ColormapPoint))
        p.colormap_count = len(colormap)
        p._keep = arr
    else:
        p.colormap = None
        p.colormap_count = 0
    return p


def build_color_table(alpha_scale, normalization_factor, scalar_range=(0.0, 1.0),
                      colormap=None) -> np.ndarray:
    out = np.empty(1024, dtype=np.float32)
    rng = (C.c_float * 2)(float(scalar_range[0]), float(scalar_range[1]))
    if colormap:
        arr = (ColormapPoint * len(colormap))()
        for i, pt in enumerate(colormap):
            arr[i].value, arr[i].red, arr[i].green, arr[i].blue, arr[i].alpha = map(float, pt)
        lib().orc_build_color_table(alpha_scale, normalization_factor, rng, arr, len(colormap),
                                    out.ctypes.data_as(C.POINTER(C.c_float)))
    else:
        lib().orc_build_color_table(alpha_scale, normalization_factor, rng, None, 0,
                                    out.ctypes.data_as(C.POINTER(C.c_float)))
    return out.reshape(256, 4)


def box_sampling(box: Box, params: PaintParams):
    sd, nf, als = C.c_float(), C.c_float(), C.c_float()
    lib().orc_box_sampling(C.byref(box), C.byref(params), C.byref(sd), C.byref(nf), C.byref(als))
    return sd.value, nf.value, als.value


def paint_box(box: Box, transform: Transform, params: PaintParams, camera: Camera,
              threads: int = 1):
    """Returns (image[H, W, 5] float32, executed cell fetches)."""
    out = np.empty((params.height, params.width, 5), dtype=np.float32)
    n = lib().orc_paint_box(C.byref(box), C.byref(transform), C.byref(params), C.byref(camera),
                            out.ctypes.data_as(C.POINTER(C.c_float)), int(threads))
    return out, int(n)


def paint_box_window(box: Box, transform: Transform, params: PaintParams, camera: Camera,
                     x0: int, y0: int, x1: int, y1: int, threads: int = 1):
    """VolumePainter::paint for the pixel window [x0, x1) x [y0, y1) of the full image.
    Returns (image[y1 - y0, x1 - x0, 5] float32, executed cell fetches in the window)."""
    out = np.empty((y1 - y0, x1 - x0, 5), dtype=np.float32)
    n = lib().orc_paint_box_window(C.byref(box), C.byref(transform), C.byref(params),
                                   C.byref(camera), int(x0), int(y0), int(x1), int(y1),
                                   out.ctypes.data_as(C.POINTER(C.c_float)), int(threads))
    return out, int(n)


def box_depth_hint(box: Box, camera: Camera) -> float:
    return float(lib().orc_box_depth_hint(C.byref(box), C.byref(camera)))


def reference_sample_distance(boxes: Sequence[Box], bounds_min, bounds_max) -> float:
    arr = (Box * len(boxes))(*boxes)
    bmin = (C.c_double * 3)(*map(float, bounds_min))
    bmax = (C.c_double * 3)(*map(float, bounds_max))
    return float(lib().orc_reference_sample_distance(arr, len(boxes), bmin, bmax))


def apply_scalar_transform(raw: float, transform: Transform) -> float:
    return float(lib().orc_apply_scalar_transform(float(raw), C.byref(transform)))


def blend_depthsort(top: np.ndarray, bottom: np.ndarray) -> np.ndarray:
    top, bottom = _f32(top), _f32(bottom)
    out = np.empty_like(top)
    lib().orc_blend_depthsort(top.ctypes.data, bottom.ctypes.data, out.ctypes.data, top.size // 5)
    return out


def blend_rgba_f32(top: np.ndarray, bottom: np.ndarray) -> np.ndarray:
    top, bottom = _f32(top), _f32(bottom)
    out = np.empty_like(top)
    lib().orc_blend_rgba_f32(top.ctypes.data, bottom.ctypes.data, out.ctypes.data, top.size // 4)
    return out


def blend_rgba_u8(top: np.ndarray, bottom: np.ndarray) -> np.ndarray:
    top = np.ascontiguousarray(top, dtype=np.uint32)
    bottom = np.ascontiguousarray(bottom, dtype=np.uint32)
    out = np.empty_like(top)
    lib().orc_blend_rgba_u8(top.ctypes.data, bottom.ctypes.data, out.ctypes.data, top.size)
    return out


_KIND = {"depthsort": (0, np.float32, 5), "rgba_f32": (1, np.float32, 4), "rgba_u8": (2, np.uint32, 1)}


def blend_regions(kind: str, top: np.ndarray, tb: int, te: int, bottom: np.ndarray, bb: int,
                  be: int):
    """ImageColorOnly::blend with regions; returns (out, begin, end)."""
    k, dt, vec = _KIND[kind]
    top = np.ascontiguousarray(top, dtype=dt).reshape(-1)
    bottom = np.ascontiguousarray(bottom, dtype=dt).reshape(-1)
    assert top.size == (te - tb) * vec and bottom.size == (be - bb) * vec
    ob, oe = min(tb, bb), max(te, be)
    out = np.empty((oe - ob) * vec, dtype=dt)
    lib().orc_blend_regions(k, top.ctypes.data, tb, te, bottom.ctypes.data, bb, be,
                            out.ctypes.data)
    return out, ob, oe


def encode_rgba_u8(rgba: np.ndarray) -> np.ndarray:
    rgba = _f32(rgba)
    out = np.empty(rgba.size // 4, dtype=np.uint32)
    lib().orc_encode_rgba_u8(rgba.ctypes.data, out.ctypes.data, out.size)
    return out


def decode_rgba_u8(enc: np.ndarray) -> np.ndarray:
    enc = np.ascontiguousarray(enc, dtype=np.uint32)
    out = np.empty((enc.size, 4), dtype=np.float32)
    lib().orc_decode_rgba_u8(enc.ctypes.data, out.ctypes.data, enc.size)
    return out


def piece_range(image_size: int, piece: int, num_pieces: int):
    b, e = C.c_int(), C.c_int()
    lib().orc_piece_range(image_size, piece, num_pieces, C.byref(b), C.byref(e))
    return b.value, e.value


def layer_order(hints, owner, local_index):
    hints = _f32(hints)
    owner = np.ascontiguousarray(owner, dtype=np.int32)
    local_index = np.ascontiguousarray(local_index, dtype=np.int32)
    n = hints.size
    order = np.empty(max(n, 1), dtype=np.int32)
    run_end = np.empty(max(n, 1), dtype=np.int32)
    r = lib().orc_layer_order(hints.ctypes.data, owner.ctypes.data, local_index.ctypes.data, n,
                              order.ctypes.data, run_end.ctypes.data)
    return order[:n].copy(), run_end[:r].copy()


def compose_layered(layers: Sequence[np.ndarray], hints, owner, local_index, n_ranks: int,
                    group_order=None, fold_variant: int = 0):
    """Simulated N-rank DirectSendBase::composeLayered + Gather.
    Returns (gathered[n_pixels, 5], piece_owner[n_pixels], n_runs)."""
    layers = [_f32(l).reshape(-1) for l in layers]
    n_layers = len(layers)
    n_pixels = layers[0].size // 5 if n_layers else 0
    ptrs = (C.c_void_p * max(n_layers, 1))(*[l.ctypes.data for l in layers])
    hints = _f32(hints)
    owner = np.ascontiguousarray(owner, dtype=np.int32)
    local_index = np.ascontiguousarray(local_index, dtype=np.int32)
    out = np.empty((n_pixels, 5), dtype=np.float32)
    po = np.empty(max(n_pixels, 1), dtype=np.int32)
    go = None
    if group_order is not None:
        go = np.ascontiguousarray(group_order, dtype=np.int32)
    runs = lib().orc_compose_layered(ptrs, hints.ctypes.data, owner.ctypes.data,
                                     local_index.ctypes.data, n_layers, n_ranks, n_pixels,
                                     go.ctypes.data if go is not None else None, fold_variant,
                                     out.ctypes.data, po.ctypes.data)
    return out, po[:n_pixels], runs


def downsample(src: np.ndarray, target_w: int, target_h: int, block: int) -> np.ndarray:
    src = _f32(src)
    assert src.size == target_w * block * target_h * block * 5
    out = np.empty((target_h, target_w, 5), dtype=np.float32)
    lib().orc_downsample(src.ctypes.data, target_w, target_h, block, out.ctypes.data)
    return out


def quantize_rgb8(src: np.ndarray, w: int, h: int) -> np.ndarray:
    src = _f32(src)
    stride = src.size // (w * h)
    out = np.empty((h, w, 3), dtype=np.uint8)
    lib().orc_quantize_rgb8(src.ctypes.data, w, h, stride, out.ctypes.data)
    return out


def fnv1a64(arr: np.ndarray) -> int:
    arr = np.ascontiguousarray(arr)
    return int(lib().orc_fnv1a64(arr.ctypes.data, arr.nbytes))


def scalar_stats(boxes: Sequence[Box]):
    """(min, max, min_positive, finite_count) over the boxes' cells."""
    arr = (Box * max(len(boxes), 1))(*boxes)
    stats = (C.c_double * 3)()
    n = lib().orc_scalar_stats(arr, len(boxes), stats)
    return stats[0], stats[1], stats[2], int(n)


def scene_transform(stats, finite_count, log_scale=False, normalize_to_data_range=True):
    """Returns (status, Transform, processed_min, processed_max, processed_range, scalar_range)."""
    st = (C.c_double * 3)(*map(float, stats))
    tr = Transform()
    pmin, pmax = C.c_double(), C.c_double()
    pr, sr = (C.c_float * 2)(), (C.c_float * 2)()
    status = lib().orc_scene_transform(st, int(finite_count), int(log_scale),
                                       int(normalize_to_data_range), C.byref(tr), C.byref(pmin),
                                       C.byref(pmax), pr, sr)
    return status, tr, pmin.value, pmax.value, (pr[0], pr[1]), (sr[0], sr[1])


def histogram(boxes: Sequence[Box], transform: Transform, range_min, range_max, bin_count):
    arr = (Box * max(len(boxes), 1))(*boxes)
    counts = np.zeros(bin_count, dtype=np.uint64)
    lib().orc_histogram(arr, len(boxes), C.byref(transform), float(range_min), float(range_max),
                        int(bin_count), counts.ctypes.data)
    return counts


def tight_bounds(boxes: Sequence[Box], fallback_min, fallback_max):
    arr = (Box * max(len(boxes), 1))(*boxes)
    fmin = (C.c_double * 3)(*map(float, fallback_min))
    fmax = (C.c_double * 3)(*map(float, fallback_max))
    omin, omax = (C.c_double * 3)(), (C.c_double * 3)()
    lib().orc_tight_bounds(arr, len(boxes), fmin, fmax, omin, omax)
    return tuple(omin), tuple(omax)


def bbox_overlay(image: np.ndarray, w: int, h: int, bounds_min, bounds_max, camera: Camera,
                 sqrt_antialiasing: int = 1) -> np.ndarray:
    """Returns a copy of the w x h depth-sort image with the wireframe blended over it."""
    out = np.array(_f32(image).reshape(-1), copy=True)
    assert out.size == w * h * 5
    bmin = (C.c_double * 3)(*map(float, bounds_min))
    bmax = (C.c_double * 3)(*map(float, bounds_max))
    lib().orc_bbox_overlay(bmin, bmax, C.byref(camera), int(sqrt_antialiasing), out.ctypes.data,
                           w, h)
    return out.reshape(h, w, 5)
