"""POD types of the volume path, mirroring Common/VolumeTypes.hpp:21-100 of the reference.

``AmrBox.values`` is a float64 tensor indexed ``[k, j, i]`` (x fastest, the order of
``amrex::Array4``); it may be a strided view (ghost cells) as long as the x stride is 1.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

from . import _capi

Vec3 = Tuple[float, float, float]


@dataclass
class ScalarTransform:
    """volume::ScalarTransform (Common/VolumeTypes.hpp:21-31)."""
    log_scale_input: bool = False
    normalize_to_unit_range: bool = False
    positive_floor: float = 0.0
    processed_min: float = 0.0
    processed_max: float = 1.0
    inverse_processed_span: float = 1.0
    normalization_min: float = 0.0
    normalization_max: float = 1.0
    inverse_normalization_span: float = 1.0

    def to_c(self) -> _capi.ScalarTransform:
        t = _capi.ScalarTransform()
        t.log_scale_input = int(bool(self.log_scale_input))
        t.normalize_to_unit_range = int(bool(self.normalize_to_unit_range))
        t.positive_floor = float(self.positive_floor)
        t.normalization_min = float(self.normalization_min)
        t.inverse_normalization_span = float(self.inverse_normalization_span)
        return t


@dataclass
class VolumeBounds:
    """volume::VolumeBounds (Common/VolumeTypes.hpp:78-81)."""
    min_corner: Vec3 = (0.0, 0.0, 0.0)
    max_corner: Vec3 = (1.0, 1.0, 1.0)


@dataclass
class CameraParameters:
    """volume::CameraParameters (Common/VolumeTypes.hpp:83-90)."""
    eye: Vec3
    look_at: Vec3
    up: Vec3 = (0.0, 1.0, 0.0)
    fov_y_degrees: float = 45.0
    near_plane: float = 0.1
    far_plane: float = 1000.0

    def to_c(self) -> _capi.Camera:
        cam = _capi.Camera()
        for c in range(3):
            cam.eye[c] = float(self.eye[c])
            cam.look_at[c] = float(self.look_at[c])
            cam.up[c] = float(self.up[c])
        cam.fov_y_degrees = float(self.fov_y_degrees)
        cam.near_plane = float(self.near_plane)
        cam.far_plane = float(self.far_plane)
        return cam


@dataclass
class ColorMapControlPoint:
    """volume::ColorMapControlPoint (Common/VolumeTypes.hpp:92-98)."""
    value: float
    red: float
    green: float
    blue: float
    alpha: float


ColorMap = List[ColorMapControlPoint]


def colormap_to_c(color_map: Optional[Sequence]) -> Tuple[Optional[C.Array], int]:
    """Accepts ColorMapControlPoint objects or (value, r, g, b, a) tuples."""
    if not color_map:
        return None, 0
    arr = (_capi.ColormapPoint * len(color_map))()
    for i, pt in enumerate(color_map):
        if isinstance(pt, ColorMapControlPoint):
            vals = (pt.value, pt.red, pt.green, pt.blue, pt.alpha)
        else:
            vals = tuple(pt)
            if len(vals) != 5:
                raise ValueError("color map entries are (value, red, green, blue, alpha)")
        arr[i].value, arr[i].red, arr[i].green, arr[i].blue, arr[i].alpha = map(float, vals)
    return arr, len(color_map)


@dataclass
class AmrBox:
    """volume::AmrBox (Common/VolumeTypes.hpp:69-76).

    values: float64 tensor [nz, ny, nx] (device for rendering; host tensors are accepted by the
    host-only helpers).  cell_dimensions is derived from its shape."""
    min_corner: Vec3
    max_corner: Vec3
    values: object = None
    level: int = 0
    dims: Optional[Tuple[int, int, int]] = None  # (nx, ny, nz) when values is None (metadata only)
    owner: int = 0                               # owning rank (sort-last partition)

    @property
    def cell_dimensions(self) -> Tuple[int, int, int]:
        if self.values is not None:
            nz, ny, nx = self.values.shape
            return int(nx), int(ny), int(nz)
        if self.dims is None:
            raise ValueError("AmrBox has neither values nor dims")
        return tuple(int(d) for d in self.dims)

    def to_c(self) -> _capi.Box:
        b = _capi.Box()
        for c in range(3):
            b.min_corner[c] = float(self.min_corner[c])
            b.max_corner[c] = float(self.max_corner[c])
        nx, ny, nz = self.cell_dimensions
        b.dims[0], b.dims[1], b.dims[2] = nx, ny, nz
        b.level = int(self.level)
        if self.values is not None:
            v = self.values
            if str(v.dtype) != "torch.float64":
                raise ValueError("AmrBox.values must be float64 (amrex::Real)")
            if v.dim() != 3 or (v.shape[2] > 1 and v.stride(2) != 1):
                raise ValueError("AmrBox.values must be [nz, ny, nx] with unit x stride")
            b.cells = v.data_ptr()
            b.jstride = int(v.stride(1))
            b.kstride = int(v.stride(0))
        else:
            b.cells = None
            b.jstride = nx
            b.kstride = nx * ny
        return b


def make_params(width: int, height: int, scalar_range=(0.0, 1.0), box_transparency: float = 0.0,
                reference_sample_distance: float = 0.0,
                bounds: Optional[VolumeBounds] = None, color_map=None) -> _capi.PaintParams:
    """The scalar arguments of VolumePainter::paint as the C ABI takes them."""
    p = _capi.PaintParams()
    p.width, p.height = int(width), int(height)
    p.scalar_range[0], p.scalar_range[1] = float(scalar_range[0]), float(scalar_range[1])
    p.box_transparency = float(box_transparency)
    p.reference_sample_distance = float(reference_sample_distance)
    bounds = bounds or VolumeBounds((-0.05,) * 3, (1.05,) * 3)
    for c in range(3):
        p.bounds_min[c] = float(bounds.min_corner[c])
        p.bounds_max[c] = float(bounds.max_corner[c])
    arr, n = colormap_to_c(color_map)
    if n:
        p.colormap = C.cast(arr, C.POINTER(_capi.ColormapPoint))
        p._keepalive = arr
    else:
        p.colormap = None
    p.colormap_count = n
    return p
