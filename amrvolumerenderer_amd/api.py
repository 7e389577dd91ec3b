"""Public entry points with the reference's option surface.

  api::Render(AmrData, RenderOptions)    VolumeRenderer/VolumeRendererApi.hpp:22-54
  python render(plotfile, **kwargs)      python/amrVolumeRenderer/module.cpp:264-303

Both entries run end to end: `render` / `run` read an AMReX plotfile without AMReX
(plotfile.py: Header, Cell_H, FAB files; our own convexify -- see DESIGN.md 6e for the one known
divergence from amrex::convexify's box list), build the scene statistics and the scalar transform
with the HIP scan kernels, and hand the fields of VolumeRenderer::SceneGeometry
(VolumeRenderer/VolumeRenderer.hpp:74-89) to `render_scene`; `render_amr_data` is api::Render
over plain per-level box lists and cell arrays (an amrex::MultiFab cannot be taken without AMReX).
Arguments are validated exactly as the reference does.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

from .types import AmrBox, CameraParameters, ColorMapControlPoint, ScalarTransform, VolumeBounds


@dataclass
class RenderOptions:
    """api::RenderOptions (VolumeRenderer/VolumeRendererApi.hpp:28-44)."""
    width: int = 512
    height: int = 512
    box_transparency: float = 0.0
    antialiasing: int = 1
    visibility_graph: bool = True
    write_visibility_graph: bool = False
    min_level: int = 0
    max_level: int = -1
    log_scale_input: bool = False
    component: int = 0
    output_filename: str = "volume-renderer.ppm"
    up_vector: Optional[Tuple[float, float, float]] = None
    scalar_range: Optional[Tuple[float, float]] = None
    camera: Optional[CameraParameters] = None
    color_map: Optional[List[ColorMapControlPoint]] = None


@dataclass
class SceneGeometry:
    """VolumeRenderer::SceneGeometry (VolumeRenderer/VolumeRenderer.hpp:74-89), hot-path fields."""
    all_boxes: List[AmrBox]            # metadata of every box (replicated), .owner set
    local_boxes: List[AmrBox]          # this rank's boxes, cell data in HBM
    scalar_transform: ScalarTransform
    bounds: VolumeBounds
    scalar_range: Tuple[float, float] = (0.0, 1.0)
    processed_scalar_range: Optional[Tuple[float, float]] = None   # after log scaling, if any


@dataclass
class AmrData:
    """api::AmrData (VolumeRenderer/VolumeRendererApi.hpp:22-26) without AMReX objects: per level
    the grids' index boxes and cell arrays instead of a MultiFab, the problem's lower corner and
    cell sizes instead of an amrex::Geometry.
    level_boxes[l] = [((ilo, jlo, klo), (ihi, jhi, khi)), ...]   inclusive, cell-centred
    level_data[l][g] = float64 array [nz, ny, nx] or [ncomp, nz, ny, nx] (numpy or torch)"""
    level_boxes: List[list]
    level_data: List[list]
    prob_lo: Tuple[float, float, float]
    cell_sizes: List[Tuple[float, float, float]]
    refinement_ratios: List[int]


def load_amr_data_geometry(ctx, data: AmrData, requested_min_level: int = 0,
                           requested_max_level: int = -1, component: int = 0,
                           log_scale_input: bool = False, normalize_to_data_range: bool = True,
                           rank: int = 0, n_ranks: int = 1, process_group=None) -> "SceneGeometry":
    """loadMultiFabGeometry (VolumeRendererApi.cpp:44-131)."""
    from . import plotfile as pf
    if not data.level_boxes or len(data.level_boxes) != len(data.level_data) or \
            len(data.cell_sizes) != len(data.level_boxes):
        raise ValueError("levelData and levelGeometry must be non-empty and of matching sizes")
    finest = len(data.level_boxes) - 1
    min_level, max_level = pf.clamp_levels(requested_min_level, requested_max_level, finest)
    if min_level > max_level:
        raise RuntimeError("minLevel must not exceed maxLevel")
    if max_level > 0 and len(data.refinement_ratios) < max_level:
        raise ValueError("refinementRatios must provide ratios for each level transition")
    if component < 0:
        raise ValueError("component index out of range")

    def fetch(level, grid_ids):
        out = {}
        for g in grid_ids:
            cells = data.level_data[level][g]
            if cells is None:
                raise ValueError("levelData contains a null MultiFab pointer")
            if cells.ndim == 4:
                if component >= cells.shape[0]:
                    raise ValueError("component index out of range")
                cells = cells[component]
            elif component != 0:
                raise ValueError("component index out of range")
            out[g] = cells
        return out

    return pf.build_scene_from_levels(
        ctx, [list(b) for b in data.level_boxes], data.cell_sizes, data.prob_lo,
        data.refinement_ratios, fetch, min_level, max_level, log_scale_input,
        normalize_to_data_range, rank, n_ranks, process_group,
        "Failed to locate any volumetric data in the provided MultiFabs.")


def build_scene_geometry(ctx, all_boxes: Sequence[AmrBox], local_boxes: Sequence[AmrBox],
                         bounds: VolumeBounds, log_scale_input: bool = False,
                         normalize_to_data_range: bool = True, process_group=None,
                         n_ranks: int = 1) -> "SceneGeometry":
    """The scalar part of detail::BuildSceneGeometry (VolumeRenderer/SceneBuilder.cpp:315-443):
    one streaming pass over the local cells (min, max, min positive, finite count), the
    MIN / MAX / SUM reductions over ranks (:327-344, :368-385) and the scalar transform.
    The geometric part (world corners, global rescale, padded bounds) needs amrex::Geometry and
    is plotfile.load_plotfile_geometry's: boxes arrive here with their corners."""
    import torch
    import torch.distributed as dist
    from . import runtime
    scene = ctx.create_scene(local_boxes, ScalarTransform())
    lo, hi, lo_pos, finite = scene.scalar_stats()
    if n_ranks > 1:
        device = ctx.device if dist.get_backend(process_group) == "nccl" else "cpu"
        mins = torch.tensor([lo, lo_pos], dtype=torch.float64, device=device)
        maxs = torch.tensor([hi], dtype=torch.float64, device=device)
        count = torch.tensor([finite], dtype=torch.int64, device=device)
        dist.all_reduce(mins, op=dist.ReduceOp.MIN, group=process_group)
        dist.all_reduce(maxs, op=dist.ReduceOp.MAX, group=process_group)
        dist.all_reduce(count, op=dist.ReduceOp.SUM, group=process_group)
        lo, lo_pos, hi, finite = mins[0].item(), mins[1].item(), maxs[0].item(), int(count.item())
    transform, processed_range, scalar_range = runtime.scene_transform_from_stats(
        (lo, hi, lo_pos), finite, log_scale_input, normalize_to_data_range)
    return SceneGeometry(list(all_boxes), list(local_boxes), transform, bounds, scalar_range,
                         processed_range)


def compute_histogram(plotfile: str, variable: Optional[str] = None, min_level: int = 0,
                      max_level: int = -1, log_scale: bool = False, bins: int = 256, ctx=None,
                      rank: int = 0, n_ranks: int = 1, process_group=None) -> dict:
    """The reference python module's compute_histogram (module.cpp:304-356, same keyword names
    and defaults; VolumeRenderer::computeScalarHistogram, VolumeRenderer.cpp:880-897): load the
    plotfile with normalisation to the data range, then bin every uncovered cell."""
    from . import plotfile as pf
    from . import runtime
    if bins <= 0:
        raise ValueError("binCount must be positive")
    if ctx is None:
        ctx, rank, n_ranks, process_group = _runtime_scope()
    scene = pf.load_plotfile_geometry(ctx, plotfile, variable or "", min_level, max_level,
                                      log_scale, True, rank, n_ranks, process_group)
    return compute_scene_histogram(ctx, scene.all_boxes, scene.local_boxes, log_scale, bins,
                                   process_group, n_ranks)


def compute_scene_histogram(ctx, all_boxes: Sequence[AmrBox], local_boxes: Sequence[AmrBox],
                            log_scale: bool = False, bins: int = 256, process_group=None,
                            n_ranks: int = 1) -> dict:
    """api::ComputeHistogram (VolumeRendererApi.cpp:397-413) for boxes already in HBM: scene statistics with
    normalisation to the data range, then the 64-bit bin counts of every cell
    (ComputeSceneHistogram, SceneBuilder.cpp:445-577).  Returns the module's dict keys."""
    import torch
    import torch.distributed as dist
    from . import runtime
    if bins <= 0:
        raise ValueError("binCount must be positive")
    scene = ctx.create_scene(local_boxes, ScalarTransform())
    lo, hi, lo_pos, finite = scene.scalar_stats()
    stage = n_ranks > 1 and dist.get_backend(process_group) != "nccl"
    if n_ranks > 1:
        device = "cpu" if stage else ctx.device
        mins = torch.tensor([lo, lo_pos], dtype=torch.float64, device=device)
        maxs = torch.tensor([hi], dtype=torch.float64, device=device)
        count = torch.tensor([finite], dtype=torch.int64, device=device)
        dist.all_reduce(mins, op=dist.ReduceOp.MIN, group=process_group)
        dist.all_reduce(maxs, op=dist.ReduceOp.MAX, group=process_group)
        dist.all_reduce(count, op=dist.ReduceOp.SUM, group=process_group)
        lo, lo_pos, hi, finite = mins[0].item(), mins[1].item(), maxs[0].item(), int(count.item())
    transform, processed_range, scalar_range = runtime.scene_transform_from_stats(
        (lo, hi, lo_pos), finite, log_scale, True)
    original = (lo, hi if hi != lo else lo + 1.0)
    counts = scene.histogram(transform, scalar_range[0], scalar_range[1], bins)
    if n_ranks > 1:
        if stage:
            host = counts.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=process_group)
            counts = host
        else:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=process_group)
    ctx.synchronize()
    host_counts = counts.cpu().numpy().astype("uint64")
    samples = int(host_counts.sum())
    if samples == 0:
        host_counts[:] = 0
    return {"counts": host_counts, "normalized_range": scalar_range,
            "processed_range": processed_range,
            "original_range": (float(__import__("numpy").float32(original[0])),
                               float(__import__("numpy").float32(original[1]))),
            "samples": samples}


def _finite(values) -> bool:
    return all(math.isfinite(float(v)) for v in values)


def validate_options(options: RenderOptions) -> None:
    """The argument checks of api::Render (VolumeRendererApi.cpp:150-255, 257-274)."""
    if not options.output_filename:
        raise ValueError("output filename must not be empty")
    if options.min_level < 0:
        raise ValueError("min level must be non-negative")
    if options.max_level < -1:
        raise ValueError("max level must be non-negative or -1 for all levels")
    if options.max_level >= 0 and options.min_level > options.max_level:
        raise RuntimeError("min level must not exceed max level")
    if options.up_vector is not None:
        if len(options.up_vector) != 3 or not _finite(options.up_vector) or \
                math.sqrt(sum(float(v) ** 2 for v in options.up_vector)) <= 0.0:
            raise ValueError("up_vector must be a finite, non-zero 3-vector")
    if options.scalar_range is not None:
        lo, hi = options.scalar_range
        if not _finite((lo, hi)) or not (lo < hi):
            raise ValueError("scalar_range must contain two values with min < max.")
    if options.color_map is not None:  # validateColorMap (VolumeRendererApi.cpp:163-196)
        if len(options.color_map) < 2:
            raise ValueError("color map must provide at least two control points")
        previous = -math.inf
        for p in options.color_map:
            vals = (p.value, p.red, p.green, p.blue, p.alpha) if isinstance(
                p, ColorMapControlPoint) else tuple(p)
            if len(vals) != 5:
                raise ValueError("color_map entries are (value, red, green, blue, alpha)")
            if not math.isfinite(float(vals[0])):
                raise ValueError("color map control point values must be finite")
            if float(vals[0]) <= previous:
                raise ValueError("color map control point values must be strictly increasing")
            previous = float(vals[0])
            for name, component in zip(("red", "green", "blue", "alpha"), vals[1:]):
                if not math.isfinite(float(component)) or not (0.0 <= float(component) <= 1.0):
                    raise ValueError(f"color map {name} components must be finite and within "
                                     "[0, 1]")
    if options.camera is not None:
        cam = options.camera
        if not _finite(cam.eye) or not _finite(cam.look_at) or not _finite(cam.up):
            raise ValueError("camera vectors must be finite")
        forward = [float(cam.look_at[a]) - float(cam.eye[a]) for a in range(3)]
        if not math.sqrt(sum(v * v for v in forward)) > 0.0:
            raise ValueError("camera eye and look-at must be distinct")
        up = [float(v) for v in cam.up]
        if not math.sqrt(sum(v * v for v in up)) > 0.0:
            raise ValueError("camera up vector must be non-zero")
        cross = (forward[1] * up[2] - forward[2] * up[1], forward[2] * up[0] - forward[0] * up[2],
                 forward[0] * up[1] - forward[1] * up[0])
        if not math.sqrt(sum(v * v for v in cross)) > 1e-6:
            raise ValueError("camera up vector must not be parallel to the view direction")
        if not (0.0 < cam.fov_y_degrees < 180.0):
            raise ValueError("camera_fov_y must be in (0, 180) degrees")
        if not (cam.near_plane > 0.0 and cam.far_plane > cam.near_plane):
            raise ValueError("camera near/far planes must satisfy 0 < near < far")


def save_ppm(rgb8, filename: str) -> bool:
    """SavePPM (Common/SavePPM.cpp:17-36): binary P6, rows already top-down in `rgb8`."""
    height, width = int(rgb8.shape[0]), int(rgb8.shape[1])
    data = rgb8.cpu().numpy().tobytes() if hasattr(rgb8, "cpu") else bytes(rgb8)
    with open(filename, "wb") as fh:
        fh.write(f"P6\n{width} {height}\n255\n".encode("ascii"))
        fh.write(data)
    return True


def save_png(rgb8, filename: str) -> bool:
    """SavePNG (Common/SavePNG.cpp:24-81): 8-bit RGB, non-interlaced, rows top-down -- the same
    pixel bytes as the PPM.  The reference hands the rows to libpng; the compressed stream
    depends on libpng's filter heuristics and zlib's settings, so parity is on the decoded
    pixels, not the file bytes.  Rows are stored with filter type 0 (None)."""
    import struct
    import zlib
    height, width = int(rgb8.shape[0]), int(rgb8.shape[1])
    rows = rgb8.cpu().numpy() if hasattr(rgb8, "cpu") else rgb8
    raw = b"".join(b"\x00" + rows[y].tobytes() for y in range(height))

    def chunk(tag: bytes, payload: bytes) -> bytes:
        return (struct.pack(">I", len(payload)) + tag + payload
                + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF))

    try:
        with open(filename, "wb") as fh:
            fh.write(b"\x89PNG\r\n\x1a\n")
            fh.write(chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 2, 0, 0, 0)))
            fh.write(chunk(b"IDAT", zlib.compress(raw, 6)))
            fh.write(chunk(b"IEND", b""))
    except OSError:
        return False
    return True


def _libm_float(name: str):
    """float f(float) of the host libm (std::sin / cos / tan / log on a float argument)."""
    import ctypes
    import ctypes.util
    lib = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    fn = getattr(lib, name)
    fn.restype = ctypes.c_float
    fn.argtypes = [ctypes.c_float]
    return lambda x: fn(float(x))


_runtime = {"refs": 0, "ctx": None, "owns_group": False}


def _ensure_runtime() -> None:
    """ensure_runtime_initialized (module.cpp:35-66): the rendering context on the local GPU and,
    when launched with one process per GPU (WORLD_SIZE > 1), the RCCL process group."""
    from . import runtime
    import torch
    if _runtime["ctx"] is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        torch.cuda.set_device(local_rank)
        if world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                _runtime["owns_group"] = True
        _runtime["ctx"] = runtime.Context(local_rank)


def initialize_runtime() -> None:
    """amrVolumeRenderer.initialize_runtime (module.cpp:103-107): set the process up once ahead
    of several render() / compute_histogram() calls (reference counted)."""
    _ensure_runtime()
    _runtime["refs"] += 1


def finalize_runtime() -> None:
    """amrVolumeRenderer.finalize_runtime (module.cpp:109-119)."""
    if _runtime["refs"] == 0:
        raise RuntimeError("amrVolumeRenderer.finalize_runtime requires a matching "
                           "initialize_runtime call")
    _runtime["refs"] -= 1
    if _runtime["refs"] == 0:
        if _runtime["ctx"] is not None:
            _runtime["ctx"].close()
            _runtime["ctx"] = None
        if _runtime["owns_group"]:
            import torch.distributed as dist
            dist.destroy_process_group()
            _runtime["owns_group"] = False


def _runtime_scope():
    """(context, rank, world size, process group) of the runtime.  Like the reference's
    RuntimeScope (module.cpp:86-101) a call without a prior initialize_runtime() initialises on
    demand -- honouring LOCAL_RANK / RANK / WORLD_SIZE, so that under a one-process-per-GPU
    launcher every process renders its own share on its own GPU.  Unlike the reference's scope
    (which finalises MPI on exit when no manual reference is held, after which nothing can be
    initialised again) the on-demand runtime stays until finalize_runtime drops the last manual
    reference or the process ends."""
    _ensure_runtime()
    rank, world, group = 0, 1, None
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world, group = dist.get_rank(), dist.get_world_size(), dist.group.WORLD
    elif int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise RuntimeError("WORLD_SIZE > 1 but no process group could be initialised")
    return _runtime["ctx"], rank, world, group


class _Mt19937:
    """std::mt19937 (the 32-bit Mersenne Twister of the C++ standard, [rand.predef])."""

    def __init__(self, seed: int):
        self.state = [0] * 624
        self.state[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            prev = self.state[i - 1]
            self.state[i] = (1812433253 * (prev ^ (prev >> 30)) + i) & 0xFFFFFFFF
        self.index = 624

    def __call__(self) -> int:
        if self.index >= 624:
            st = self.state
            for i in range(624):
                y = (st[i] & 0x80000000) | (st[(i + 1) % 624] & 0x7FFFFFFF)
                st[i] = st[(i + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.index = 0
        y = self.state[self.index]
        self.index += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def _uniform_float(rng: _Mt19937, a, b):
    """libstdc++'s std::uniform_real_distribution<float>(a, b)(mt19937): generate_canonical
    takes one 32-bit draw, converts it to float, divides by 2^32 and steps a result of 1.0 down
    to the float below; then canonical * (b - a) + a, all in float."""
    import numpy as np
    f32 = np.float32
    canonical = f32(f32(rng()) / f32(4294967296.0))
    if canonical >= f32(1.0):
        canonical = np.nextafter(f32(1.0), f32(0.0))
    return f32(f32(canonical * f32(f32(b) - f32(a))) + f32(a))


def automatic_camera(bounds: VolumeBounds, camera_seed: int = 91021,
                     up_vector: Optional[Sequence[float]] = None) -> CameraParameters:
    """The camera renderScene places when none is given (VolumeRenderer.cpp:974-1023): on a
    sphere around the bounds' centre, azimuth and altitude drawn from std::mt19937(cameraSeed).
    Float arithmetic and the host libm's sin / cos / tan as in the reference's host code."""
    import numpy as np
    f32 = np.float32
    k_pi = f32(3.14159265358979323846)
    k_two_pi = f32(f32(2.0) * k_pi)
    # amrex::RealVect is double: 0.5f * (min + max) promotes to double
    center = [0.5 * (bounds.min_corner[a] + bounds.max_corner[a]) for a in range(3)]
    half = [0.5 * (bounds.max_corner[a] - bounds.min_corner[a]) for a in range(3)]
    radius = f32(math.sqrt(half[0] * half[0] + half[1] * half[1] + half[2] * half[2]))
    if radius <= f32(0.0):
        radius = f32(1.0)
    fov_y = f32(k_pi * f32(0.25))
    max_altitude = f32(k_pi * f32(0.25))
    half_fov = f32(fov_y * f32(0.5))
    sinf, cosf, tanf = _libm_float("sinf"), _libm_float("cosf"), _libm_float("tanf")
    min_distance = f32(radius / f32(tanf(half_fov))) if half_fov > 0 else radius
    safety = max(f32(f32(0.25) * radius), f32(0.5))
    distance = f32(min_distance + safety)
    rng = _Mt19937(camera_seed)
    azimuth = _uniform_float(rng, f32(0.0), k_two_pi)
    altitude = _uniform_float(rng, -max_altitude, max_altitude)
    cos_altitude, sin_altitude = f32(cosf(altitude)), f32(sinf(altitude))
    sin_azimuth, cos_azimuth = f32(sinf(azimuth)), f32(cosf(azimuth))
    eye = (center[0] + float(f32(f32(distance * cos_altitude) * sin_azimuth)),
           center[1] + float(f32(distance * sin_altitude)),
           center[2] + float(f32(f32(distance * cos_altitude) * cos_azimuth)))

    def normalize(v):
        length = math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
        if length > 0.0 and math.isfinite(length):
            return (v[0] / length, v[1] / length, v[2] / length)
        return (0.0, 0.0, -1.0)

    def cross_length(a, b):
        c = (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
        return math.sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2])

    up = tuple(float(v) for v in up_vector) if up_vector is not None else (0.0, 1.0, 0.0)
    view_dir = normalize(tuple(center[a] - eye[a] for a in range(3)))
    if cross_length(view_dir, up) <= float(f32(1e-4)):
        up = (0.0, 0.0, 1.0)
        if cross_length(view_dir, up) <= float(f32(1e-4)):
            up = (1.0, 0.0, 0.0)
    up = normalize(up)
    return CameraParameters(eye, tuple(center), up,
                            float(f32(f32(fov_y * f32(180.0)) / k_pi)), float(f32(0.1)),
                            float(f32(distance * f32(4.0))))


def render_scene(ctx, scene: SceneGeometry, options: RenderOptions, rank: int = 0,
                 n_ranks: int = 1, process_group=None, stage_through_host: bool = False) -> int:
    """renderScene (VolumeRenderer.cpp:947-1101 -> renderSingleTrial) with an explicit camera or
    the automatic one: paints, composites, gathers and writes the image on rank 0.  Returns 0 on
    success like the reference."""
    from .renderer import FrameRenderer, RenderParameters
    validate_options(options)
    camera = options.camera
    if camera is None:
        camera = automatic_camera(scene.bounds, up_vector=options.up_vector)
    extension = os.path.splitext(options.output_filename)[1].lower()
    renderer = FrameRenderer(ctx, scene.all_boxes, scene.local_boxes, scene.scalar_transform,
                             scene.bounds, scene.scalar_range, rank, n_ranks, process_group,
                             color_map=options.color_map, stage_through_host=stage_through_host)
    _, rgb8 = renderer.render(
        RenderParameters(options.width, options.height, options.box_transparency,
                         options.antialiasing, options.visibility_graph,
                         write_visibility_graph=options.write_visibility_graph), camera)
    renderer.synchronize()
    if rank == 0:
        # any other extension falls back to PPM (VolumeRenderer.cpp:1316-1327)
        writer = save_png if extension == ".png" else save_ppm
        if not writer(rgb8, options.output_filename):
            return 1
    return 0


def render(plotfile: str, width: int = 512, height: int = 512, box_transparency: float = 0.0,
           antialiasing: int = 1, visibility_graph: bool = True,
           write_visibility_graph: bool = False, variable: Optional[str] = None,
           min_level: int = 0, max_level: int = -1, log_scale: bool = False,
           up_vector: Optional[Sequence[float]] = None, output: Optional[str] = None,
           scalar_range: Optional[Sequence[float]] = None,
           camera_eye: Optional[Sequence[float]] = None,
           camera_look_at: Optional[Sequence[float]] = None,
           camera_up: Optional[Sequence[float]] = None, camera_fov_y: Optional[float] = None,
           camera_near: Optional[float] = None, camera_far: Optional[float] = None,
           color_map: Optional[Sequence[Sequence[float]]] = None) -> int:
    """The reference's python entry (python/amrVolumeRenderer/module.cpp:275-303), same keyword
    names and defaults: validates the arguments as the reference does, reads the plotfile
    (plotfile.py), renders on cuda:0 and writes the image.  Multi-rank use: call run() with the
    rank, world size and process group."""
    camera = None
    if camera_eye is not None or camera_look_at is not None:
        if camera_eye is None or camera_look_at is None:
            raise ValueError("camera_eye and camera_look_at must be given together")
        camera = CameraParameters(tuple(camera_eye), tuple(camera_look_at),
                                  tuple(camera_up) if camera_up is not None else (0.0, 1.0, 0.0),
                                  45.0 if camera_fov_y is None else float(camera_fov_y),
                                  0.1 if camera_near is None else float(camera_near),
                                  1000.0 if camera_far is None else float(camera_far))
    cmap = None
    if color_map is not None:
        cmap = []
        for entry in color_map:
            if len(tuple(entry)) != 5:
                raise ValueError("color_map entries are (value, red, green, blue, alpha)")
            cmap.append(ColorMapControlPoint(*[float(v) for v in entry]))
    options = RenderOptions(
        width=width, height=height, box_transparency=box_transparency, antialiasing=antialiasing,
        visibility_graph=visibility_graph, write_visibility_graph=write_visibility_graph,
        min_level=min_level, max_level=max_level, log_scale_input=log_scale,
        output_filename=output if output is not None else "volume-renderer.ppm",
        up_vector=tuple(up_vector) if up_vector is not None else None,
        scalar_range=tuple(scalar_range) if scalar_range is not None else None,
        camera=camera, color_map=cmap)
    validate_options(options)
    from .renderer import validate_render_parameters, RenderParameters
    validate_render_parameters(RenderParameters(width, height, box_transparency, antialiasing))
    if not plotfile:
        raise RuntimeError("plotfile path is required")
    if not os.path.exists(plotfile):
        raise RuntimeError(f"plotfile path '{plotfile}' does not exist")
    ctx, rank, world, group = _runtime_scope()
    return run(plotfile, options, variable or "", ctx, rank, world, group)


def run(plotfile: str, options: RenderOptions, variable_name: str = "", ctx=None, rank: int = 0,
        n_ranks: int = 1, process_group=None, stage_through_host: bool = False) -> int:
    """VolumeRenderer::run(RunOptions) after its argument checks (VolumeRenderer.cpp:1469-1576):
    load the plotfile, apply a scalar-range override and convert the colour map's physical
    values to normalised ones, then renderScene."""
    from . import plotfile as pf
    from . import runtime
    if ctx is None:
        ctx = runtime.Context(0)
    has_override = options.scalar_range is not None
    scene = pf.load_plotfile_geometry(ctx, plotfile, variable_name, options.min_level,
                                      options.max_level, options.log_scale_input,
                                      not has_override, rank, n_ranks, process_group)
    return _render_loaded_scene(ctx, scene, options, rank, n_ranks, process_group,
                                stage_through_host)


def render_amr_data(data: AmrData, options: RenderOptions, ctx=None, rank: int = 0,
                    n_ranks: int = 1, process_group=None) -> int:
    """api::Render(const AmrData&, const RenderOptions&) (VolumeRendererApi.cpp:257-395)."""
    from . import runtime
    validate_options(options)
    if ctx is None:
        ctx = runtime.Context(0)
    scene = load_amr_data_geometry(ctx, data, options.min_level, options.max_level,
                                   options.component, options.log_scale_input,
                                   options.scalar_range is None, rank, n_ranks, process_group)
    return _render_loaded_scene(ctx, scene, options, rank, n_ranks, process_group)


def _render_loaded_scene(ctx, scene: SceneGeometry, options: RenderOptions, rank: int,
                         n_ranks: int, process_group, stage_through_host: bool = False) -> int:
    """The common tail of VolumeRenderer::run and api::Render: scalar-range override, colour-map
    values from physical to normalised, camera-up normalisation, renderScene."""
    import numpy as np
    f32 = np.float32
    has_override = options.scalar_range is not None
    if scene.processed_scalar_range is None:
        raise RuntimeError("Internal error: processed scalar range unavailable for color mapping.")
    processed_min, processed_max = (f32(v) for v in scene.processed_scalar_range)
    span = f32(processed_max - processed_min)
    if not (span > 0.0) or not np.isfinite(span):
        raise RuntimeError("Failed to establish a finite scalar range for color mapping.")
    logf = _libm_float("logf")

    def to_processed(physical) -> np.float32:
        physical = f32(physical)
        if not np.isfinite(physical):
            raise ValueError("color_map scalar values must be finite.")
        if options.log_scale_input:
            if not (physical > 0.0):
                raise ValueError("color_map scalar values must be positive when log scaling is "
                                 "enabled.")
            return f32(logf(physical))
        return physical

    norm_min, norm_max = processed_min, processed_max
    if has_override:
        norm_min = to_processed(options.scalar_range[0])
        norm_max = to_processed(options.scalar_range[1])
        if not (norm_min < norm_max):
            raise ValueError("scalar_range must contain two values with min < max.")
    norm_span = f32(norm_max - norm_min)
    if not (norm_span > 0.0) or not np.isfinite(norm_span):
        raise RuntimeError("Failed to establish a finite scalar range for color mapping.")
    if has_override:
        # SetSceneNormalizationRange (SceneBuilder.cpp:427-443), amrex::Real arithmetic
        lo, hi = float(norm_min), float(norm_max)
        transform = scene.scalar_transform
        transform.normalize_to_unit_range = True
        transform.normalization_min = lo
        transform.inverse_normalization_span = 1.0 / (hi - lo)
        scene.scalar_range = (0.0, 1.0)
    if options.color_map is not None:
        converted = []
        for point in options.color_map:
            value = f32(f32(to_processed(point.value) - norm_min) / norm_span)
            if not np.isfinite(value):
                raise ValueError("color_map produced a non-finite normalized scalar value.")
            value = min(max(value, f32(0.0)), f32(1.0))
            converted.append(ColorMapControlPoint(float(value), point.red, point.green,
                                                  point.blue, point.alpha))
        options = _replace(options, color_map=converted)
    if options.camera is not None:
        cam = options.camera
        length = math.sqrt(sum(float(v) ** 2 for v in cam.up))
        up = tuple(float(v) / length for v in cam.up) if (length > 0.0 and math.isfinite(length)) \
            else (0.0, 0.0, -1.0)
        options = _replace(options, camera=CameraParameters(cam.eye, cam.look_at, up,
                                                            cam.fov_y_degrees, cam.near_plane,
                                                            cam.far_plane))
    return render_scene(ctx, scene, options, rank, n_ranks, process_group, stage_through_host)


def _replace(options: RenderOptions, **changes) -> RenderOptions:
    import dataclasses
    return dataclasses.replace(options, **changes)
