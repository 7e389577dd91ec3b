"""Public entry points with the reference's option surface.

  api::Render(AmrData, RenderOptions)    VolumeRenderer/VolumeRendererApi.hpp:22-54
  python render(plotfile, **kwargs)      python/amrVolumeRenderer/module.cpp:264-303

What is built this round is the hot path below these entries.  Scene construction from AMReX
data (SceneBuilder, plotfile reading) is not, so `render_scene` takes
the scene in the form the hot path consumes -- the list of AmrBox with cell data in HBM, the
scalar transform, the bounds and the normalised scalar range, i.e. the fields of
VolumeRenderer::SceneGeometry (VolumeRenderer/VolumeRenderer.hpp:74-89) -- and `render` validates
its arguments exactly as the reference does and then reports that plotfile ingestion is not
available.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
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


def build_scene_geometry(ctx, all_boxes: Sequence[AmrBox], local_boxes: Sequence[AmrBox],
                         bounds: VolumeBounds, log_scale_input: bool = False,
                         normalize_to_data_range: bool = True, process_group=None,
                         n_ranks: int = 1) -> "SceneGeometry":
    """The scalar part of detail::BuildSceneGeometry (VolumeRenderer/SceneBuilder.cpp:315-443):
    one streaming pass over the local cells (min, max, min positive, finite count), the
    MIN / MAX / SUM reductions over ranks (:327-344, :368-385) and the scalar transform.
    The geometric part (world corners, global rescale, padded bounds) needs amrex::Geometry and
    is the caller's: boxes arrive with their corners."""
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
    transform, _, scalar_range = runtime.scene_transform_from_stats(
        (lo, hi, lo_pos), finite, log_scale_input, normalize_to_data_range)
    return SceneGeometry(list(all_boxes), list(local_boxes), transform, bounds, scalar_range)


def compute_histogram(ctx, all_boxes: Sequence[AmrBox], local_boxes: Sequence[AmrBox],
                      log_scale: bool = False, bins: int = 256, process_group=None,
                      n_ranks: int = 1) -> dict:
    """api::ComputeHistogram / python compute_histogram (VolumeRendererApi.cpp:397-413,
    python/amrVolumeRenderer/module.cpp:304-356) for boxes already in HBM: scene statistics with
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
    if options.color_map is not None:
        if len(options.color_map) == 0:
            raise ValueError("color_map must contain at least one control point")
        for p in options.color_map:
            vals = (p.value, p.red, p.green, p.blue, p.alpha) if isinstance(
                p, ColorMapControlPoint) else tuple(p)
            if len(vals) != 5 or not _finite(vals):
                raise ValueError("color_map entries are finite (value, red, green, blue, alpha)")
    if options.camera is not None:
        cam = options.camera
        if not _finite(cam.eye) or not _finite(cam.look_at) or not _finite(cam.up):
            raise ValueError("camera vectors must be finite")
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


def render_scene(ctx, scene: SceneGeometry, options: RenderOptions, rank: int = 0,
                 n_ranks: int = 1, process_group=None) -> int:
    """renderScene with an explicit camera (VolumeRenderer.cpp:1062-1101 -> renderSingleTrial):
    paints, composites, gathers and writes the image on rank 0.  Returns 0 on success like the
    reference."""
    from .renderer import FrameRenderer, RenderParameters
    validate_options(options)
    if options.camera is None:
        raise ValueError("render_scene needs an explicit camera (automatic placement is part of "
                         "the frame driver that is out of scope, VolumeRenderer.cpp:974-1023)")
    extension = os.path.splitext(options.output_filename)[1].lower()
    renderer = FrameRenderer(ctx, scene.all_boxes, scene.local_boxes, scene.scalar_transform,
                             scene.bounds, scene.scalar_range, rank, n_ranks, process_group,
                             color_map=options.color_map)
    _, rgb8 = renderer.render(RenderParameters(options.width, options.height,
                                               options.box_transparency, options.antialiasing,
                                               options.visibility_graph), options.camera)
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
    names and defaults.  Arguments are validated as the reference validates them; the plotfile
    itself cannot be read yet (no AMReX plotfile reader in this round)."""
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
        raise ValueError("plotfile path must not be empty")
    raise NotImplementedError(
        "reading AMReX plotfiles (VolumeRenderer.cpp:588-714) is outside this round's scope "
        "(SURVEY.md 8(f-1)); build a SceneGeometry and call render_scene()")
