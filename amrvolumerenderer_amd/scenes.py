"""Synthetic AMR scenes of the benchmark configurations.

Domain [0,1]^3 (so SceneBuilder's global rescale, VolumeRenderer/SceneBuilder.cpp:229-254, is
the identity).  Level l has cell size 1/(n0 * 2^l) and covers the centred cube of half the
extent of level l-1; a level keeps only the boxes not covered by the next level ("convexified",
VolumeRenderer/VolumeRendererApi.cpp:98).  Boxes are aligned cubes of `box_cells` cells.

Only metadata lives here; cell data is produced per box on demand, as numpy (tests, CPU
baseline) or as torch tensors on a device (bench), from the same closed-form fields.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from dataclasses import field as dc_field
from typing import List, Optional, Tuple

import numpy as np

from .types import AmrBox, CameraParameters, ScalarTransform, VolumeBounds

# fixed Gaussian blobs of the `smooth` field: (centre, sigma)
_BLOBS = (((0.5, 0.5, 0.5), 0.08), ((0.3, 0.6, 0.4), 0.15), ((0.7, 0.35, 0.6), 0.12),
          ((0.45, 0.3, 0.75), 0.2))
_NOISE_SEED = 0x5EED


@dataclass
class BoxMeta:
    level: int
    lo: Tuple[int, int, int]        # index of the first cell at this level (i, j, k)
    dims: Tuple[int, int, int]      # (nx, ny, nz)
    min_corner: Tuple[float, float, float]
    max_corner: Tuple[float, float, float]
    owner: int = 0


@dataclass
class SceneSpec:
    name: str
    n0: int                         # base-level cells per side
    levels: int
    box_cells: int
    field: str = "smooth"           # "smooth" | "noise" | "radial"
    extent: float = 1.0             # domain edge length
    boxes: List[BoxMeta] = dc_field(default_factory=list)
    bounds: VolumeBounds = dc_field(default_factory=lambda: VolumeBounds((-0.05,) * 3, (1.05,) * 3))
    transform: ScalarTransform = dc_field(default_factory=lambda: ScalarTransform(
        normalize_to_unit_range=True, normalization_min=0.0, normalization_max=1.0,
        inverse_normalization_span=1.0))
    scalar_range: Tuple[float, float] = (0.0, 1.0)

    @property
    def total_cells(self) -> int:
        return sum(b.dims[0] * b.dims[1] * b.dims[2] for b in self.boxes)


def make_amr_scene(n0: int, levels: int, box_cells: int, field_name: str = "smooth",
                   name: Optional[str] = None, extent: float = 1.0) -> SceneSpec:
    """Nested-centre AMR hierarchy.  `extent` scales the domain (1.0 gives power-of-two cell
    sizes; another value exercises the general IEEE-division indexing path)."""
    if n0 % box_cells != 0:
        raise ValueError("n0 must be a multiple of box_cells")
    per_side = n0 // box_cells
    if levels > 1 and (per_side % 4 != 0):
        raise ValueError("refined scenes need n0/box_cells divisible by 4 (aligned nesting)")
    spec = SceneSpec(name=name or f"amr{levels}l_{n0}", n0=n0, levels=levels,
                     box_cells=box_cells, field=field_name, extent=extent)
    spec.bounds = VolumeBounds((-0.05 * extent,) * 3, (1.05 * extent,) * 3)
    for level in range(levels):
        h = extent / (n0 * (1 << level))
        # first cell (at this level's resolution) of the level's region
        region_lo = 0
        for l in range(1, level + 1):
            region_lo = (region_lo + n0 // 4) * 2  # centred half cube, refined by 2
        # boxes of this level covered by the next level: the centred half of the region
        hole_lo, hole_hi = per_side // 4, per_side - per_side // 4
        for bz in range(per_side):
            for by in range(per_side):
                for bx in range(per_side):
                    covered = (level + 1 < levels and hole_lo <= bx < hole_hi
                               and hole_lo <= by < hole_hi and hole_lo <= bz < hole_hi)
                    if covered:
                        continue
                    lo = (region_lo + bx * box_cells, region_lo + by * box_cells,
                          region_lo + bz * box_cells)
                    spec.boxes.append(BoxMeta(
                        level=level, lo=lo, dims=(box_cells,) * 3,
                        min_corner=tuple(l * h for l in lo),
                        max_corner=tuple((l + box_cells) * h for l in lo)))
    return spec


# ---- the benchmark configurations (BASELINE.json "configs") -----------------------------------

def config1(field_name: str = "radial") -> SceneSpec:
    """insitu_example scaled to 64^3 / 8 boxes of 32^3 (Examples/RenderFromMultiFab.cpp:20-46)."""
    return make_amr_scene(64, 1, 32, field_name, "config1_insitu_64")


def config2(field_name: str = "smooth") -> SceneSpec:
    """Single-level 512^3, 64 boxes of 128^3 (1.07 GB)."""
    return make_amr_scene(512, 1, 128, field_name, "config2_uniform_512")


def config3(field_name: str = "smooth") -> SceneSpec:
    """3-level AMR, 256^3 base, 176 boxes of 64^3 (369 MB)."""
    return make_amr_scene(256, 3, 64, field_name, "config3_amr3_256")


def config4(field_name: str = "smooth") -> SceneSpec:
    """3-level AMR, 512^3 base, 176 boxes of 128^3 (2.95 GB): the headline configuration."""
    return make_amr_scene(512, 3, 128, field_name, "config4_amr3_512")


def config5(field_name: str = "smooth") -> SceneSpec:
    """4-level AMR, 1024^3 base, 1856 boxes of 128^3 (31.1 GB)."""
    return make_amr_scene(1024, 4, 128, field_name, "config5_amr4_1024")


def default_camera() -> CameraParameters:
    """The fixed benchmark camera."""
    return CameraParameters(eye=(2.2, 1.6, 2.9), look_at=(0.5, 0.5, 0.5), up=(0.0, 1.0, 0.0),
                            fov_y_degrees=45.0, near_plane=0.1, far_plane=20.0)


def orbit_camera(view: int, n_views: int = 16) -> CameraParameters:
    """View `view` of an n_views orbit around the domain centre, same distance and elevation
    as the default camera."""
    base = default_camera()
    ox, oy, oz = (base.eye[c] - base.look_at[c] for c in range(3))
    radius = math.hypot(ox, oz)
    theta0 = math.atan2(ox, oz)
    theta = theta0 + 2.0 * math.pi * view / n_views
    eye = (base.look_at[0] + radius * math.sin(theta), base.look_at[1] + oy,
           base.look_at[2] + radius * math.cos(theta))
    return CameraParameters(eye=eye, look_at=base.look_at, up=base.up,
                            fov_y_degrees=base.fov_y_degrees, near_plane=base.near_plane,
                            far_plane=base.far_plane)


# ---- ownership (sort-last partition of boxes over ranks) ---------------------------------------

def _morton3(x: int, y: int, z: int) -> int:
    code = 0
    for bit in range(21):
        code |= ((x >> bit) & 1) << (3 * bit) | ((y >> bit) & 1) << (3 * bit + 1) | \
                ((z >> bit) & 1) << (3 * bit + 2)
    return code


def box_cost(meta: BoxMeta, pixels_per_unit: float) -> float:
    """Estimated GPU time (ms) of one box per frame: the classify pass streams every cell, the
    march takes rays x steps with rays ~ projected face area.  Constants measured on MI355X
    (profiles/experiments_rounds_1_to_3.md section 4): 0.52 ms / 369 M cells, 1.31 ms / 759 M samples."""
    ext = [meta.max_corner[c] - meta.min_corner[c] for c in range(3)]
    cells = meta.dims[0] * meta.dims[1] * meta.dims[2]
    face = (ext[0] * ext[1] * ext[1] * ext[2] * ext[0] * ext[2]) ** (1.0 / 3.0)
    steps = 2.0 * max(meta.dims) * 1.2
    samples = face * pixels_per_unit * pixels_per_unit * steps
    return 1.41e-9 * cells + 1.73e-9 * samples


def assign_owners(spec: SceneSpec, n_ranks: int, policy: str = "morton",
                  pixels_per_unit: float = 787.0) -> None:
    """morton: boxes sorted by the Morton code of their centre (all levels merged), contiguous
    chunks of ceil(B / N) per rank.  morton_cost: the same curve cut where the accumulated
    estimated cost (box_cost) reaches k/N of the total -- spatially compact AND balanced;
    pixels_per_unit = image height / (2 d tan(fovY/2)) of the intended view.
    morton_pairs: the curve cut into 2N stretches of equal estimated cost, rank r takes stretch r
    and stretch 2N-1-r -- the two ends of the curve are opposite corners of the domain, so every
    rank owns a near and a far region whatever the view direction (with plain chunks the rank
    that owns the octant next to the eye marches 1.6x the samples of the one behind it:
    config-4, N = 8, 121 M against 74 M).
    level_pairs: the pairing applied to every AMR level on its own (that level's boxes along the
    curve, 2N stretches of equal count): every rank then owns the same number of boxes OF EVERY
    LEVEL -- equal classify bytes, equal mix of long coarse and short fine rays -- in a near and a
    far region per level.  (Cutting the merged curve cannot do both: a coarse box holds 16x the
    samples of a fine one of the same cell count.)
    round_robin: box b -> rank b % N (stress variant).  block: level-major chunks."""
    n = len(spec.boxes)
    if policy == "round_robin":
        for i, b in enumerate(spec.boxes):
            b.owner = i % n_ranks
        return
    if policy == "block":
        chunk = -(-n // n_ranks)
        for i, b in enumerate(spec.boxes):
            b.owner = min(i // chunk, n_ranks - 1)
        return
    if policy not in ("morton", "morton_cost", "morton_pairs", "level_pairs"):
        raise ValueError(f"unknown ownership policy {policy!r}")
    finest = spec.n0 * (1 << (spec.levels - 1))
    extent = spec.extent

    def key(i: int) -> int:
        b = spec.boxes[i]
        centre = [0.5 * (b.min_corner[c] + b.max_corner[c]) / extent for c in range(3)]
        q = [min(int(c * finest), finest - 1) for c in centre]
        return _morton3(*q)

    ranked = sorted(range(n), key=lambda i: (key(i), i))
    if policy == "level_pairs":
        for level in sorted({b.level for b in spec.boxes}):
            members = [i for i in ranked if spec.boxes[i].level == level]
            for pos, i in enumerate(members):
                k = min(int((pos + 0.5) * 2 * n_ranks / len(members)), 2 * n_ranks - 1)
                spec.boxes[i].owner = k if k < n_ranks else 2 * n_ranks - 1 - k
        return
    if policy in ("morton_cost", "morton_pairs"):
        costs = [box_cost(spec.boxes[i], pixels_per_unit) for i in ranked]
        total = sum(costs)
        stretches = n_ranks if policy == "morton_cost" else 2 * n_ranks
        running = 0.0
        for i, c in zip(ranked, costs):
            # the stretch whose share [k/S, (k+1)/S) of the total cost holds this box's midpoint
            mid = running + 0.5 * c
            k = min(int(mid * stretches / total), stretches - 1) if total > 0 else 0
            spec.boxes[i].owner = k if k < n_ranks else 2 * n_ranks - 1 - k
            running += c
        return
    chunk = -(-n // n_ranks)
    for pos, i in enumerate(ranked):
        spec.boxes[i].owner = min(pos // chunk, n_ranks - 1)


def local_box_indices(spec: SceneSpec, rank: int) -> List[int]:
    """geometry.localBoxes of `rank`: level-major, then box order (SceneBuilder.cpp:134-188)."""
    return [i for i, b in enumerate(spec.boxes) if b.owner == rank]


# ---- fields ---------------------------------------------------------------------------------------

def _cell_centres(meta: BoxMeta, spec: SceneSpec, xp, **kw):
    n_level = spec.n0 * (1 << meta.level)
    axes = []
    for c in range(3):
        idx = xp.arange(meta.lo[c], meta.lo[c] + meta.dims[c], dtype=xp.float64, **kw)
        axes.append((idx + 0.5) / n_level)
    return axes  # x, y, z in [0,1] (domain-normalised coordinates)


def _splitmix64_numpy(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def box_cells_numpy(spec: SceneSpec, index: int) -> np.ndarray:
    """float64 [nz, ny, nx] cell data of box `index`."""
    meta = spec.boxes[index]
    if spec.field == "noise":
        i = np.arange(meta.lo[0], meta.lo[0] + meta.dims[0], dtype=np.uint64)
        j = np.arange(meta.lo[1], meta.lo[1] + meta.dims[1], dtype=np.uint64)
        k = np.arange(meta.lo[2], meta.lo[2] + meta.dims[2], dtype=np.uint64)
        key = (np.uint64(_NOISE_SEED) ^ (np.uint64(meta.level) << np.uint64(60))
               ^ (k[:, None, None] << np.uint64(40)) ^ (j[None, :, None] << np.uint64(20))
               ^ i[None, None, :])
        bits = _splitmix64_numpy(key)
        return np.ascontiguousarray((bits >> np.uint64(11)).astype(np.float64) / float(1 << 53))
    x, y, z = _cell_centres(meta, spec, np)
    X, Y, Z = x[None, None, :], y[None, :, None], z[:, None, None]
    if spec.field == "radial":
        # the in-situ example's field x^2+y^2+z^2 (RenderFromMultiFab.cpp:42-45), scaled to [0,1]
        return np.ascontiguousarray((X * X + Y * Y + Z * Z) / 3.0)
    if spec.field != "smooth":
        raise ValueError(f"unknown field {spec.field!r}")
    value = 0.5 * (X * X + Y * Y + Z * Z) / 3.0
    for (cx, cy, cz), sigma in _BLOBS:
        d2 = (X - cx) ** 2 + (Y - cy) ** 2 + (Z - cz) ** 2
        value = value + 0.125 * np.exp(-d2 / (sigma * sigma))
    return np.ascontiguousarray(value)


def box_cells_torch(spec: SceneSpec, index: int, device):
    """Same fields evaluated with torch on `device` (float64 [nz, ny, nx]).  Values agree with
    box_cells_numpy to rounding, not bit for bit -- always feed ONE of them to both sides of a
    comparison."""
    import torch
    meta = spec.boxes[index]
    if spec.field == "noise":
        mask34 = (1 << 34) - 1
        mask37 = (1 << 37) - 1
        mask33 = (1 << 33) - 1

        def wrap(v: int) -> int:  # two's-complement int64 view of a uint64 constant
            return v - (1 << 64) if v >= (1 << 63) else v

        i = torch.arange(meta.lo[0], meta.lo[0] + meta.dims[0], dtype=torch.int64, device=device)
        j = torch.arange(meta.lo[1], meta.lo[1] + meta.dims[1], dtype=torch.int64, device=device)
        k = torch.arange(meta.lo[2], meta.lo[2] + meta.dims[2], dtype=torch.int64, device=device)
        key = (wrap((_NOISE_SEED ^ (meta.level << 60)) & ((1 << 64) - 1))
               ^ (k[:, None, None] << 40) ^ (j[None, :, None] << 20) ^ i[None, None, :])
        z = key + wrap(0x9E3779B97F4A7C15)
        z = (z ^ ((z >> 30) & mask34)) * wrap(0xBF58476D1CE4E5B9)
        z = (z ^ ((z >> 27) & mask37)) * wrap(0x94D049BB133111EB)
        z = z ^ ((z >> 31) & mask33)
        top53 = (z >> 11) & ((1 << 53) - 1)
        return (top53.to(torch.float64) / float(1 << 53)).contiguous()
    x, y, z = _cell_centres(meta, spec, torch, device=device)
    X, Y, Z = x[None, None, :], y[None, :, None], z[:, None, None]
    if spec.field == "radial":
        return ((X * X + Y * Y + Z * Z) / 3.0).contiguous()
    if spec.field != "smooth":
        raise ValueError(f"unknown field {spec.field!r}")
    value = 0.5 * (X * X + Y * Y + Z * Z) / 3.0
    for (cx, cy, cz), sigma in _BLOBS:
        d2 = (X - cx) ** 2 + (Y - cy) ** 2 + (Z - cz) ** 2
        value = value + 0.125 * torch.exp(-d2 / (sigma * sigma))
    return value.contiguous()


def amr_box(spec: SceneSpec, index: int, values) -> AmrBox:
    meta = spec.boxes[index]
    return AmrBox(min_corner=meta.min_corner, max_corner=meta.max_corner, values=values,
                  level=meta.level, owner=meta.owner)


def metadata_box(spec: SceneSpec, index: int) -> AmrBox:
    """An AmrBox without cell data (host-only quantities: hints, sampling constants)."""
    meta = spec.boxes[index]
    return AmrBox(min_corner=meta.min_corner, max_corner=meta.max_corner, values=None,
                  level=meta.level, dims=meta.dims, owner=meta.owner)
