"""AMReX plotfile ingestion without AMReX.

What the reference does through amrex::PlotFileData, amrex::convexify and
detail::BuildSceneGeometry (VolumeRenderer/VolumeRenderer.cpp:587-711,
VolumeRenderer/SceneBuilder.cpp:112-313) is rebuilt here from the on-disk format:

  <plotfile>/Header                 text: variables, levels, geometry, grids
  <plotfile>/Level_<n>/Cell_H       text: VisMF header (box array, FabOnDisk file + offset)
  <plotfile>/Level_<n>/Cell_D_xxxxx binary FABs: "FAB ((8, (64 11 52 0 1 12 0 1023)),
                                    (8, (8 7 6 5 4 3 2 1)))((lo) (hi) (type)) ncomp\\n" + the
                                    values, x fastest, component slowest

AMReX itself (26.04, fetched by the reference's CMakeLists.txt:43-52) is not vendored in the
reference tree; the format is the published "HyperCLaw-V1.1" plotfile / VisMF layout.  The file
IO is host code (numpy); the cells go to HBM once and the boxes the renderer sees are strided
views into those grids (AmrBox carries jstride / kstride like amrex::Array4).

`convexify` here is our own re-boxing: a coarse grid minus the coarsened grids of the next finer
level, cut with AMReX's published boxDiff splitting order.  The reference calls amrex::convexify,
whose exact box list (order, merging) is not reproduced, so frames rendered from the same
plotfile can differ from the reference's at the pixels where rays cross a box seam.
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

IntBox = Tuple[Tuple[int, int, int], Tuple[int, int, int]]  # inclusive (lo, hi), cell-centred

_BOX_RE = re.compile(r"\(\((-?\d+),(-?\d+),(-?\d+)\) \((-?\d+),(-?\d+),(-?\d+)\) \((\d+),(\d+),(\d+)\)\)")
_FAB_RE = re.compile(rb"FAB \(\((\d+), \(([\d ]+)\)\),\((\d+), \(([\d ]+)\)\)\)"
                     rb"\(\((-?\d+),(-?\d+),(-?\d+)\) \((-?\d+),(-?\d+),(-?\d+)\) "
                     rb"\((\d+),(\d+),(\d+)\)\) (\d+)\n")


def _parse_box(text: str) -> IntBox:
    m = _BOX_RE.search(text)
    if not m:
        raise RuntimeError(f"cannot parse box {text!r}")
    v = [int(x) for x in m.groups()]
    if v[6:9] != [0, 0, 0]:
        raise RuntimeError("only cell-centred data is supported")
    return (v[0], v[1], v[2]), (v[3], v[4], v[5])


@dataclass
class _LevelHeader:
    n_comp: int = 0
    n_ghost: int = 0
    boxes: List[IntBox] = field(default_factory=list)
    fab_on_disk: List[Tuple[str, int]] = field(default_factory=list)


class PlotFileData:
    """The part of amrex::PlotFileData the renderer uses (VolumeRenderer.cpp:598-666)."""

    def __init__(self, path: str):
        self.path = path
        header = os.path.join(path, "Header")
        if not os.path.isfile(header):
            raise RuntimeError(f"'{path}' is not a plotfile: no Header")
        with open(header) as fh:
            lines = [line.rstrip("\n") for line in fh]
        it = iter(lines)
        self.version = next(it).strip()
        n_comp = int(next(it))
        self.var_names = [next(it).strip() for _ in range(n_comp)]
        self.space_dim = int(next(it))
        self.time = float(next(it))
        self.finest_level = int(next(it))
        n_levels = self.finest_level + 1
        self.prob_lo = tuple(float(x) for x in next(it).split())
        self.prob_hi = tuple(float(x) for x in next(it).split())
        self.ref_ratio = [int(x) for x in next(it).split()][:self.finest_level]
        domain_line = next(it)
        self.prob_domain = [_parse_box(m.group(0)) for m in _BOX_RE.finditer(domain_line)]
        self.level_steps = [int(x) for x in next(it).split()]
        self.cell_size = [tuple(float(x) for x in next(it).split()) for _ in range(n_levels)]
        self.coord_sys = int(next(it))
        next(it)  # boundary width
        self.level_paths: List[str] = []
        self.grid_counts: List[int] = []
        for level in range(n_levels):
            parts = next(it).split()
            if int(parts[0]) != level:
                raise RuntimeError("malformed plotfile Header: level block out of order")
            n_grids = int(parts[1])
            self.grid_counts.append(n_grids)
            next(it)  # level step
            for _ in range(n_grids * self.space_dim):
                next(it)  # physical extent of the grid
            self.level_paths.append(next(it).strip())
        self._levels: Dict[int, _LevelHeader] = {}

    # -- VisMF header of one level -------------------------------------------------------------
    def _level(self, level: int) -> _LevelHeader:
        if level in self._levels:
            return self._levels[level]
        if not (0 <= level <= self.finest_level):
            raise IndexError("level out of range")
        path = os.path.join(self.path, self.level_paths[level] + "_H")
        with open(path) as fh:
            lines = [line.rstrip("\n") for line in fh]
        it = iter(lines)
        next(it)  # VisMF version
        next(it)  # how the FABs were distributed over files
        out = _LevelHeader()
        out.n_comp = int(next(it))
        ghost = next(it).strip()
        out.n_ghost = max(int(x) for x in re.findall(r"-?\d+", ghost))
        n_boxes = int(next(it).strip().lstrip("(").split()[0])
        for _ in range(n_boxes):
            out.boxes.append(_parse_box(next(it)))
        next(it)  # ")"
        n_fabs = int(next(it))
        for _ in range(n_fabs):
            _, name, offset = next(it).split()
            out.fab_on_disk.append((name, int(offset)))
        if n_fabs != n_boxes:
            raise RuntimeError("malformed VisMF header: FabOnDisk count differs from the box array")
        self._levels[level] = out
        return out

    def boxes(self, level: int) -> List[IntBox]:
        return list(self._level(level).boxes)

    def get(self, level: int, name: str, grids: Optional[Sequence[int]] = None
            ) -> Dict[int, np.ndarray]:
        """PlotFileData::get(level, name): the component's cells of the requested grids (all by
        default) as float64 arrays [nz, ny, nx] over the valid box."""
        if name not in self.var_names:
            raise RuntimeError(f"Variable '{name}' not found in plotfile '{self.path}'.")
        comp = self.var_names.index(name)
        header = self._level(level)
        wanted = range(len(header.boxes)) if grids is None else grids
        out: Dict[int, np.ndarray] = {}
        directory = os.path.dirname(os.path.join(self.path, self.level_paths[level]))
        for g in wanted:
            file_name, offset = header.fab_on_disk[g]
            with open(os.path.join(directory, file_name), "rb") as fh:
                fh.seek(offset)
                head = fh.readline()
                m = _FAB_RE.match(head)
                if not m:
                    raise RuntimeError(f"cannot parse FAB header {head!r}")
                real_bytes = int(m.group(3))
                order = [int(x) for x in m.group(4).split()]
                lo = tuple(int(m.group(i)) for i in (5, 6, 7))
                hi = tuple(int(m.group(i)) for i in (8, 9, 10))
                n_comp = int(m.group(14))
                if comp >= n_comp:
                    raise RuntimeError("FAB holds fewer components than the Header names")
                little = order == list(range(real_bytes, 0, -1))
                if not little and order != list(range(1, real_bytes + 1)):
                    raise RuntimeError(f"unsupported byte order {order}")
                dtype = np.dtype({4: "f4", 8: "f8"}[real_bytes]).newbyteorder("<" if little else ">")
                dims = tuple(h - l + 1 for l, h in zip(lo, hi))
                n = dims[0] * dims[1] * dims[2]
                fh.seek(comp * n * real_bytes, os.SEEK_CUR)
                data = np.fromfile(fh, dtype=dtype, count=n)
            if data.size != n:
                raise RuntimeError("truncated FAB data")
            cells = data.astype(np.float64).reshape(dims[2], dims[1], dims[0])
            vlo, vhi = header.boxes[g]
            if (lo, hi) != (vlo, vhi):  # the FAB on disk carries ghost cells: keep the valid box
                sl = tuple(slice(vlo[a] - lo[a], vhi[a] - lo[a] + 1) for a in (2, 1, 0))
                cells = np.ascontiguousarray(cells[sl])
            out[g] = cells
        return out


# ---- writer (fixtures for the tests, export of synthetic scenes) ---------------------------------

def write_plotfile(path: str, var_names: Sequence[str], levels: Sequence[dict],
                   prob_lo: Sequence[float], prob_hi: Sequence[float],
                   ref_ratio: Sequence[int], time: float = 0.0, grids_per_file: int = 4,
                   real_bytes: int = 8) -> None:
    """Writes a HyperCLaw-V1.1 plotfile.  levels[l] = {"domain": IntBox, "boxes": [IntBox],
    "data": [array [ncomp, nz, ny, nx]]}; cell sizes follow from prob_lo/hi and the domain."""
    os.makedirs(path, exist_ok=True)
    n_levels = len(levels)
    n_comp = len(var_names)
    dx = []
    for lev in levels:
        dlo, dhi = lev["domain"]
        dx.append(tuple((prob_hi[a] - prob_lo[a]) / (dhi[a] - dlo[a] + 1) for a in range(3)))

    def box_text(box: IntBox) -> str:
        lo, hi = box
        return f"(({lo[0]},{lo[1]},{lo[2]}) ({hi[0]},{hi[1]},{hi[2]}) (0,0,0))"

    def real(v: float) -> str:
        return repr(float(v))

    with open(os.path.join(path, "Header"), "w") as fh:
        fh.write("HyperCLaw-V1.1\n")
        fh.write(f"{n_comp}\n")
        for name in var_names:
            fh.write(name + "\n")
        fh.write("3\n")
        fh.write(real(time) + "\n")
        fh.write(f"{n_levels - 1}\n")
        fh.write(" ".join(real(v) for v in prob_lo) + " \n")
        fh.write(" ".join(real(v) for v in prob_hi) + " \n")
        fh.write(" ".join(str(int(r)) for r in ref_ratio[:n_levels - 1]) + " \n")
        fh.write(" ".join(box_text(lev["domain"]) for lev in levels) + " \n")
        fh.write(" ".join("0" for _ in levels) + " \n")
        for d in dx:
            fh.write(" ".join(real(v) for v in d) + " \n")
        fh.write("0\n0\n")
        for level, lev in enumerate(levels):
            fh.write(f"{level} {len(lev['boxes'])} {real(time)}\n0\n")
            for lo, hi in lev["boxes"]:
                for a in range(3):
                    fh.write(f"{real(prob_lo[a] + lo[a] * dx[level][a])} "
                             f"{real(prob_lo[a] + (hi[a] + 1) * dx[level][a])}\n")
            fh.write(f"Level_{level}/Cell\n")

    descriptor = {8: b"((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))",
                  4: b"((8, (32 8 23 0 1 9 0 127)),(4, (4 3 2 1)))"}[real_bytes]
    dtype = {8: "<f8", 4: "<f4"}[real_bytes]
    for level, lev in enumerate(levels):
        directory = os.path.join(path, f"Level_{level}")
        os.makedirs(directory, exist_ok=True)
        on_disk = []
        open_files: Dict[str, int] = {}
        for g, (box, data) in enumerate(zip(lev["boxes"], lev["data"])):
            name = f"Cell_D_{g // max(grids_per_file, 1):05d}"
            lo, hi = box
            dims = tuple(h - l + 1 for l, h in zip(lo, hi))
            arr = np.asarray(data, dtype=np.float64)
            if arr.shape != (n_comp, dims[2], dims[1], dims[0]):
                raise ValueError(f"grid {g} of level {level}: data shape {arr.shape} does not "
                                 f"match its box {box}")
            mode = "ab" if name in open_files else "wb"
            with open(os.path.join(directory, name), mode) as fh:
                offset = open_files.get(name, 0)
                head = b"FAB " + descriptor + box_text(box).encode() + f" {n_comp}\n".encode()
                fh.write(head)
                payload = arr.astype(dtype).tobytes()
                fh.write(payload)
                open_files[name] = offset + len(head) + len(payload)
            on_disk.append((name, offset))
        with open(os.path.join(directory, "Cell_H"), "w") as fh:
            fh.write(f"1\n0\n{n_comp}\n0\n")
            fh.write(f"({len(lev['boxes'])} 0\n")
            for box in lev["boxes"]:
                fh.write(box_text(box) + "\n")
            fh.write(")\n")
            fh.write(f"{len(on_disk)}\n")
            for name, offset in on_disk:
                fh.write(f"FabOnDisk: {name} {offset}\n")
            fh.write("\n")
            for reducer in (np.min, np.max):
                fh.write(f"{len(lev['boxes'])},{n_comp}\n")
                for data in lev["data"]:
                    arr = np.asarray(data, dtype=np.float64)
                    fh.write(",".join(repr(float(reducer(arr[c]))) for c in range(n_comp)) + ",\n")
                fh.write("\n")


# ---- convexify ---------------------------------------------------------------------------------

def _intersects(a: IntBox, b: IntBox) -> bool:
    return all(a[0][d] <= b[1][d] and b[0][d] <= a[1][d] for d in range(3))


def box_diff(b1: IntBox, b2: IntBox) -> List[IntBox]:
    """b1 minus b2 as disjoint boxes, in amrex::boxDiff's order: dimensions from the last to the
    first, the slab below b2 then the slab above it, the remainder shrinking as it goes."""
    if not _intersects(b1, b2):
        return [b1]
    lo, hi = list(b1[0]), list(b1[1])
    out: List[IntBox] = []
    for d in (2, 1, 0):
        if lo[d] < b2[0][d] <= hi[d]:
            piece_hi = list(hi)
            piece_hi[d] = b2[0][d] - 1
            out.append((tuple(lo), tuple(piece_hi)))
            lo[d] = b2[0][d]
        if lo[d] <= b2[1][d] < hi[d]:
            piece_lo = list(lo)
            piece_lo[d] = b2[1][d] + 1
            out.append((tuple(piece_lo), tuple(hi)))
            hi[d] = b2[1][d]
    return out


def _coarsen(box: IntBox, ratio: int) -> IntBox:
    return (tuple(v // ratio for v in box[0]), tuple(v // ratio for v in box[1]))  # floor division


def uncovered_parts(grid: IntBox, fine_boxes: Sequence[IntBox], ratio: int) -> List[IntBox]:
    """The cells of `grid` that no box of the next finer level covers."""
    parts = [grid]
    for fine in fine_boxes:
        coarse = _coarsen(fine, ratio)
        if not _intersects(grid, coarse):
            continue
        parts = [piece for part in parts for piece in box_diff(part, coarse)]
        if not parts:
            break
    return parts


def convexify(level_boxes: Sequence[Sequence[IntBox]], ref_ratio: Sequence[int]
              ) -> List[List[Tuple[int, IntBox]]]:
    """Per level, the list of (parent grid index, sub-box): every level keeps only the cells the
    next finer level does not cover; the finest level is kept whole (what amrex::convexify
    delivers to VolumeRenderer.cpp:668-669, with our own box list)."""
    n_levels = len(level_boxes)
    out: List[List[Tuple[int, IntBox]]] = []
    for level in range(n_levels):
        if level == n_levels - 1:
            out.append([(g, box) for g, box in enumerate(level_boxes[level])])
            continue
        entries = []
        for g, grid in enumerate(level_boxes[level]):
            for part in uncovered_parts(grid, level_boxes[level + 1], ref_ratio[level]):
                entries.append((g, part))
        out.append(entries)
    return out


# ---- scene construction --------------------------------------------------------------------------

def _morton3(x: int, y: int, z: int, bits: int = 20) -> int:
    key = 0
    for b in range(bits):
        key |= ((x >> b) & 1) << (3 * b) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b + 2)
    return key


def assign_box_owners(boxes, n_ranks: int) -> None:
    """Sort-last partition of the boxes (our design; the reference inherits AMReX's
    DistributionMapping).  Per AMR level: that level's boxes along the Morton curve of their
    centres, cut into 2N stretches of equal cell count; rank r takes stretch r and stretch
    2N-1-r.  The two ends of the curve are opposite corners of the domain, so every rank owns, on
    every level, a region near the eye and one far from it whatever the view direction, the same
    number of cells (classify bytes) and the same mix of long coarse and short fine rays (a coarse
    box holds 16x the samples of a fine box of the same cell count; with one chunk of the merged
    curve per rank the rank next to the eye marched 1.6x the samples of the one behind it).
    scenes.assign_owners calls the same rule "level_pairs"."""
    if not boxes:
        return
    lo = [min(b.min_corner[a] for b in boxes) for a in range(3)]
    hi = [max(b.max_corner[a] for b in boxes) for a in range(3)]
    span = [max(hi[a] - lo[a], 1e-300) for a in range(3)]

    def key(i: int) -> int:
        b = boxes[i]
        q = [min(int(((0.5 * (b.min_corner[a] + b.max_corner[a]) - lo[a]) / span[a]) * (1 << 20)),
                 (1 << 20) - 1) for a in range(3)]
        return _morton3(*q)

    stretches = 2 * n_ranks
    for level in sorted({b.level for b in boxes}):
        ranked = sorted((i for i in range(len(boxes)) if boxes[i].level == level),
                        key=lambda i: (key(i), i))
        cells = [boxes[i].cell_dimensions[0] * boxes[i].cell_dimensions[1] *
                 boxes[i].cell_dimensions[2] for i in ranked]
        total = float(sum(cells))
        running = 0.0
        for i, c in zip(ranked, cells):
            mid = running + 0.5 * c
            k = min(int(mid * stretches / total), stretches - 1) if total > 0 else 0
            boxes[i].owner = k if k < n_ranks else stretches - 1 - k
            running += c


def build_scene_from_levels(ctx, level_boxes, cell_sizes, prob_lo, ref_ratio, fetch_grids,
                            min_level: int, max_level: int, log_scale_input: bool,
                            normalize_to_data_range: bool, rank: int, n_ranks: int,
                            process_group, no_data_error: str):
    """amrex::convexify + detail::BuildSceneGeometry (SceneBuilder.cpp:112-425) for levels given
    as box lists: level_boxes[l] = [IntBox], cell_sizes[l] = (dx, dy, dz), fetch_grids(level,
    [grid indices]) -> {grid index: float64 array [nz, ny, nx] (numpy or a tensor on ctx.device)}.
    Only the grids this rank's boxes come from are fetched."""
    import torch
    from . import api
    from .types import AmrBox, VolumeBounds
    convex = convexify(level_boxes[:max_level + 1], list(ref_ratio)[:max_level])

    # world-space boxes, level-major then grid order (MFIter order, SceneBuilder.cpp:134-188)
    entries = []  # (level, parent grid, sub-box)
    boxes = []
    for level in range(min_level, max_level + 1):
        dx = cell_sizes[level]
        for parent, (lo, hi) in convex[level]:
            dims = tuple(hi[a] - lo[a] + 1 for a in range(3))
            if min(dims) <= 0:
                continue
            min_corner = tuple(prob_lo[a] + float(lo[a]) * dx[a] for a in range(3))
            max_corner = tuple(prob_lo[a] + float(hi[a] + 1) * dx[a] for a in range(3))
            entries.append((level, parent, (lo, hi)))
            boxes.append(AmrBox(min_corner, max_corner, level=level, dims=dims))
    if not boxes:
        raise RuntimeError(no_data_error)

    # global rescale: the shortest edge of the data's bounding box becomes 1 (:229-254)
    gmin = [min(b.min_corner[a] for b in boxes) for a in range(3)]
    gmax = [max(b.max_corner[a] for b in boxes) for a in range(3)]
    min_extent = float("inf")
    for a in range(3):
        length = abs(gmax[a] - gmin[a])
        if length > 0.0 and np.isfinite(length):
            min_extent = min(min_extent, length)
    scale = 1.0 / min_extent if (min_extent > 0.0 and np.isfinite(min_extent)) else 1.0
    if not np.isfinite(scale) or not (scale > 0.0):
        scale = 1.0
    if scale != 1.0:
        for b in boxes:
            b.min_corner = tuple(v * scale for v in b.min_corner)
            b.max_corner = tuple(v * scale for v in b.max_corner)
    gmin = [min(b.min_corner[a] for b in boxes) for a in range(3)]
    gmax = [max(b.max_corner[a] for b in boxes) for a in range(3)]
    extent = [gmax[a] - gmin[a] for a in range(3)]
    max_extent = max(extent)
    padding = max_extent * 0.05 if max_extent > 0.0 else 1.0
    bounds = VolumeBounds(tuple(v - padding for v in gmin), tuple(v + padding for v in gmax))

    assign_box_owners(boxes, n_ranks)

    # cells: each parent grid of a local box is fetched and uploaded once; boxes are views into it
    needed = {}
    for (level, parent, _), b in zip(entries, boxes):
        if b.owner == rank:
            needed.setdefault(level, set()).add(parent)
    grids = {}
    for level, parents in needed.items():
        for parent, cells in fetch_grids(level, sorted(parents)).items():
            tensor = cells if isinstance(cells, torch.Tensor) else torch.from_numpy(
                np.ascontiguousarray(cells, dtype=np.float64))
            grids[(level, parent)] = tensor.to(device=ctx.device, dtype=torch.float64)
    local = []
    for (level, parent, (lo, hi)), b in zip(entries, boxes):
        if b.owner != rank:
            continue
        glo, ghi = level_boxes[level][parent]
        grid = grids[(level, parent)]
        if tuple(grid.shape) != tuple(ghi[a] - glo[a] + 1 for a in (2, 1, 0)):
            raise ValueError(f"grid {parent} of level {level} does not match its box")
        view = grid[lo[2] - glo[2]:hi[2] - glo[2] + 1, lo[1] - glo[1]:hi[1] - glo[1] + 1,
                    lo[0] - glo[0]:hi[0] - glo[0] + 1]
        local.append(AmrBox(b.min_corner, b.max_corner, view, b.level, owner=rank))
    return api.build_scene_geometry(ctx, boxes, local, bounds, log_scale_input,
                                    normalize_to_data_range, process_group, n_ranks)


def clamp_levels(requested_min_level: int, requested_max_level: int, finest: int):
    """The level clamps of loadPlotFileGeometry / loadMultiFabGeometry
    (VolumeRenderer.cpp:626-642, VolumeRendererApi.cpp:58-72)."""
    min_level = min(max(requested_min_level, 0), finest)
    max_level = requested_max_level
    if max_level < 0 or max_level > finest:
        max_level = finest
    return min_level, max_level


def load_plotfile_geometry(ctx, plotfile_path: str, variable_name: str = "",
                           requested_min_level: int = 0, requested_max_level: int = -1,
                           log_scale_input: bool = False, normalize_to_data_range: bool = True,
                           rank: int = 0, n_ranks: int = 1, process_group=None):
    """VolumeRenderer::loadPlotFileGeometry (VolumeRenderer.cpp:587-711) + the geometric part of
    detail::BuildSceneGeometry (SceneBuilder.cpp:112-313): reads the requested component,
    removes the cells covered by finer levels, builds world-space boxes (probLo + index * cell
    size, rescaled so that the shortest domain edge is 1), the padded bounds and the scalar
    transform.  Every rank reads only the grids its own boxes come from."""
    if not plotfile_path:
        raise ValueError("Plotfile path must not be empty.")
    plotfile = PlotFileData(plotfile_path)
    if plotfile.space_dim != 3:
        raise RuntimeError(f"Plotfile '{plotfile_path}' has space dimension {plotfile.space_dim}. "
                           "The volume renderer currently expects 3D data.")
    if not plotfile.var_names:
        raise RuntimeError("Plotfile contains no cell variables to render.")
    component = variable_name or plotfile.var_names[0]
    if component not in plotfile.var_names:
        raise RuntimeError(f"Variable '{component}' not found in plotfile '{plotfile_path}'.")
    min_level, max_level = clamp_levels(requested_min_level, requested_max_level,
                                        plotfile.finest_level)
    if min_level > max_level:
        raise RuntimeError(f"Minimum AMR level {min_level} exceeds available maximum level "
                           f"{max_level}.")
    level_boxes = [plotfile.boxes(level) for level in range(max_level + 1)]
    return build_scene_from_levels(
        ctx, level_boxes, plotfile.cell_size, plotfile.prob_lo, plotfile.ref_ratio,
        lambda level, grid_ids: plotfile.get(level, component, grid_ids), min_level, max_level,
        log_scale_input, normalize_to_data_range, rank, n_ranks, process_group,
        "Failed to locate any volumetric data within the plotfile.")
