"""CPU restatement of BuildVisibilityOrderedGroup (TEST INFRASTRUCTURE ONLY).

Follows Common/VisibilityOrdering.cpp:63-632 of the reference line by line in numpy float32 /
float64 scalars (pure-Python loops: use it on small box sets).  Only tests/ may import it.

The MPI allgathers of the reference (:86-152) concatenate every rank's localBoxes in rank order;
here `boxes` is that concatenation already: a list of (min_corner, max_corner, owner) in rank-major
order.  Third-party arithmetic restated (AMReX 26.04, fetched by the reference's CMakeLists.txt:43-52,
not vendored): amrex::RealVect is 3 doubles (vectorLength = sqrt of the sum of squares,
crossProduct, dotProduct as published); amrex::SmallMatrix<float,N,M> operator* accumulates
r(i,j) += a(i,k) * b(k,j) for k = 0..N-1 from a zero-initialised result, in float.

Parity: unpinned -- the reference has no test or vector for this function.  SURVEY.md 8c probe (1)
shows the result is pixel-neutral (any group order composites to the same bits).
"""
from __future__ import annotations

import ctypes
import ctypes.util
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

f32 = np.float32
INF = f32(np.inf)

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.tanf.restype = ctypes.c_float
_libm.tanf.argtypes = [ctypes.c_float]


def _tanf(x) -> np.float32:
    """std::tan(float) of the host libm the reference's host code would call."""
    return f32(_libm.tanf(float(x)))


def _safe_normalize(v):  # Common/CameraUtils.hpp:17-23 (double)
    length = math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    if length > 0.0 and math.isfinite(length):
        return [v[0] / length, v[1] / length, v[2] / length]
    return [0.0, 0.0, -1.0]


def _cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def _dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def make_view_matrix(eye, look_at, up) -> np.ndarray:  # Common/CameraUtils.hpp:25-63
    forward = _safe_normalize([look_at[c] - eye[c] for c in range(3)])
    right = _cross(forward, up)
    right_length = math.sqrt(_dot(right, right))
    if right_length > 0.0 and math.isfinite(right_length):
        right = [r / right_length for r in right]
    else:
        right = [1.0, 0.0, 0.0]
    up_ortho = _cross(right, forward)
    view = np.identity(4, dtype=np.float32)
    for c in range(3):
        view[c, 0] = f32(right[c])
        view[c, 1] = f32(up_ortho[c])
        view[c, 2] = f32(-forward[c])
        view[c, 3] = f32(0.0)
    view[3, 0] = f32(-_dot(right, eye))
    view[3, 1] = f32(-_dot(up_ortho, eye))
    view[3, 2] = f32(_dot(forward, eye))
    view[3, 3] = f32(1.0)
    return view


def make_perspective_matrix(fov_y_degrees, aspect, near_plane, far_plane) -> np.ndarray:
    """VisibilityOrdering.cpp:35-59, float."""
    m = np.identity(4, dtype=np.float32)
    k_pi = f32(3.14159265358979323846)
    fov_y_degrees, aspect = f32(fov_y_degrees), f32(aspect)
    near_plane, far_plane = f32(near_plane), f32(far_plane)
    # std::tan(float) -> float; the argument is evaluated left to right in float
    fov_tangent = _tanf(f32(f32(f32(fov_y_degrees * k_pi) / f32(180.0)) * f32(0.5)))
    size = f32(near_plane * fov_tangent)
    left, right = f32(f32(-size) * aspect), f32(size * aspect)
    bottom, top = f32(-size), size
    two = f32(2.0)
    m[0, 0] = f32(f32(two * near_plane) / f32(right - left))
    m[1, 1] = f32(f32(two * near_plane) / f32(top - bottom))
    m[0, 2] = f32(f32(right + left) / f32(right - left))
    m[1, 2] = f32(f32(top + bottom) / f32(top - bottom))
    m[2, 2] = f32(f32(-f32(far_plane + near_plane)) / f32(far_plane - near_plane))
    m[3, 2] = f32(-1.0)
    m[2, 3] = f32(f32(-f32(f32(two * far_plane) * near_plane)) / f32(far_plane - near_plane))
    m[3, 3] = f32(0.0)
    return m


def _matvec(m, v):  # SmallMatrix operator*: r(i) = ((0 + m(i,0) v0) + m(i,1) v1) + ...
    out = []
    for i in range(4):
        acc = f32(0.0)
        for k in range(4):
            acc = f32(acc + f32(m[i, k] * v[k]))
        out.append(acc)
    return out


class _Box:
    __slots__ = ("lo", "hi", "owner", "min_depth", "max_depth")

    def __init__(self, lo, hi, owner):
        self.lo = [float(f32(x)) for x in lo]   # floats travel through MPI_FLOAT (:107-119)
        self.hi = [float(f32(x)) for x in hi]
        self.owner = owner
        self.min_depth = INF
        self.max_depth = INF

    def copy(self):
        b = _Box(self.lo, self.hi, self.owner)
        b.min_depth, b.max_depth = self.min_depth, self.max_depth
        return b


def _depth_range(modelview, projection, lo, hi):  # :165-192
    min_depth, max_depth = INF, f32(-np.inf)
    for corner_index in range(8):
        corner = [hi[0] if corner_index & 1 else lo[0], hi[1] if corner_index & 2 else lo[1],
                  hi[2] if corner_index & 4 else lo[2]]
        homogeneous = [f32(corner[0]), f32(corner[1]), f32(corner[2]), f32(1.0)]
        clip = _matvec(projection, _matvec(modelview, homogeneous))
        if clip[3] != f32(0.0):
            with np.errstate(all="ignore"):
                depth = f32(clip[2] / clip[3])
            # std::min(a, b) = (b < a) ? b : a; std::max(a, b) = (a < b) ? b : a
            min_depth = depth if depth < min_depth else min_depth
            max_depth = depth if max_depth < depth else max_depth
    if not np.isfinite(min_depth) or not np.isfinite(max_depth):
        min_depth, max_depth = INF, INF
    return min_depth, max_depth


def _nearly_equal(a, b):  # :213-216
    a, b = f32(a), f32(b)
    scale = max(f32(1.0), abs(a), abs(b))
    return abs(f32(a - b)) <= f32(f32(1e-5) * scale)


def _overlaps(a_min, a_max, b_min, b_max):  # :218-230
    a_min, a_max, b_min, b_max = f32(a_min), f32(a_max), f32(b_min), f32(b_max)
    overlap_min = max(a_min, b_min)
    overlap_max = min(a_max, b_max)
    scale = max(f32(1.0), abs(a_min), abs(a_max), abs(b_min), abs(b_max), abs(overlap_min),
                abs(overlap_max))
    return f32(overlap_max - overlap_min) > f32(f32(1e-5) * scale)


K_DIRECTION_TOLERANCE = f32(1e-6)


def visibility_order(boxes: Sequence[Tuple[Sequence[float], Sequence[float], int]], n_ranks: int,
                     eye, look_at, up, fov_y_degrees, near_plane, far_plane, aspect,
                     use_visibility_graph: bool = True,
                     dot_files: Optional[List[str]] = None) -> Tuple[List[int], int, bool]:
    """Returns (rank order, number of cycle-breaking splits, graph ordering succeeded).
    dot_files, if a list, receives the text of every exported graph (:318-350)."""
    default_order = list(range(n_ranks))
    if not use_visibility_graph:
        return default_order, 0, True
    if len(boxes) <= 0:
        return default_order, 0, True
    modelview = make_view_matrix(eye, look_at, up)
    projection = make_perspective_matrix(fov_y_degrees, aspect, near_plane, far_plane)
    current = [_Box(lo, hi, owner) for lo, hi, owner in boxes]
    for b in current:
        b.min_depth, b.max_depth = _depth_range(modelview, projection, b.lo, b.hi)
    view_dir = _safe_normalize([look_at[c] - eye[c] for c in range(3)])

    def compare_key(index):  # compareBoxes (:240-258) is a lexicographic total order
        b = current[index]
        finite = bool(np.isfinite(b.min_depth))
        return (0 if finite else 1, float(b.min_depth) if finite else 0.0,
                float(b.max_depth) if finite else 0.0, b.owner, index)

    def rebuild_adjacency():  # :262-316
        n = len(current)
        adjacency = [[] for _ in range(n)]
        indegree = [0] * n

        def add_edge(a, b):
            if a == b:
                return
            if b not in adjacency[a]:
                adjacency[a].append(b)
                indegree[b] += 1

        for i in range(n):
            a = current[i]
            for j in range(i + 1, n):
                b = current[j]
                for axis in range(3):
                    axis1, axis2 = (axis + 1) % 3, (axis + 2) % 3
                    if not _overlaps(a.lo[axis1], a.hi[axis1], b.lo[axis1], b.hi[axis1]):
                        continue
                    if not _overlaps(a.lo[axis2], a.hi[axis2], b.lo[axis2], b.hi[axis2]):
                        continue
                    dir_component = f32(view_dir[axis])
                    if _nearly_equal(a.hi[axis], b.lo[axis]):
                        if dir_component > K_DIRECTION_TOLERANCE:
                            add_edge(j, i)
                        elif dir_component < -K_DIRECTION_TOLERANCE:
                            add_edge(i, j)
                    elif _nearly_equal(b.hi[axis], a.lo[axis]):
                        if dir_component > K_DIRECTION_TOLERANCE:
                            add_edge(i, j)
                        elif dir_component < -K_DIRECTION_TOLERANCE:
                            add_edge(j, i)
        return adjacency, indegree

    def export_graph(adjacency):  # :318-350
        if dot_files is None:
            return
        lines = ["digraph VisibilityGraph {", "  rankdir=LR;"]
        for idx, info in enumerate(current):
            lines.append(f'  box{idx} [label="box {idx}\\nrank {info.owner}'
                         f'\\nminDepth {_fixed6(info.min_depth)}'
                         f'\\nmaxDepth {_fixed6(info.max_depth)}"];')
        for a, edges in enumerate(adjacency):
            for b in edges:
                lines.append(f"  box{a} -> box{b};")
        lines.append("}")
        dot_files.append("\n".join(lines) + "\n")

    def topo_sort(adjacency, indegree):  # :358-399
        indegree = list(indegree)
        ready = sorted((i for i in range(len(current)) if indegree[i] == 0), key=compare_key)
        order = []
        while ready:
            node = ready.pop(0)
            order.append(node)
            for nxt in adjacency[node]:
                indegree[nxt] -= 1
                if indegree[nxt] == 0:
                    ready.append(nxt)
            ready.sort(key=compare_key)
        return len(order) == len(current), order, indegree

    def find_cycle(adjacency, residual):  # :401-445
        n = len(adjacency)
        state, parent, cycle = [0] * n, [-1] * n, []

        def dfs(node):
            state[node] = 1
            for nxt in adjacency[node]:
                if state[nxt] == 0:
                    parent[nxt] = node
                    if dfs(nxt):
                        return True
                elif state[nxt] == 1:
                    cycle.clear()
                    cycle.append(nxt)
                    cur = node
                    while cur != nxt and cur != -1:
                        cycle.append(cur)
                        cur = parent[cur]
                    cycle.reverse()
                    return True
            state[node] = 2
            return False

        for node in range(n):
            if residual[node] > 0 and state[node] == 0:
                if dfs(node):
                    break
        return cycle

    def break_cycle(cycle):  # :447-569
        if len(cycle) < 2:
            return False
        chosen_axis = 0
        best_alignment = f32(abs(view_dir[0]))
        for axis in (1, 2):
            alignment = f32(abs(view_dir[axis]))
            if alignment > best_alignment:
                best_alignment, chosen_axis = alignment, axis
        if best_alignment <= K_DIRECTION_TOLERANCE:
            widest = f32(-1.0)
            for axis in range(3):
                for index in cycle:
                    length = f32(current[index].hi[axis] - current[index].lo[axis])
                    if length > widest:
                        widest, chosen_axis = length, axis
        dir_component = f32(view_dir[chosen_axis])
        if abs(dir_component) <= K_DIRECTION_TOLERANCE:
            return False
        min_length_tolerance = f32(1e-6)
        target_index, target_length = cycle[0], f32(-1.0)
        for index in cycle:
            length = f32(current[index].hi[chosen_axis] - current[index].lo[chosen_axis])
            if length > target_length and length > min_length_tolerance:
                target_length, target_index = length, index
        if target_length <= min_length_tolerance:
            return False
        target = current[target_index].copy()
        min_val, max_val = f32(target.lo[chosen_axis]), f32(target.hi[chosen_axis])
        length = f32(max_val - min_val)
        epsilon = max(f32(f32(1e-5) * length), f32(1e-6))
        candidates = []
        for index in cycle:
            if index == target_index:
                continue
            other_min = f32(current[index].lo[chosen_axis])
            other_max = f32(current[index].hi[chosen_axis])
            if other_min > f32(min_val + epsilon) and other_min < f32(max_val - epsilon):
                candidates.append(other_min)
            if other_max > f32(min_val + epsilon) and other_max < f32(max_val - epsilon):
                candidates.append(other_max)
        split = f32(f32(0.5) * f32(min_val + max_val))
        if candidates:
            split = max(candidates) if dir_component > f32(0.0) else min(candidates)
        if split <= f32(min_val + epsilon):
            split = f32(min_val + epsilon)
        if split >= f32(max_val - epsilon):
            split = f32(max_val - epsilon)
        if not (split > min_val and split < max_val):
            return False
        near_box, far_box = target.copy(), target.copy()
        if dir_component > f32(0.0):
            near_box.hi[chosen_axis] = float(split)
            far_box.lo[chosen_axis] = float(split)
        else:
            near_box.lo[chosen_axis] = float(split)
            far_box.hi[chosen_axis] = float(split)
        for b in (near_box, far_box):
            b.min_depth, b.max_depth = _depth_range(modelview, projection, b.lo, b.hi)
        current[target_index] = near_box
        current.append(far_box)
        return True

    max_iterations = max(len(boxes), 1) * 8 + 32
    splits = 0
    for _ in range(max_iterations):
        adjacency, indegree = rebuild_adjacency()
        export_graph(adjacency)
        ok, order, residual = topo_sort(adjacency, indegree)
        if ok:
            visited = [0] * n_ranks
            rank_order = []
            for index in order:
                owner = current[index].owner
                if owner >= 0 and not visited[owner]:
                    visited[owner] = 1
                    rank_order.append(owner)
            for owner in default_order:
                if not visited[owner]:
                    visited[owner] = 1
                    rank_order.append(owner)
            return rank_order, splits, True
        cycle = find_cycle(adjacency, residual)
        if not cycle:
            return default_order, splits, False
        if not break_cycle(cycle):
            return default_order, splits, False
        splits += 1
    return default_order, splits, False


def _fixed6(value) -> str:
    """operator<< of a float under std::fixed << std::setprecision(6)."""
    v = float(value)
    if math.isinf(v):
        return "inf" if v > 0 else "-inf"
    if math.isnan(v):
        return "nan"
    return f"{v:.6f}"
