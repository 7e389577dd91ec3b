"""Shared helpers of the parity tests: the same numpy cell arrays feed the oracle (CPU) and the
HIP path (uploaded to HBM), and results are compared bit for bit."""
from __future__ import annotations

import numpy as np

from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.types import (AmrBox, CameraParameters, ScalarTransform, VolumeBounds,
                                         make_params)


def bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(got: np.ndarray, want: np.ndarray, what: str = "") -> None:
    got = np.ascontiguousarray(got, dtype=np.float32).reshape(-1)
    want = np.ascontiguousarray(want, dtype=np.float32).reshape(-1)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
    if bad.size:
        i = int(bad[0])
        raise AssertionError(
            f"{what}: {bad.size} of {got.size} floats differ; first at {i} "
            f"(pixel {i // 5}, comp {i % 5}): got {got[i]!r} want {want[i]!r}")


def oracle_camera(O, cam: CameraParameters):
    return O.make_camera(cam.eye, cam.look_at, cam.up, cam.fov_y_degrees, cam.near_plane,
                         cam.far_plane)


def oracle_transform(O, tr: ScalarTransform):
    return O.make_transform(tr.log_scale_input, tr.normalize_to_unit_range, tr.positive_floor,
                            tr.normalization_min, tr.inverse_normalization_span)


def oracle_params(O, width, height, scalar_range, box_transparency, ref_dist,
                  bounds: VolumeBounds, color_map=None):
    return O.make_params(width, height, scalar_range, box_transparency, ref_dist,
                         bounds.min_corner, bounds.max_corner, color_map)


def device_box(ctx, cells: np.ndarray, min_corner, max_corner, level=0, owner=0) -> AmrBox:
    import torch
    t = torch.from_numpy(np.ascontiguousarray(cells)).to(ctx.device)
    return AmrBox(tuple(min_corner), tuple(max_corner), t, level=level, owner=owner)


def scene_cells(spec):
    return [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(worker, nprocs: int, args_for_port) -> None:
    """torch.multiprocessing.spawn with a rendezvous port probed here.  Between the probe and rank
    0's bind the kernel may hand the port to somebody else (EADDRINUSE: seen once in some thirty
    runs of the suite); that attempt has queued nothing yet and is made again on another port."""
    import torch.multiprocessing as mp
    for attempt in range(4):
        try:
            mp.spawn(worker, args=args_for_port(free_port()), nprocs=nprocs, join=True)
            return
        except Exception as error:   # (mp.ProcessRaisedException carries the rank's traceback as text)
            if "EADDRINUSE" not in str(error) or attempt == 3:
                raise
