#!/usr/bin/env python3
"""Render an AMReX plotfile with the reference python module's signature, or -- without a path --
write a small two-level synthetic plotfile first and render that.

    python examples/render_plotfile.py [plt00000] [--variable density] [--output frame.png]
    python examples/render_plotfile.py --orbit 36          # frames of a camera orbit (static data:
                                                           # the classified volume is kept)
"""
import argparse
import math
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from amrvolumerenderer_amd import api, plotfile as pf, runtime
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters
from amrvolumerenderer_amd.types import CameraParameters


def synthetic_plotfile(path: str, n0: int = 32) -> None:
    def field(box, level):
        lo, hi = box
        n = n0 * 2 ** level
        z, y, x = np.meshgrid(*[(np.arange(lo[a], hi[a] + 1) + 0.5) / n for a in (2, 1, 0)],
                              indexing="ij")
        r2 = (x - 0.5) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2
        return (np.exp(-r2 / 0.05) + 0.15 * np.sin(9 * x) * np.sin(7 * y))[None]

    half = n0 // 2
    coarse = [((i * half, j * half, k * half),
               (i * half + half - 1, j * half + half - 1, k * half + half - 1))
              for k in range(2) for j in range(2) for i in range(2)]
    fine = [((n0 // 2, n0 // 2, n0 // 2), (3 * n0 // 2 - 1,) * 3)]  # the centre, refined by 2
    levels = [{"domain": ((0, 0, 0), (n0 - 1,) * 3), "boxes": coarse,
               "data": [field(b, 0) for b in coarse]},
              {"domain": ((0, 0, 0), (2 * n0 - 1,) * 3), "boxes": fine,
               "data": [field(b, 1) for b in fine]}]
    pf.write_plotfile(path, ["density"], levels, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), [2])


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("plotfile", nargs="?")
    ap.add_argument("--variable")
    ap.add_argument("--output", default="frame.png")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--transparency", type=float, default=0.6)
    ap.add_argument("--orbit", type=int, default=0, help="render this many views around the data")
    args = ap.parse_args()
    path = args.plotfile
    scratch = None
    if path is None:
        scratch = tempfile.TemporaryDirectory()
        path = os.path.join(scratch.name, "plt00000")
        synthetic_plotfile(path)
    if args.orbit <= 0:
        return api.render(path, width=args.size, height=args.size, variable=args.variable,
                          box_transparency=args.transparency, output=args.output)
    ctx = runtime.Context(0)
    scene = pf.load_plotfile_geometry(ctx, path, args.variable or "")
    renderer = FrameRenderer(ctx, scene.all_boxes, scene.local_boxes, scene.scalar_transform,
                             scene.bounds, scene.scalar_range, cache_classification=True)
    centre = [0.5 * (a + b) for a, b in zip(scene.bounds.min_corner, scene.bounds.max_corner)]
    base, ext = os.path.splitext(args.output)
    for view in range(args.orbit):
        angle = 2.0 * math.pi * view / args.orbit
        eye = (centre[0] + 3.0 * math.sin(angle), centre[1] + 1.0, centre[2] + 3.0 * math.cos(angle))
        camera = CameraParameters(eye, tuple(centre), (0.0, 1.0, 0.0), 45.0, 0.1, 20.0)
        _, rgb8 = renderer.render(RenderParameters(args.size, args.size, args.transparency), camera)
        renderer.synchronize()
        writer = api.save_png if ext.lower() == ".png" else api.save_ppm
        writer(rgb8, f"{base}_{view:03d}{ext}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
