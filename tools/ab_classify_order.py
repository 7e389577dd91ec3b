#!/usr/bin/env python3
"""Experiment: does the ORDER in which the classify pass streams the boxes
matter beside the march?  The classify pass of frame i+1 runs beside the march of frame i; the
march retires its tiles centre-first (cost order: the long rays through the fine levels), the
classify pass takes the boxes in scene order (level-major: coarse first).  The scene's box list is
permuted before it is handed to the driver -- results are the same frame (same boxes), only the
classify pass's sweep order and the tie-break of equal depth hints change.

  python tools/ab_classify_order.py [--frames 300]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--orders", nargs="+",
                default=["level_major", "finest_first", "near_first", "far_first", "level_major"])
args = ap.parse_args()

import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import build_scene_on_device

cam = scenes.default_camera()
kw = dict(use_visibility_graph=True, draw_bounds=False)
ctx = runtime.Context(0)
reference = None
for order in args.orders:
    spec = scenes.config4("smooth")
    scenes.assign_owners(spec, 1, "morton")

    def depth(b):
        c = [0.5 * (b.min_corner[a] + b.max_corner[a]) for a in range(3)]
        return sum((c[a] - cam.eye[a]) ** 2 for a in range(3))

    if order == "finest_first":
        spec.boxes.sort(key=lambda b: -b.level)
    elif order == "near_first":
        spec.boxes.sort(key=depth)
    elif order == "far_first":
        spec.boxes.sort(key=lambda b: -depth(b))
    all_boxes, local = build_scene_on_device(ctx, spec, 0)
    r = runtime.NativeRenderer(0, local, spec.transform, spec.bounds, spec.scalar_range, 0, 1, None)
    r.set_options(-1, False)
    counter = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    _, rgb = r.render(2048, 2048, 0.97, 1, cam, samples=counter, **kw)
    r.synchronize()
    if reference is None:
        reference = rgb.clone()
    same = bool(torch.equal(rgb, reference))
    begin = time.perf_counter()
    n = 0
    while n < 64 or (time.perf_counter() - begin < 3.0 and not r.corun_state()["settled"]):
        r.render(2048, 2048, 0.97, 1, cam, **kw)
        n += 1
    r.synchronize()
    r.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.frames):
        r.render(2048, 2048, 0.97, 1, cam, **kw)
    r.synchronize()
    dt = (time.perf_counter() - t0) / args.frames
    c, m, b, _ = r.timings()
    print(json.dumps(dict(order=order, frame_ms=round(1e3 * dt, 4), classify_ms=round(c, 4),
                          march_ms=round(m, 4), union_ms=round(b, 4), samples=int(counter.item()),
                          same_bytes_as_first=same, corun=r.corun_state())), flush=True)
    r.close()
    del r, all_boxes, local
    torch.cuda.empty_cache()
