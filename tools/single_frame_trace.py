#!/usr/bin/env python3
"""A few single, synchronised frames (render + synchronize, a new camera each) for a kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/single_frame_trace.py --chunks K
and, with --read TRACE.csv, the timeline of the last frames in that trace (us from the frame's
first kernel).  What a drop-in caller of Render() gets, kernel by kernel."""
import argparse
import csv
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="config4")
ap.add_argument("--chunks", type=int, default=-1)
ap.add_argument("--frames", type=int, default=6)
ap.add_argument("--transparency", type=float, default=0.97)
ap.add_argument("--read", default=None)
ap.add_argument("--layout", type=int, default=-1)
ap.add_argument("--share", type=int, default=-1, help="fixed LDS reserve of the classify pass (bytes)")
ap.add_argument("--show", type=int, default=2)
args = ap.parse_args()

if args.read:
    rows = []
    for r in csv.DictReader(open(args.read)):
        m = re.search(r"avr::(?:\(anonymous namespace\)::)?(\w+)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                     m.group(1) if m else r["Kernel_Name"][:40], int(r["Queue_Id"])))
    rows.sort()
    folds = [i for i, r in enumerate(rows) if r[2].startswith("fold_plan")]
    # a frame = everything after the previous fold up to and including this fold
    for which in folds[-args.show:]:
        prev = max([f for f in folds if f < which], default=-1)
        frame = rows[prev + 1:which + 1]
        t0 = frame[0][0]
        print(f"-- frame ending at fold #{which}: {(frame[-1][1] - t0) / 1e3:.1f} us of kernels")
        for s, e, name, q in frame:
            print(f"   q{q} {name:28s} {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f})")
    sys.exit(0)

import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

IMAGE = {"config2": (1024, 1024), "config3": (2048, 2048), "config4": (2048, 2048)}
spec = getattr(scenes, args.config)("smooth")
scenes.assign_owners(spec, 1, "level_pairs")
width, height = IMAGE[args.config]
ctx = runtime.Context(0)
all_boxes, local_boxes = build_scene_on_device(ctx, spec, 0)
renderer = FrameRenderer(ctx, all_boxes, local_boxes, spec.transform, spec.bounds,
                         spec.scalar_range, 0, 1, None)
renderer.native.set_frame_chunks(args.chunks)
if args.layout >= 0:
    renderer.native.set_overlap(args.layout)
if args.share >= 0:
    renderer.native.set_overlap(1)
    renderer.native.set_classify_share(args.share)
p = RenderParameters(width=width, height=height, box_transparency=args.transparency,
                     antialiasing=1, draw_bounds=False)
times = []
for i in range(args.frames):
    cam = scenes.orbit_camera(7 * i + 3, 3600)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    renderer.render(p, cam)
    renderer.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
print("chunks", args.chunks, "layout", args.layout, "share", args.share, "single frames ms:", " ".join(f"{t:.3f}" for t in times))
