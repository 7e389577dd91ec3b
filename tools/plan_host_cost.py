#!/usr/bin/env python3
"""Host cost of one frame's planning for a camera that never repeats (no GPU needed):
avr_frame_plan_create and avr_frame_plan_tighten for config-4's 176 boxes at 2048^2, a new orbit
camera every call.  (the fly-through regime.)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.compositor import FramePlan, make_box_array, make_owner_array
from amrvolumerenderer_amd.types import make_params

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--ownership", default="morton")
ap.add_argument("--contiguous", action="store_true", help="the reference's piece ranges instead "
                "of the frame driver's row bands")
args = ap.parse_args()

for n in args.ranks:
    spec = scenes.config4("smooth")
    scenes.assign_owners(spec, n, args.ownership)
    meta = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    ref = runtime.reference_sample_distance(meta, spec.bounds.min_corner, spec.bounds.max_corner)
    params = make_params(2048, 2048, spec.scalar_range, 0.97, ref, spec.bounds)
    boxes, owners = make_box_array(meta), make_owner_array(meta)
    for rank in sorted({0, n - 1}):
        create = tighten = 0.0
        sends = []
        for f in range(args.frames):
            cam = scenes.orbit_camera(f, args.frames)
            t0 = time.perf_counter()
            plan = FramePlan(meta, params, cam, rank, n, _box_array=boxes, _owner_array=owners,
                             piece_layout=0 if args.contiguous else 1, band_rows=8)
            t1 = time.perf_counter()
            if n > 1:
                plan.tighten()
            t2 = time.perf_counter()
            create += t1 - t0
            tighten += t2 - t1
            sends.append(plan.send_floats)
            plan.close()
        print(f"N={n} rank {rank}: plan_create {1e6 * create / args.frames:7.1f} us  "
              f"tighten {1e6 * tighten / args.frames:7.1f} us per new camera "
              f"(send {4e-6 * sum(sends) / len(sends):.1f} MB)", flush=True)
