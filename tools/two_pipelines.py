#!/usr/bin/env python3
"""Experiment: how much frame rate do tails and cross-stream launch gaps cost a rank?  The same
share of the config-4 frame is rendered by ONE frame driver, and by TWO drivers (each with its own
three streams and classified volumes, sharing the cells) that take the frames alternately -- twice
the kernels in flight, so the tail of one march overlaps the head of the next.

  python tools/two_pipelines.py N_RANKS RANK [--ownership level_pairs]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("n_ranks", type=int)
ap.add_argument("rank", type=int)
ap.add_argument("--ownership", default="level_pairs")
ap.add_argument("--frames", type=int, default=400)
ap.add_argument("--drivers", type=int, nargs="+", default=[1, 2, 3])
args = ap.parse_args()

import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import build_scene_on_device

cam = scenes.default_camera()
spec = scenes.config4("smooth")
scenes.assign_owners(spec, args.n_ranks, args.ownership)
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, args.rank)
merged, mine = [], iter(local)
for b in all_boxes:
    merged.append(next(mine) if b.owner == args.rank else b)
kw = dict(use_visibility_graph=True, draw_bounds=False)


def make():
    comm = runtime.Comm.solo(args.rank, args.n_ranks) if args.n_ranks > 1 else None
    r = runtime.NativeRenderer(0, merged, spec.transform, spec.bounds, spec.scalar_range, args.rank,
                               args.n_ranks, comm)
    r.set_options(-1, False)
    return r, comm


for n_drivers in args.drivers:
    drivers = [make() for _ in range(n_drivers)]
    rs = [d[0] for d in drivers]

    def frames(n):
        for i in range(n):
            rs[i % n_drivers].render(2048, 2048, 0.97, 1, cam, **kw)
            if i % 64 == 63:
                for r in rs:
                    r.synchronize()
        for r in rs:
            r.synchronize()

    begin = time.perf_counter()
    frames(64)
    while time.perf_counter() - begin < 4.0 and not all(r.corun_state()["settled"] for r in rs):
        frames(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.frames):
        rs[i % n_drivers].render(2048, 2048, 0.97, 1, cam, **kw)
    for r in rs:
        r.synchronize()
    dt = (time.perf_counter() - t0) / args.frames
    print(json.dumps(dict(n_ranks=args.n_ranks, rank=args.rank, drivers=n_drivers,
                          frame_ms=round(1e3 * dt, 4), corun=[r.corun_state() for r in rs])),
          flush=True)
    for r, c in drivers:
        r.close()
