#!/usr/bin/env python3
"""How much of the sparse exchange volume is empty pixels?  For every rank of an N-rank config-4
frame: the floats of its send buffer (run rectangles cut into pieces, avr_frame_plan) against
the pixels in it that carry anything (alpha != 0) -- the bound of any exact compression of the
exchange (empty pixels are the identity of the depth-sort blend) -- and against the tightened
layout of avr_frame_plan_tighten (per-row extents of the runs' boxes)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--size", type=int, default=2048)
args = ap.parse_args()
cam = scenes.default_camera()
for n in args.ranks:
    spec = scenes.config4("smooth")
    scenes.assign_owners(spec, n, "morton")
    total_px = total_full = total_tight = 0
    worst = 0.0
    for rank in range(n):
        ctx = runtime.Context(0)
        all_boxes, local = build_scene_on_device(ctx, spec, rank)
        r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range,
                          rank, n, None, native=False, stage_through_host=True)
        p = RenderParameters(width=args.size, height=args.size, box_transparency=0.97,
                             antialiasing=1, draw_bounds=False)
        params, _ = r.make_params(p)
        order = r.visibility.order(cam, 1.0, True, None) if r.visibility is not None else None
        plan = r.plan(params, cam, order)
        send = r.paint(plan)
        r.march_ctx.synchronize()
        torch.cuda.synchronize()
        px = plan.send_floats // 5
        layers = send[:px * 5].view(px, 5)
        full = int((layers[:, 3] != 0).sum().item())
        total_px += px
        total_full += full
        worst = max(worst, px * 20 / 1e6)
        plan.tighten()
        tight_px = plan.send_floats // 5
        total_tight += tight_px
        print(f"  N={n} rank {rank}: {px * 20 / 1e6:7.2f} MB in rectangles, {100.0 * full / max(px, 1):5.1f} % "
              f"of the pixels non-empty ({full * 20 / 1e6:7.2f} MB); tightened layout "
              f"{tight_px * 20 / 1e6:7.2f} MB")
        del r, send, layers, local, all_boxes
        torch.cuda.empty_cache()
    print(f"N={n}: {total_px * 20 / 1e6:.1f} MB in all, {100.0 * total_full / total_px:.1f} % non-empty; "
          f"tightened (avr_frame_plan_tighten) {total_tight * 20 / 1e6:.1f} MB = "
          f"{100.0 * total_tight / total_px:.1f} %")
