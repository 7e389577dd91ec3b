#!/usr/bin/env python3
"""What a drop-in caller of Render() gets (the reference renders ONE frame per call and returns:
VolumeRenderer/VolumeRenderer.cpp:1103-1339, Examples/RenderFromMultiFab.cpp:17-62,
python/amrVolumeRenderer/module.cpp:252-255), beside bench.py's pipelined steady state:

  single_frame_ms   a camera the driver has never seen, render + synchronize, host wall clock
                    (plan + classify + march + fold, nothing overlapped with another frame);
                    median / min / max of `--singles` frames of ONE renderer, the first (which
                    also allocates every buffer) reported apart
  first_frames      a FRESH renderer, `--burst` frames of one camera queued back to back:
                    mean period of frames 1..N, of the last quarter, and the frame at which the
                    driver's co-run search reported "settled"

  python tools/latency_probe.py --config config4 [--transparency 0.97] [--json out.json]
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

IMAGE = {"config1": (256, 256), "config2": (1024, 1024), "config3": (2048, 2048),
         "config4": (2048, 2048), "config5": (4096, 4096)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config4", choices=sorted(IMAGE))
    ap.add_argument("--transparency", type=float, default=0.97)
    ap.add_argument("--antialiasing", type=int, default=1)
    ap.add_argument("--singles", type=int, default=10)
    ap.add_argument("--burst", type=int, default=100)
    ap.add_argument("--repeat-camera", action="store_true",
                    help="single frames of ONE camera (the plan is cached after the first)")
    ap.add_argument("--full-search", action="store_true", help="avr_renderer_set_corun_balance(0)")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()

    import torch
    from amrvolumerenderer_amd import build as avr_build
    avr_build.build()
    from amrvolumerenderer_amd import runtime, scenes
    from amrvolumerenderer_amd.renderer import (FrameRenderer, RenderParameters,
                                                build_scene_on_device)

    spec = getattr(scenes, args.config)("smooth")
    scenes.assign_owners(spec, 1, "level_pairs")
    width, height = IMAGE[args.config]
    ctx = runtime.Context(0)
    all_boxes, local_boxes = build_scene_on_device(ctx, spec, 0)
    torch.cuda.synchronize()
    rparams = RenderParameters(width=width, height=height, box_transparency=args.transparency,
                               antialiasing=args.antialiasing, draw_bounds=False)

    def fresh():
        renderer = FrameRenderer(ctx, all_boxes, local_boxes, spec.transform, spec.bounds,
                                 spec.scalar_range, 0, 1, None)
        if args.full_search:
            renderer.native.set_corun_balance(0)
        return renderer

    # ---- single synchronised frames ---------------------------------------------------------
    renderer = fresh()
    singles = []
    for i in range(args.singles + 1):
        cam = (scenes.default_camera() if args.repeat_camera
               else scenes.orbit_camera(7 * i + 3, 3600))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        renderer.render(rparams, cam)
        renderer.synchronize()
        singles.append((time.perf_counter() - t0) * 1e3)
    first_single, singles = singles[0], singles[1:]
    corun_single = renderer.native.corun_state() if renderer.native is not None else None
    del renderer

    # ---- the first frames of a fresh renderer, queued back to back ---------------------------
    renderer = fresh()
    cam = scenes.default_camera()
    torch.cuda.synchronize()
    stamps, settled_at = [], None
    t0 = time.perf_counter()
    for i in range(args.burst):
        renderer.render(rparams, cam)
        # (the driver keeps the host a few frames ahead at most: the host clock after frame i is
        # the GPU's within that lead)
        stamps.append(time.perf_counter() - t0)
        if settled_at is None and renderer.native is not None and \
                renderer.native.corun_state()["settled"]:
            settled_at = i + 1
    renderer.synchronize()
    total = time.perf_counter() - t0
    corun_burst = renderer.native.corun_state() if renderer.native is not None else None
    quarter = max(args.burst // 4, 1)
    tail = (total - stamps[-quarter - 1]) / quarter * 1e3 if args.burst > quarter else None

    out = {
        "config": args.config, "transparency": args.transparency, "image": [width, height],
        "single_frame_ms": {"median": round(statistics.median(singles), 4),
                            "min": round(min(singles), 4), "max": round(max(singles), 4),
                            "n": len(singles), "first_with_allocations": round(first_single, 3),
                            "camera": "one camera (plan cached)" if args.repeat_camera
                                      else "a new camera per frame (plan made per frame)",
                            "corun": corun_single},
        "first_frames": {"frames": args.burst, "mean_period_ms": round(total / args.burst * 1e3, 4),
                         "last_quarter_period_ms": round(tail, 4) if tail else None,
                         "frames_to_settle": settled_at, "corun": corun_burst},
    }
    text = json.dumps(out)
    print(text, flush=True)
    if args.json:
        with open(args.json, "w") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main()
