#!/usr/bin/env python3
"""The frame pipeline in its steady state for a few seconds, for tools/corun_counters (a
rocprofiler-sdk tool that samples device-wide counters meanwhile): prints the steady phase's begin,
end (CLOCK_MONOTONIC) and frame count as JSON.
  ROCP_TOOL_LIBRARIES=$PWD/tools/corun_counters/libcorun_counters.so AVR_COUNTER_LOG=out.log \\
  AVR_COUNTER_GROUPS="SQ_WAVES SQ_BUSY_CYCLES;TA_TA_BUSY_sum" python3 tools/corun_counters/steady.py --mode corun
modes: corun (the product's pipeline), march (the classification cached: the march alone),
       back_to_back (the two kernels alternating on one stream: each alone, time-shared)"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="corun", choices=["corun", "march", "back_to_back"])
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--transparency", type=float, default=0.97)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    from amrvolumerenderer_amd import runtime, scenes
    from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device
    spec = scenes.config4("smooth")
    scenes.assign_owners(spec, 1, "level_pairs")
    ctx = runtime.Context(0)
    all_boxes, local = build_scene_on_device(ctx, spec, 0)
    r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range, 0, 1, None,
                      cache_classification=(args.mode == "march"))
    if args.mode == "back_to_back":
        r.native.set_overlap(0)
    p = RenderParameters(width=2048, height=2048, box_transparency=args.transparency, antialiasing=1,
                         draw_bounds=False)
    cam = scenes.default_camera()
    t0 = time.monotonic()
    frames = 0
    while True:       # settle: the driver's co-run search
        for _ in range(16):
            r.render(p, cam)
            frames += 1
        if frames >= 32 and (r.native.corun_state()["settled"] or time.monotonic() - t0 > 4.0):
            break
    r.synchronize()
    begin = time.monotonic()
    steady = 0
    while time.monotonic() - begin < args.seconds:
        for _ in range(64):
            r.render(p, cam)
        steady += 64
    r.synchronize()
    end = time.monotonic()
    out = {"mode": args.mode, "begin": begin, "end": end, "frames": steady,
           "ms_per_frame": (end - begin) / steady * 1e3, "corun": r.native.corun_state()}
    text = json.dumps(out)
    print(text, flush=True)
    if args.out:
        open(args.out, "w").write(text + "\n")

if __name__ == "__main__":
    main()
