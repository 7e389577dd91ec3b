#!/usr/bin/env python3
"""What a speculative frame's classify pass saves, kernel by kernel on ONE stream (no pipelining):
plain classify + march against flagged classify + checking march + the two gated repair launches.
  python tools/speculation_probe.py [config4] [transparency=0.0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amrvolumerenderer_amd import build as avr_build
avr_build.build()
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

IMAGE = {"config2": (1024, 1024), "config3": (2048, 2048), "config4": (2048, 2048)}
config = sys.argv[1] if len(sys.argv) > 1 else "config4"
transparency = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
spec = getattr(scenes, config)("smooth")
scenes.assign_owners(spec, 1, "level_pairs")
w, h = IMAGE[config]
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, 0)
r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range, 0, 1, None, native=False)
p = RenderParameters(width=w, height=h, box_transparency=transparency, antialiasing=1, draw_bounds=False)
params, _ = r.make_params(p)
plan = r.plan(params, scenes.default_camera(), None)
scene = r.scene
n = len(scene.boxes)
out = torch.empty(max(plan.send_floats, 1), device=ctx.device)
visited = torch.zeros(n, dtype=torch.uint8, device=ctx.device)
scene.classify_plan(ctx, plan, 0)
scene.march_plan_speculative(ctx, plan, 0, out, visited=visited)
ctx.synchronize()
print(f"{config} transparency {transparency}: {int(visited.sum())} of {n} boxes sampled")


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    ctx.synchronize()
    with torch.cuda.stream(ctx.stream):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(ctx.stream)
        for _ in range(reps):
            fn()
        b.record(ctx.stream)
    ctx.synchronize()
    return a.elapsed_time(b) / reps


missed = torch.zeros(n, dtype=torch.uint8, device=ctx.device)
count = torch.zeros(1, dtype=torch.int32, device=ctx.device)
scratch = torch.zeros(n, dtype=torch.uint8, device=ctx.device)
print("plain classify          %.4f ms" % timed(lambda: scene.classify_plan(ctx, plan, 0)))
print("flagged classify        %.4f ms" % timed(lambda: scene.classify_plan_flagged(ctx, plan, 0, visited)))
positions = torch.nonzero(visited.cpu()).flatten().tolist()
print("classify, those positions %.4f ms" % timed(lambda: scene.classify_plan_positions(ctx, plan, 0, positions)))
print("plain march             %.4f ms" % timed(lambda: scene.march_plan(ctx, plan, 0, out)))
print("checking march          %.4f ms" % timed(lambda: scene.march_plan_speculative(
    ctx, plan, 0, out, classified=visited, visited=scratch, missed=missed, miss_count=count)))
print("gated classify (idle)   %.4f ms" % timed(lambda: scene.classify_plan_flagged(ctx, plan, 0, missed, gate=count)))
print("gated march (idle)      %.4f ms" % timed(lambda: scene.march_plan_speculative(ctx, plan, 0, out, gate=count)))
print("misses", int(count.item()))
