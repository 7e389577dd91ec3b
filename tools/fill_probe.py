#!/usr/bin/env python3
"""Total time of bursts of K frames started on a drained, settled pipeline (K = 1 ... 40), and
the same after 50 ms of idle: the start-up cost behind the gap between 20-step and 200-step runs."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device
spec = scenes.config4("smooth"); scenes.assign_owners(spec, 1, "morton")
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, 0)
r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range)
p = RenderParameters(2048, 2048, 0.97, 1, draw_bounds=False)
cam = scenes.default_camera()
for i in range(700): r.render(p, cam)
r.synchronize(); torch.cuda.synchronize()
def run(n, gap=0.0):
    r.synchronize(); torch.cuda.synchronize()
    if gap: time.sleep(gap)
    t0 = time.perf_counter()
    for i in range(n): r.render(p, cam)
    r.synchronize(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for n in (1, 2, 3, 5, 10, 20, 40):
    print(n, "frames:", " ".join("%.3f" % run(n) for _ in range(6)), " after 50 ms idle:", " ".join("%.3f" % run(n, 0.05) for _ in range(3)))
