#!/usr/bin/env python3
"""What the per-frame HIP timing events of bench.py cost the frame (avr_renderer_set_timing on / off),
for 20 and 300 timed frames on a settled pipeline: ~0.3-0.5 % over 300 frames; 20-frame runs
scatter by +-2 % (pipeline fill)."""
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
print(r.native.corun_state())
def run(n, timing):
    if timing: r.native.set_timing(True)
    r.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): r.render(p, cam)
    r.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    if timing:
        r.native.timings(); r.native.set_timing(False)
    return dt
for rep in range(3):
    print("steps 20: timing off %.4f  on %.4f | steps 300: off %.4f on %.4f" % (run(20, False), run(20, True), run(300, False), run(300, True)))
