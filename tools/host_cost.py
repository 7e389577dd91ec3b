#!/usr/bin/env python3
"""Host cost of one frame: enqueue time of FrameRenderer.render for a scene whose GPU work is
negligible (176 boxes of 4^3 cells, 64x64 image), measured without cProfile."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

spec = scenes.make_amr_scene(16, 3, 4, "smooth")
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, 0)
r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range)
p = RenderParameters(64, 64, 0.97, 1)
cam = scenes.default_camera()
for _ in range(20):
    r.render(p, cam)
r.synchronize(); torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    r.render(p, cam)
t1 = time.perf_counter()
r.synchronize(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"boxes={len(all_boxes)} enqueue {1e6*(t1-t0)/n:.1f} us/frame, total {1e6*(t2-t0)/n:.1f} us/frame")
