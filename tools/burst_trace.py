#!/usr/bin/env python3
"""Kernel timeline of ONE short burst of frames on a settled pipeline (run under
rocprofv3 --kernel-trace; tools/burst_trace.sh): where the first frames after a drain lose time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
spec = scenes.config4("smooth"); scenes.assign_owners(spec, 1, "morton")
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, 0)
r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range)
r.native.set_classify_share(24576)
r.native.set_overlap(1)
p = RenderParameters(2048, 2048, 0.97, 1, draw_bounds=False)
cam = scenes.default_camera()
for i in range(300): r.render(p, cam)
r.synchronize(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n): r.render(p, cam)
r.synchronize(); torch.cuda.synchronize()
print("burst of", n, "frames: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
