#!/usr/bin/env python3
"""Which host call of a frame blocks when the GPU is the bottleneck (config4, N=1)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amrvolumerenderer_amd import runtime, scenes, compositor
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

spec = scenes.config4("smooth")
scenes.assign_owners(spec, 1, "morton")
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, 0)
r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range)
p = RenderParameters(2048, 2048, 0.97, 1, draw_bounds=False)
cam = scenes.default_camera()
log = []
def wrap(obj, name):
    fn = getattr(obj, name)
    def timed(*a, **k):
        t = time.perf_counter(); out = fn(*a, **k); dt = time.perf_counter() - t
        log.append((name, dt)); return out
    setattr(obj, name, timed)
wrap(r, "plan"); wrap(r.scene, "classify_plan"); wrap(r.scene, "march_plan")
wrap(r.compositor, "compose"); wrap(r.compositor, "gather")
for i in range(5): r.render(p, cam)
r.synchronize(); torch.cuda.synchronize()
log.clear()
t0 = time.perf_counter()
for i in range(40):
    n = len(log)
    t = time.perf_counter(); r.render(p, cam); dt = time.perf_counter() - t
    if dt > 0.6e-3:
        print(f"frame {i} at {1e3*(t-t0):7.2f} ms took {1e3*dt:6.2f} ms:", ", ".join(f"{k} {1e3*v:.2f}" for k, v in log[n:]))
r.synchronize(); torch.cuda.synchronize()
print(f"40 frames: {1e3*(time.perf_counter()-t0):.2f} ms")
