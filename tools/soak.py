#!/usr/bin/env python3
"""Soak: thousands of pipelined frames with a moving camera; host RSS and device memory must
stay flat, the last frame must equal the first frame of the same view."""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
spec = scenes.config3("smooth")
scenes.assign_owners(spec, 1, "morton")
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, 0)
r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range)
p = RenderParameters(1024, 1024, 0.9, 1)
views = [scenes.orbit_camera(v, 64) for v in range(64)]
_, first = r.render(p, views[0]); r.synchronize(); first = first.cpu().numpy().copy()
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
for i in range(200): r.render(p, views[i % 64])
r.synchronize()
rss0, dev0 = rss(), torch.cuda.memory_reserved()
t0 = time.perf_counter()
for i in range(frames): r.render(p, views[i % 64])
r.synchronize()
dt = time.perf_counter() - t0
_, last = r.render(p, views[0]); r.synchronize()
print(f"{frames} frames in {dt:.2f} s ({1e3*dt/frames:.3f} ms/frame); host RSS {rss0:.0f} -> {rss():.0f} MiB; "
      f"torch reserved {dev0/2**20:.0f} -> {torch.cuda.memory_reserved()/2**20:.0f} MiB; "
      f"same image: {np.array_equal(first, last.cpu().numpy())}")
