#!/usr/bin/env python3
"""config-4 with a domain edge that is not a power of two (cell spacings 1.25 / 512 ...: every box
takes the reciprocal index path, IndexMode kReciprocal), the camera scaled with it: the pipelined
frame of the C++ driver.  python tools/reciprocal_probe.py [extent=1.25] [frames=400]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amrvolumerenderer_amd import build as avr_build
avr_build.build()
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device
from amrvolumerenderer_amd.types import CameraParameters

extent = float(sys.argv[1]) if len(sys.argv) > 1 else 1.25
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 400
spec = scenes.make_amr_scene(512, 3, 128, "smooth", "config4_scaled", extent=extent)
scenes.assign_owners(spec, 1, "level_pairs")
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, 0)
r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range, 0, 1, None)
base = scenes.default_camera()
cam = CameraParameters(eye=tuple(extent * v for v in base.eye), look_at=tuple(extent * v for v in base.look_at),
                       up=base.up, fov_y_degrees=base.fov_y_degrees, near_plane=base.near_plane * extent,
                       far_plane=base.far_plane * extent)
p = RenderParameters(width=2048, height=2048, box_transparency=0.97, antialiasing=1, draw_bounds=False)
samples = torch.zeros(1, dtype=torch.int64, device="cuda")
r.render(p, cam, samples=samples)
r.synchronize()
t0 = time.monotonic()
n = 0
while True:
    for _ in range(16):
        r.render(p, cam)
        n += 1
    if n >= 32 and (r.native.corun_state()["settled"] or time.monotonic() - t0 > 4.0):
        break
r.synchronize()
t0 = time.monotonic()
for _ in range(frames):
    r.render(p, cam)
r.synchronize()
ms = (time.monotonic() - t0) / frames * 1e3
print(f"extent {extent}: {int(samples.item())} samples per frame, {ms:.4f} ms per frame, {r.native.corun_state()}")
