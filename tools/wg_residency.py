#!/usr/bin/env python3
"""Per-workgroup timeline of the two paint kernels in the running pipeline (diagnostic build:
git apply tools/patches/r5_timeline.patch, built to tools/_variants/timeline.so; run with
AVR_HIP_LIBRARY=tools/_variants/timeline.so): how many workgroups of each kernel are resident over
time (chip-wide and per CU), how long a workgroup lives, alone and side by side.

  AVR_HIP_LIBRARY=$PWD/tools/_variants/timeline.so python tools/wg_residency.py [--layout 1 --share 24576]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--layout", type=int, default=1)
ap.add_argument("--share", type=int, default=24576)
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--record", type=int, default=6, help="frames recorded at the end")
ap.add_argument("--transparency", type=float, default=0.97)
args = ap.parse_args()

import torch
from amrvolumerenderer_amd import _capi, runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

lib = _capi.lib()
spec = scenes.config4("smooth")
scenes.assign_owners(spec, 1, "level_pairs")
ctx = runtime.Context(0)
all_boxes, local_boxes = build_scene_on_device(ctx, spec, 0)
renderer = FrameRenderer(ctx, all_boxes, local_boxes, spec.transform, spec.bounds, spec.scalar_range, 0, 1, None)
renderer.native.set_overlap(args.layout)
if args.share >= 0:
    renderer.native.set_classify_share(args.share)
p = RenderParameters(width=2048, height=2048, box_transparency=args.transparency, antialiasing=1, draw_bounds=False)
cam = scenes.default_camera()
for _ in range(args.frames):
    renderer.render(p, cam)
capacity = 400000
buf = torch.zeros(capacity * 4, dtype=torch.int64, device=ctx.device)
renderer.synchronize()
torch.cuda.synchronize()
lib.avr_debug_timeline.argtypes = [C.c_void_p, C.c_uint]
lib.avr_debug_timeline_count.argtypes = [C.POINTER(C.c_uint)]
assert lib.avr_debug_timeline(C.c_void_p(buf.data_ptr()), capacity) == 0
for _ in range(args.record + 12):   # back into the steady state, recording
    renderer.render(p, cam)
renderer.synchronize()
torch.cuda.synchronize()
n = C.c_uint()
lib.avr_debug_timeline_count(C.byref(n))
lib.avr_debug_timeline(None, 0)
rows = buf.cpu().numpy().reshape(-1, 4)[:min(n.value, capacity)]
kind, t0, t1, hw = rows[:, 0], rows[:, 1].astype(np.int64), rows[:, 2].astype(np.int64), rows[:, 3]
print(f"layout {args.layout} share {args.share}: {len(rows)} workgroups recorded (of {n.value})")
# HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9); xcc in the upper word
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | ((hw >> 32) << 8)
# take a window in the middle of the recording: a few frame periods
lo, hi = np.percentile(t0, 60), np.percentile(t0, 90)
span_us = (hi - lo) / 100.0
for k, name in ((1, "classify"), (2, "march")):
    m = (kind == k) & (t0 >= lo) & (t0 < hi)
    if not m.any():
        print(f"  {name}: none in the window")
        continue
    dur = (t1[m] - t0[m]) / 100.0   # us
    weight = 16.0 if k == 1 else 1.0   # (every 16th classify workgroup is recorded)
    resident = weight * dur.sum() / span_us   # average workgroups resident chip-wide
    cus = len(np.unique(cu[m]))
    print(f"  {name:8s}: {m.sum():6d} workgroups in {span_us:7.1f} us; lifetime mean {dur.mean():7.2f} us, "
          f"median {np.median(dur):7.2f}, p90 {np.percentile(dur, 90):7.2f}; resident on average "
          f"{resident:7.1f} chip-wide = {resident / max(cus, 1):5.2f} per CU ({cus} CUs seen)")
