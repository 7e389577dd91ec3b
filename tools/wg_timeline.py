#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DAVR_TIMELINE, e.g.
  cd amrvolumerenderer_amd/csrc && hipcc $(CXXFLAGS) -DAVR_TIMELINE --offload-arch=gfx950 -shared -o ../../build/variants/timeline.so *.hip *.cpp
and AVR_HIP_LIBRARY=build/variants/timeline.so): per-workgroup start / end times of the
march of the config-4 frame -> how many workgroups are resident over time, the distribution of
their durations and how much of the kernel's span is tail."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
stats = len(sys.argv) > 3
spec = scenes.config4("smooth")
scenes.assign_owners(spec, ranks, "morton")
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, rank)
r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range, rank, ranks,
                  None, march_workgroups_per_cu=0, cache_classification=True)
p = RenderParameters(2048, 2048, 0.97, 1, draw_bounds=False)
cam = scenes.default_camera()
params, _ = r.make_params(p)
plan = r.plan(params, cam)
for _ in range(3):
    r.paint(plan, None, 0)
r.synchronize()
N = 1 << 18
buf = torch.zeros(4 * N, dtype=torch.int64, device=ctx.device)
samples = torch.zeros(1, dtype=torch.int64, device=ctx.device) if stats else None
r.march_ctx.set_march_counters(buf)
r.paint(plan, samples, 0)
r.synchronize()
r.march_ctx.set_march_counters(None)
t = buf.cpu().numpy().reshape(-1, 4)
t = t[t[:, 1] != 0]
start, end, hw, fetch = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64), t[:, 2], t[:, 3]
t0 = start.min()
start = (start - t0) / 100.0   # us (100 MHz)
end = (end - t0) / 100.0
dur = end - start
span = end.max()
print(f"workgroups {len(t)}  span {span:.1f} us  sum of durations {dur.sum():.0f} us  "
      f"mean {dur.mean():.1f} us  median {np.median(dur):.1f}  p90 {np.percentile(dur, 90):.1f}  max {dur.max():.1f}")
print(f"average resident workgroups over the span: {dur.sum() / span:.1f}  (capacity 2048)")
edges = np.linspace(0, span, 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    resident = int(((start <= mid) & (end > mid)).sum())
    print(f"  t = {mid:7.1f} us: {resident:5d} resident {'#' * (resident // 40)}")
xcc = (hw >> 32) & 0xF
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (xcc << 7)   # cu_id, se_id, xcc
per_cu_end = {}
for c, e in zip(cu, end):
    per_cu_end[c] = max(per_cu_end.get(c, 0.0), e)
ends = np.array(sorted(per_cu_end.values()))
print(f"distinct CUs seen {len(ends)}; CU finish times: min {ends.min():.1f} median {np.median(ends):.1f} max {ends.max():.1f} us")
if stats:
    print(f"samples {int(samples.item())}; per workgroup mean {fetch.mean():.0f} max {fetch.max()}")
    rate = fetch / np.maximum(dur, 1e-3)
    print(f"lane-samples per us per workgroup: median {np.median(rate):.0f} p10 {np.percentile(rate,10):.0f} p90 {np.percentile(rate,90):.0f}")
