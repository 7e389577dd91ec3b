#!/usr/bin/env python3
"""How much of the march's time depends on where its classified bricklets are when it first touches
them: the march alone (config-4, volume classified once), timed with events on its stream,
  (a) back to back -- what is left of the 369 MB volume in the 256 MB memory-side cache stays;
  (b) after a kernel that writes S MB of other memory first (S = 64 ... 2048): the cache holds
      something else, every first touch goes to HBM -- but to an idle HBM;
beside the co-run figure of the bench (march 0.945 ms beside the classify pass).  Diagnostics."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from amrvolumerenderer_amd import runtime, scenes  # noqa: E402
from amrvolumerenderer_amd.compositor import FramePlan  # noqa: E402
from amrvolumerenderer_amd.renderer import build_scene_on_device  # noqa: E402
from amrvolumerenderer_amd.types import make_params  # noqa: E402

spec = scenes.config4("smooth")
scenes.assign_owners(spec, 1, "morton")
base = runtime.Context(0)
meta, local = build_scene_on_device(base, spec, 0)
march = runtime.Context(0, priority=-1)
scene = march.create_scene(local, spec.transform)
ref = runtime.reference_sample_distance(meta, spec.bounds.min_corner, spec.bounds.max_corner)
params = make_params(2048, 2048, spec.scalar_range, 0.97, ref, spec.bounds)
plan = FramePlan(meta, params, scenes.default_camera(), 0, 1)
with torch.cuda.stream(march.stream):
    send = torch.empty(max(plan.send_floats, 1), device=base.device)
    junk = torch.empty(2048 << 20, dtype=torch.uint8, device=base.device)
scene.classify_plan(march, plan, 0)
for _ in range(20):
    scene.march_plan(march, plan, 0, send)
march.synchronize()
K = 40
for flush_mb in (0, 64, 128, 256, 512, 1024, 2048):
    spans = []
    for _ in range(K):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(march.stream):
            if flush_mb:
                junk[: flush_mb << 20].fill_(1)
            e0.record()
        scene.march_plan(march, plan, 0, send)
        with torch.cuda.stream(march.stream):
            e1.record()
        spans.append((e0, e1))
    march.synchronize()
    torch.cuda.synchronize()
    times = sorted(a.elapsed_time(b) for a, b in spans)
    print(f"march alone after writing {flush_mb:5d} MB elsewhere: median {times[K // 2]:.4f} ms  "
          f"(min {times[0]:.4f}, max {times[-1]:.4f})", flush=True)
