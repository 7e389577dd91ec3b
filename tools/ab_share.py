#!/usr/bin/env python3
"""A/B helper: the march alone (classified volume built once) for single ranks' shares of the
N-rank config-4 frame, and the classify pass alone -- kernel times by HIP events.  Uses only the
round-1 part of the C ABI, so any build of the library can be compared (AVR_HIP_LIBRARY)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.compositor import FramePlan
from amrvolumerenderer_amd.renderer import build_scene_on_device
from amrvolumerenderer_amd.types import make_params

n_ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ranks = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 3, 7]
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 0
spec = scenes.config4("smooth")
scenes.assign_owners(spec, n_ranks, "morton")
cam = scenes.default_camera()
tag = os.path.basename(os.environ.get("AVR_HIP_LIBRARY", "tree"))
for rank in ranks:
    ctx = runtime.Context(0)
    ctx.set_march_occupancy(cap)
    all_boxes, local = build_scene_on_device(ctx, spec, rank)
    scene = ctx.create_scene(local, spec.transform)
    ref = runtime.reference_sample_distance(all_boxes, spec.bounds.min_corner, spec.bounds.max_corner)
    params = make_params(2048, 2048, spec.scalar_range, 0.97, ref, spec.bounds)
    plan = FramePlan(all_boxes, params, cam, rank, n_ranks)
    out = ctx.empty(max(plan.send_floats, 1))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tc = tm = 0.0
    n = 40
    for it in range(n + 5):
        if it == 5:
            tc = tm = 0.0
        ev[0].record(ctx.stream)
        scene.classify_plan(ctx, plan, 0)
        ev[1].record(ctx.stream)
        scene.march_plan(ctx, plan, 0, out)
        ev[2].record(ctx.stream)
        ctx.synchronize()
        tc += ev[0].elapsed_time(ev[1])
        tm += ev[1].elapsed_time(ev[2])
    print(f"{tag} N={n_ranks} rank {rank} cap {cap}: classify alone {1e3 * tc / n:7.1f} us  "
          f"march alone {1e3 * tm / n:7.1f} us  (runs {plan.n_local_runs}, send {plan.send_floats * 4 / 1e6:.1f} MB)")
    del scene, local, all_boxes, out
    torch.cuda.empty_cache()
