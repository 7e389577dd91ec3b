#!/usr/bin/env python3
"""Where the host time of one frame goes (same tiny scene as host_cost.py)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

spec = scenes.make_amr_scene(16, 3, 4, "smooth")
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, 0)
r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range)
p = RenderParameters(int(sys.argv[1]) if len(sys.argv) > 1 else 64, 64, 0.97, 1)
cam = scenes.default_camera()
T = collections.defaultdict(float)
def timed(name, fn, *a, **k):
    t = time.perf_counter(); out = fn(*a, **k); T[name] += time.perf_counter() - t; return out
n = 1000
for it in range(n + 20):
    if it == 20: T.clear()
    params, root = timed("make_params", r.make_params, p)
    plan = timed("FramePlan", r.plan, params, cam)
    c, comm = r.ctx, r.comm_ctx
    t = time.perf_counter()
    c.join()
    with torch.cuda.stream(c.stream):
        T["streams"] += time.perf_counter() - t
        send = timed("render_plan", r.paint, plan, None, it & 1)
        t = time.perf_counter()
        ev = torch.cuda.Event(); ev.record(c.stream)
    with torch.cuda.stream(comm.stream):
        comm.stream.wait_event(ev)
        T["streams"] += time.perf_counter() - t
        piece, rgb = timed("fold_plan", r.compositor.compose, plan, send, True, True)
        flat = piece
        out = timed("flip", torch.flip, rgb.view(p.height, p.width, 3), [0])
        t = time.perf_counter()
    T["streams"] += time.perf_counter() - t
    timed("plan_del", plan.close)
r.synchronize()
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f"{k:14s} {1e6*v/n:7.1f} us")
print("sum", round(1e6*sum(T.values())/n, 1))
