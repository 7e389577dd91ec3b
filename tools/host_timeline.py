#!/usr/bin/env python3
"""Average host time of the calls inside FrameRenderer.render for one simulated rank of a
config-4 frame (collectives stubbed)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

ranks, rank, policy = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
spec = scenes.config4("smooth")
scenes.assign_owners(spec, ranks, policy)
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, rank)
r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range, rank, ranks, None)
p = RenderParameters(2048, 2048, 0.97, 1, draw_bounds=False)
cam = scenes.default_camera()
stubs = {}
def planned_receive(plan, send):
    n = max(plan.recv_floats, 5)
    if n not in stubs:
        buf = torch.zeros(n, device=ctx.device).view(-1, 5); buf[:, 4] = float("inf"); stubs[n] = buf.reshape(-1)
    return stubs[n]
r.compositor.exchange = planned_receive
r.compositor.gather = lambda plan, piece, dst=0: None
T = collections.defaultdict(float)
def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    def timed(*a, **k):
        t = time.perf_counter(); out = fn(*a, **k); T[label or name] += time.perf_counter() - t; return out
    setattr(obj, name, timed)
wrap(r.scene, "classify_plan"); wrap(r.scene, "march_plan"); wrap(r.comm_ctx, "fold_plan")
wrap(r, "paint")
wrap(r.comm_ctx, "empty", "comm.empty")
from amrvolumerenderer_amd import _capi
lib = _capi.lib()
class LibProxy:
    def __getattr__(self, name):
        fn = getattr(lib, name)
        def timed(*a):
            t = time.perf_counter(); out = fn(*a); T["C:" + name] += time.perf_counter() - t; return out
        return timed
_capi.lib = lambda: LibProxy()
for _ in range(30): r.render(p, cam)
r.synchronize(); T.clear()
n = 400
t0 = time.perf_counter()
for _ in range(n): r.render(p, cam)
loop = time.perf_counter() - t0
ts = time.perf_counter()
r.synchronize()
print(f"drain after the loop: {1e6 * (time.perf_counter() - ts):.0f} us")
print(f"loop {1e6*loop/n:.1f} us/frame")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f"  {k:14s} {1e6*v/n:7.1f} us")
