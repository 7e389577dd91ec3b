import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import build_scene_on_device
mode = sys.argv[1]
spec = scenes.config4("smooth"); scenes.assign_owners(spec, 1, "morton")
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, 0)
dummies = []
if mode == "dummy_hi2_lo1":
    dummies = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
elif mode == "dummy_lo1":
    dummies = [torch.cuda.Stream(priority=0)]
elif mode == "dummy_hi2":
    dummies = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)]
for d in dummies:
    with torch.cuda.stream(d):
        torch.zeros(1, device="cuda")
torch.cuda.synchronize()
r = runtime.NativeRenderer(0, local, spec.transform, spec.bounds, spec.scalar_range, 0, 1, None)
r.set_options(0, False)
cam = scenes.default_camera()
kw = dict(use_visibility_graph=True, draw_bounds=False)
t_end = time.perf_counter() + 0.5
i = 0
while time.perf_counter() < t_end:
    r.render(2048, 2048, 0.97, 1, cam, **kw); i += 1
    if i % 16 == 0: r.synchronize()
r.synchronize()
r.set_timing(True)
t0 = time.perf_counter()
for _ in range(300):
    r.render(2048, 2048, 0.97, 1, cam, **kw)
r.synchronize()
dt = (time.perf_counter() - t0) / 300
c, m, b, n = r.timings()
print(f"{mode}: frame {1e3*dt:.3f} ms classify {c:.3f} march {m:.3f} union {b:.3f}")
