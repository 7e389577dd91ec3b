import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device
spec = scenes.config4("smooth")
scenes.assign_owners(spec, 1, "morton")
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, 0)
r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range)
p = RenderParameters(2048, 2048, 0.97, 1, draw_bounds=False)
cam = scenes.default_camera()
torch.cuda.synchronize()
for trial in range(3):
    r.kernel_events = []
    ref = torch.cuda.Event(enable_timing=True); ref.record(r.march_ctx.stream)
    t0 = time.perf_counter()
    hostt = []
    for i in range(12):
        r.render(p, cam)
        hostt.append(time.perf_counter() - t0)
    r.synchronize(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ev = r.kernel_events; r.kernel_events = None
    print(f"trial {trial}: wall {wall*1e3:.2f} ms for 12 frames")
    for i, (c0, c1, m0, m1) in enumerate(ev):
        print(f"  f{i}: host {hostt[i]*1e3:6.2f}  classify {ref.elapsed_time(c0):6.2f}..{ref.elapsed_time(c1):6.2f}  march {ref.elapsed_time(m0):6.2f}..{ref.elapsed_time(m1):6.2f}")
    time.sleep(0.5 if trial == 1 else 0)
