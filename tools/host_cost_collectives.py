#!/usr/bin/env python3
"""Host cost of one frame with and without the RCCL calls (one-rank nccl group, tiny scene so
that the GPU is never the bottleneck)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
spec = scenes.make_amr_scene(16, 3, 4, "smooth")   # 176 boxes of 4^3: config-4's box count
scenes.assign_owners(spec, 1, "morton")
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, 0)
p = RenderParameters(64, 64, 0.97, 1, draw_bounds=False)
cam = scenes.default_camera()
for force in (False, True):
    r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range, 0, 1,
                      dist.group.WORLD, force_collectives=force)
    for _ in range(50):
        r.render(p, cam)
    r.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        r.render(p, cam)
    host = time.perf_counter() - t0
    r.synchronize()
    total = time.perf_counter() - t0
    print(f"collectives {force}: host {1e6 * host / n:.1f} us/frame, wall {1e6 * total / n:.1f} us/frame")
dist.destroy_process_group()
