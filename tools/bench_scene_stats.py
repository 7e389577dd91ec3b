#!/usr/bin/env python3
"""Throughput of the scene-statistics kernels on the config-4 scene (2.95 GB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import build_scene_on_device

field = sys.argv[1] if len(sys.argv) > 1 else "smooth"
spec = scenes.config4(field)
ctx = runtime.Context(0)
_, local = build_scene_on_device(ctx, spec, 0)
scene = ctx.create_scene(local, spec.transform)
nbytes = spec.total_cells * 8
torch.cuda.synchronize()
for _ in range(2):
    stats = scene.scalar_stats()
t0 = time.perf_counter()
for _ in range(5):
    stats = scene.scalar_stats()
t = (time.perf_counter() - t0) / 5
print(f"scalar_stats: {stats}  {t*1e3:.3f} ms  {nbytes/t/1e12:.2f} TB/s (host-timed incl. sync)")
tr, _, rng = runtime.scene_transform_from_stats(stats[:3], stats[3], False, True)
for bins in (256, 4096):
    counts = torch.zeros(bins, dtype=torch.int64, device=ctx.device)
    for _ in range(2):
        scene.histogram(tr, rng[0], rng[1], bins, counts)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ctx.stream)
    for _ in range(5):
        scene.histogram(tr, rng[0], rng[1], bins, counts)
    e1.record(ctx.stream)
    ctx.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"histogram[{field}, {bins} bins]: {ms:.3f} ms  {nbytes/ms/1e9:.2f} TB/s")

# host cost of one histogram call (enqueue only)
counts = torch.zeros(256, dtype=torch.int64, device=ctx.device)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    scene.histogram(tr, rng[0], rng[1], 256, counts)
t1 = time.perf_counter()
ctx.synchronize()
print(f"histogram host enqueue: {(t1 - t0) / 20 * 1e6:.0f} us per call")
