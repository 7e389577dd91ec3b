#!/usr/bin/env python3
"""Upper bound of what a PERSISTENT classify kernel could buy (task 4), without
building one: in an experiment build (-DAVR_EXPERIMENT_CLASSIFY_REPEAT) ONE classify launch sweeps
the frame's tiles K times over -- a classify pass with no kernel boundary, no event, no doorbell
and no wait of any kind between its "frames" -- beside K marches queued back to back on the march
stream (each reading a volume classified earlier: timing only).  Whatever a persistent design does
about its frame queue, it cannot be faster than this.  Period = the later of the two streams' ends
/ K, for a sweep of the classify workgroups' LDS reserve; beside it the marches alone and the long
classify pass alone.  usage (GPU box): AVR_HIP_LIBRARY=build/variants/repeat.so python
tools/upper_bound_persistent.py [--frames K]"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=100)
ap.add_argument("--config", default="config4")
ap.add_argument("--reserves", type=int, nargs="+",
                default=[0, 8192, 16384, 20480, 22528, 24576, 26624, 28672, 32768, 40960])
args = ap.parse_args()

import torch  # noqa: E402
from amrvolumerenderer_amd import _capi, runtime, scenes  # noqa: E402
from amrvolumerenderer_amd.compositor import FramePlan  # noqa: E402
from amrvolumerenderer_amd.renderer import build_scene_on_device  # noqa: E402
from amrvolumerenderer_amd.types import make_params  # noqa: E402

L = _capi.lib()
spec = getattr(scenes, args.config)("smooth")
scenes.assign_owners(spec, 1, "morton")
size = {"config2": 1024}.get(args.config, 2048)
base = runtime.Context(0)
meta, local = build_scene_on_device(base, spec, 0)
march = runtime.Context(0, priority=-1)
classify = runtime.Context(0, priority=0)
scene = march.create_scene(local, spec.transform)
ref = runtime.reference_sample_distance(meta, spec.bounds.min_corner, spec.bounds.max_corner)
params = make_params(size, size, spec.scalar_range, 0.97, ref, spec.bounds)
plan = FramePlan(meta, params, scenes.default_camera(), 0, 1)
with torch.cuda.stream(march.stream):
    send = torch.empty(max(plan.send_floats, 1), device=base.device)
K = args.frames


def reserve(ctx, nbytes):
    _capi.check(L.avr_context_set_classify_lds_reserve(ctx._handle, int(nbytes)))


def sync():
    march.synchronize()
    classify.synchronize()
    torch.cuda.synchronize()


os.environ["AVR_CLASSIFY_REPEAT"] = "1"
scene.classify_plan(march, plan, 0)
scene.classify_plan(march, plan, 1)
for _ in range(20):      # clocks up
    scene.march_plan(march, plan, 0, send)
sync()

t0 = time.perf_counter()
for _ in range(K):
    scene.march_plan(march, plan, 0, send)
sync()
march_alone = (time.perf_counter() - t0) / K
os.environ["AVR_CLASSIFY_REPEAT"] = str(K)
reserve(classify, 0)
t0 = time.perf_counter()
scene.classify_plan(classify, plan, 1)
sync()
classify_alone = (time.perf_counter() - t0) / K
print(f"{args.config}: {K} marches alone {1e3 * march_alone:.4f} ms each; one classify launch of {K} "
      f"sweeps alone {1e3 * classify_alone:.4f} ms per sweep", flush=True)
for nbytes in args.reserves:
    reserve(classify, nbytes)
    rows = []
    for rep in range(2):
        t0 = time.perf_counter()
        scene.classify_plan(classify, plan, 1)
        for _ in range(K):
            scene.march_plan(march, plan, 0, send)
        march.synchronize()
        t_march = time.perf_counter() - t0
        classify.synchronize()
        t_all = time.perf_counter() - t0
        rows.append((t_march / K, t_all / K))
    text = "  ".join(f"marches done {1e3 * a:.4f}, both done {1e3 * b:.4f}" for a, b in rows)
    print(f"  LDS reserve {nbytes:6d}: per frame (ms) {text}", flush=True)
