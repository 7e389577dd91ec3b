#!/usr/bin/env python3
"""How many boxes a march wave meets (needs a library built with tools/patches/r5_visit_counts.patch:
counters[5] candidates after the 64-at-a-time cull, [6] cull trips, [7] wave-level box visits that
march, [8] lanes that march in them).  AVR_HIP_LIBRARY=tools/_variants/visits.so python tools/visit_counts.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amrvolumerenderer_amd import build as avr_build
avr_build.build()
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

IMAGE = {"config2": (1024, 1024), "config3": (2048, 2048), "config4": (2048, 2048), "config5": (4096, 4096)}
for name in sys.argv[1:] or ["config4"]:
    config, _, t = name.partition(":")
    transparency = float(t) if t else 0.97
    spec = getattr(scenes, config)("smooth")
    scenes.assign_owners(spec, 1, "level_pairs")
    w, h = IMAGE[config]
    ctx = runtime.Context(0)
    all_boxes, local = build_scene_on_device(ctx, spec, 0)
    r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range, 0, 1, None, native=False)
    counters = torch.zeros(16, dtype=torch.int64, device="cuda")
    samples = torch.zeros(1, dtype=torch.int64, device="cuda")
    r.march_ctx.set_march_counters(counters)
    p = RenderParameters(width=w, height=h, box_transparency=transparency, antialiasing=1, draw_bounds=False)
    r.render(p, scenes.default_camera(), samples=samples)
    r.synchronize(); torch.cuda.synchronize()
    c = counters.tolist(); n = int(samples.item())
    waves = (w // 8) * (h // 8)
    print(f"{name}: samples {n}  waves {waves}  candidates/wave {c[5]/waves:.2f}  cull trips/wave {c[6]/waves:.2f}  "
          f"marching visits/wave {c[7]/waves:.2f}  lanes per marching visit {c[8]/max(c[7],1):.1f}  "
          f"samples per marching lane-visit {n/max(c[8],1):.1f}")
    del r
