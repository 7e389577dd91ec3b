#!/usr/bin/env python3
"""cProfile of the host side of one simulated rank's frames (no collectives)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

ranks, rank = int(sys.argv[1]), int(sys.argv[2])
spec = scenes.make_amr_scene(64, 3, 16, "smooth") if len(sys.argv) > 3 else scenes.config4("smooth")
scenes.assign_owners(spec, ranks, "morton")
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, rank)
r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range, rank, ranks, None)
p = RenderParameters(2048, 2048, 0.97, 1, draw_bounds=False)
cam = scenes.default_camera()
params, _ = r.make_params(p)
plan = r.plan(params, cam)
recv = torch.zeros(max(plan.recv_floats, 5), device=ctx.device).view(-1, 5)
recv[:, 4] = float("inf")
recv = recv.reshape(-1)
r.compositor.exchange = lambda plan, send: recv      # no collective: planned-size receive buffer
r.compositor.gather = lambda plan, piece, dst=0: None
for _ in range(20):
    r.render(p, cam)
r.synchronize()
prof = cProfile.Profile()
prof.enable()
for _ in range(300):
    r.render(p, cam)
prof.disable()
r.synchronize()
st = pstats.Stats(prof)
st.sort_stats("cumulative").print_stats(28)
