#!/usr/bin/env python3
"""Plays each rank of an N-rank frame on ONE GPU without the collectives: plan -> paint ->
fold (over a receive buffer of empty pixels of the planned size).  Reports, per rank, the GPU
time of the paint and fold stages, the bytes the rank would send / receive, and the host time of
one pipelined frame.  An estimate tool for ownership policies -- not a bench line."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters, build_scene_on_device

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="config4")
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--ownership", default="morton")
ap.add_argument("--size", type=int, default=2048)
ap.add_argument("--frames", type=int, default=20)
ap.add_argument("--only-rank", type=int, default=-1)
ap.add_argument("--transparency", type=float, default=0.97)
ap.add_argument("--march-occupancy", type=int, default=None)
ap.add_argument("--no-join", action="store_true", help="experiment: skip the caller-stream join")
ap.add_argument("--priorities", default="-1,-1,0", help="march,comm,classify stream priorities")
ap.add_argument("--pipeline", type=int, default=0,
                help="also time this many unsynchronised frames (the renderer's three-stream "
                     "pipeline without the collectives)")
args = ap.parse_args()

spec = getattr(scenes, args.config)("smooth")
scenes.assign_owners(spec, args.ranks, args.ownership)
cam = scenes.default_camera()
p = RenderParameters(args.size, args.size, args.transparency, 1, draw_bounds=False)
print(f"{args.config} {args.ranks} ranks, ownership {args.ownership}, {args.size}^2")
worst = 0.0
for rank in (range(args.ranks) if args.only_rank < 0 else [args.only_rank]):
    ctx = runtime.Context(0)
    all_boxes, local = build_scene_on_device(ctx, spec, rank)
    r = FrameRenderer(ctx, all_boxes, local, spec.transform, spec.bounds, spec.scalar_range, rank,
                      args.ranks, None, march_workgroups_per_cu=args.march_occupancy,
                      stream_priorities=tuple(int(v) for v in args.priorities.split(",")))
    mctx = r.march_ctx
    params, _ = r.make_params(p)
    plan = r.plan(params, cam)
    recv = torch.zeros(max(plan.recv_floats, 5), device=ctx.device).view(-1, 5)
    recv[:, 4] = float("inf")
    recv = recv.reshape(-1)
    counter = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    r.paint(plan, counter, 0)
    r.synchronize()
    samples = int(counter.item())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_paint = t_fold = 0.0
    host = 0.0
    for it in range(args.frames + 3):
        if it == 3:
            t_paint = t_fold = host = 0.0
        t0 = time.perf_counter()
        plan = r.plan(params, cam)
        with torch.cuda.stream(mctx.stream):
            ev[0].record(r.classify_ctx.stream)
            send = r.paint(plan, None, it & 1)
            ev[1].record(mctx.stream)
            piece, rgb = mctx.fold_plan(plan, recv, True, sync_streams=False)
            ev[2].record(mctx.stream)
        host += time.perf_counter() - t0
        r.synchronize()
        t_paint += ev[0].elapsed_time(ev[1])
        t_fold += ev[1].elapsed_time(ev[2])
        last = plan
    n = args.frames
    worst = max(worst, t_paint / n)
    print(f"rank {rank}: boxes {len(local):3d} paint {t_paint / n:6.3f} ms  fold {t_fold / n:6.3f} ms  "
          f"send {last.send_floats * 4 / 1e6:6.2f} MB recv {last.recv_floats * 4 / 1e6:6.2f} MB  "
          f"runs {len(last.runs())}  host {1e3 * host / n:6.3f} ms  samples {samples / 1e6:7.1f} M")
    if args.pipeline:
        # the renderer's own frame loop with the collectives stubbed out: the exchange hands back
        # a receive buffer of the planned size, the gather nothing
        F = args.pipeline
        stubs = {}

        def planned_receive(plan, send):
            n = max(plan.recv_floats, 5)
            if n not in stubs:
                buf = torch.zeros(n, device=ctx.device).view(-1, 5)
                buf[:, 4] = float("inf")
                stubs[n] = buf.reshape(-1)
            return stubs[n]

        r.compositor.exchange = planned_receive
        if args.no_join:
            r.classify_ctx.join = lambda: None
        r.compositor.gather = lambda plan, piece, dst=0: None
        for _ in range(20):
            r.render(p, cam)
        r.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(F):
            r.render(p, cam)
        host = time.perf_counter() - t0
        r.synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / F
        print(f"    pipelined, 3 streams      : {1e3 * dt:6.3f} ms / frame (host {1e3 * host / F:.3f} ms)")
    del r, local, all_boxes, recv
    torch.cuda.empty_cache()
print(f"slowest paint {worst:.3f} ms")
