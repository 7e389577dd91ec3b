#!/usr/bin/env python3
"""What does a live RCCL communicator do to the one-rank frame?  

Round 2 saw the config-4 frame go from 1.06 to 1.33 ms once ncclCommInitRank had run in the
process and never found out why.  This probe plays the same frame in ONE process per mode:

  none        no RCCL in the process
  comm_first  one-rank RCCL communicator created BEFORE the renderer (and its three streams)
  comm_last   renderer created and used first, communicator afterwards
  comm_used   comm_first + one grouped send/recv round through it before the frames

and for each prints three fixed schedules (no search): the two kernels back to back on one
stream, side by side with the given LDS reserve, and the driver's own search.  Run it under
`rocprofv3 --kernel-trace` with --trace-frames to see which hardware queue each kernel lands on.

  python tools/rccl_resident.py MODE [--reserve BYTES] [--frames N]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("mode", choices=["none", "comm_first", "comm_last", "comm_used"])
ap.add_argument("--reserve", type=int, default=24576)
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--trace-frames", type=int, default=0,
                help="only play this many side-by-side frames (for a kernel trace) and exit")
ap.add_argument("--n-ranks", type=int, default=1, help="play rank --rank of this many (solo exchange)")
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--ownership", default="morton")
args = ap.parse_args()

import torch
from amrvolumerenderer_amd import _capi, runtime, scenes
from amrvolumerenderer_amd.renderer import build_scene_on_device

device = torch.device("cuda", 0)
cam = scenes.default_camera()
spec = scenes.config4("smooth")
scenes.assign_owners(spec, args.n_ranks, args.ownership)
ctx = runtime.Context(0)
all_boxes, local = build_scene_on_device(ctx, spec, args.rank)
merged, mine = [], iter(local)
for b in all_boxes:
    merged.append(next(mine) if b.owner == args.rank else b)


def one_rank_comm():
    return runtime.Comm(0, 0, 1, lambda ident: ident)


def use(comm):
    L = _capi.lib()
    hints = (C.c_float * 1)(1.0)
    owner = (C.c_int32 * 1)(0)
    plan = C.c_void_p()
    _capi.check(L.avr_layered_plan_create(hints, owner, 1, 1, 0, None, 64, 64, C.byref(plan)))
    with torch.cuda.stream(ctx.stream):
        send = torch.zeros(64 * 64 * 5, device=device)
        recv = torch.zeros(64 * 64 * 5, device=device)
    for _ in range(4):
        _capi.check(L.avr_exchange(ctx._handle, plan, comm._handle, C.c_void_p(send.data_ptr()),
                                   C.c_void_p(recv.data_ptr())))
    ctx.synchronize()
    L.avr_frame_plan_destroy(plan)


keep = None
if args.mode in ("comm_first", "comm_used"):
    keep = one_rank_comm()
    if args.mode == "comm_used":
        use(keep)

solo = runtime.Comm.solo(args.rank, args.n_ranks) if args.n_ranks > 1 else None
r = runtime.NativeRenderer(0, merged, spec.transform, spec.bounds, spec.scalar_range, args.rank,
                           args.n_ranks, solo)
r.set_options(-1, False)
kw = dict(use_visibility_graph=True, draw_bounds=False)


def frames(n):
    for i in range(n):
        r.render(2048, 2048, 0.97, 1, cam, **kw)
    r.synchronize()


if args.mode == "comm_last":
    frames(32)
    keep = one_rank_comm()


def timed(label, overlap, share):
    r.set_overlap(overlap)
    r.set_classify_share(share)
    frames(64)
    if overlap < 0:   # let the driver's search finish (bounded); no drain in between: a drained
        begin = time.perf_counter()   # pipeline voids the window the search is timing
        while not r.corun_state()["settled"] and time.perf_counter() - begin < 4.0:
            r.render(2048, 2048, 0.97, 1, cam, **kw)
        r.synchronize()
    torch.cuda.synchronize()
    r.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.frames):
        r.render(2048, 2048, 0.97, 1, cam, **kw)
    r.synchronize()
    dt = (time.perf_counter() - t0) / args.frames
    c, m, b, _ = r.timings()
    r.set_timing(False)
    out = dict(mode=args.mode, schedule=label, frame_ms=round(1e3 * dt, 4), classify_ms=round(c, 4),
               march_ms=round(m, 4), union_ms=round(b, 4), corun=r.corun_state(),
               hw_queues=os.environ.get("GPU_MAX_HW_QUEUES", "default"))
    print(json.dumps(out), flush=True)


if args.trace_frames:
    r.set_overlap(1)
    r.set_classify_share(args.reserve)
    frames(args.trace_frames)
    raise SystemExit(0)

timed("back_to_back", 0, -1)
timed(f"side_by_side_{args.reserve}", 1, args.reserve)
timed("searched", -1, -1)
