#!/usr/bin/env python3
"""What ONE GPU can say about the N-GPU frame: every rank's share of the config-4 frame is
played alone on this GPU through the C++ frame driver (avr_renderer, the three-stream pipeline
bench.py runs) with the solo communicator -- the rank classifies, marches, folds, converts and
"gathers" exactly what it would in the N-rank frame, only the peers' blocks do not move -- and the
exchange / gather are added from a per-link model:

  t_exchange = RCCL launch latency + max over peers (bytes to / from that peer) / link rate
  t_gather   = RCCL launch latency + (bytes into rank 0 from the busiest peer) / link rate

xGMI is point to point (one link per GPU pair, ~153 GB/s per the MI355X guide; 64 GB/s is used
as a pessimistic achieved rate), the all-to-all uses all 7 links of a GPU at once, so the
busiest link sets the time.  The collectives run on the compositing stream, one frame behind
the march, so in the pipelined frame they are hidden unless they exceed the paint stage; both
readings are printed:

  serial    share + exchange + gather          (nothing overlaps: an upper bound)
  pipelined max(share, exchange + fold + gather)   (what the three streams are built for)

An estimate tool -- not a bench line.  The 8-GPU measurement is the driver's.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.renderer import build_scene_on_device
from amrvolumerenderer_amd.types import CameraParameters

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="config4")
ap.add_argument("--ranks", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--ownership", default="morton")
ap.add_argument("--size", type=int, default=2048)
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--transparency", type=float, default=0.97)
ap.add_argument("--march-occupancy", type=int, default=-1)
ap.add_argument("--link-gbs", type=float, nargs="+", default=[153.0, 64.0])
ap.add_argument("--rccl-latency-us", type=float, default=None,
                help="launch-to-completion latency of a small grouped send/recv round; default: "
                     "measured here on a one-rank RCCL communicator")
ap.add_argument("--only-rank", type=int, default=-1)
ap.add_argument("--overlap", type=int, default=-1, help="avr_renderer_set_overlap")
args = ap.parse_args()

cam = scenes.default_camera()
device = torch.device("cuda", 0)


def measure_rccl_latency():
    """Launch + completion of one small grouped ncclSend / ncclRecv round (one-rank communicator:
    the only RCCL configuration a one-GPU box can run)."""
    import ctypes as C
    import numpy as np
    from amrvolumerenderer_amd import _capi
    L = _capi.lib()
    ctx = runtime.Context(0)
    comm = runtime.Comm(0, 0, 1, lambda ident: ident)
    hints = (C.c_float * 1)(1.0)
    owner = (C.c_int32 * 1)(0)
    plan = C.c_void_p()
    _capi.check(L.avr_layered_plan_create(hints, owner, 1, 1, 0, None, 64, 64, C.byref(plan)))
    with torch.cuda.stream(ctx.stream):
        send = torch.zeros(64 * 64 * 5, device=device)
        recv = torch.zeros(64 * 64 * 5, device=device)
    ctx.synchronize()
    times = []
    for it in range(60):
        t0 = time.perf_counter()
        _capi.check(L.avr_exchange(ctx._handle, plan, comm._handle, C.c_void_p(send.data_ptr()),
                                   C.c_void_p(recv.data_ptr())))
        ctx.synchronize()
        if it >= 10:
            times.append(time.perf_counter() - t0)
    L.avr_frame_plan_destroy(plan)
    comm.close()
    return 1e6 * float(np.median(times))


print(f"{args.config}, {args.size}^2, ownership {args.ownership}")

summary = []
for n_ranks in args.ranks:
    spec = getattr(scenes, args.config)("smooth")
    scenes.assign_owners(spec, n_ranks, args.ownership)
    worst = {"share": 0.0, "host": 0.0}
    per_rank = []
    for rank in (range(n_ranks) if args.only_rank < 0 else [args.only_rank]):
        ctx = runtime.Context(0)
        all_boxes, local = build_scene_on_device(ctx, spec, rank)
        merged, mine = [], iter(local)
        for b in all_boxes:
            merged.append(next(mine) if b.owner == rank else b)
        comm = runtime.Comm.solo(rank, n_ranks) if n_ranks > 1 else None
        r = runtime.NativeRenderer(0, merged, spec.transform, spec.bounds, spec.scalar_range, rank,
                                   n_ranks, comm)
        r.set_options(args.march_occupancy, False)
        r.set_overlap(args.overlap)
        counter = torch.zeros(1, dtype=torch.int64, device=device)
        kw = dict(use_visibility_graph=True, draw_bounds=False)
        r.render(args.size, args.size, args.transparency, 1, cam, samples=counter, **kw)
        r.synchronize()
        samples = int(counter.item())
        info = r.plan_info()
        warm_until = time.perf_counter() + 0.5   # clocks and allocator pools settle
        warm = 0
        while time.perf_counter() < warm_until or warm < 30:
            r.render(args.size, args.size, args.transparency, 1, cam, **kw)
            warm += 1
            if warm % 16 == 0:
                r.synchronize()
        r.synchronize()
        torch.cuda.synchronize()
        r.set_timing(True)
        r.host_profile(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.frames):
            r.render(args.size, args.size, args.transparency, 1, cam, **kw)
        host = (time.perf_counter() - t0) / args.frames
        r.synchronize()
        torch.cuda.synchronize()
        share = (time.perf_counter() - t0) / args.frames
        sections, _ = r.host_profile()
        classify_ms, march_ms, busy_ms, _ = r.timings()
        r.set_timing(False)
        # exchange volume per peer from the plan (floats -> bytes)
        send_mb = info.send_floats * 4 / 1e6
        recv_mb = info.recv_floats * 4 / 1e6
        per_rank.append(dict(rank=rank, boxes=len(local), share_ms=1e3 * share, host_ms=1e3 * host,
                             classify_ms=classify_ms, march_ms=march_ms, busy_ms=busy_ms,
                             send_mb=send_mb, recv_mb=recv_mb, runs=info.n_local_runs,
                             samples=samples, piece_px=info.piece_end - info.piece_begin))
        print(f"  N={n_ranks} rank {rank}: boxes {len(local):3d} runs {info.n_local_runs:3d} "
              f"share {1e3 * share:6.3f} ms (host {1e3 * host:5.3f}; classify {classify_ms:5.3f} "
              f"march {march_ms:5.3f} union {busy_ms:5.3f})  send {send_mb:6.2f} MB recv "
              f"{recv_mb:6.2f} MB  samples {samples / 1e6:6.1f} M")
        print("      host us/frame inside the C ABI: " +
              "  ".join(f"{k} {v:.0f}" for k, v in sections.items()))
        del r, comm, local, all_boxes, merged
        torch.cuda.empty_cache()
    slowest = max(per_rank, key=lambda d: d["share_ms"])
    # busiest link: a rank sends / receives at most its whole buffer split over N-1 peers; the
    # pessimistic reading puts a rank's whole receive volume on ONE link
    line = dict(n=n_ranks, share=slowest["share_ms"], host=max(d["host_ms"] for d in per_rank))
    for gbs in args.link_gbs:
        if n_ranks == 1:
            ex = ga = 0.0
        else:
            worst_even = max(max(d["send_mb"], d["recv_mb"]) for d in per_rank) / (n_ranks - 1)
            worst_one = max(max(d["send_mb"], d["recv_mb"]) for d in per_rank)
            ex = worst_even / gbs      # MB / (GB/s) = ms; the RCCL latency is added below
            ex_pess = worst_one / gbs
            ga = (args.size * args.size * 3 / n_ranks / 1e6) / gbs
            line[f"ex_pess@{gbs:g}"] = ex_pess
        line[f"ex@{gbs:g}"] = ex
        line[f"ga@{gbs:g}"] = ga
    summary.append(line)

# RCCL is initialised only now: merely creating a communicator in this process was measured to
# change how kernels of different streams share the GPU afterwards (one rank, classify beside
# march: 1.06 ms per frame before, 1.33 ms after) -- which is also why a rank of an N-rank frame
# runs its two kernels back to back by default (avr_renderer_set_overlap).
latency_us = args.rccl_latency_us if args.rccl_latency_us is not None else measure_rccl_latency()
print(f"RCCL round latency {latency_us:.1f} us "
      f"({'given' if args.rccl_latency_us is not None else 'measured, one-rank communicator'})")
for line in summary:
    n_ranks = line["n"]
    for gbs in args.link_gbs:
        if n_ranks == 1:
            continue
        line[f"ex@{gbs:g}"] += latency_us * 1e-3
        line[f"ex_pess@{gbs:g}"] += latency_us * 1e-3
        line[f"ga@{gbs:g}"] += latency_us * 1e-3

print("\nestimate of the N-GPU frame from one GPU (ms; speed-up vs N = 1):")
base = summary[0]["share"] if summary and summary[0]["n"] == 1 else None
for line in summary:
    n = line["n"]
    text = f"  N={n}: slowest share {line['share']:.3f} (host {line['host']:.3f})"
    for gbs in args.link_gbs:
        ex, ga = line[f"ex@{gbs:g}"], line[f"ga@{gbs:g}"]
        serial = line["share"] + ex + ga
        piped = max(line["share"], ex + ga + 0.03)
        text += f" | @{gbs:g} GB/s: exch {ex:.3f} gather {ga:.3f} -> serial {serial:.3f}"
        text += f" pipelined {piped:.3f}"
        if base:
            text += f" ({base / serial:.2f}x / {base / piped:.2f}x)"
        if n > 1:
            text += f" [all on one link: exch {line[f'ex_pess@{gbs:g}']:.3f}]"
    print(text)
