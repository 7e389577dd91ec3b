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

Every share is measured in a process of its own, as a rank of the N-GPU job would run it: HIP
maps streams onto a few hardware queues in creation order, and in a process that had already
created and destroyed other renderers the classify and the march stream of a later one were seen
to land on one queue -- "side by side" then means one after the other, and the driver's co-run
search (avr_renderer_corun_state) rightly but unrepresentatively chose "back to back".

An estimate tool -- not a bench line.  The 8-GPU measurement is the driver's.
"""
import argparse
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

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
ap.add_argument("--no-rccl", action="store_true",
                help="do NOT keep a live one-rank RCCL communicator in every share process (round "
                     "2's way; the default since round 3 is the N-GPU job's situation: RCCL "
                     "initialised and used before the renderer exists)")
ap.add_argument("--contiguous-pieces", action="store_true",
                help="the reference's contiguous pieces instead of the driver's row bands")
ap.add_argument("--fly-through", action="store_true",
                help="every frame a camera the driver has never seen (0.1 degree of orbit per "
                     "frame): visibility order, frame plan, tightened exchange layout and per-box "
                     "prologue are made per frame")
ap.add_argument("--through-rccl", type=int, default=0, metavar="PERCENT",
                help="play every rank's exchange and gather through a one-rank RCCL communicator "
                     "to the rank itself instead of moving only its own block: this share of "
                     "every peer's block (100 = all of it, one block after the other over ONE "
                     "connection where the node has N - 1 links)")
ap.add_argument("--no-plan-ahead", action="store_true",
                help="--fly-through: make every plan on the frames' own thread")
ap.add_argument("--overlap", type=int, default=-1, help="avr_renderer_set_overlap")
ap.add_argument("--classify-share", type=int, default=-1, help="avr_renderer_set_classify_share")
ap.add_argument("--worker", type=int, nargs=2, metavar=("N_RANKS", "RANK"), default=None,
                help="internal: measure this one share and print it as a JSON line")
ap.add_argument("--latency-worker", action="store_true", help="internal: RCCL round latency")
args = ap.parse_args()


def measure_share(n_ranks, rank):
    import torch
    from amrvolumerenderer_amd import runtime, scenes
    from amrvolumerenderer_amd.renderer import build_scene_on_device

    cam = scenes.default_camera()
    device = torch.device("cuda", 0)
    spec = getattr(scenes, args.config)("smooth")
    scenes.assign_owners(spec, n_ranks, args.ownership)
    ctx = runtime.Context(0)
    all_boxes, local = build_scene_on_device(ctx, spec, rank)
    merged, mine = [], iter(local)
    for b in all_boxes:
        merged.append(next(mine) if b.owner == rank else b)
    resident = None
    if not args.no_rccl:
        # what a rank of the real job has in its process: an initialised, used RCCL communicator
        # (a one-rank one is all a single GPU can hold)
        import ctypes as C
        from amrvolumerenderer_amd import _capi
        resident = runtime.Comm(0, 0, 1, lambda ident: ident)
        L = _capi.lib()
        hints, owner1, plan1 = (C.c_float * 1)(1.0), (C.c_int32 * 1)(0), C.c_void_p()
        _capi.check(L.avr_layered_plan_create(hints, owner1, 1, 1, 0, None, 64, 64, C.byref(plan1)))
        with torch.cuda.stream(ctx.stream):
            a = torch.zeros(64 * 64 * 5, device=device)
            b = torch.zeros(64 * 64 * 5, device=device)
        for _ in range(4):
            _capi.check(L.avr_exchange(ctx._handle, plan1, resident._handle, C.c_void_p(a.data_ptr()),
                                       C.c_void_p(b.data_ptr())))
        ctx.synchronize()
        L.avr_frame_plan_destroy(plan1)
    comm = None
    if n_ranks > 1:
        # --through-rccl: the rank's exchange and gather go through a one-rank RCCL communicator to
        # the rank itself (RCCL's launch, kernel and bytes beside the paint kernels; not the links)
        # (--through-rccl -1: "one link" -- only the largest peer block, whole: RCCL works off the
        # operations for ONE peer one after the other, which the node does not have)
        comm = (runtime.Comm.solo_rccl(0, rank, n_ranks, max(args.through_rccl, 0))
                if args.through_rccl else runtime.Comm.solo(rank, n_ranks))
    r = runtime.NativeRenderer(0, merged, spec.transform, spec.bounds, spec.scalar_range, rank,
                               n_ranks, comm)
    r.set_options(args.march_occupancy, False)
    r.set_overlap(args.overlap)
    r.set_classify_share(args.classify_share)
    if args.contiguous_pieces:
        r.set_piece_layout(0, 1)
    counter = torch.zeros(1, dtype=torch.int64, device=device)
    kw = dict(use_visibility_graph=True, draw_bounds=False)
    view = [0]

    class NextCamera:   # `cam` below: the default camera, or one that never repeats
        def __call__(self):
            if not args.fly_through:
                return scenes.default_camera()
            view[0] += 1
            return scenes.orbit_camera(view[0], 3600)
    next_cam = NextCamera()
    r.render(args.size, args.size, args.transparency, 1, cam, samples=counter, **kw)
    r.synchronize()
    samples = int(counter.item())
    # clocks and allocator pools settle, and the driver finishes measuring how the rank's two
    # kernels share the GPU (bounded at 3 s)
    # a scripted fly-through knows its next camera: its plan is made one frame ahead on a helper
    # thread (avr_renderer_prepare) unless --no-plan-ahead -- also while the driver's search runs
    # (with the host as the limit every candidate would read the same)
    ahead = None
    if args.fly_through and n_ranks > 1 and not args.no_plan_ahead:
        ahead = runtime.PlanAhead(r)
    upcoming = [next_cam()]

    def frame():
        this_cam, upcoming[0] = upcoming[0], next_cam()
        if ahead is not None:
            ahead.submit(args.size, args.size, args.transparency, 1, upcoming[0], **kw)
        r.render(args.size, args.size, args.transparency, 1, this_cam, **kw)

    begin = time.perf_counter()
    warm = 0
    while True:
        elapsed = time.perf_counter() - begin
        if warm >= 30 and elapsed >= 0.5 and (elapsed >= 3.0 or r.corun_state()["settled"]):
            break
        frame()
        warm += 1
    r.synchronize()
    torch.cuda.synchronize()
    r.set_timing(True)
    r.host_profile(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.frames):
        frame()
    host = (time.perf_counter() - t0) / args.frames
    if ahead is not None:
        ahead.close()
    r.synchronize()
    torch.cuda.synchronize()
    share = (time.perf_counter() - t0) / args.frames
    sections, _ = r.host_profile()
    info = r.plan_info()   # (after the first frame the driver tightens the exchange layout)
    classify_ms, march_ms, busy_ms, _ = r.timings()
    r.set_timing(False)
    return dict(rank=rank, boxes=len(local), share_ms=1e3 * share, host_ms=1e3 * host,
                classify_ms=classify_ms, march_ms=march_ms, busy_ms=busy_ms,
                send_mb=info.send_floats * 4 / 1e6, recv_mb=info.recv_floats * 4 / 1e6,
                runs=info.n_local_runs, samples=samples,
                piece_px=info.piece_end - info.piece_begin, corun=r.corun_state(),
                host_us={k: round(v) for k, v in sections.items()})


def measure_rccl_latency():
    """Launch + completion of one small grouped ncclSend / ncclRecv round (one-rank communicator:
    the only RCCL configuration a one-GPU box can run)."""
    import ctypes as C
    import numpy as np
    import torch
    from amrvolumerenderer_amd import _capi, runtime
    device = torch.device("cuda", 0)
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


if args.worker is not None:
    print(json.dumps(measure_share(*args.worker)), flush=True)
    raise SystemExit(0)
if args.latency_worker:
    print(json.dumps({"latency_us": measure_rccl_latency()}), flush=True)
    raise SystemExit(0)


def child(extra):
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    for line in out.stdout.splitlines():
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"worker failed: {' '.join(cmd)}\n{out.stdout[-2000:]}\n{out.stderr[-2000:]}")


print(f"{args.config}, {args.size}^2, ownership {args.ownership}, "
      f"{'contiguous pieces' if args.contiguous_pieces else 'row-band pieces'}, "
      f"{'no RCCL in the share processes' if args.no_rccl else 'live one-rank RCCL communicator in every share process'}"
      f"{f', EXCHANGE ({args.through_rccl} % of every block) AND GATHER THROUGH RCCL (to the rank itself)' if args.through_rccl > 0 else ', EXCHANGE (the busiest link: the largest peer block, whole) AND GATHER (one peer) THROUGH RCCL (to the rank itself)' if args.through_rccl < 0 else ''}"
      f"{', FLY-THROUGH (a new camera every frame)' if args.fly_through else ''}",
      flush=True)
summary = []
for n_ranks in args.ranks:
    per_rank = []
    for rank in (range(n_ranks) if args.only_rank < 0 else [args.only_rank]):
        d = child(["--worker", str(n_ranks), str(rank)])
        per_rank.append(d)
        corun = d["corun"]
        print(f"  N={n_ranks} rank {rank}: boxes {d['boxes']:3d} runs {d['runs']:3d} "
              f"share {d['share_ms']:6.3f} ms (host {d['host_ms']:5.3f}; classify "
              f"{d['classify_ms']:5.3f} march {d['march_ms']:5.3f} union {d['busy_ms']:5.3f})  send "
              f"{d['send_mb']:6.2f} MB recv {d['recv_mb']:6.2f} MB  samples "
              f"{d['samples'] / 1e6:6.1f} M")
        print(f"      classify {corun['classify']}, LDS reserve {corun['lds_reserve_bytes']} "
              f"({'settled' if corun['settled'] else 'still searching'} after "
              f"{corun['timed_windows']} timed windows); host us/frame inside the C ABI: " +
              "  ".join(f"{k} {v}" for k, v in d["host_us"].items()), flush=True)
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
            ga = (args.size * args.size * 3 / n_ranks / 1e6) / gbs
            line[f"ex_pess@{gbs:g}"] = worst_one / gbs
        line[f"ex@{gbs:g}"] = ex
        line[f"ga@{gbs:g}"] = ga
    summary.append(line)

# (RCCL in a process of its own too: merely creating a communicator was measured to change how
# kernels of different streams share the GPU afterwards -- one rank, classify beside march,
# 1.06 ms per frame before and 1.33 ms after -- one more reason why the driver MEASURES whether
# a rank's two kernels run back to back or side by side: on the 8-GPU node each rank decides
# with RCCL present.)
latency_us = (args.rccl_latency_us if args.rccl_latency_us is not None
              else child(["--latency-worker"])["latency_us"])
print(f"RCCL round latency {latency_us:.1f} us "
      f"({'given' if args.rccl_latency_us is not None else 'measured, one-rank communicator'})")
for line in summary:
    if line["n"] == 1:
        continue
    for gbs in args.link_gbs:
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
