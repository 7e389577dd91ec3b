#!/usr/bin/env python3
"""Benchmark of the hot path on BASELINE.json's metric: Mray-samples/s (+ frames/s) for a
2048^2 render of the 512^3-base 3-level AMR scene (config-4 of SURVEY.md 8d), translucent
transfer function (box_transparency 0.97: full traversal, the throughput regime).

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches it under torch.distributed.run (one rank per GPU over RCCL); started
bare (no WORLD_SIZE in the environment) it launches those N ranks itself, as child processes and
before anything in this process touches the GPU, relays rank 0's JSON line and exits with the
children's status.

A "step" is one frame: fused paint + run fold -> DirectSend exchange -> fold -> gather ->
8-bit conversion, with all cell data already resident in HBM.  Strong scaling: the frame is
fixed, boxes are partitioned over ranks (Morton chunks).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="config4",
                    choices=["config1", "config2", "config3", "config4", "config5", "tiny", "hostbound"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--field", default="smooth", choices=["smooth", "noise", "radial"])
    ap.add_argument("--transparency", type=float, default=0.97)
    ap.add_argument("--ownership", default="level_pairs", choices=["morton", "morton_cost", "morton_pairs", "level_pairs", "round_robin",
                             "block"])
    ap.add_argument("--antialiasing", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=60.0,
                    help="cap of the cpu_baseline leg's CPU time (config-4 needs ~35 s for its "
                         "whole, unextrapolated frame; what does not fit is extrapolated and says so)")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the untimed single-frame / first-100-frames figures (`latency`)")
    ap.add_argument("--orbit", type=int, default=0, help="average over this many orbit views")
    ap.add_argument("--fly-through", action="store_true",
                    help="every frame of the timed region (and of the warm-up) has a camera the "
                         "driver has never seen: plan cache useless, host planning (visibility "
                         "order, frame plan, exchange layout, per-box prologue) paid per frame")
    ap.add_argument("--cache-classification", action="store_true",
                    help="NOT the headline configuration: keep the classified volumes across "
                         "frames (static data, moving camera) instead of re-reading the f64 "
                         "cells every frame")
    ap.add_argument("--priorities", default="-1,-1,0",
                    help="HIP stream priorities of the march, compositing and classify streams")
    ap.add_argument("--march-occupancy", type=int, default=None,
                    help="resident march workgroups per CU (0 = uncapped, the default); see DESIGN.md")
    ap.add_argument("--autotune", action="store_true",
                    help="pick the march occupancy cap (0 or 5) by timing 30 frames of each first")
    ap.add_argument("--classify-share", type=int, default=-1,
                    help="LDS reserve (bytes) per classify workgroup beside the march; -1 = "
                         "balanced by the C++ driver (avr_renderer_set_classify_share)")
    ap.add_argument("--layout", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="avr_renderer_set_overlap: -1 measured by the driver (default), 0 back to "
                         "back, 1 side by side, 2 paired (A/B only)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="diagnostics: no HIP timing events around the two paint kernels in the "
                         "timed region (roofline.kernel_ms is then the frame time itself)")
    ap.add_argument("--debug-stall-rank", type=int, default=-1,
                    help="test hook: this rank stops rendering after the settling frames and sleeps "
                         "(a peer that hangs): the others must run into AVR_FRAME_TIMEOUT_MS, say "
                         "where, and the launch must end non-zero")
    ap.add_argument("--occlusion-culling", type=int, default=-1,
                    help="avr_renderer_set_occlusion_culling: -1 / 0 off (the driver's default), "
                         "k >= 2: every frame in k culled chunks")
    ap.add_argument("--no-speculation", action="store_true",
                    help="A/B only: avr_renderer_set_visibility_speculation(0) -- every frame classifies "
                         "every box (the driver's default speculates where a standing camera's rays "
                         "sample at most 85 % of the boxes: the opaque regime)")
    ap.add_argument("--corun-full-search", action="store_true",
                    help="A/B only: one rank times every candidate of the co-run search (rounds 2-4) "
                         "instead of balancing the two kernels by their durations "
                         "(avr_renderer_set_corun_balance)")
    ap.add_argument("--no-coordination", action="store_true",
                    help="N > 1, A/B only: every rank runs the co-run search on its own (round 3) "
                         "instead of all ranks as one system (avr_renderer_set_corun_coordination)")
    ap.add_argument("--no-plan-ahead", action="store_true",
                    help="--fly-through at N > 1: make every frame plan on the thread that queues "
                         "the frames instead of one frame ahead on a helper thread")
    ap.add_argument("--no-self-check", action="store_true",
                    help="experiment builds only (tools/ab_*.sh variants that change results)")
    ap.add_argument("--check-collectives", action="store_true",
                    help="N > 1: also render one untimed frame through the torch.distributed "
                         "collectives (all_to_all_single / gather) and require rank 0's bytes to "
                         "equal the C++ driver's (grouped ncclSend / ncclRecv)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="debug: all ranks share cuda:0 (RCCL refuses two ranks on one device) and "
                         "the C++ frame driver's exchange / gather go through a shared-memory "
                         "segment (avr_comm_create_shared): the whole N > 1 flow of this script -- "
                         "gloo control plane, native driver, checks, JSON line -- on a 1-GPU box; "
                         "not a measurement")
    ap.add_argument("--rehearse-python-pipeline", action="store_true",
                    help="with --rehearse-on-one-gpu: the torch.distributed frame loop over gloo "
                         "through host copies instead of the C++ driver")
    return ap.parse_args()


CONFIG_IMAGE = {"config1": (256, 256), "config2": (1024, 1024), "config3": (2048, 2048),
                "config4": (2048, 2048), "config5": (4096, 4096), "tiny": (256, 256),
                "hostbound": (2048, 2048)}


def cpu_baseline(spec, local_boxes, renderer, rparams, camera, seconds, frame_samples):
    """The reference's frame in the reference's shape, on the host cores of this box, through the
    oracle (the CPU restatement, oracle/avr_oracle.c) -- reported beside the GPU number; never the
    thing measured or shipped.

    renderSingleTrial times three stages (reportStageTime, VolumeRenderer.cpp:1121-1136):
    per-box rendering (:1200-1219: VolumePainter::paint into one full-frame layer per box),
    compositing (:1231-1253: composeLayered) and gather + write; the reference is serial per rank
    (AMReX_OMP OFF) and parallel over MPI ranks.  Unextrapolated wherever the budget allows (it
    does for config-1 to config-4): EVERY box is painted at full resolution on T threads (OpenMP
    over image rows stands in for T ranks painting their own boxes) and, as one reference rank,
    again on one thread; ALL full-frame layers are folded by orc_compose_layered on one thread (what
    one rank does for its piece, times the ranks); the result is converted to bytes.  `seconds`
    caps the CPU time (config-5's 1856 layers of 1.34 GB cannot be held): what did
    not fit is extrapolated by samples (paint) and by layers (compose), and the line says so."""
    import numpy as np
    from oracle import oracle as O
    threads = min(os.cpu_count() or 1, 16)
    params, _ = renderer.make_params(rparams)
    ocam = O.make_camera(camera.eye, camera.look_at, camera.up, camera.fov_y_degrees,
                         camera.near_plane, camera.far_plane)
    tr = spec.transform
    otr = O.make_transform(tr.log_scale_input, tr.normalize_to_unit_range, tr.positive_floor,
                           tr.normalization_min, tr.inverse_normalization_span)
    op = O.make_params(params.width, params.height, spec.scalar_range, rparams.box_transparency,
                       renderer.reference_sample_distance, spec.bounds.min_corner,
                       spec.bounds.max_corner)
    n = len(local_boxes)
    layer_bytes = params.width * params.height * 20
    # (the layers are kept for the fold: at most ~24 GB of them)
    keep = max(8, min(n, int(24e9 // layer_bytes)))
    begin = time.perf_counter()
    deadline = begin + seconds
    layers, hints = [], []
    paint_s = paint_samples = 0
    serial_s = serial_samples = 0
    painted = 0
    o_boxes = []
    for i in range(n):
        if i >= keep or (time.perf_counter() > deadline * 0.5 + begin * 0.5 and painted >= 8):
            break   # (half the budget belongs to the one-rank pass and the fold)
        box = local_boxes[i]
        ob = O.make_box(box.values.cpu().numpy(), box.min_corner, box.max_corner)
        o_boxes.append(ob)
        t0 = time.perf_counter()
        layer, ns = O.paint_box(ob, otr, op, ocam, threads=threads)
        paint_s += time.perf_counter() - t0
        paint_samples += ns
        layers.append(layer)
        hints.append(O.box_depth_hint(ob, ocam))
        painted += 1
    serial_boxes = 0
    for ob in o_boxes:   # one reference rank: the same boxes on one thread
        if time.perf_counter() > deadline and serial_boxes >= 4:
            break
        t0 = time.perf_counter()
        _, ns1 = O.paint_box(ob, otr, op, ocam, threads=1)
        serial_s += time.perf_counter() - t0
        serial_samples += ns1
        serial_boxes += 1
    used = len(layers)
    t0 = time.perf_counter()
    image, _, _ = O.compose_layered(layers, hints, [0] * used, list(range(used)), 1)
    compose_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.quantize_rgb8(image, params.width, params.height)
    bytes_s = time.perf_counter() - t0
    total = float(frame_samples)
    whole = used == n and serial_boxes == n
    # the whole frame (identity where every box was painted): paint scales with the samples, the
    # fold with the layers
    paint_frame = paint_s * total / max(paint_samples, 1)
    paint_serial = serial_s * total / max(serial_samples, 1)
    compose_serial = compose_s * n / used
    compose_ranks = compose_serial / threads    # every rank folds its own 1/T of the pixels
    frame = paint_frame + compose_ranks + bytes_s
    frame_serial = paint_serial + compose_serial + bytes_s
    return {
        "value": round(total / frame / 1e6, 3), "unit": "Mray-samples/s", "cores": threads,
        "kind": "port", "frames_per_s": round(1.0 / frame, 4),
        "extrapolated": not whole,
        "stages_s": {"per_box_rendering": round(paint_frame, 3),
                     "compositing": round(compose_ranks, 3),
                     "gather_and_bytes": round(bytes_s, 3)},
        "one_rank": {"frames_per_s": round(1.0 / frame_serial, 4),
                     "value": round(total / frame_serial / 1e6, 3),
                     "stages_s": {"per_box_rendering": round(paint_serial, 3),
                                  "compositing": round(compose_serial, 3),
                                  "gather_and_bytes": round(bytes_s, 3)}},
        "sample": f"oracle frame in the reference's stages (VolumeRenderer.cpp:1121-1136) at "
                  f"{params.width}x{params.height}, same camera and transfer function: "
                  f"VolumePainter::paint of {used} of the {n} boxes ({paint_samples} of the frame's "
                  f"{int(total)} samples in {paint_s:.2f} s on {threads} OpenMP threads over image "
                  f"rows) and of {serial_boxes} of them on one thread = one reference rank "
                  f"({serial_samples} samples in {serial_s:.2f} s), orc_compose_layered of those "
                  f"{used} full-frame layers on one thread ({compose_s:.2f} s; divided by {threads} "
                  f"for {threads} ranks folding their own pieces), 8-bit conversion {bytes_s:.3f} s; "
                  + ("nothing extrapolated" if whole else
                     "stage times extrapolated to the whole frame by samples / layers "
                     "(the --cpu-seconds budget)"),
    }


def latency(ctx, spec, all_boxes, local_boxes, rparams, cameras):
    """What a drop-in caller of Render() gets, beside the pipelined headline (untimed, after it):
    the reference renders ONE frame per call and returns (VolumeRenderer.cpp:1103-1339,
    Examples/RenderFromMultiFab.cpp:17-62, module.cpp:252-255).  single_frame_ms: a FRESH renderer,
    a camera it has never seen per frame, render + synchronize on the host clock (plan + classify +
    march + fold, nothing overlapped with another frame), median of 10 after the frame that
    allocates; first_100: another fresh renderer, 100 frames of one camera queued back to back --
    their mean period and the frame at which the driver reported its co-run layout settled."""
    import statistics
    import torch
    from amrvolumerenderer_amd import scenes
    from amrvolumerenderer_amd.renderer import FrameRenderer

    def fresh():
        return FrameRenderer(ctx, all_boxes, local_boxes, spec.transform, spec.bounds,
                             spec.scalar_range, 0, 1, None)

    renderer = fresh()
    singles = []
    for i in range(11):
        cam = scenes.orbit_camera(7 * i + 3, 3600)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        renderer.render(rparams, cam)
        renderer.synchronize()
        singles.append((time.perf_counter() - t0) * 1e3)
    del renderer
    renderer = fresh()
    settled_at = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(100):
        renderer.render(rparams, cameras[0])
        if settled_at is None and renderer.native.corun_state()["settled"]:
            settled_at = i + 1
    renderer.synchronize()
    mean_100 = (time.perf_counter() - t0) * 1e3 / 100
    while settled_at is None and i < 2000:   # (bounded: the full search takes ~700 frames)
        i += 1
        renderer.render(rparams, cameras[0])
        if renderer.native.corun_state()["settled"]:
            settled_at = i + 1
    renderer.synchronize()
    corun = renderer.native.corun_state()
    del renderer
    return {"single_frame_ms": round(statistics.median(singles[1:]), 4),
            "single_frame_ms_min_max": [round(min(singles[1:]), 4), round(max(singles[1:]), 4)],
            "first_frame_with_allocations_ms": round(singles[0], 3),
            "first_100_frames_mean_ms": round(mean_100, 4),
            "frames_to_settle": settled_at, "settled_on": corun}


PMC_DIR = os.path.join("profiles", "r5_final")
PMC_SOURCES = ("avr_kernels.hip", "avr_device.h", "avr_renderer.cpp")


def kernel_sources_sha256():
    """Identity of what the PMC counters were taken on: the two paint kernels and the driver that
    schedules them (tools/pmc_passes.sh writes the same digest into its summary)."""
    import hashlib
    digest = hashlib.sha256()
    for name in PMC_SOURCES:
        with open(os.path.join(ROOT, "amrvolumerenderer_amd", "csrc", name), "rb") as fh:
            digest.update(fh.read())
    return digest.hexdigest()


def workload_key(args, world):
    """Names the workloads whose PMC summaries are committed (profiles/r5_final/pmc_<key>.txt):
    one rank, the smooth field, the default camera, the configuration's own image size."""
    if (world != 1 or args.field != "smooth" or args.width or args.height or args.orbit
            or args.fly_through or args.cache_classification):
        return None
    if args.antialiasing != (4 if args.config == "config5" else 1):
        return None
    regime = {0.97: "translucent", 0.0: "opaque"}.get(args.transparency)
    if regime is None or args.config not in ("config2", "config3", "config4", "config5"):
        return None
    return f"{args.config}_{regime}"


def profiled_traffic(args, world):
    """HBM bytes per paint-stage launch from the committed rocprofv3 PMC summary of THIS workload
    (separate --pmc passes, tools/pmc_passes.sh) -- only if that summary was taken on the kernels
    of this tree: its first line carries the sha256 of the kernel and driver sources, and a tree
    whose sources differ gets (None, "stale ...") instead of somebody else's bytes.  FETCH_SIZE and
    WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B/lane)
    coalesced stream, so the classify kernel's reads are doubled and the march's byte gathers
    are not (MI355X_MICROARCH.md, HBM)."""
    key = workload_key(args, world)
    if key is None or getattr(args, "no_speculation", False):   # (an A/B run: the summaries are the default driver's)
        return None, None
    summary = os.path.join(PMC_DIR, f"pmc_{key}.txt")
    path = os.path.join(ROOT, summary)
    if not os.path.exists(path):
        return None, f"no PMC summary ({summary})"
    counters, kernel, recorded = {}, None, None
    for line in open(path):
        if line.startswith("# sources sha256:"):
            recorded = line.split(":", 1)[1].split()[0]
            continue
        if line.startswith("#"):
            continue
        if not line.startswith(" "):
            kernel = line.strip()
            continue
        parts = line.split()
        if len(parts) >= 3 and parts[-1].startswith("mean="):
            counters[(kernel, parts[0])] = float(parts[-1][5:])
    if recorded != kernel_sources_sha256():
        return None, (f"stale: {summary} was taken on other kernel sources "
                      f"({(recorded or 'no digest')[:12]}...): rerun tools/pmc_passes.sh")
    try:
        kib = (2.0 * counters[("classify_kernel", "FETCH_SIZE")]
               + counters[("classify_kernel", "WRITE_SIZE")]
               + counters[("render_runs_kernel", "FETCH_SIZE")]
               + counters[("render_runs_kernel", "WRITE_SIZE")])
    except KeyError:
        return None, f"{summary} lacks FETCH_SIZE / WRITE_SIZE"
    return int(kib * 1024), summary


def launcher_command(n_ranks, port, argv):
    """The command the driver itself uses for N > 1 (one rank per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1", "--master-port",
            str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(args, argv, run=None):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as CHILD processes (never
    an exec, and nothing here has touched the GPU or even imported torch), relay rank 0's JSON
    line, return the children's exit status.  `run` is injectable for the launcher's unit test."""
    import subprocess
    # the native library is built once here (hipcc needs no GPU) instead of by N racing children
    from amrvolumerenderer_amd import build as avr_build
    avr_build.build()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this platform
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    command = launcher_command(args.gpus, free_port(), argv)
    run = run or subprocess.run
    done = run(command, env=env, stdout=subprocess.PIPE, text=True)
    lines = [line for line in (done.stdout or "").splitlines() if line.startswith('{"metric"')]
    for line in (done.stdout or "").splitlines():
        if not line.startswith('{"metric"'):
            print(line, file=sys.stderr)
    if done.returncode != 0:
        print(f"bench.py: the {args.gpus}-rank launch failed with status {done.returncode}",
              file=sys.stderr)
        return done.returncode or 1
    if len(lines) != 1:
        print(f"bench.py: expected one result line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    state = {}
    try:
        run(args, state)
    except BaseException as error:  # noqa: BLE001 -- reported, then the process ends at once
        if isinstance(error, SystemExit) and error.code in (0, None):
            raise
        # A frame of several ranks either completes or errors (avr_set_frame_timeout_ms): say which
        # rank failed where and how its co-run search stood, then leave WITHOUT the interpreter's
        # shutdown -- a stream that does not move would hang torch's teardown (and with it the job,
        # until the launcher's own limit); the launcher ends the other ranks.
        import traceback
        renderer = state.get("renderer")
        native = getattr(renderer, "native", None) if renderer is not None else None
        lines = [f"bench.py: rank {os.environ.get('RANK', '0')} of "
                 f"{os.environ.get('WORLD_SIZE', '1')} failed: {type(error).__name__}: {error}"]
        if native is not None:
            try:
                lines.append(f"  renderer: {native.failure() or 'not failed (the error came from elsewhere)'}")
                lines.append(f"  corun_state: {native.corun_state()}")
            except Exception as inner:  # noqa: BLE001
                lines.append(f"  (no renderer state: {inner})")
        print("\n".join(lines), file=sys.stderr, flush=True)
        traceback.print_exc()
        sys.stderr.flush()
        code = error.code if isinstance(error, SystemExit) and isinstance(error.code, int) else 4
        os._exit(code or 4)


def run(args, state):
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The process group is the CONTROL plane only -- RCCL's 128-byte id, the barriers, the
        # sample-count and max-time reductions -- and runs over gloo, so that the one RCCL
        # communicator in the process is the C++ driver's own (a second one, torch's, would
        # share the GPU's queues with it for nothing).
        # Its collectives carry a timeout of their own (gloo's default is 30 minutes): a rank
        # that hangs must end the job within minutes, also where the wait is a barrier or a
        # reduction of this script and not the C++ driver's (whose control rounds run under
        # AVR_FRAME_TIMEOUT_MS anyway, avr_comm_control_allgather).
        import datetime
        limit_ms = int(os.environ.get("AVR_FRAME_TIMEOUT_MS", "0") or 0) or 30000
        dist.init_process_group("gloo", rank=rank, world_size=world,
                                timeout=datetime.timedelta(milliseconds=max(4 * limit_ms, 120000)))
        group = dist.group.WORLD

    from amrvolumerenderer_amd import build as avr_build
    if rank == 0:
        avr_build.build()
    if world > 1:
        dist.barrier()
    from amrvolumerenderer_amd import runtime, scenes
    from amrvolumerenderer_amd.renderer import (FrameRenderer, RenderParameters,
                                                build_scene_on_device)

    if args.config == "tiny":
        spec = scenes.make_amr_scene(64, 2, 16, args.field, "tiny_amr2_64")
    elif args.config == "hostbound":
        # 176 boxes' worth of host planning, but only the 8 finest boxes... keep it simple: a
        # 3-level scene of 8^3 boxes whose GPU work is small next to the host cost of a frame
        spec = scenes.make_amr_scene(32, 3, 8, args.field, "hostbound_amr3_32")
    else:
        spec = getattr(scenes, args.config)(args.field)
    scenes.assign_owners(spec, world, args.ownership)
    width, height = CONFIG_IMAGE[args.config]
    width = args.width or width
    height = args.height or height

    ctx = runtime.Context(local_rank)
    all_boxes, local_boxes = build_scene_on_device(ctx, spec, rank)
    torch.cuda.synchronize()
    rehearsal_comm = None
    if world > 1 and args.rehearse_on_one_gpu and not args.rehearse_python_pipeline:
        # one segment per job: its name travels from rank 0 over the control plane
        token = [f"/avr_bench_{os.getpid()}_{int(time.time() * 1e3) & 0xffffff:x}"]
        dist.broadcast_object_list(token, src=0, group=group)
        rehearsal_comm = runtime.Comm.shared(token[0], rank, world, 512 << 20)
        # the control plane the real run uses (gloo through the C ABI's callback), so that the
        # rehearsal exercises it too: plan agreement and co-run windows travel over it
        rehearsal_comm.set_control(runtime.control_over_process_group(group))
    renderer = FrameRenderer(ctx, all_boxes, local_boxes, spec.transform, spec.bounds,
                             spec.scalar_range, rank, world, group, comm=rehearsal_comm,
                             stage_through_host=(args.rehearse_on_one_gpu and
                                                 rehearsal_comm is None),
                             march_workgroups_per_cu=args.march_occupancy,
                             stream_priorities=tuple(int(v) for v in args.priorities.split(",")),
                             cache_classification=args.cache_classification)
    state["renderer"] = renderer
    if renderer.native is not None and args.occlusion_culling >= 0:
        renderer.native.set_occlusion_culling(args.occlusion_culling)
    if renderer.native is not None and args.corun_full_search:
        renderer.native.set_corun_balance(0)
    if renderer.native is not None and args.no_speculation:
        renderer.native.set_visibility_speculation(0)
    if renderer.native is not None and args.no_coordination:
        renderer.native.set_corun_coordination(0)
    if renderer.native is not None and args.classify_share >= 0:
        renderer.native.set_classify_share(args.classify_share)
    if renderer.native is not None and args.layout >= 0:
        renderer.native.set_overlap(args.layout)
    rparams = RenderParameters(width=width, height=height, box_transparency=args.transparency,
                               antialiasing=args.antialiasing,
                               draw_bounds=False)  # not part of the metric
    cameras = ([scenes.orbit_camera(v, args.orbit) for v in range(args.orbit)]
               if args.orbit > 0 else [scenes.default_camera()])
    # --fly-through: a camera that never repeats (0.1 degree of orbit per frame).  The timed
    # frames are views FLY_TIMED + i; everything untimed before them uses views counted up from 0
    # (never reaching FLY_TIMED), so no timed frame finds a plan in the driver's cache.
    FLY_VIEWS, FLY_TIMED = 3600, 1_000_000
    if args.fly_through:
        if args.orbit:
            raise SystemExit("--fly-through and --orbit exclude each other")
        cameras = [scenes.orbit_camera(FLY_TIMED + i, FLY_VIEWS) for i in range(args.steps)]
    fly_view = [0]

    # ---- untimed: sample count of one frame per camera (device-side counter, stats build) ----
    samples_dev = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    frame_samples = []
    for cam in cameras:
        samples_dev.zero_()
        renderer.render(rparams, cam, samples=samples_dev)
        renderer.synchronize()
        torch.cuda.synchronize()
        s = samples_dev.clone()
        if world > 1:
            s = s.cpu()
            dist.all_reduce(s, group=group)
        frame_samples.append(int(s.item()))
    local_runs = renderer.last_plan.n_local_runs
    total_runs = renderer.last_plan.n_runs_total
    send_floats = renderer.last_plan.send_floats

    # ---- untimed self-checks of the N-rank frame (no oracle here: properties only) ------------
    checks = {}
    default_scene = (args.config == "config4" and args.field == "smooth" and not args.width
                     and not args.height and args.transparency == 0.97 and args.antialiasing == 1
                     and args.orbit == 0 and not args.fly_through)
    if default_scene and not args.no_self_check:
        # the ranks' shares of the frame add up to the one-rank frame's samples, whatever N
        # (the count tests/test_full_size_gpu.py checks against the oracle)
        checks["samples_sum_equals_one_rank_frame"] = frame_samples[0] == 759136367
    if world > 1 and renderer.native is not None and args.check_collectives:
        # the same frame through the other implementation of the collectives: the C++ driver's
        # grouped ncclSend / ncclRecv (avr_exchange / avr_gather) against torch.distributed's
        # all_to_all_single / gather on the same RCCL transport -- rank 0's bytes must agree
        twin_group = group if args.rehearse_on_one_gpu else dist.new_group(backend="nccl")
        # (the twin is the torch.distributed frame loop: over gloo through host copies when the
        # ranks share one GPU)
        twin = FrameRenderer(ctx, all_boxes, local_boxes, spec.transform, spec.bounds,
                             spec.scalar_range, rank, world, twin_group, native=False,
                             stage_through_host=args.rehearse_on_one_gpu)
        _, a = renderer.render(rparams, cameras[0])
        _, b = twin.render(rparams, cameras[0])
        renderer.synchronize()
        twin.synchronize()
        torch.cuda.synchronize()
        same = torch.ones(1)
        if rank == 0:
            same[0] = float(torch.equal(a, b) and bool(a.any()))
        dist.broadcast(same, 0, group=group)
        checks["native_collectives_equal_torch_distributed"] = bool(same.item())
        del twin
    if not all(checks.values()):
        if rank == 0:
            print(f"bench.py: self-check failed: {checks}", file=sys.stderr)
        raise SystemExit(3)

    if args.fly_through and renderer.native is not None:
        renderer.native.set_tighten(True)   # forgets the plans the sample counting left behind
        if world > 1:
            # every frame of a fly-through has a NEW plan: agreeing on each over the control plane
            # (a gloo allgather: a few tenths of a millisecond) would cost more than a rank of
            # eight's frame.  The ranks' first plans were agreed on above (same scene, settings and
            # camera path on every rank by construction of this script), so the check is off from
            # here on -- what avr_renderer_set_plan_check(0) is for; the deadline still ends a
            # frame whose ranks were driven apart.
            renderer.native.set_plan_check(False)

    # A scripted fly-through knows its next camera: for N > 1 the plan of frame f + 1 is made on a
    # helper thread while frame f is queued (avr_renderer_prepare; at N = 1 a plan costs the host
    # 15 us of a 1 ms frame and nobody needs it).  --no-plan-ahead: every plan on the frames' thread.
    ahead = None
    if (args.fly_through and world > 1 and renderer.native is not None
            and not args.no_plan_ahead):
        from amrvolumerenderer_amd.runtime import PlanAhead
        ahead = PlanAhead(renderer.native)

    def plan_next(camera):
        ahead.submit(rparams.width, rparams.height, rparams.box_transparency,
                     rparams.antialiasing, camera, rparams.use_visibility_graph,
                     rparams.draw_bounds, rparams.write_visibility_graph)

    def step(i, timed=False):
        if args.fly_through and not timed:
            fly_view[0] += 1
            if ahead is not None:
                plan_next(scenes.orbit_camera(fly_view[0] + 1, FLY_VIEWS))
            return renderer.render(rparams, scenes.orbit_camera(fly_view[0], FLY_VIEWS))
        if ahead is not None and i + 1 < len(cameras):
            plan_next(cameras[i + 1])
        return renderer.render(rparams, cameras[i % len(cameras)])

    # Untimed setup (optional): pick the march occupancy cap for this workload
    if args.autotune and world == 1:
        renderer.autotune(rparams, cameras[0], frames=30)
    # Untimed setup, like the sample counting above: the frames in which the C++ frame driver
    # settles how this rank's two kernels share the GPU (avr_renderer_corun_state; bounded at 2 s)
    # -- they also bring the GPU to its working clocks and fill the allocator pools, so that a run
    # timing very few steps (the driver's --steps 20 is a 20 ms timed region) measures the steady
    # state a 200-step run does; then the W warm-up steps of the contract.
    burst_begin = time.perf_counter()
    burst = 0

    def settled():
        return renderer.native is None or renderer.native.corun_state()["settled"]

    # (N ranks: every frame is a collective, so every rank must run the same number of them --
    # a fixed count, well past the <= 2900 frames the search takes at short frames, instead of a
    # clock)
    # (a one-GPU rehearsal moves every block through host memory: it is not there to settle)
    fixed_burst = (48 if args.rehearse_on_one_gpu else 4000) if world > 1 else None
    while True:
        for _ in range(16):
            step(burst)
            burst += 1
        elapsed = time.perf_counter() - burst_begin
        if fixed_burst is not None:
            done = burst >= fixed_burst
        else:
            # (one rank: the driver balances its two kernels within ~85 frames,
            # avr_renderer_set_corun_balance; the full search of rounds 2-4 took 600-1500)
            done = burst >= 32 and (elapsed >= 2.0 or settled())
        if done:
            break
        # (No synchronise in here for the C++ driver: its queue is a few frames deep at most
        # (descriptor ring, back-pressure), so the clock above means GPU time anyway -- and a drain
        # every 64 frames voided every window of the driver's search that is longer than that: the
        # finalists' double windows at frames under 0.5 ms (config-2, config-3: 44 + 44 frames), so
        # that the search only finished inside the timed region and a run read 0.38 or 0.43 ms
        # depending on which finalist -- "back to back" among them -- the timed frames fell on.)
        if renderer.native is None and burst % 64 == 0:
            renderer.synchronize()   # keep the queue short so the clock check means GPU time
    renderer.synchronize()
    params, _ = renderer.make_params(rparams)
    native = renderer.native is not None
    if args.debug_stall_rank == rank and world > 1:
        print(f"bench.py: rank {rank} stalls on purpose (--debug-stall-rank)", file=sys.stderr, flush=True)
        time.sleep(3600)
    for i in range(args.warmup):
        step(i)

    # ---- timed region: EXACTLY `steps` frames ----------------------------------------------
    # the two paint kernels' own durations are taken with HIP events on the streams they are
    # launched on (classify on classify_ctx.stream, march on march_ctx.stream).
    # Nothing but the contract's synchronise (+ barrier) lies between the last warm-up frame and
    # t0: this GPU drops its clocks when it idles and takes milliseconds to come back (measured
    # with a sleep here: 0.5 ms idle 1.011 ms per frame over 20 steps, 5 ms 1.095, 50 ms 1.117 --
    # the first march after the pause runs 1.25 ms instead of 0.97), so the untimed bookkeeping
    # is not added to that pause.
    if native:
        renderer.native.set_timing(not args.no_kernel_timing)   # drains the streams, records the epoch
        renderer.native.host_profile(reset=True)
    else:
        renderer.synchronize()
        renderer.kernel_events = []
        epoch = torch.cuda.Event(enable_timing=True)
        epoch.record(renderer.march_ctx.stream)
    torch.cuda.synchronize()
    t0 = time.monotonic()
    if world > 1:
        # The ranks leave a gloo barrier a few hundred microseconds apart, which is a tenth of
        # the 3 ms a rank of eight times over 20 frames: after the barrier they agree on ONE start
        # instant (CLOCK_MONOTONIC is the same clock in every process of the node) and every rank
        # measures from it -- a rank that hears of it late starts late and is counted late.
        dist.barrier()
        start = torch.tensor([time.monotonic() + 3e-4], dtype=torch.float64)
        dist.broadcast(start, src=0, group=group)
        t0 = float(start.item())
        while time.monotonic() < t0:
            pass
    for i in range(args.steps):
        step(i, timed=True)
    renderer.synchronize()
    torch.cuda.synchronize()
    # this rank's K frames are complete (every frame is a collective, so they are complete on every
    # rank within an exchange of each other); the closing barrier's own latency is not frame time:
    # the maximum over the ranks, below, is what makes the figure the slowest rank's
    elapsed = time.monotonic() - t0
    if world > 1:
        dist.barrier()
    planned_ahead = ahead is not None
    if ahead is not None:
        ahead.close()

    t = torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    elapsed = float(t.item())
    # The classify pass of frame i+1 runs beside the march of frame i, so each kernel's own
    # duration (what rocprofv3 --stats reports) includes the time it shared the GPU with the
    # other and their sum exceeds the frame time.  The paint stage's GPU time per frame is the
    # length of the union of the kernels' execution intervals over the timed region / frames.
    host_us = None
    if native:
        sections, profiled = renderer.native.host_profile()
        if profiled == args.steps:
            host_us = {k: round(v, 1) for k, v in sections.items()}
        if args.no_kernel_timing:
            classify_ms = march_ms = float("nan")
            kernel_ms, n_events = elapsed * 1e3 / args.steps, args.steps
        else:
            classify_ms, march_ms, kernel_ms, n_events = renderer.native.timings()
        renderer.native.set_timing(False)
        assert n_events == args.steps
    else:
        kernel_events, renderer.kernel_events = renderer.kernel_events, None
        n_events = max(len(kernel_events), 1)
        classify_ms = sum(c0.elapsed_time(c1) for c0, c1, _, _ in kernel_events) / n_events
        march_ms = sum(m0.elapsed_time(m1) for _, _, m0, m1 in kernel_events) / n_events
        spans = []
        for c0, c1, m0, m1 in kernel_events:
            spans.append((epoch.elapsed_time(c0), epoch.elapsed_time(c1)))
            spans.append((epoch.elapsed_time(m0), epoch.elapsed_time(m1)))
        spans.sort()
        busy, cursor = 0.0, float("-inf")
        for begin, end in spans:
            if end > cursor:
                busy += end - max(begin, cursor)
                cursor = end
        kernel_ms = busy / n_events

    samples_total = sum(frame_samples[i % len(cameras)] for i in range(args.steps))
    ms_per_step = elapsed * 1e3 / args.steps
    value = samples_total / elapsed / 1e6

    # ---- roofline of the paint stage (classify_kernel + render_runs_kernel, launched back to
    # back by one avr_render_plan call), this rank's launch ----------------------------------
    # algorithmic bytes per launch = 8 B per executed cell fetch (one amrex::Real) + 20 B per
    # emitted layer pixel; samples of THIS rank's launch:
    samples_dev.zero_()
    renderer.render(rparams, cameras[0], samples=samples_dev)
    renderer.synchronize()
    my_samples = int(samples_dev.item())
    if args.fly_through and world == 1:   # the timed frames' own average, not the first view's
        my_samples = round(sum(frame_samples) / len(frame_samples))
    n_pixels = params.width * params.height
    send_floats = renderer.last_plan.send_floats   # (N > 1: the tightened layout by now)
    algo_bytes = 8.0 * my_samples + 4.0 * send_floats
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_source = profiled_traffic(args, world)
    # (a speculating driver -- one rank, a standing camera, at most 85 % of the boxes sampled:
    # the opaque regime -- classifies only the boxes an earlier frame's rays sampled; the boxes of
    # BASELINE's configurations are all of one size, so the fraction of boxes is that of the cells)
    speculation = renderer.native.speculation_state() if native else None
    classified_fraction = (speculation["sampled_fraction"]
                           if speculation and speculation["state"] == "speculating"
                           and speculation["sampled_fraction"] is not None else 1.0)
    roofline = {
        "bound": "hbm", "kernel": "classify_kernel + render_runs_kernel (the paint stage of one frame)",
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        # `frac` is the ALGORITHMIC fraction the task defines (8 B per sample + 20 B per stored
        # layer pixel over the paint stage's GPU time).  The kernels move far fewer bytes -- the
        # f64 cells are read once per frame by the classify pass and the march gathers one
        # classified byte per sample from cache -- so the fraction of the HBM peak the MEASURED
        # traffic amounts to is reported beside it and the two must be quoted together.
        "frac": round(achieved / HBM_PEAK_GBS, 5), "algorithmic_frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": traffic, "traffic_source": traffic_source,
        "traffic_frac": (round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                         if traffic else None),
        "kernel_ms": round(kernel_ms, 4), "classify_ms": round(classify_ms, 4),
        "march_ms": round(march_ms, 4), "algorithmic_bytes": int(algo_bytes),
        "samples_this_rank": my_samples,
        "classified_fraction": classified_fraction,
        "compulsory_bytes": int(sum(b.values.numel() for b in local_boxes) * 8 * classified_fraction
                                + 4 * send_floats),
    }
    # the third reading SURVEY.md 8(d) asks for: what the frame cannot avoid moving (every f64 cell
    # the frame's rays can reach once + the stored layer pixels) over the same kernel time
    roofline["compulsory_frac"] = round(
        roofline["compulsory_bytes"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)

    corun_ranks = [renderer.native.corun_state() if native else None]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, corun_ranks[0], group=group)
        corun_ranks = gathered
    out = {
        "metric": f"Mray-samples/s ({width}x{height} render of {spec.n0}^3-base "
                  f"{spec.levels}-level AMR" + (f", antialiasing {args.antialiasing}"
                                                 if args.antialiasing > 1 else "") + ")",
        "value": round(value, 3), "unit": "Mray-samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "frames_per_s": round(1e3 / ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32 march / f64 cell fetch+transform", "data": "synthetic",
        "config": {
            "workload": f"{spec.name}: {len(spec.boxes)} boxes of {spec.box_cells}^3 "
                        f"({spec.total_cells * 8 / 1e9:.2f} GB f64), {spec.levels} AMR levels, "
                        f"field={spec.field}, {width}x{height}, antialiasing={args.antialiasing}, "
                        f"box_transparency={args.transparency}, default jet map, "
                        f"{len(cameras)} view(s)",
            "ownership": args.ownership, "runs_total": total_runs,
            "frame_driver": ("C++ (avr_renderer: 3 HIP streams" +
                             ((", exchange + gather through shared memory (one-GPU rehearsal)"
                               if rehearsal_comm is not None else ", RCCL exchange + gather")
                              if world > 1 else "") + ")") if native
                            else ("Python pipeline (torch streams; gloo through host copies)"
                                  if args.rehearse_on_one_gpu else
                                  "Python pipeline with torch.distributed RCCL collectives -- the "
                                  f"C++ driver's communicator failed: {renderer.native_error}"),
            "march_workgroups_per_cu": renderer.march_workgroups_per_cu,
            "corun": corun_ranks[0] if native else None,
            # one rank, a standing camera: frames may classify only the boxes an earlier frame's
            # rays sampled (checked by the march, repaired if wrong: results never change)
            "visibility_speculation": renderer.native.speculation_state() if native else None,
            # N > 1: what every rank's driver holds (searched as one system: the same on all)
            "corun_ranks": corun_ranks if (native and world > 1) else None,
            "exchange": ({"rank0_send_mb": round(renderer.last_plan.send_floats * 4 / 1e6, 2),
                          "layout": "per-row extents of the runs (avr_frame_plan_tighten)"
                          if native else "run rectangles"} if world > 1 else None),
            "classification": ("cached across frames (cells not re-read: not the headline "
                               "configuration)" if args.cache_classification else "every frame"),
            "samples_per_frame": (frame_samples[0] if len(frame_samples) == 1 else
                                  (round(sum(frame_samples) / len(frame_samples))
                                   if args.fly_through else frame_samples)),
            "camera": ("fly-through: every frame a camera the driver has never seen (0.1 degree "
                       "of orbit per frame; plan cache useless"
                       + ("; each frame's plan made one frame ahead on a helper thread, "
                          "avr_renderer_prepare)" if planned_ahead else ")")
                       if args.fly_through else
                       f"orbit of {len(cameras)} views (plans cached)" if args.orbit else
                       "static (the frame plan is made once)"),
            # frames rendered before the W warm-up steps, all untimed: the sample count of every
            # camera (stats build of the march), then the frames that bring the clocks up and let
            # the C++ driver finish measuring how the two kernels share the GPU
            # (corun.timed_windows windows of frames that fill 8 ms each)
            "untimed_frames": {"sample_count": len(cameras), "settle": burst, "warmup": args.warmup},
            "host_us_per_frame": host_us,
            "self_checks": checks,
        },
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_latency and native:
        # (the headline's renderer goes first: a second set of classified volumes beside it is
        # 1.1 GB for config-4, 11 GB for config-5)
        state.pop("renderer", None)
        del renderer
        out["latency"] = latency(ctx, spec, all_boxes, local_boxes, rparams, cameras)
        renderer = FrameRenderer(ctx, all_boxes, local_boxes, spec.transform, spec.bounds,
                                 spec.scalar_range, rank, world, group)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(spec, local_boxes, renderer, rparams, cameras[0],
                                           args.cpu_seconds, frame_samples[0])
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
