"""GPU rehearsal of the N > 1 frame: 3 processes share cuda:0 (within the box's limit of 6),
each owning a third of the boxes, run FrameRenderer.render with the HIP kernels and exchange
over gloo through host copies (RCCL cannot place two ranks on one device).  Rank 0's frame must
be bit-identical to the oracle's 3-rank layered compose."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import spawn_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 120, 72


def _worker(rank, world, port, policy, antialiasing, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from amrvolumerenderer_amd import runtime, scenes
        from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters
        from helpers import device_box
        from test_frame_plan import local_indices, oracle_overlay, painted_scene

        root = int(round(antialiasing ** 0.5))
        spec = scenes.make_amr_scene(32, 2, 8, "smooth")
        cam = scenes.orbit_camera(3)
        cells, layers, hints, ref = painted_scene(O, spec, cam, W * root, H * root, 0.85)
        scenes.assign_owners(spec, world, policy)
        ctx = runtime.Context(0)
        meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
        local = [device_box(ctx, cells[i], spec.boxes[i].min_corner, spec.boxes[i].max_corner,
                            spec.boxes[i].level, rank)
                 for i in scenes.local_box_indices(spec, rank)]
        renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range,
                                 rank, world, dist.group.WORLD, stage_through_host=True)
        samples = torch.zeros(1, dtype=torch.int64, device=ctx.device)
        image, rgb8 = renderer.render(RenderParameters(W, H, 0.85, antialiasing), cam,
                                      samples=samples, want_image=True)
        renderer.synchronize()
        if rank == 0:
            owners = [b.owner for b in spec.boxes]
            want, _, _ = O.compose_layered(layers, hints, owners, local_indices(owners, world),
                                           world)
            if root > 1:
                want = O.downsample(want, W, H, root).reshape(-1, 5)
            # RenderParameters.draw_bounds defaults to the reference's behaviour: every rank
            # overlays its own piece (or rank 0 the downsampled image)
            want = oracle_overlay(O, spec, cells, cam, want, W, H)
            ok = np.array_equal(image.cpu().numpy().reshape(-1, 5).view(np.uint32),
                                want.view(np.uint32))
            ok8 = np.array_equal(rgb8.cpu().numpy(), O.quantize_rgb8(want, W, H))
            with open(out_path, "w") as fh:
                fh.write(f"{int(ok)} {int(ok8)}")
        else:
            assert image is None and rgb8 is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("policy,antialiasing", [("morton", 1), ("round_robin", 4)])
def test_three_ranks_on_one_gpu(tmp_path, policy, antialiasing):
    out = tmp_path / "result.txt"
    spawn_ranks(_worker, 3, lambda port: (3, port, policy, antialiasing, str(out)))
    ok, ok8 = out.read_text().split()
    assert ok == "1", "rank 0's frame differs from the oracle's 3-rank compose"
    assert ok8 == "1", "RGB8 bytes differ"


def _rccl_worker(rank, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from oracle import oracle as O
        from amrvolumerenderer_amd import runtime, scenes
        from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters
        from helpers import device_box
        from test_frame_plan import oracle_overlay, painted_scene

        results = []
        for antialiasing, (w, h) in ((1, (120, 72)), (4, (60, 36)), (1, (75, 43))):
            root = int(round(antialiasing ** 0.5))
            spec = scenes.make_amr_scene(32, 2, 8, "smooth")
            cam = scenes.orbit_camera(5)
            cells, layers, hints, ref = painted_scene(O, spec, cam, w * root, h * root, 0.85)
            ctx = runtime.Context(0)
            meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
            local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level)
                     for c, m in zip(cells, spec.boxes)]
            renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds,
                                     spec.scalar_range, 0, 1, dist.group.WORLD,
                                     force_collectives=True)
            frames = [renderer.render(RenderParameters(w, h, 0.85, antialiasing), cam,
                                      want_image=True) for _ in range(3)]  # pipelined burst
            renderer.synchronize()
            want, _, _ = O.compose_layered(layers, hints, [0] * len(layers),
                                           np.arange(len(layers)), 1)
            if root > 1:
                want = O.downsample(want, w, h, root).reshape(-1, 5)
            want = oracle_overlay(O, spec, cells, cam, want, w, h)
            for image, rgb8 in frames:
                results.append(np.array_equal(image.cpu().numpy().reshape(-1, 5).view(np.uint32),
                                              want.view(np.uint32)))
                results.append(np.array_equal(rgb8.cpu().numpy(), O.quantize_rgb8(want, w, h)))
        t = torch.ones(1, device="cuda:0")
        dist.all_reduce(t)
        dist.barrier()
        with open(out_path, "w") as fh:
            fh.write(" ".join(str(int(r)) for r in results))
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_with_one_rank(tmp_path):
    """The RCCL code path itself (all_to_all_single with split lists, gather of uint8 and float
    pieces into views, three streams, process-group stream ordering) on a one-rank "nccl" group:
    the only RCCL configuration a one-GPU box can run."""
    out = tmp_path / "result.txt"
    spawn_ranks(_rccl_worker, 1, lambda port: (port, str(out)))
    flags = out.read_text().split()
    assert flags and all(f == "1" for f in flags), flags


def _plotfile_worker(rank, world, port, plot_path, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from amrvolumerenderer_amd import api, runtime
        from amrvolumerenderer_amd.types import CameraParameters
        ctx = runtime.Context(0)
        cam = CameraParameters((2.4, 1.7, 2.2), (0.5, 0.5, 0.5), (0.0, 1.0, 0.0), 40.0, 0.05, 30.0)
        options = api.RenderOptions(width=80, height=56, box_transparency=0.5, camera=cam,
                                    output_filename=out_path)
        assert api.run(plot_path, options, "density", ctx, rank, world, dist.group.WORLD,
                       stage_through_host=True) == 0
    finally:
        dist.destroy_process_group()


def test_two_ranks_render_a_plotfile(tmp_path):
    """Multi-rank plotfile ingestion end to end: both ranks read only their own grids, the scene
    statistics are reduced over the ranks, the frame is composited and rank 0 writes the file --
    against the oracle's 2-rank compose with the same ownership."""
    from oracle import oracle as O
    from amrvolumerenderer_amd import plotfile as pf
    from amrvolumerenderer_amd.types import AmrBox, CameraParameters, VolumeBounds
    from amrvolumerenderer_amd import scenes
    from test_frame_plan import local_indices, oracle_camera, oracle_params, oracle_transform
    from test_plotfile import two_level_scene
    levels = two_level_scene(np.random.default_rng(9))
    plot = str(tmp_path / "plt_two")
    pf.write_plotfile(plot, ["density", "noise"], levels, (0.0, 0.0, 0.0), (2.0, 2.0, 2.0), [2])
    out = str(tmp_path / "frame.ppm")
    spawn_ranks(_plotfile_worker, 2, lambda port: (2, port, plot, out))

    W, H, scale = 80, 56, 0.5
    convex = pf.convexify([lev["boxes"] for lev in levels], [2])
    meta, oboxes, cells_all = [], [], []
    for level, lev in enumerate(levels):
        dx = 2.0 / (16 * 2 ** level)
        for parent, (lo, hi) in convex[level]:
            glo = lev["boxes"][parent][0]
            cells = np.ascontiguousarray(
                lev["data"][parent][0][lo[2] - glo[2]:hi[2] - glo[2] + 1,
                                       lo[1] - glo[1]:hi[1] - glo[1] + 1,
                                       lo[0] - glo[0]:hi[0] - glo[0] + 1])
            mn = tuple(lo[a] * dx * scale for a in range(3))
            mx = tuple((hi[a] + 1) * dx * scale for a in range(3))
            meta.append(AmrBox(mn, mx, level=level, dims=cells.shape[::-1]))
            oboxes.append(O.make_box(cells, mn, mx))
            cells_all.append(cells)
    pf.assign_box_owners(meta, 2)  # the partition policy, not arithmetic
    owners = [b.owner for b in meta]
    assert set(owners) == {0, 1}
    lo_v, hi_v = min(c.min() for c in cells_all), max(c.max() for c in cells_all)
    bounds = VolumeBounds((-0.05,) * 3, (1.05,) * 3)
    cam = CameraParameters((2.4, 1.7, 2.2), (0.5, 0.5, 0.5), (0.0, 1.0, 0.0), 40.0, 0.05, 30.0)
    transform = scenes.ScalarTransform(normalize_to_unit_range=True, normalization_min=lo_v,
                                       inverse_normalization_span=1.0 / (hi_v - lo_v))
    ref = O.reference_sample_distance(oboxes, bounds.min_corner, bounds.max_corner)
    op = oracle_params(O, W, H, (0.0, 1.0), 0.5, ref, bounds)
    ocam, otr = oracle_camera(O, cam), oracle_transform(O, transform)
    layers = [O.paint_box(ob, otr, op, ocam)[0] for ob in oboxes]
    hints = [O.box_depth_hint(ob, ocam) for ob in oboxes]
    want, _, _ = O.compose_layered(layers, hints, owners, local_indices(owners, 2), 2)
    tight = O.tight_bounds(oboxes, bounds.min_corner, bounds.max_corner)
    want = O.bbox_overlay(want, W, H, tight[0], tight[1], ocam, 1).reshape(-1, 5)
    data = open(out, "rb").read()
    header = f"P6\n{W} {H}\n255\n".encode()
    assert data.startswith(header)
    got = np.frombuffer(data[len(header):], np.uint8).reshape(H, W, 3)
    assert np.array_equal(got, O.quantize_rgb8(want, W, H))


@pytest.mark.gpu
def test_native_rccl_exchange_and_gather_with_one_rank(O):
    """The C ABI's own collectives (avr_comm_create / avr_exchange / avr_gather: grouped ncclSend /
    ncclRecv on the context's stream) on a one-rank RCCL communicator -- there the block a rank
    keeps for itself is deliberately routed through RCCL -- together with the generic layered
    plan (avr_layered_plan_create, avr_pack_layers, avr_fold_plan), against the oracle."""
    import ctypes as C
    import torch
    from amrvolumerenderer_amd import _capi, runtime
    from test_oracle_compose import synthetic_layers
    L = _capi.lib()
    ctx = runtime.Context(0)
    comm = runtime.Comm(0, 0, 1, lambda ident: ident)
    W, H, n_layers = 61, 43, 7
    layers, hints = synthetic_layers(n_layers, W, H)
    want, _, _ = O.compose_layered(layers, hints, [0] * n_layers, list(range(n_layers)), 1)
    hints_c = (C.c_float * n_layers)(*[float(h) for h in hints])
    owner_c = (C.c_int32 * n_layers)(*([0] * n_layers))
    plan = C.c_void_p()
    _capi.check(L.avr_layered_plan_create(hints_c, owner_c, n_layers, 1, 0, None, W, H,
                                          C.byref(plan)))
    info = _capi.FramePlanInfo()
    _capi.check(L.avr_frame_plan_get_info(plan, C.byref(info)))
    assert info.n_runs_total == 1 and info.send_floats == W * H * 5 == info.recv_floats
    # the communicator's in-band control plane (no caller's allgather installed): a grouped round
    # of tiny ncclSend / ncclRecv on the context's stream -- here to the rank itself
    word = bytes(range(16))
    assert comm.control_allgather(word, ctx) == [word]
    _capi.check(L.avr_frame_plan_agree(plan, comm._handle, ctx._handle, 7))
    assert comm.control_rounds() == 2
    dev = [torch.from_numpy(np.ascontiguousarray(l)).to(ctx.device) for l in layers]
    pointers = (C.c_void_p * n_layers)(*[t.data_ptr() for t in dev])
    with torch.cuda.stream(ctx.stream):
        send = torch.full((info.send_floats,), float("nan"), device=ctx.device)
        recv = torch.full((info.recv_floats,), float("nan"), device=ctx.device)
        piece = torch.empty((W * H, 5), device=ctx.device)
        rgb8 = torch.empty((W * H, 3), dtype=torch.uint8, device=ctx.device)
        full = torch.zeros((W * H, 5), device=ctx.device)
        full8 = torch.zeros((W * H, 3), dtype=torch.uint8, device=ctx.device)
    ctx.join()
    _capi.check(L.avr_pack_layers(ctx._handle, plan, pointers, n_layers, C.c_void_p(send.data_ptr())))
    _capi.check(L.avr_exchange(ctx._handle, plan, comm._handle, C.c_void_p(send.data_ptr()),
                               C.c_void_p(recv.data_ptr())))
    _capi.check(L.avr_fold_plan(ctx._handle, plan, C.c_void_p(recv.data_ptr()),
                                C.c_void_p(piece.data_ptr()), C.c_void_p(rgb8.data_ptr())))
    _capi.check(L.avr_gather(ctx._handle, plan, comm._handle, C.c_void_p(piece.data_ptr()), 20,
                             C.c_void_p(full.data_ptr()), 0))
    _capi.check(L.avr_gather(ctx._handle, plan, comm._handle, C.c_void_p(rgb8.data_ptr()), 3,
                             C.c_void_p(full8.data_ptr()), 0))
    ctx.synchronize()
    assert torch.equal(recv.view(torch.int32), send.view(torch.int32))   # through ncclSend/ncclRecv
    assert np.array_equal(full.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(full8.cpu().numpy().reshape(H, W, 3)[::-1], O.quantize_rgb8(want, W, H))
    L.avr_frame_plan_destroy(plan)
    comm.close()


def _native_worker(rank, world, port, policy, antialiasing, name, out_path, bytes_only=False,
                   contiguous=False):
    """One rank PROCESS of the C++ frame driver on the shared GPU: avr_renderer with the
    cross-process rehearsal communicator (avr_comm_create_shared) -- the plans, offsets and
    ordering of the RCCL flavour, blocks through a shared-memory segment."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from amrvolumerenderer_amd import runtime, scenes
        from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters
        from helpers import device_box
        from test_frame_plan import local_indices, oracle_overlay, painted_scene

        root = int(round(antialiasing ** 0.5))
        spec = scenes.make_amr_scene(32, 2, 8, "smooth")
        cams = [scenes.orbit_camera(3), scenes.default_camera(), scenes.orbit_camera(3)]
        scenes.assign_owners(spec, world, policy)
        owners = [b.owner for b in spec.boxes]
        ctx = runtime.Context(0)
        cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
        meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
        local = [device_box(ctx, cells[i], spec.boxes[i].min_corner, spec.boxes[i].max_corner,
                            spec.boxes[i].level, rank)
                 for i in scenes.local_box_indices(spec, rank)]
        comm = runtime.Comm.shared(name, rank, world, 64 << 20)
        renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range,
                                 rank, world, dist.group.WORLD, comm=comm)
        assert renderer.native is not None
        H = globals()["H"] + (1 if contiguous == 2 else 0)   # 73 rows: a third / half cuts a row
        if contiguous:   # the reference's contiguous pieces
            renderer.native.set_piece_layout(0, 1)
        # pipelined: three frames (two cameras, the first one again) without a host sync between
        # (bytes_only: no float image is asked for, so the RGB8 pieces of a frame travel to rank 0
        # with the NEXT frame's grouped round and the last frame's with the synchronise: every
        # frame's bytes must still land in that frame's own tensor)
        frames = [renderer.render(RenderParameters(W, H, 0.85, antialiasing), cam,
                                  want_image=not bytes_only)
                  for cam in cams]
        renderer.synchronize()
        info = renderer.native.plan_info()
        assert info.piece_layout == (0 if contiguous else 1), (info.piece_layout, info.band_rows)
        assert contiguous or info.band_rows == 8      # the driver's row bands
        if rank == 0:
            flags = []
            for cam, (image, rgb8) in zip(cams, frames):
                _, layers, hints, _ = painted_scene(O, spec, cam, W * root, H * root, 0.85)
                want, _, _ = O.compose_layered(layers, hints, owners,
                                               local_indices(owners, world), world)
                if root > 1:
                    want = O.downsample(want, W, H, root).reshape(-1, 5)
                want = oracle_overlay(O, spec, cells, cam, want, W, H)
                flags.append(bytes_only or
                             np.array_equal(image.cpu().numpy().reshape(-1, 5).view(np.uint32),
                                            want.view(np.uint32)))
                flags.append(np.array_equal(rgb8.cpu().numpy(), O.quantize_rgb8(want, W, H)))
            with open(out_path, "w") as fh:
                fh.write(" ".join(str(int(f)) for f in flags))
        else:
            assert all(image is None and rgb8 is None for image, rgb8 in frames)
        dist.barrier()
        renderer.native.close()
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,policy,antialiasing,bytes_only,contiguous", [
    (3, "level_pairs", 1, False, False), (2, "morton", 4, False, False),
    (4, "round_robin", 1, False, False), (3, "level_pairs", 1, True, False),
    (4, "morton", 1, True, False),
    # the reference's contiguous pieces: over 73 rows they cut through rows (2: the root's own
    # piece then travels through the gathered buffer), over 72 they do not (1: read in place)
    (3, "morton", 1, True, 2), (2, "level_pairs", 1, True, 1)])
def test_native_driver_across_processes(tmp_path, world, policy, antialiasing, bytes_only,
                                        contiguous):
    """The C++ frame driver as N rank PROCESSES on one GPU (what `bench.py --gpus N` and the
    reference's MPI ranks are), wired by the shared-memory rehearsal communicator: row-band pieces,
    exchange layout tightened from the first frame, frames pipelined; rank 0's frames are the
    oracle's N-rank compose bit for bit."""
    out = tmp_path / "result.txt"
    name = f"/avr_test_{os.getpid()}_{world}_{antialiasing}_{int(bytes_only)}_{int(contiguous)}"
    spawn_ranks(_native_worker, world, lambda port: (world, port, policy, antialiasing, name, str(out), bytes_only, contiguous))
    flags = out.read_text().split()
    assert len(flags) == 6 and all(f == "1" for f in flags), flags


def _failfast_worker(rank, world, port, name, mode, out_dir):
    """A rank process of the fail-fast rehearsal (test_ranks_fail_fast_...)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from amrvolumerenderer_amd import _capi, runtime, scenes
        from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters
        from helpers import device_box

        runtime.set_frame_timeout_ms(3000)
        spec = scenes.make_amr_scene(32, 2, 8, "smooth")
        scenes.assign_owners(spec, world, "level_pairs")
        ctx = runtime.Context(0)
        cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
        meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
        local = [device_box(ctx, cells[i], spec.boxes[i].min_corner, spec.boxes[i].max_corner,
                            spec.boxes[i].level, rank)
                 for i in scenes.local_box_indices(spec, rank)]
        comm = runtime.Comm.shared(name, rank, world, 64 << 20)
        renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range,
                                 rank, world, dist.group.WORLD, comm=comm)
        p = RenderParameters(W, H, 0.85, 1)
        good = scenes.default_camera()
        for _ in range(3):   # the ranks agree: frames run
            renderer.render(p, good)
        renderer.synchronize()
        rounds_before = comm.control_rounds()
        begin = time.monotonic()
        error = ""
        try:
            if mode == "different_plan":
                # a new camera for everybody -- but rank 1 is handed another one: its plan
                # describes another exchange
                renderer.render(p, scenes.orbit_camera(5 if rank == 1 else 3))
                renderer.synchronize()
            elif mode == "different_calls":
                # only rank 1 is handed a new camera: it goes to agree on a plan nobody else has,
                # while the others queue the frame of the plan they already agreed on
                renderer.render(p, scenes.orbit_camera(5) if rank == 1 else good)
                renderer.synchronize()
            elif rank != 1:   # "missing_rank": rank 1 simply stops rendering
                for _ in range(6):
                    renderer.render(p, good)
                renderer.synchronize()
        except _capi.AvrError as failure:
            error = str(failure)
        elapsed = time.monotonic() - begin
        failed = renderer.native.failure() or ""
        later = ""
        if error:
            try:   # a failed renderer refuses further frames at once
                renderer.render(p, good)
            except _capi.AvrError as failure:
                later = str(failure)
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as fh:
            fh.write("\n".join([error.replace("\n", " "), f"{elapsed:.3f}", failed.replace("\n", " "),
                                later.replace("\n", " "),
                                str(comm.control_rounds() - rounds_before)]))
        dist.barrier()
        renderer.native.close()
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["different_plan", "different_calls", "missing_rank"])
def test_ranks_fail_fast_instead_of_hanging(tmp_path, mode):
    """The N-rank frame completes or errors (DirectSendBase.cpp:206-220, 277), three rank
    processes over the shared-memory rehearsal communicator:
    * one rank is handed a deliberately different plan (another camera): the agreement check of a
      new plan (avr_frame_plan_agree, one control-plane allgather) makes EVERY rank return an
      error before anything of that frame is queued -- no hang, no deadline needed;
    * only one rank is handed a new camera, so it alone goes to agree on a plan while the others
      start the frame's exchange: the rehearsal communicators notice that the ranks are in different
      collectives and every rank errors at once (over RCCL such rounds never end: the deadline);
    * one rank stops rendering: its peers' waits run into AVR_FRAME_TIMEOUT_MS (3 s here) and
      return an error naming rank, frame, stage and co-run state; the renderer is failed for good."""
    world = 3
    name = f"/avr_failfast_{os.getpid()}_{mode}"
    spawn_ranks(_failfast_worker, world, lambda port: (world, port, name, mode, str(tmp_path)))
    for rank in range(world):
        error, elapsed, failed, later, rounds = (tmp_path / f"rank{rank}.txt").read_text().split("\n")
        if mode == "different_plan":
            assert "frame plan: rank 1's plan differs from rank 0's" in error, (rank, error)
            assert float(elapsed) < 2.0 and failed == "" and int(rounds) == 1
        elif mode == "different_calls":
            assert "the ranks' calls differ" in error, (rank, error)
            assert ("differ: rank 1 is in a control-plane allgather" in error) == (rank != 1), (rank, error)
            assert float(elapsed) < 2.0
        elif rank == 1:
            assert error == ""
        else:
            assert "AVR_FRAME_TIMEOUT_MS" in error and f"rank {rank} of 3" in error, (rank, error)
            assert "stage:" in error and "co-run:" in error
            assert 2.5 < float(elapsed) < 12.0, elapsed
            assert failed == error and later == error


def _lockstep_worker(rank, world, port, name, frames, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import json
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from amrvolumerenderer_amd import runtime, scenes
        from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters
        from helpers import device_box

        spec = scenes.make_amr_scene(32, 2, 8, "smooth")
        scenes.assign_owners(spec, world, "level_pairs")
        ctx = runtime.Context(0)
        cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
        meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
        local = [device_box(ctx, cells[i], spec.boxes[i].min_corner, spec.boxes[i].max_corner,
                            spec.boxes[i].level, rank)
                 for i in scenes.local_box_indices(spec, rank)]
        comm = runtime.Comm.shared(name, rank, world, 64 << 20)
        renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range,
                                 rank, world, dist.group.WORLD, comm=comm)
        native = renderer.native
        native.set_corun_history(frames)
        p = RenderParameters(W, H, 0.85, 1)
        cam = scenes.default_camera()
        first = None
        for f in range(frames):
            if rank == 2 and f in (57, 131):
                # ONE rank's pipeline drains (what a buffer that grows does): only its window is
                # void.  (Not synchronize(): with a frame's gather pending that is a collective.)
                native.set_timing(False)
            out = renderer.render(p, cam)
            first = first or out
        renderer.synchronize()
        last = renderer.render(p, cam)
        renderer.synchronize()
        same = bool(rank != 0 or torch.equal(first[1], last[1]))
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
            json.dump({"history": native.corun_history(), "state": native.corun_state(),
                       "rounds": comm.control_rounds(), "same": same}, fh)
        dist.barrier()
        native.close()
        comm.close()
    finally:
        dist.destroy_process_group()


def test_ranks_search_their_corun_layout_as_one_system(tmp_path):
    """Four rank processes on one GPU (the box admits six processes on the card; eight ranks as
    threads: tests/cxx/adapter_test): the co-run search is coordinated -- every rank holds the same
    candidate in EVERY frame, although one rank's pipeline drains on its own twice, and the windows'
    periods are agreed on over the control plane (one allgather per window).  Scheduling only:
    the frames stay the same."""
    import json
    world, frames = 4, 400
    name = f"/avr_lockstep_{os.getpid()}"
    spawn_ranks(_lockstep_worker, world, lambda port: (world, port, name, frames, str(tmp_path)))
    got = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(world)]
    assert all(g["same"] for g in got)
    assert len(got[0]["history"]) == frames
    for r in range(1, world):
        assert got[r]["history"] == got[0]["history"], f"rank {r} held other candidates than rank 0"
        assert got[r]["state"] == got[0]["state"]
        assert got[r]["rounds"] == got[0]["rounds"]
    assert len(set(got[0]["history"])) >= 5, "the search moved through its candidates"
    windows = got[0]["state"]["timed_windows"]
    # one control-plane round per timed or void window, plus the plan agreement
    assert windows >= 10 and got[0]["rounds"] >= windows + 1


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_bench_multi_rank_flow_on_one_gpu(n_ranks):
    """`python bench.py --gpus N` end to end where one GPU is all there is: the script starts its
    ranks itself (torch.distributed.run, as the driver does), the control plane runs over gloo,
    every rank drives the C++ frame driver, and -- RCCL refusing two ranks on one device -- the
    exchange and the gather go through the shared-memory rehearsal communicator.  Checks the
    N > 1 code of the script, not a speed: rank 0's line, the sample-count identity, and that
    the driver's collectives give the bytes of the torch.distributed frame loop."""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    done = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks),
         "--rehearse-on-one-gpu", "--config", "tiny", "--steps", "4", "--warmup", "1",
         "--check-collectives"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
        text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [l for l in done.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, done.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == n_ranks and line["steps"] == 4 and line["scaling"] == "strong"
    assert line["config"]["frame_driver"].startswith("C++ (avr_renderer")
    assert "shared memory" in line["config"]["frame_driver"]
    assert line["config"]["self_checks"] == {"native_collectives_equal_torch_distributed": True}
    assert line["config"]["exchange"]["layout"].startswith("per-row extents")
    assert line["value"] > 0 and line["cpu_baseline"] is None


def test_bench_ends_with_an_error_when_a_rank_hangs():
    """`bench.py --gpus 3` (rehearsed on one GPU) with one rank that stops rendering and sleeps: its
    peers run into AVR_FRAME_TIMEOUT_MS (4 s here), print which rank failed where and how its co-run
    search stood, and the launch ends non-zero within seconds instead of sitting to the launcher's
    own limit."""
    import subprocess
    import time
    env = dict(os.environ, AVR_FRAME_TIMEOUT_MS="4000")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    begin = time.monotonic()
    done = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--rehearse-on-one-gpu",
         "--config", "tiny", "--steps", "4", "--warmup", "2", "--debug-stall-rank", "1"],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    elapsed = time.monotonic() - begin
    assert done.returncode != 0
    assert not [l for l in done.stdout.splitlines() if l.startswith('{"metric"')]
    assert "rank 1 stalls on purpose" in done.stderr
    assert "failed: AvrError" in done.stderr and "AVR_FRAME_TIMEOUT_MS" in done.stderr
    assert "renderer:" in done.stderr and "corun_state:" in done.stderr and "stage:" in done.stderr
    assert elapsed < 120, elapsed


@pytest.mark.gpu
@pytest.mark.parametrize("percent", [100, 25, 0])
def test_one_rank_of_four_played_through_rccl(percent):
    """avr_comm_create_solo_rccl (timing studies, tools/rank_share.py --through-rccl): a rank of
    four played alone with its grouped send / receive round and its gather going through a one-rank
    RCCL communicator to the rank itself.  The frames are not images; they must run, pipelined, and
    plan what the plain solo communicator plans."""
    import torch
    from amrvolumerenderer_amd import runtime, scenes
    from amrvolumerenderer_amd.renderer import build_scene_on_device
    ctx = runtime.Context(0)
    n_ranks = 4
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    scenes.assign_owners(spec, n_ranks, "level_pairs")
    cams = [scenes.orbit_camera(v, 12) for v in range(12)]
    plans = {}
    for rank, flavour in ((1, "solo"), (1, "rccl"), (0, "rccl")):
        all_boxes, local = build_scene_on_device(ctx, spec, rank)
        merged, mine = [], iter(local)
        for b in all_boxes:
            merged.append(next(mine) if b.owner == rank else b)
        comm = (runtime.Comm.solo(rank, n_ranks) if flavour == "solo"
                else runtime.Comm.solo_rccl(0, rank, n_ranks, percent))
        r = runtime.NativeRenderer(0, merged, spec.transform, spec.bounds, spec.scalar_range, rank,
                                   n_ranks, comm)
        seen = []
        for i in range(36):   # back to back
            r.render(160, 120, 0.9, 1, cams[i % len(cams)], draw_bounds=False, want_image=(i % 5 == 0))
            info = r.plan_info()
            seen.append((info.send_floats, info.recv_floats, info.n_runs_total))
        r.synchronize()
        torch.cuda.synchronize()
        plans[(rank, flavour)] = seen
        r.close()
    assert plans[(1, "solo")] == plans[(1, "rccl")]
    assert all(s > 0 for s, _, _ in plans[(0, "rccl")])
    with pytest.raises(Exception):
        runtime.Comm.solo_rccl(0, 0, 4, 101)
