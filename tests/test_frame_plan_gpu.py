"""GPU parity of the plan path: avr_render_plan (classify + march into the sparse send buffer)
and avr_fold_plan, for N simulated ranks on one GPU, against the oracle's per-box layers and
layered DirectSend compose.  Also the full FrameRenderer frame (1 rank) incl. antialiasing."""
import numpy as np
import pytest
import torch

from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.compositor import FramePlan
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters
from amrvolumerenderer_amd.types import CameraParameters, make_params

import plan_helpers as PH
from helpers import assert_bit_equal, device_box
from test_frame_plan import local_indices, oracle_overlay, painted_scene

import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_ranks,policy,size,transparency,scene_shape", [
    (1, "morton", (96, 64), 0.9, (32, 2, 8)), (2, "morton", (75, 43), 0.8, (32, 2, 8)),
    (4, "round_robin", (75, 43), 0.8, (32, 2, 8)), (8, "morton", (128, 128), 0.0, (32, 2, 8)),
    (3, "block", (50, 70), 0.97, (32, 2, 8)),
    # 960 boxes dealt round-robin: several hundred runs (the fold scans its run list in chunks
    # of 256) and many (tile, run) march items per rank
    (6, "round_robin", (64, 48), 0.9, (32, 2, 4))])
def test_render_and_fold_plan(O, ctx, n_ranks, policy, size, transparency, scene_shape):
    W, H = size
    spec = scenes.make_amr_scene(*scene_shape, "smooth")
    cam = scenes.default_camera()
    cells, layers, hints, ref = painted_scene(O, spec, cam, W, H, transparency)
    scenes.assign_owners(spec, n_ranks, policy)
    owners = [b.owner for b in spec.boxes]
    want, _, _ = O.compose_layered(layers, hints, owners, local_indices(owners, n_ranks), n_ranks)
    meta = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    params = make_params(W, H, spec.scalar_range, transparency, ref, spec.bounds)

    plans, sends = [], []
    for r in range(n_ranks):
        plan = FramePlan(meta, params, cam, r, n_ranks)
        local = [device_box(ctx, cells[i], spec.boxes[i].min_corner, spec.boxes[i].max_corner,
                            spec.boxes[i].level, r)
                 for i in scenes.local_box_indices(spec, r)]
        scene = ctx.create_scene(local, spec.transform)
        out = torch.full((max(plan.send_floats, 1),), float("nan"), device=ctx.device)
        send = scene.render_plan(plan, out=out)
        ctx.synchronize()
        send = send.cpu().numpy()
        # the march must have written exactly the layout documented in include/avr_hip.h
        run_layers = PH.oracle_run_layers(O, layers, plan)
        expect = PH.pack_send_buffer(plan, run_layers)
        written = ~np.isnan(expect)
        assert_bit_equal(send[written], expect[written], f"send buffer of rank {r}")
        assert np.all(np.isnan(send[~written][: max(plan.send_floats - written.sum(), 0)]))
        plans.append(plan)
        sends.append(send)

    if scene_shape[2] == 4:
        assert plans[0].n_runs_total > 256
    recvs = PH.route(plans, sends)
    got = np.zeros((W * H, 5), np.float32)
    got8 = np.zeros((W * H, 3), np.uint8)
    for plan, recv in zip(plans, recvs):
        dev = torch.from_numpy(np.ascontiguousarray(recv)).to(ctx.device)
        if dev.numel() == 0:
            dev = torch.zeros(1, device=ctx.device)
        piece, rgb8 = ctx.fold_plan(plan, dev, want_rgb8=True)
        ctx.synchronize()
        got[plan.piece_begin:plan.piece_end] = piece.cpu().numpy()
        got8[plan.piece_begin:plan.piece_end] = rgb8.cpu().numpy()
        # avr_fold_plan_own: the rank's block for itself read from its send buffer instead (the
        # receive buffer holds poison there, as after avr_exchange_peers)
        poisoned = np.array(recv, dtype=np.float32, copy=True)
        begin = sum(plan.recv_splits[:plan.rank])
        poisoned[begin:begin + plan.recv_splits[plan.rank]] = np.nan
        dev2 = torch.from_numpy(np.ascontiguousarray(poisoned)).to(ctx.device)
        own = torch.from_numpy(np.ascontiguousarray(sends[plan.rank])).to(ctx.device)
        if dev2.numel() == 0:
            dev2 = torch.zeros(1, device=ctx.device)
        if own.numel() == 0:
            own = torch.zeros(1, device=ctx.device)
        piece2, rgb82 = ctx.fold_plan(plan, dev2, want_rgb8=True, own_send=own)
        ctx.synchronize()
        assert torch.equal(piece2.view(torch.int32), piece.view(torch.int32)), plan.rank
        assert torch.equal(rgb82, rgb8), plan.rank
    assert_bit_equal(got, want, f"{n_ranks} ranks {policy}")
    assert np.array_equal(got8, O.quantize_rgb8(want, W, H)[::-1].reshape(-1, 3))
    if n_ranks == 1:
        # avr_fold_plan_image: the same bytes as the output file's rows (top-down), in one pass;
        # and it is for one rank only
        dev = torch.from_numpy(np.ascontiguousarray(recvs[0])).to(ctx.device)
        image8 = ctx.fold_plan_image(plans[0], dev)
        ctx.synchronize()
        assert np.array_equal(image8.cpu().numpy(), O.quantize_rgb8(want, W, H))
    else:
        with pytest.raises(Exception):
            ctx.fold_plan_image(plans[0], torch.zeros(max(plans[0].recv_floats, 1),
                                                     device=ctx.device))


@pytest.mark.parametrize("antialiasing,draw_bounds", [(1, False), (4, False), (1, True), (4, True)])
def test_frame_renderer_single_rank(O, ctx, antialiasing, draw_bounds):
    root = int(round(antialiasing ** 0.5))
    W, H = 64, 40
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    cam = scenes.default_camera()
    cells, layers, hints, ref = painted_scene(O, spec, cam, W * root, H * root, 0.5)
    owners = [0] * len(cells)
    want, _, _ = O.compose_layered(layers, hints, owners, np.arange(len(cells)), 1)
    if root > 1:
        want = O.downsample(want, W, H, root).reshape(-1, 5)
    if draw_bounds:  # VolumeRenderer.cpp:1311-1314: tight bounds, radius scale 1, after the AA
        want = oracle_overlay(O, spec, cells, cam, want, W, H)
    want8 = O.quantize_rgb8(want, W, H)

    meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
    local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level) for c, m in
             zip(cells, spec.boxes)]
    renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range)
    assert np.float32(renderer.reference_sample_distance) == np.float32(ref)
    image, rgb8 = renderer.render(
        RenderParameters(W, H, 0.5, antialiasing, draw_bounds=draw_bounds), cam, want_image=True)
    renderer.synchronize()
    assert_bit_equal(image.cpu().numpy(), want, "frame image")
    assert np.array_equal(rgb8.cpu().numpy(), want8)


def test_render_parameter_validation(ctx):
    from amrvolumerenderer_amd.renderer import validate_render_parameters
    with pytest.raises(ValueError):
        validate_render_parameters(RenderParameters(64, 64, 0.0, 3))  # not a perfect square
    with pytest.raises(ValueError):
        validate_render_parameters(RenderParameters(0, 64))
    with pytest.raises(ValueError):
        validate_render_parameters(RenderParameters(64, 64, 1.5))
    assert validate_render_parameters(RenderParameters(64, 64, 0.2, 9)) == 3


def test_pipelined_frames_reuse_buffers_safely(O, ctx):
    """Frames are pipelined over two streams with double-buffered send layouts: a burst of
    frames with different cameras / sizes must each equal its own single-frame result."""
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
    local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level) for c, m in
             zip(cells, spec.boxes)]
    renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range)
    jobs = [(scenes.orbit_camera(v), RenderParameters(96 + 16 * (v % 3), 64, 0.3 * (v % 3), 1))
            for v in range(7)]
    burst = [renderer.render(p, cam, want_image=True) for cam, p in jobs]  # no sync in between
    renderer.synchronize()
    burst = [(img.cpu().numpy(), rgb.cpu().numpy()) for img, rgb in burst]
    for (cam, p), (img, rgb) in zip(jobs, burst):
        one_img, one_rgb = renderer.render(p, cam, want_image=True)
        renderer.synchronize()
        assert np.array_equal(img.view(np.uint32), one_img.cpu().numpy().view(np.uint32))
        assert np.array_equal(rgb, one_rgb.cpu().numpy())


def test_cells_written_on_the_default_stream_are_seen_without_a_host_sync(ctx):
    """The caller fills the cells (and zeroes the sample counter) on torch's default stream --
    handle 0, which the C ABI's input_stream cannot name -- and renders at once: the frame must
    be ordered after that work on the driver's non-blocking streams."""
    from amrvolumerenderer_amd import runtime as rt
    n = 96
    lo, hi = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    from amrvolumerenderer_amd.types import AmrBox, ScalarTransform, VolumeBounds
    bounds = VolumeBounds(lo, hi)
    cells = torch.zeros((n, n, n), dtype=torch.float64, device=ctx.device)
    box = AmrBox(lo, hi, cells)
    tr = ScalarTransform(normalize_to_unit_range=True)
    r = rt.NativeRenderer(0, [box], tr, bounds)
    cam = scenes.default_camera()
    torch.cuda.synchronize()
    counter = torch.ones(1, dtype=torch.int64, device=ctx.device)
    results = []
    for value in (0.25, 0.75, 0.5):
        # a long queue on the default stream, the cell fill at its very end
        scratch = torch.zeros(1 << 26, device=ctx.device)
        for _ in range(20):
            scratch.add_(1.0)
        cells.fill_(value)
        counter.zero_()
        img, rgb = r.render(128, 128, 0.5, 1, cam, draw_bounds=False, samples=counter,
                            want_image=True)
        # the caller's side of the contract: its next writes (and its read of the counter) are
        # ordered after the frame's classify pass and march
        default = torch.cuda.current_stream(ctx.device)
        default.wait_stream(r.streams[0])
        default.wait_stream(r.streams[1])
        results.append((img, rgb, counter.clone()))
    r.synchronize()
    torch.cuda.synchronize()
    fresh = []
    for value in (0.25, 0.75, 0.5):
        cells.fill_(value)
        counter.zero_()
        torch.cuda.synchronize()
        img, rgb = r.render(128, 128, 0.5, 1, cam, draw_bounds=False, samples=counter,
                            want_image=True)
        r.synchronize()
        torch.cuda.synchronize()
        fresh.append((img, rgb, counter.clone()))
    assert not torch.equal(fresh[0][0], fresh[1][0])
    for (img, rgb, count), (want_img, want_rgb, want_count) in zip(results, fresh):
        assert torch.equal(img.view(torch.int32), want_img.view(torch.int32))
        assert torch.equal(rgb, want_rgb)
        assert int(want_count.item()) > 0 and int(count.item()) == int(want_count.item())
    r.close()


def test_two_phase_plan_calls_equal_the_fused_call(ctx):
    """avr_classify_plan + avr_march_plan (either classified slot, classify on another
    context's stream) produce the same send buffer as avr_render_plan."""
    from amrvolumerenderer_amd import runtime as rt
    spec = scenes.make_amr_scene(32, 2, 8, "noise")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
    local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level) for c, m in
             zip(cells, spec.boxes)]
    scene = ctx.create_scene(local, spec.transform)
    ref = rt.reference_sample_distance(meta, spec.bounds.min_corner, spec.bounds.max_corner)
    params = make_params(80, 56, spec.scalar_range, 0.7, ref, spec.bounds)
    plan = FramePlan(meta, params, scenes.default_camera(), 0, 1)
    want = scene.render_plan(plan).clone()
    ctx.synchronize()
    other = rt.Context(0)
    for slot in (0, 1, 0):
        plan2 = FramePlan(meta, params, scenes.default_camera(), 0, 1)
        scene.classify_plan(other, plan2, slot)
        done = torch.cuda.Event()
        done.record(other.stream)
        ctx.stream.wait_event(done)
        out = torch.zeros_like(want)
        scene.march_plan(ctx, plan2, slot, out)
        ctx.synchronize()
        assert torch.equal(out.view(torch.int32), want.view(torch.int32))


def test_classification_cache_is_exact_and_invalidates(O, ctx):
    """avr_scene_set_classification_cache: cached frames equal uncached ones bit for bit, a
    changed scalar range re-classifies, and cells changed in place need invalidate()."""
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
    local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level) for c, m in
             zip(cells, spec.boxes)]
    p = RenderParameters(96, 64, 0.7, 1)

    def frames(renderer, cams):
        out = []
        for cam in cams:
            image, rgb8 = renderer.render(p, cam, want_image=True)
            renderer.synchronize()
            out.append((image.cpu().numpy().view(np.uint32).copy(), rgb8.cpu().numpy().copy()))
        return out

    cams = [scenes.orbit_camera(v) for v in range(5)]
    plain = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range)
    cached = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range,
                           cache_classification=True)
    want = frames(plain, cams)
    got = frames(cached, cams)
    for (a, a8), (b, b8) in zip(want, got):
        assert np.array_equal(a, b) and np.array_equal(a8, b8)
    # another scalar range changes the table indices: the key differs, no stale re-use
    wide = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, (-0.5, 1.5))
    cached.scalar_range = (-0.5, 1.5)
    for (a, a8), (b, b8) in zip(frames(wide, cams[:2]), frames(cached, cams[:2])):
        assert np.array_equal(a, b) and np.array_equal(a8, b8)
    cached.scalar_range = spec.scalar_range
    # cells changed in place: stale until invalidate() (three frames: every one of the driver's
    # AVR_CLASSIFIED_SLOTS classified volumes then holds the original cells under the current key)
    before = frames(cached, cams[:3])
    local[0].values.mul_(0.25)
    ctx.synchronize()
    torch.cuda.synchronize()
    stale = frames(cached, cams[:3])
    assert all(np.array_equal(a, b) for (a, _), (b, _) in zip(before, stale))
    cached.invalidate()
    fresh = frames(cached, cams[:3])
    truth = frames(FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range),
                   cams[:3])
    assert all(np.array_equal(a, b) for (a, _), (b, _) in zip(truth, fresh))
    assert not all(np.array_equal(a, b) for (a, _), (b, _) in zip(before, fresh))


@pytest.mark.parametrize("native", [True, False])
def test_growing_send_buffers_under_allocator_churn(O, ctx, native):
    """Frames whose sparse send layout grows from frame to frame (larger images), rendered back
    to back without synchronising while the caller allocates and frees device memory on its own
    stream in between: the driver's buffers are replaced while earlier frames may still be in
    flight on the other streams, and every frame must still equal a fresh renderer's."""
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
    local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level) for c, m in
             zip(cells, spec.boxes)]
    sizes = [(40, 32), (72, 48), (72, 48), (130, 96), (200, 150), (200, 150), (333, 240)]
    cams = [scenes.orbit_camera(v) for v in range(len(sizes))]
    renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range,
                             native=native)
    junk, frames = [], []
    for (w, h), cam in zip(sizes, cams):
        frames.append(renderer.render(RenderParameters(w, h, 0.8, 1), cam, want_image=True))
        junk.append(torch.full((1 << 20,), 7.0, device=ctx.device))   # churn on the caller's stream
        if len(junk) > 2:
            junk.pop(0)
    renderer.synchronize()
    torch.cuda.synchronize()
    for (w, h), cam, (image, rgb8) in zip(sizes, cams, frames):
        fresh = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range,
                              native=native)
        want_image, want_rgb8 = fresh.render(RenderParameters(w, h, 0.8, 1), cam, want_image=True)
        fresh.synchronize()
        assert torch.equal(image.view(torch.int32), want_image.view(torch.int32)), (w, h)
        assert torch.equal(rgb8, want_rgb8), (w, h)


def test_corun_tuning_never_changes_results(O, ctx):
    """avr_renderer_set_classify_share / avr_renderer_set_overlap: where the classify pass runs
    (before or beside the march) and the LDS reserve that caps it there are scheduling only --
    frames rendered back to back while the driver tries its candidates, and with either choice
    fixed by the caller, are bit-identical to a fresh renderer's; the driver times windows of
    frames once the pipeline is running and reports what it used."""
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
    local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level) for c, m in
             zip(cells, spec.boxes)]
    p = RenderParameters(160, 120, 0.9, 1)
    cams = [scenes.orbit_camera(v) for v in range(3)]

    def run(share, overlap, frames):
        renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range)
        assert renderer.native is not None
        renderer.native.set_classify_share(share)
        renderer.native.set_overlap(overlap)
        out = []
        for f in range(frames):   # back to back: no synchronisation between frames
            out.append(renderer.render(p, cams[f % len(cams)], want_image=True))
        renderer.synchronize()
        torch.cuda.synchronize()
        return renderer, out

    _, want = run(0, 0, 3)
    # (2: the paired layout -- every frame's two kernels on one stream, the frames alternating
    # between two streams; with the reserve fixed, then searched; with host-side back-pressure)
    # (frame counts: a window of this scene's 0.1 ms frames is 40 frames after 40 let pass, the
    # finalists' windows 160: a whole search is 28 x 80 + 3 x 200 = 2840 frames, back to back
    # against one fixed reserve 2 x 80 + 3 x 200 = 760)
    for share, overlap, frames in ((61440, 1, 12), (-1, -1, 3600), (4096, -1, 1100), (8192, 2, 13),
                                   (-1, 2, 600)):
        renderer, got = run(share, overlap, frames)
        state = renderer.native.corun_state()
        if overlap == 1:
            assert state == {"classify": "beside the march", "lds_reserve_bytes": share,
                             "settled": True, "timed_windows": 0}
        elif overlap == 2 and share >= 0:
            assert state == {"classify": "before its march, the frames alternating between two "
                                         "streams", "lds_reserve_bytes": share, "settled": True,
                             "timed_windows": 0}
        elif overlap == 2:
            assert state["classify"].startswith("before its march, the frames alternating")
            assert state["timed_windows"] >= 4
        elif share < 0:
            # (a whole search is 29 windows; the held candidate's windows are twice as long as the
            # search's and a suspected drift is checked over four times as many frames, so the
            # 0.1 ms frames of this small scene no longer start a new search by themselves)
            assert state["timed_windows"] >= 29
            assert state["settled"]
            assert 0 <= state["lds_reserve_bytes"] <= 57344
        else:
            assert state["timed_windows"] >= 2 and state["settled"]
            assert state["lds_reserve_bytes"] in (0, share)
        for f, (image, rgb8) in enumerate(got):
            w_image, w_rgb8 = want[f % len(cams)]
            assert torch.equal(image.view(torch.int32), w_image.view(torch.int32)), (share, f)
            assert torch.equal(rgb8, w_rgb8), (share, f)
    with pytest.raises(Exception):
        renderer.native.set_classify_share(61441)
    # bursts too short for anything to be timed (a synchronisation every five frames): the driver
    # keeps the one-rank default -- side by side with the reserve its balancing starts from
    # (CoRunTuner::kBalanceSeed, 24 KiB) -- and the frames stay exact
    renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range)
    for burst in range(30):
        got = [renderer.render(p, cams[f % len(cams)], want_image=True) for f in range(5)]
        renderer.synchronize()
        torch.cuda.synchronize()
        for f, (image, rgb8) in enumerate(got):
            assert torch.equal(image.view(torch.int32), want[f % len(cams)][0].view(torch.int32))
            assert torch.equal(rgb8, want[f % len(cams)][1])
    state = renderer.native.corun_state()
    assert state == {"classify": "beside the march", "lds_reserve_bytes": 24576, "settled": False,
                     "timed_windows": 0}


def test_timing_toggled_under_the_paired_layout(ctx):
    """What bench.py does around its timed region, under the layout ranks of four and eight settle
    on: timing on, frames, timing off (which destroys the frames' timing events), frames again.
    The paired layout orders every classify pass after the previous frame's: that hand-over must
    not go through an event the toggle destroyed (round 3 advisor finding)."""
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
    local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level) for c, m in
             zip(cells, spec.boxes)]
    p = RenderParameters(160, 120, 0.9, 1)
    cam = scenes.default_camera()
    renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range)
    native = renderer.native
    assert native is not None
    native.set_overlap(0)
    want = renderer.render(p, cam, want_image=True)
    renderer.synchronize()
    native.set_overlap(2)
    native.set_classify_share(8192)
    got = []
    for cycle in range(3):
        native.set_timing(True)
        got += [renderer.render(p, cam, want_image=True) for _ in range(7)]
        renderer.synchronize()
        _, _, busy_ms, frames = native.timings()
        assert frames == 7 and busy_ms > 0.0
        native.set_timing(False)
        got += [renderer.render(p, cam, want_image=True) for _ in range(5)]   # no drain before them
    renderer.synchronize()
    torch.cuda.synchronize()
    assert native.corun_state()["classify"].startswith("before its march, the frames alternating")
    for image, rgb8 in got:
        assert torch.equal(image.view(torch.int32), want[0].view(torch.int32))
        assert torch.equal(rgb8, want[1])


def test_a_stream_that_does_not_move_ends_in_an_error_not_a_hang():
    """avr_set_frame_timeout_ms on a REAL stream: the compositing stream is kept busy for 1.2 s by
    a bounded stall kernel (it always ends by itself) while the deadline is 0.25 s -- the
    synchronise returns AVR_ERR_RUNTIME naming that stream, the rank, the frame, the stage and the
    co-run state; the renderer is failed for good and its destruction does not wait either.  In a
    child process, so that the leaked renderer never meets the other tests."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import torch
        from amrvolumerenderer_amd import _capi, runtime, scenes
        from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters
        from helpers import device_box
        ctx = runtime.Context(0)
        spec = scenes.make_amr_scene(32, 1, 16, "smooth")
        cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
        meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
        local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level) for c, m in zip(cells, spec.boxes)]
        renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range)
        native = renderer.native
        p, cam = RenderParameters(96, 64, 0.9, 1), scenes.default_camera()
        for _ in range(4):          # every rotating buffer has its size
            renderer.render(p, cam)
        renderer.synchronize()
        runtime.set_frame_timeout_ms(250)
        _capi.check(_capi.lib().avr_debug_stall_stream(native.streams[2].cuda_stream, 1200))
        t0 = time.monotonic()
        try:
            renderer.render(p, cam)
            renderer.synchronize()
            print("NO ERROR")
        except _capi.AvrError as error:
            print("ERROR", round(time.monotonic() - t0, 3), str(error))
        try:
            renderer.render(p, cam)
            print("NO ERROR")
        except _capi.AvrError as error:
            print("LATER", str(error) == native.failure())
        t0 = time.monotonic()
        native.close()
        print("CLOSED", round(time.monotonic() - t0, 3))
        time.sleep(1.5)            # the stall kernel ends by itself; then the process may go
        torch.cuda.synchronize()
        print("DRAINED")
    """) % (ROOT, os.path.join(ROOT, "tests"))
    done = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = done.stdout.strip().splitlines()
    assert lines[0].startswith("ERROR ") and "the compositing stream" in lines[0], lines
    assert "AVR_FRAME_TIMEOUT_MS" in lines[0] and "rank 0 of 1" in lines[0] and "co-run:" in lines[0]
    assert 0.2 < float(lines[0].split()[1]) < 1.0, lines[0]
    assert lines[1] == "LATER True" and lines[2].startswith("CLOSED") and lines[3] == "DRAINED"
    assert float(lines[2].split()[1]) < 0.2, lines[2]


def test_plans_made_ahead_on_another_thread_change_nothing(ctx):
    """avr_renderer_prepare / runtime.PlanAhead: the frame plans of a camera path made one (and
    several) frames ahead on a helper thread, while the frames are queued -- more cameras than the
    driver keeps plans (32), so plans are evicted under the frames' feet.  One rank: every frame
    equals the one a renderer planning on its own thread delivers.  A rank of four (played alone:
    its frames hold nothing defined, its plans do): every frame's plan, tightened exchange layout
    included, is the one made without the helper."""
    from amrvolumerenderer_amd.renderer import build_scene_on_device
    cams = [scenes.orbit_camera(v, 90) for v in range(90)]
    args = (160, 120, 0.9, 1)
    kw = dict(use_visibility_graph=True, draw_bounds=False)

    def frames(n_ranks, lookahead):
        spec = scenes.make_amr_scene(32, 2, 8, "smooth")
        scenes.assign_owners(spec, n_ranks, "level_pairs")
        all_boxes, local = build_scene_on_device(ctx, spec, 0)
        merged, mine = [], iter(local)
        for b in all_boxes:
            merged.append(next(mine) if b.owner == 0 else b)
        r = runtime.NativeRenderer(0, merged, spec.transform, spec.bounds, spec.scalar_range, 0,
                                   n_ranks, runtime.Comm.solo(0, n_ranks) if n_ranks > 1 else None)
        ahead = runtime.PlanAhead(r) if lookahead else None
        out, plans = [], []
        for i, cam in enumerate(cams):
            if ahead is not None:   # the first frame asks for the next `lookahead`, later ones for one more
                for j in ([i + lookahead] if i else range(1, lookahead + 1)):
                    if j < len(cams):
                        ahead.submit(*args, cams[j], **kw)
            out.append(r.render(*args, cam, want_image=True, **kw))
            info = r.plan_info()
            plans.append((info.n_runs_total, info.n_local_runs, info.send_floats, info.recv_floats,
                          info.piece_begin, info.piece_end))
        if ahead is not None:
            ahead.close()
        r.synchronize()
        torch.cuda.synchronize()
        out = [(img.clone(), rgb.clone()) for img, rgb in out]
        # preparing what is already there, and on the frames' own thread, is fine too
        r.prepare(*args, cams[-1], **kw)
        r.prepare(*args, cams[0], **kw)
        with pytest.raises(Exception):
            r.prepare(0, 120, 0.9, 1, cams[0], **kw)
        r.close()
        return out, plans

    want, want_plans = frames(1, 0)
    assert len({p for p in want_plans}) >= 1 and bool((want[0][0][..., 3] > 0).any())
    for lookahead in (1, 3):
        got, got_plans = frames(1, lookahead)
        assert got_plans == want_plans
        for f, ((img, rgb), (w_img, w_rgb)) in enumerate(zip(got, want)):
            assert torch.equal(img.view(torch.int32), w_img.view(torch.int32)), (lookahead, f)
            assert torch.equal(rgb, w_rgb), (lookahead, f)
    _, want_plans = frames(4, 0)
    assert len(set(want_plans)) > 10 and all(p[2] > 0 for p in want_plans)
    for lookahead in (1, 3):
        assert frames(4, lookahead)[1] == want_plans, lookahead


def _simulated_frame(O, ctx, spec, cam, W, H, transparency, n_ranks, tighten, cells=None, bands=0):
    """The N-rank frame with the ranks played one after the other on this GPU: per rank its frame
    plan (tightened or not), classify + march into the send buffer, the all-to-all by hand, the
    fold.  Returns (image [W*H, 5], floats sent by all ranks, non-empty pixels the march found
    outside a tightened row span)."""
    from test_frame_plan import painted_scene
    import plan_helpers as PH
    if cells is None:
        cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    meta = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    oboxes = [O.make_box(c, m.min_corner, m.max_corner) for c, m in zip(cells, spec.boxes)]
    ref = O.reference_sample_distance(oboxes, spec.bounds.min_corner, spec.bounds.max_corner)
    params = make_params(W, H, spec.scalar_range, transparency, ref, spec.bounds)
    counters = torch.zeros(5, dtype=torch.int64, device=ctx.device)
    samples = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    ctx.set_march_counters(counters)
    plans, sends = [], []
    try:
        for r in range(n_ranks):
            # (bands: the frame driver's row-band pieces instead of the reference's ranges)
            plan = FramePlan(meta, params, cam, r, n_ranks, piece_layout=1 if bands else 0,
                             band_rows=bands or 1)
            if tighten:
                plan.tighten()
            local = [device_box(ctx, cells[i], spec.boxes[i].min_corner, spec.boxes[i].max_corner,
                                spec.boxes[i].level) for i in scenes.local_box_indices(spec, r)]
            scene = ctx.create_scene(local, spec.transform)
            send = scene.render_plan(plan, samples=samples)
            ctx.synchronize()
            plans.append(plan)
            sends.append(send[:plan.send_floats].cpu().numpy())
    finally:
        ctx.set_march_counters(None)
    image = np.zeros((W * H, 5), np.float32)
    for plan, recv in zip(plans, PH.route(plans, sends)):
        dev = torch.from_numpy(np.ascontiguousarray(recv)).to(ctx.device)
        if dev.numel() == 0:
            dev = torch.zeros(1, device=ctx.device)
        piece, _ = ctx.fold_plan(plan, dev, want_rgb8=True)
        ctx.synchronize()
        image[PH.piece_pixels(plan)] = piece.cpu().numpy()
    return image, sum(p.send_floats for p in plans), int(counters[4].item())


@pytest.mark.parametrize("n_ranks,policy", [(2, "morton"), (3, "round_robin"), (5, "morton"),
                                            (8, "round_robin")])
def test_tightened_exchange_layout_is_exact(O, ctx, n_ranks, policy):
    """avr_frame_plan_tighten: per-row extents instead of the runs' rectangles.  For views from
    outside, from inside the domain (boxes reach behind the eye: those keep their rectangle) and
    grazing along a face, the frame of N simulated ranks is the oracle's N-rank compose bit for
    bit, the march finds no non-empty pixel outside a span, and fewer floats travel."""
    from test_frame_plan import local_indices, painted_scene
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    scenes.assign_owners(spec, n_ranks, policy)
    owners = [b.owner for b in spec.boxes]
    W, H, transparency = 131, 97, 0.85
    cams = [scenes.default_camera(), scenes.orbit_camera(5),
            CameraParameters((0.45, 0.55, 0.5), (0.9, 0.4, 0.1), (0.0, 1.0, 0.0), 70.0, 0.05, 20.0),
            CameraParameters((0.5, 1.0, 2.6), (0.5, 1.0, 0.0), (0.0, 1.0, 0.0), 35.0, 0.1, 20.0)]
    saved = []
    try:
        for cam in cams:
            cells, layers, hints, _ = painted_scene(O, spec, cam, W, H, transparency)
            want, _, _ = O.compose_layered(layers, hints, owners, local_indices(owners, n_ranks),
                                           n_ranks)
            loose, loose_floats, _ = _simulated_frame(O, ctx, spec, cam, W, H, transparency,
                                                      n_ranks, False, cells)
            tight, tight_floats, dropped = _simulated_frame(O, ctx, spec, cam, W, H, transparency,
                                                            n_ranks, True, cells)
            assert_bit_equal(loose, want, "rectangular layout")
            assert_bit_equal(tight, want, "tightened layout")
            assert dropped == 0
            assert tight_floats <= loose_floats
            # the frame driver's pieces: bands of rows dealt round-robin (97 rows: the last cycle
            # is partial, and with 8 ranks x 16 rows some pieces have no rows at all)
            for bands in (4, 16):
                banded, banded_floats, dropped = _simulated_frame(
                    O, ctx, spec, cam, W, H, transparency, n_ranks, bands == 4, cells, bands=bands)
                assert_bit_equal(banded, want, f"row bands of {bands}")
                assert dropped == 0
            saved.append(1.0 - tight_floats / loose_floats)
        assert saved[0] > 0.1 and saved[1] > 0.1   # the oblique views from outside
    finally:
        scenes.assign_owners(spec, 1, "morton")


def test_tightened_layout_random_views(O, ctx):
    """Random cameras -- far, close with strong perspective, eyes inside boxes, views nearly along
    an axis -- for a 3-rank frame: the tightened layout gives the rectangular layout's image bit
    for bit and the march never finds a non-empty pixel outside a row span."""
    rng = np.random.default_rng(2026)
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    scenes.assign_owners(spec, 3, "round_robin")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    W, H = 113, 79
    saved = 0.0
    try:
        for case in range(24):
            kind = case % 4
            centre = np.array([0.5, 0.5, 0.5])
            if kind == 0:      # from outside, anywhere on a sphere
                d = rng.normal(size=3)
                eye = centre + d / np.linalg.norm(d) * rng.uniform(1.2, 4.0)
                look, fov = centre + rng.uniform(-0.2, 0.2, size=3), rng.uniform(20.0, 60.0)
            elif kind == 1:    # close to a face: strong perspective, wide angle
                d = rng.normal(size=3)
                eye = centre + d / np.linalg.norm(d) * rng.uniform(0.75, 1.0)
                look, fov = centre, rng.uniform(70.0, 110.0)
            elif kind == 2:    # inside the domain: boxes reach behind the eye
                eye = rng.uniform(0.1, 0.9, size=3)
                look, fov = eye + rng.normal(size=3), rng.uniform(40.0, 100.0)
            else:              # nearly along an axis
                axis = int(rng.integers(0, 3))
                eye = centre.copy()
                eye[axis] += rng.choice([-1.0, 1.0]) * rng.uniform(1.5, 3.0)
                eye += rng.uniform(-1e-3, 1e-3, size=3)
                look, fov = centre, rng.uniform(25.0, 50.0)
            forward = look - eye
            up = (0.0, 1.0, 0.0) if abs(forward[1]) < 0.9 * np.linalg.norm(forward) else (1.0, 0.0, 0.0)
            cam = CameraParameters(tuple(float(v) for v in eye), tuple(float(v) for v in look), up,
                                   float(fov), 0.05, 30.0)
            loose, loose_floats, _ = _simulated_frame(O, ctx, spec, cam, W, H, 0.9, 3, False, cells)
            tight, tight_floats, dropped = _simulated_frame(O, ctx, spec, cam, W, H, 0.9, 3, True,
                                                            cells)
            assert_bit_equal(tight, loose, f"view {case}")
            assert dropped == 0, (case, dropped)
            assert tight_floats <= loose_floats
            saved += 1.0 - tight_floats / max(loose_floats, 1)
        assert saved / 24 > 0.05
    finally:
        scenes.assign_owners(spec, 1, "morton")
