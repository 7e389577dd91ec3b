"""GPU parity of a frame classified and marched in depth-ordered chunks (avr_classify_plan_chunked /
avr_march_plan_chunked, avr_renderer_set_frame_chunks): the run's left fold
(DirectSendBase.cpp:413-426) is cut between launches, never re-associated, so every send buffer,
sample count, image and byte must equal the single-launch frame's -- which the other GPU tests
compare with the oracle."""
import numpy as np
import pytest
import torch

from amrvolumerenderer_amd import scenes
from amrvolumerenderer_amd.compositor import FramePlan
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters
from amrvolumerenderer_amd.types import make_params
from amrvolumerenderer_amd import runtime

from helpers import device_box

pytestmark = pytest.mark.gpu


def _local_scene(ctx, spec, cells, rank):
    local = [device_box(ctx, cells[i], spec.boxes[i].min_corner, spec.boxes[i].max_corner,
                        spec.boxes[i].level, rank)
             for i in scenes.local_box_indices(spec, rank)]
    return ctx.create_scene(local, spec.transform)


@pytest.mark.parametrize("n_ranks,policy,transparency,tighten", [
    (1, "morton", 0.97, False), (1, "morton", 0.0, False), (3, "round_robin", 0.9, False),
    (4, "morton", 0.5, True), (8, "level_pairs", 0.97, True)])
def test_chunked_plan_calls_equal_the_single_launch(ctx, n_ranks, policy, transparency, tighten):
    W, H = 150, 110
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    scenes.assign_owners(spec, n_ranks, policy)
    meta = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    ref = runtime.reference_sample_distance(meta, spec.bounds.min_corner, spec.bounds.max_corner)
    params = make_params(W, H, spec.scalar_range, transparency, ref, spec.bounds)
    cam = scenes.orbit_camera(5, 24)
    for r in range(n_ranks):
        plan = FramePlan(meta, params, cam, r, n_ranks)
        if tighten:
            plan.tighten()
        scene = _local_scene(ctx, spec, cells, r)
        n = max(plan.send_floats, 1)
        want = torch.full((n,), float("nan"), device=ctx.device)
        want_samples = torch.zeros(1, dtype=torch.int64, device=ctx.device)
        scene.classify_plan(ctx, plan, 0)
        scene.march_plan(ctx, plan, 0, want, want_samples)
        ctx.synchronize()
        assert int(want_samples.item()) > 0 or plan.n_local_runs == 0
        for chunks in (2, 3, 5, 16):
            got = torch.full((n,), float("nan"), device=ctx.device)
            got_samples = torch.zeros(1, dtype=torch.int64, device=ctx.device)
            # another classified volume, so that nothing of the single-launch frame is read
            scene.classify_plan_chunked(ctx, plan, 1, chunks, first_alone=(chunks == 3))
            scene.march_plan_chunked(ctx, plan, 1, got, chunks, got_samples)
            ctx.synchronize()
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (r, chunks)
            assert int(got_samples.item()) == int(want_samples.item()), (r, chunks)


def test_chunks_on_two_streams_ordered_by_their_events(ctx):
    """The frame driver's shape: classify launches on one stream, march launches on another,
    launch k of the march waiting for the event recorded behind launch k of the classify pass."""
    W, H = 200, 160
    spec = scenes.make_amr_scene(64, 2, 16, "smooth")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    scenes.assign_owners(spec, 1, "morton")
    meta = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    ref = runtime.reference_sample_distance(meta, spec.bounds.min_corner, spec.bounds.max_corner)
    params = make_params(W, H, spec.scalar_range, 0.95, ref, spec.bounds)
    cam = scenes.default_camera()
    plan = FramePlan(meta, params, cam, 0, 1)
    scene = _local_scene(ctx, spec, cells, 0)
    other = runtime.Context(ctx.device_index, priority=-1)
    n = max(plan.send_floats, 1)
    want = torch.full((n,), float("nan"), device=ctx.device)
    scene.classify_plan(ctx, plan, 0)
    scene.march_plan(ctx, plan, 0, want)
    ctx.synchronize()
    for chunks in (2, 4, 6):
        events = [torch.cuda.Event() for _ in range(chunks)]
        for e in events:   # (a torch event has no handle before its first record)
            e.record(torch.cuda.current_stream(ctx.device))
        got = torch.full((n,), float("nan"), device=ctx.device)
        torch.cuda.synchronize()
        scene.classify_plan_chunked(ctx, plan, 2, chunks, events=events, first_alone=True)
        scene.march_plan_chunked(other, plan, 2, got, chunks, events=events)
        other.synchronize()
        ctx.synchronize()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), chunks


@pytest.mark.parametrize("transparency", [0.97, 0.0])
def test_frames_of_a_caller_who_waits_equal_pipelined_frames(ctx, transparency):
    """avr_renderer_set_frame_chunks: frames waited for one by one (the reference's call shape: one
    Render() per frame) or queued back to back, in one launch per kernel or cut into chunks: the
    images and bytes are the same."""
    spec = scenes.make_amr_scene(64, 3, 16, "smooth")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
    local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level) for c, m in
             zip(cells, spec.boxes)]
    p = RenderParameters(320, 200, transparency, 1, draw_bounds=False)
    cams = [scenes.orbit_camera(v, 12) for v in range(4)]

    def fresh(chunks):
        renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range)
        assert renderer.native is not None
        renderer.native.set_frame_chunks(chunks)
        return renderer

    # never chunked, frames queued back to back
    plain = fresh(1)
    want = [plain.render(p, cam, want_image=True) for cam in cams]
    plain.synchronize()
    assert plain.native.last_frame_chunks() == 1

    # a caller who waits for every frame (the reference's call shape), default settings
    waiting = fresh(-1)
    for f, cam in enumerate(cams):
        image, rgb8 = waiting.render(p, cam, want_image=True)
        assert waiting.native.last_frame_chunks() == 1, f   # (chunks are opt-in: they do not pay)
        waiting.synchronize()
        assert torch.equal(image.view(torch.int32), want[f][0].view(torch.int32)), f
        assert torch.equal(rgb8, want[f][1]), f
    # ... and one whose every frame is cut in four, waited for
    got = []
    cut = fresh(4)
    for f, cam in enumerate(cams):
        got.append(cut.render(p, cam, want_image=True))
        assert cut.native.last_frame_chunks() == 4, f
        cut.synchronize()
    # every frame in three / seven chunks
    for chunks in (3, 7):
        forced = fresh(chunks)
        out = [forced.render(p, cam, want_image=True) for cam in cams]
        forced.synchronize()
        assert forced.native.last_frame_chunks() == chunks
        got.extend(out)
    torch.cuda.synchronize()
    for f, (image, rgb8) in enumerate(got):
        w_image, w_rgb8 = want[f % len(cams)]
        assert torch.equal(image.view(torch.int32), w_image.view(torch.int32)), f
        assert torch.equal(rgb8, w_rgb8), f


@pytest.mark.parametrize("n_ranks,policy,transparency", [
    (1, "morton", 0.0), (1, "morton", 0.3), (1, "morton", 0.97), (3, "round_robin", 0.0),
    (4, "level_pairs", 0.1)])
def test_culled_frame_equals_the_single_launch(ctx, n_ranks, policy, transparency):
    """avr_render_plan_culled: boxes that no ray can still sample are left out of the classify pass
    -- decided by the march's own skip test, so the run layers and the sample count are the
    single-launch frame's bit for bit; an opaque transfer function really leaves boxes out, and a
    box that is left out is one the march never reads (its bricklets are poisoned here)."""
    W, H = 190, 140
    spec = scenes.make_amr_scene(64, 3, 16, "smooth")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    scenes.assign_owners(spec, n_ranks, policy)
    meta = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    ref = runtime.reference_sample_distance(meta, spec.bounds.min_corner, spec.bounds.max_corner)
    params = make_params(W, H, spec.scalar_range, transparency, ref, spec.bounds)
    culled_somewhere = False
    for view in (0, 5, 11):
        cam = scenes.orbit_camera(view, 16)
        for r in range(n_ranks):
            plan = FramePlan(meta, params, cam, r, n_ranks)
            scene = _local_scene(ctx, spec, cells, r)
            n = max(plan.send_floats, 1)
            want = torch.full((n,), float("nan"), device=ctx.device)
            want_samples = torch.zeros(1, dtype=torch.int64, device=ctx.device)
            scene.classify_plan(ctx, plan, 0)
            scene.march_plan(ctx, plan, 0, want, want_samples)
            ctx.synchronize()
            for chunks in (2, 4, 9):
                # classified volume 1 first holds OTHER table indices for every cell (the same
                # scene under another scalar range): whatever the culled frame does not classify
                # stays wrong, and a march that read it would show
                other = FramePlan(meta, make_params(W, H, (0.2, 0.6), 0.5, ref, spec.bounds),
                                  scenes.orbit_camera(view + 3, 16), r, n_ranks)
                scratch = torch.empty(max(other.send_floats, 1), device=ctx.device)
                scene.classify_plan(ctx, other, 1)
                got = torch.full((n,), float("nan"), device=ctx.device)
                got_samples = torch.zeros(1, dtype=torch.int64, device=ctx.device)
                flags = scene.render_plan_culled(ctx, plan, 1, got, chunks, got_samples)
                ctx.synchronize()
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (view, r, chunks)
                assert int(got_samples.item()) == int(want_samples.item()), (view, r, chunks)
                # (row k: boxes behind chunk k still visible; the last row is never written)
                visible = flags[:chunks - 1].cpu().numpy()
                if visible.shape[1] and (visible[-1] == 0).any() and plan.n_local_runs > 0:
                    culled_somewhere = True
                del scratch
    if transparency == 0.0:
        assert culled_somewhere, "an opaque frame left no box out"
