"""GPU parity of SPECULATIVE frames (avr_classify_plan_flagged / avr_march_plan_speculative,
avr_renderer_set_visibility_speculation): a frame classifies only the boxes an earlier frame of the
same plan sampled; the march checks every box it needs, and a gated repair pass redoes the frame
when the guess was wrong.  Either way every send buffer must equal the plain frame's bit for bit --
which the other GPU tests compare with the oracle."""
import numpy as np
import pytest
import torch

from amrvolumerenderer_amd import scenes
from amrvolumerenderer_amd.compositor import FramePlan
from amrvolumerenderer_amd.types import make_params
from amrvolumerenderer_amd import runtime

from helpers import device_box

pytestmark = pytest.mark.gpu


def _local_scene(ctx, spec, cells, rank):
    local = [device_box(ctx, cells[i], spec.boxes[i].min_corner, spec.boxes[i].max_corner,
                        spec.boxes[i].level, rank)
             for i in scenes.local_box_indices(spec, rank)]
    return ctx.create_scene(local, spec.transform)


def _plain(ctx, scene, plan, slot):
    want = torch.full((max(plan.send_floats, 1),), float("nan"), device=ctx.device)
    scene.classify_plan(ctx, plan, slot)
    scene.march_plan(ctx, plan, slot, want)
    ctx.synchronize()
    return want


def _speculative(ctx, scene, plan, slot, flags, dirty=False):
    """The protocol of include/avr_hip.h: flagged classify, checking march, gated repair (dirty:
    the repair redoes only the workgroups that met an unclassified box)."""
    n = max(len(scene.boxes), 1)
    marks = (torch.zeros(scene.march_plan_workgroups(plan), dtype=torch.uint8, device=ctx.device)
             if dirty else None)
    got = torch.full((max(plan.send_floats, 1),), float("nan"), device=ctx.device)
    visited = torch.zeros(n, dtype=torch.uint8, device=ctx.device)
    missed = torch.zeros(n, dtype=torch.uint8, device=ctx.device)
    miss_count = torch.zeros(1, dtype=torch.int32, device=ctx.device)
    scene.classify_plan_flagged(ctx, plan, slot, flags)
    scene.march_plan_speculative(ctx, plan, slot, got, classified=flags, visited=visited,
                                 missed=missed, miss_count=miss_count, dirty_workgroups=marks)
    scene.classify_plan_flagged(ctx, plan, slot, missed, gate=miss_count)
    scene.march_plan_speculative(ctx, plan, slot, got, visited=visited, gate=miss_count,
                                 dirty_workgroups=marks)
    ctx.synchronize()
    if marks is not None and int(miss_count.item()) > 0:
        assert 0 < int(marks.sum().item()) <= marks.numel()
    return got, visited, missed, int(miss_count.item())


@pytest.mark.parametrize("n_ranks,policy,transparency", [
    (1, "morton", 0.0), (1, "morton", 0.3), (1, "level_pairs", 0.97), (3, "round_robin", 0.0),
    (4, "morton", 0.1)])
def test_speculative_frame_equals_the_plain_frame(ctx, n_ranks, policy, transparency):
    W, H = 190, 140
    spec = scenes.make_amr_scene(64, 3, 16, "smooth")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    scenes.assign_owners(spec, n_ranks, policy)
    meta = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    ref = runtime.reference_sample_distance(meta, spec.bounds.min_corner, spec.bounds.max_corner)
    params = make_params(W, H, spec.scalar_range, transparency, ref, spec.bounds)
    left_out = False
    for view in (0, 5, 11):
        cam = scenes.orbit_camera(view, 16)
        for r in range(n_ranks):
            plan = FramePlan(meta, params, cam, r, n_ranks)
            scene = _local_scene(ctx, spec, cells, r)
            if plan.n_local_runs == 0:
                continue
            want = _plain(ctx, scene, plan, 0)
            n = max(len(scene.boxes), 1)
            # a first frame that only records what it samples
            first = torch.full_like(want, float("nan"))
            visited = torch.zeros(n, dtype=torch.uint8, device=ctx.device)
            scene.march_plan_speculative(ctx, plan, 0, first, visited=visited)
            ctx.synchronize()
            assert torch.equal(first.view(torch.int32), want.view(torch.int32)), (view, r)
            # classified volume 1 holds OTHER table indices for every cell (another scalar range):
            # whatever the flagged pass leaves out stays wrong, and a march that read it would show
            other = FramePlan(meta, make_params(W, H, (0.2, 0.6), 0.5, ref, spec.bounds),
                              scenes.orbit_camera(view + 3, 16), r, n_ranks)
            scene.classify_plan(ctx, other, 1)
            got, again, missed, misses = _speculative(ctx, scene, plan, 1, visited)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (view, r)
            assert misses == 0 and int(missed.sum().item()) == 0, (view, r, misses)
            assert torch.equal(again, visited), (view, r)   # the same frame samples the same boxes
            # ... and the same set held on the host: a launch of exactly those boxes' tiles
            scene.classify_plan(ctx, other, 1)
            listed = torch.full_like(want, float("nan"))
            missed.zero_()
            count = torch.zeros(1, dtype=torch.int32, device=ctx.device)
            scene.classify_plan_positions(ctx, plan, 1, torch.nonzero(visited.cpu()).flatten().tolist())
            scene.march_plan_speculative(ctx, plan, 1, listed, classified=visited, missed=missed,
                                         miss_count=count)
            ctx.synchronize()
            assert torch.equal(listed.view(torch.int32), want.view(torch.int32)), (view, r)
            assert int(count.item()) == 0, (view, r)
            # ... with the flags handed over in host memory (staged with the launch)
            scene.classify_plan(ctx, other, 1)
            scene.classify_plan_positions(ctx, plan, 1, torch.nonzero(visited.cpu()).flatten().tolist())
            hosted = torch.full_like(want, float("nan"))
            scene.march_plan_speculative(ctx, plan, 1, hosted, classified=visited.cpu(), missed=missed,
                                         miss_count=count)
            ctx.synchronize()
            assert torch.equal(hosted.view(torch.int32), want.view(torch.int32)), (view, r)
            assert int(count.item()) == 0, (view, r)
            if int((visited == 0).sum().item()) > 0:
                left_out = True
    if transparency == 0.0 and n_ranks == 1:   # (a rank of several folds short runs: little to hide)
        assert left_out, "an opaque frame sampled every box"


@pytest.mark.parametrize("dirty", [False, True])
@pytest.mark.parametrize("guess", ["nothing", "front_only", "random"])
def test_a_wrong_guess_is_repaired(ctx, guess, dirty):
    """Flags that leave out boxes the frame needs (the cells changed since the frame they come
    from): the march raises them, the gated repair pass classifies them and marches again."""
    W, H = 170, 130
    spec = scenes.make_amr_scene(64, 3, 16, "smooth")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    scenes.assign_owners(spec, 1, "morton")
    meta = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    ref = runtime.reference_sample_distance(meta, spec.bounds.min_corner, spec.bounds.max_corner)
    rng = np.random.default_rng(7)
    for transparency in (0.0, 0.6):
        params = make_params(W, H, spec.scalar_range, transparency, ref, spec.bounds)
        plan = FramePlan(meta, params, scenes.orbit_camera(3, 16), 0, 1)
        scene = _local_scene(ctx, spec, cells, 0)
        want = _plain(ctx, scene, plan, 0)
        n = len(scene.boxes)
        truth = torch.zeros(n, dtype=torch.uint8, device=ctx.device)
        scratch = torch.empty_like(want)
        scene.march_plan_speculative(ctx, plan, 0, scratch, visited=truth)
        ctx.synchronize()
        needed = truth.cpu().numpy().astype(bool)
        if guess == "nothing":
            flags = np.zeros(n, np.uint8)
        elif guess == "front_only":
            flags = np.zeros(n, np.uint8)
            flags[np.flatnonzero(needed)[:max(1, needed.sum() // 3)]] = 1
        else:
            flags = (rng.random(n) < 0.5).astype(np.uint8)
        other = FramePlan(meta, make_params(W, H, (0.2, 0.6), 0.5, ref, spec.bounds),
                          scenes.orbit_camera(9, 16), 0, 1)
        scene.classify_plan(ctx, other, 1)   # poison
        got, visited, missed, misses = _speculative(ctx, scene, plan, 1,
                                                    torch.from_numpy(flags).to(ctx.device), dirty)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (guess, transparency)
        lacking = needed & (flags == 0)
        assert (misses > 0) == bool(lacking.any()), (guess, transparency, misses)
        # every box the true frame needs is classified after the repair
        covered = (flags != 0) | (missed.cpu().numpy() != 0)
        assert not (needed & ~covered).any(), (guess, transparency)
        # and what the final march sampled is recorded (a superset is allowed: the first pass)
        assert not (needed & (visited.cpu().numpy() == 0)).any(), (guess, transparency)


def test_argument_checks(ctx):
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    scenes.assign_owners(spec, 1, "morton")
    meta = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    ref = runtime.reference_sample_distance(meta, spec.bounds.min_corner, spec.bounds.max_corner)
    plan = FramePlan(meta, make_params(64, 48, spec.scalar_range, 0.5, ref, spec.bounds),
                     scenes.default_camera(), 0, 1)
    scene = _local_scene(ctx, spec, cells, 0)
    out = torch.empty(max(plan.send_floats, 1), device=ctx.device)
    n = len(scene.boxes)
    flags = torch.ones(n, dtype=torch.uint8, device=ctx.device)
    with pytest.raises(ValueError):
        scene.classify_plan_flagged(ctx, plan, 0, flags[:n - 1])
    with pytest.raises(ValueError):   # positions must ascend
        scene.classify_plan_positions(ctx, plan, 0, [1, 0])
    with pytest.raises(ValueError):   # flags to check, nowhere to report a miss
        scene.march_plan_speculative(ctx, plan, 0, out, classified=flags)


# ---- the frame driver (avr_renderer_set_visibility_speculation) -----------------------------------

def _renderer_scene(ctx):
    from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters
    spec = scenes.make_amr_scene(64, 3, 16, "smooth")
    cells = [scenes.box_cells_numpy(spec, i) for i in range(len(spec.boxes))]
    meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
    local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level) for c, m in zip(cells, spec.boxes)]

    def fresh(mode):
        renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range)
        assert renderer.native is not None
        renderer.native.set_visibility_speculation(mode)
        # (16-cell boxes hide little: every plan that leaves ANY box unsampled speculates here)
        renderer.native.debug_set_speculation_threshold(0.999)
        return renderer
    return spec, local, fresh, RenderParameters


def _same(a, b):
    return torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)) and torch.equal(a[1], b[1])


def test_driver_speculates_on_a_standing_camera_and_repairs_after_the_cells_change(ctx):
    spec, local, fresh, RenderParameters = _renderer_scene(ctx)
    p = RenderParameters(320, 200, 0.0, 1, draw_bounds=False)   # the reference's default: opaque
    cam = scenes.orbit_camera(2, 12)
    plain = fresh(0)
    want = plain.render(p, cam, want_image=True)
    plain.synchronize()
    assert plain.native.speculation_state()["state"] == "off"

    guessing = fresh(-1)
    frames = [guessing.render(p, cam, want_image=True) for _ in range(14)]
    guessing.synchronize()
    for f, frame in enumerate(frames):
        assert _same(frame, want), f
    state = guessing.native.speculation_state()
    assert state["state"] == "speculating", state
    assert state["speculative_frames"] > 0 and state["repaired_frames"] == 0, state
    assert 0.0 < state["sampled_fraction"] < 1.0, state

    # the cells change under the standing camera: everything becomes the range's minimum, which the
    # default map makes (nearly) transparent -- rays now reach boxes the flags leave out
    saved = [box.values.clone() for box in local]
    try:
        for box in local:
            box.values.fill_(float(spec.scalar_range[0]))
        torch.cuda.synchronize()
        want_changed = plain.render(p, cam, want_image=True)
        plain.synchronize()
        assert not _same(want_changed, want)
        for f in range(5):
            frame = guessing.render(p, cam, want_image=True)
            guessing.synchronize()
            assert _same(frame, want_changed), f
        state = guessing.native.speculation_state()
        assert state["repaired_frames"] >= 1, state
        # (with every box in view now the driver gives the guessing up: "not worth it")
        assert state["state"] in ("suspended after repairs", "speculating", "not worth it"), state
    finally:
        for box, cells in zip(local, saved):
            box.values.copy_(cells)
        torch.cuda.synchronize()
    # ... and back: the frames stay right while the driver finds its way again
    for f in range(6):
        frame = guessing.render(p, cam, want_image=True)
        guessing.synchronize()
        assert _same(frame, want), f


def test_driver_leaves_translucent_frames_and_moving_cameras_alone(ctx):
    spec, local, fresh, RenderParameters = _renderer_scene(ctx)
    # a translucent frame samples every box: decided once, then nothing is recorded any more
    p = RenderParameters(320, 200, 0.97, 1, draw_bounds=False)
    cam = scenes.orbit_camera(2, 12)
    plain = fresh(0)
    want = plain.render(p, cam, want_image=True)
    plain.synchronize()
    guessing = fresh(-1)
    frames = [guessing.render(p, cam, want_image=True) for _ in range(10)]
    guessing.synchronize()
    for f, frame in enumerate(frames):
        assert _same(frame, want), f
    state = guessing.native.speculation_state()
    assert state["state"] == "not worth it" and state["speculative_frames"] == 0, state
    # a camera that never repeats: no plan has a frame to guess from
    opaque = RenderParameters(320, 200, 0.0, 1, draw_bounds=False)
    cams = [scenes.orbit_camera(v, 12) for v in range(6)]
    want = [plain.render(opaque, c, want_image=True) for c in cams]
    plain.synchronize()
    got = [guessing.render(opaque, c, want_image=True) for c in cams]
    guessing.synchronize()
    for f in range(len(cams)):
        assert _same(got[f], want[f]), f
    assert guessing.native.speculation_state()["speculative_frames"] == 0


def test_driver_carries_what_it_learnt_to_a_moving_camera(ctx):
    """The memory is per box, not per plan: a camera that turns a little every frame keeps the set,
    boxes that come into view are missed once (and repaired), then remembered."""
    spec, local, fresh, RenderParameters = _renderer_scene(ctx)
    p = RenderParameters(320, 200, 0.0, 1, draw_bounds=False)
    cams = [scenes.orbit_camera(v, 240) for v in range(40)]          # 1.5 degrees per frame
    plain = fresh(0)
    want = [plain.render(p, c, want_image=True) for c in cams]
    plain.synchronize()
    guessing = fresh(-1)
    got = [guessing.render(p, c, want_image=True) for c in cams]
    guessing.synchronize()
    for f in range(len(cams)):
        assert _same(got[f], want[f]), f
    state = guessing.native.speculation_state()
    assert state["speculative_frames"] > 0, state
    # ... and a jump to the other side of the volume: whatever the set lacks is repaired
    far = [scenes.orbit_camera(v, 240) for v in (120, 121, 122, 200, 60)]
    want = [plain.render(p, c, want_image=True) for c in far]
    plain.synchronize()
    for f, c in enumerate(far):
        frame = guessing.render(p, c, want_image=True)
        guessing.synchronize()
        assert _same(frame, want[f]), f


def test_driver_forgets_what_it_learnt_when_the_transfer_function_changes(ctx):
    """Another scalar range maps the cells to other opacities: what was sampled under the old one says
    nothing (observations still in flight are ignored when they arrive)."""
    spec, local, fresh, RenderParameters = _renderer_scene(ctx)
    p = RenderParameters(320, 200, 0.0, 1, draw_bounds=False)
    cam = scenes.orbit_camera(2, 12)
    plain, guessing = fresh(0), fresh(-1)
    for scalar_range in (spec.scalar_range, (0.2, 0.6), (0.0, 0.3), spec.scalar_range):
        plain.scalar_range = scalar_range
        guessing.scalar_range = scalar_range     # (no synchronise: frames of the old range in flight)
        want = plain.render(p, cam, want_image=True)
        plain.synchronize()
        frames = [guessing.render(p, cam, want_image=True) for _ in range(9)]
        guessing.synchronize()
        for f, frame in enumerate(frames):
            assert _same(frame, want), (scalar_range, f)
    assert guessing.native.speculation_state()["speculative_frames"] > 0


def test_soak_random_cameras_and_changing_cells(ctx):
    """150 frames of a guessing driver against a plain one on the same data, in bursts of five
    queued frames: the camera stands, turns a little or jumps, and between bursts, now and then, a
    random third of the boxes turns transparent or back -- every frame's image and bytes are the
    plain frame's."""
    spec, local, fresh, RenderParameters = _renderer_scene(ctx)
    p = RenderParameters(256, 160, 0.0, 1, draw_bounds=False)
    plain, guessing = fresh(0), fresh(-1)
    rng = np.random.default_rng(11)
    original = [box.values.clone() for box in local]
    transparent = set()
    view = 0
    try:
        for burst in range(30):
            if burst % 5 == 4:      # (nothing is in flight: the bursts end synchronised)
                for b in rng.choice(len(local), size=len(local) // 3, replace=False):
                    if b in transparent:
                        local[b].values.copy_(original[b])
                        transparent.discard(b)
                    else:
                        local[b].values.fill_(float(spec.scalar_range[0]))
                        transparent.add(b)
                torch.cuda.synchronize()
            cams = []
            for _ in range(5):
                move = rng.random()
                if 0.5 <= move < 0.9:
                    view = (view + 1) % 240                  # turns 1.5 degrees
                elif move >= 0.9:
                    view = int(rng.integers(0, 240))         # jumps
                cams.append(scenes.orbit_camera(view, 240))  # (else: the camera stands)
            want = [plain.render(p, cam, want_image=True) for cam in cams]
            got = [guessing.render(p, cam, want_image=True) for cam in cams]
            plain.synchronize()
            guessing.synchronize()
            for f in range(5):
                assert _same(got[f], want[f]), (burst, f)
        state = guessing.native.speculation_state()
        assert state["speculative_frames"] > 0, state
    finally:
        torch.cuda.synchronize()
        for box, cells in zip(local, original):
            box.values.copy_(cells)
        torch.cuda.synchronize()
