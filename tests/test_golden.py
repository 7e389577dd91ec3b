"""Committed golden vectors (tests/golden/oracle_vectors.npz, made by make_golden.py from the
oracle): the oracle must still reproduce them (CPU), and the HIP frame must match them (GPU)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G  # noqa: E402

from helpers import assert_bit_equal  # noqa: E402

GOLDEN = np.load(os.path.join(HERE, "golden", "oracle_vectors.npz"))


@pytest.mark.parametrize("name", list(G.cases()))
def test_oracle_reproduces_golden(O, name):
    got = G.compute(O, name)
    for key, value in got.items():
        want = GOLDEN[f"{name}/{key}"]
        if np.asarray(value).dtype == np.float32:
            assert_bit_equal(np.asarray(value), want, f"{name}/{key}")
        else:
            assert np.array_equal(np.asarray(value), want), f"{name}/{key}"


def test_color_tables_golden(O, avr_lib):
    from amrvolumerenderer_amd import runtime
    assert_bit_equal(O.build_color_table(1.0, 1.0), GOLDEN["table/jet_nf1"], "jet")
    assert_bit_equal(runtime.build_color_table(0.03, 0.25), GOLDEN["table/jet_nf0.25_a0.03"], "jet")
    assert_bit_equal(runtime.build_color_table(0.85, 0.5, (0.0, 1.0), G.LAB_MAP),
                     GOLDEN["table/lab_nf0.5"], "lab")


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(G.cases()))
def test_hip_frame_matches_golden(ctx, name):
    """Simulated N ranks on one GPU through the plan path, compared with the committed frame."""
    import torch
    from amrvolumerenderer_amd import runtime, scenes
    from amrvolumerenderer_amd.compositor import FramePlan
    from amrvolumerenderer_amd.types import make_params
    import plan_helpers as PH
    from helpers import device_box, scene_cells
    kw, (W, H), transparency, cmap, n_ranks, policy = G.cases()[name]
    spec = scenes.make_amr_scene(**kw)
    cam = scenes.default_camera()
    if kw.get("extent", 1.0) != 1.0:
        cam.eye = tuple(c * kw["extent"] for c in cam.eye)
        cam.look_at = tuple(c * kw["extent"] for c in cam.look_at)
    cells = scene_cells(spec)
    scenes.assign_owners(spec, n_ranks, policy)
    meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
    ref = runtime.reference_sample_distance(meta, spec.bounds.min_corner, spec.bounds.max_corner)
    assert np.float32(ref) == GOLDEN[f"{name}/reference_sample_distance"]
    params = make_params(W, H, spec.scalar_range, transparency, ref, spec.bounds, cmap)
    plans, sends, total = [], [], 0
    for r in range(n_ranks):
        plan = FramePlan(meta, params, cam, r, n_ranks)
        local = [device_box(ctx, cells[i], spec.boxes[i].min_corner, spec.boxes[i].max_corner,
                            spec.boxes[i].level, r) for i in scenes.local_box_indices(spec, r)]
        samples = torch.zeros(1, dtype=torch.int64, device=ctx.device)
        send = ctx.create_scene(local, spec.transform).render_plan(plan, samples=samples)
        ctx.synchronize()
        total += int(samples.item())
        plans.append(plan)
        sends.append(send.cpu().numpy())
    assert plans[0].n_runs_total == int(GOLDEN[f"{name}/runs"])
    if transparency >= 0.5:
        assert total == int(GOLDEN[f"{name}/samples"])
    got = np.zeros((W * H, 5), np.float32)
    got8 = np.zeros((W * H, 3), np.uint8)
    for plan, recv in zip(plans, PH.route(plans, sends)):
        dev = torch.from_numpy(np.ascontiguousarray(recv)).to(ctx.device)
        if dev.numel() == 0:
            dev = torch.zeros(1, device=ctx.device)
        piece, rgb8 = ctx.fold_plan(plan, dev, want_rgb8=True)
        ctx.synchronize()
        got[plan.piece_begin:plan.piece_end] = piece.cpu().numpy()
        got8[plan.piece_begin:plan.piece_end] = rgb8.cpu().numpy()
    assert_bit_equal(got, GOLDEN[f"{name}/frame"], name)
    assert np.array_equal(got8.reshape(H, W, 3)[::-1], GOLDEN[f"{name}/rgb8"])
