"""scene scalar statistics, the BuildSceneGeometry scalar transform and the
histogram -- host logic against the oracle on the CPU, kernels against the oracle on the GPU."""
import numpy as np
import pytest

from amrvolumerenderer_amd import runtime


def _close_transform(t, ot):
    assert bool(t.log_scale_input) == bool(ot.log_scale_input)
    assert bool(t.normalize_to_unit_range) == bool(ot.normalize_to_unit_range)
    assert t.positive_floor == ot.positive_floor
    assert t.normalization_min == ot.normalization_min
    assert t.inverse_normalization_span == ot.inverse_normalization_span


@pytest.mark.parametrize("stats,count,log,norm", [
    ((0.25, 7.5, 0.25), 1000, False, True), ((-3.0, 7.5, 0.001), 1000, False, False),
    ((-3.0, 7.5, 0.001), 1000, True, True), ((2.0, 2.0, 2.0), 10, False, True),
    ((1e-300, 1e300, 1e-300), 5, True, True), ((5.0, 5.0, 5.0), 3, True, False)])
def test_scene_transform_matches_oracle(O, avr_lib, stats, count, log, norm):
    status, otr, pmin, pmax, oproc, orange = O.scene_transform(stats, count, log, norm)
    assert status == 0
    tr, proc, rng = runtime.scene_transform_from_stats(stats, count, log, norm)
    _close_transform(tr, otr)
    assert tr.processed_min == pmin and tr.processed_max == pmax
    assert np.float32(proc[0]) == np.float32(oproc[0]) and np.float32(proc[1]) == np.float32(oproc[1])
    assert rng == orange
    if norm:
        assert rng == (0.0, 1.0)  # SetSceneNormalizationRange


def test_scene_transform_errors(O, avr_lib):
    inf = float("inf")
    assert O.scene_transform((-2.0, -1.0, inf), 10, True, True)[0] == 1
    with pytest.raises(RuntimeError, match="no positive"):
        runtime.scene_transform_from_stats((-2.0, -1.0, inf), 10, True, True)
    assert O.scene_transform((inf, -inf, inf), 0, False, True)[0] == 2
    with pytest.raises(RuntimeError):
        runtime.scene_transform_from_stats((inf, -inf, inf), 0, False, True)


def _boxes(rng, shapes, strided=False):
    out = []
    for nz, ny, nx in shapes:
        if strided:
            storage = rng.normal(1.0, 2.0, (nz + 3, ny + 2, nx + 5))
            out.append((storage, (slice(1, 1 + nz), slice(2, 2 + ny), slice(3, 3 + nx))))
        else:
            out.append((rng.normal(1.0, 2.0, (nz, ny, nx)), None))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("strided", [False, True])
def test_scalar_stats_and_histogram_kernels(O, ctx, strided):
    import torch
    from amrvolumerenderer_amd.types import AmrBox, ScalarTransform
    rng = np.random.default_rng(42)
    specs = _boxes(rng, [(16, 16, 16), (5, 9, 131), (12, 7, 300), (4, 4, 3)], strided)
    specs[0][0].reshape(-1)[::17] = np.nan
    specs[1][0].reshape(-1)[::23] = np.inf
    specs[2][0].reshape(-1)[::29] = -np.inf
    dev_boxes, orc_boxes = [], []
    for storage, sl in specs:
        t = torch.from_numpy(storage).to(ctx.device)
        view = t if sl is None else t[sl]
        host = storage if sl is None else storage[sl]
        dev_boxes.append(AmrBox((0, 0, 0), (1, 1, 1), view))
        orc_boxes.append(O.make_box(host, (0, 0, 0), (1, 1, 1)))
    scene = ctx.create_scene(dev_boxes, ScalarTransform())
    got = scene.scalar_stats()
    want = O.scalar_stats(orc_boxes)
    assert got == want, (got, want)

    for log_scale, bins in [(False, 256), (True, 64), (False, 7), (False, 5000)]:
        tr, _, rng_pair = runtime.scene_transform_from_stats(got[:3], got[3], log_scale, True)
        status, otr, *_ = O.scene_transform(want[:3], want[3], log_scale, True)
        assert status == 0
        counts = scene.histogram(tr, rng_pair[0], rng_pair[1], bins)
        ctx.synchronize()
        want_counts = O.histogram(orc_boxes, otr, rng_pair[0], rng_pair[1], bins)
        got_counts = counts.cpu().numpy().astype(np.uint64)
        if log_scale:
            # device log vs glibc log: a cell exactly on a bin edge may move (none observed)
            assert np.abs(got_counts.astype(np.int64) - want_counts.astype(np.int64)).sum() <= 4
        else:
            assert np.array_equal(got_counts, want_counts)
        assert int(got_counts.sum()) == sum(int(np.prod(s[0].shape if s[1] is None else
                                                       s[0][s[1]].shape)) for s in specs)
    # accumulates into an existing buffer; degenerate range leaves it untouched
    again = scene.histogram(tr, rng_pair[0], rng_pair[1], 5000, counts)
    ctx.synchronize()
    assert np.array_equal(again.cpu().numpy().astype(np.uint64), 2 * got_counts)
    untouched = scene.histogram(tr, 1.0, 1.0, 16)
    assert int(untouched.sum().item()) == 0
    with pytest.raises(ValueError):
        scene.histogram(tr, 0.0, 1.0, 0)


@pytest.mark.gpu
def test_build_scene_geometry_and_compute_histogram(O, ctx):
    import torch
    from amrvolumerenderer_amd import api, scenes
    from helpers import device_box, scene_cells
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    cells = [c * 40.0 - 3.0 for c in scene_cells(spec)]   # physical values, not in [0,1]
    local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level) for c, m in
             zip(cells, spec.boxes)]
    meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
    orc_boxes = [O.make_box(c, m.min_corner, m.max_corner) for c, m in zip(cells, spec.boxes)]
    stats = O.scalar_stats(orc_boxes)
    geometry = api.build_scene_geometry(ctx, meta, local, spec.bounds)
    _, otr, *_ = O.scene_transform(stats[:3], stats[3], False, True)
    _close_transform(geometry.scalar_transform, otr)
    assert geometry.scalar_range == (0.0, 1.0)
    result = api.compute_scene_histogram(ctx, meta, local, log_scale=False, bins=128)
    want = O.histogram(orc_boxes, otr, 0.0, 1.0, 128)
    assert np.array_equal(result["counts"], want)
    assert result["samples"] == sum(c.size for c in cells)
    assert result["normalized_range"] == (0.0, 1.0)
    assert np.float32(result["original_range"][0]) == np.float32(stats[0])
    assert np.float32(result["original_range"][1]) == np.float32(stats[1])
