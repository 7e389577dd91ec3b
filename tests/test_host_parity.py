"""CPU tests: the product's host prologue (C ABI, no GPU needed) against the oracle, the
library's exported symbols, and the reference's own known-answer blend fixtures."""
import ctypes as C
import re
import os

import numpy as np
import pytest

from amrvolumerenderer_amd import _capi, runtime, scenes
from amrvolumerenderer_amd.types import AmrBox, CameraParameters, VolumeBounds, make_params

from helpers import assert_bit_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LAB_MAP = [(0.0, 0.0, 0.0, 0.2, 0.0), (0.25, 0.1, 0.3, 0.9, 0.1), (0.5, 0.9, 0.9, 0.2, 0.4),
           (0.8, 1.0, 0.3, 0.0, 0.7), (1.0, 1.0, 1.0, 1.0, 1.0)]


def test_library_exports_every_declared_symbol(avr_lib):
    # the drop-in boundary (avr_hip.h) and the test hooks / diagnostics kept apart from it
    declared = set()
    for name in ("avr_hip.h", "avr_hip_debug.h"):
        header = open(os.path.join(ROOT, "include", name)).read()
        found = set(re.findall(r"\b(avr_[a-z0-9_]+)\s*\(", header))
        assert found, f"no declarations found in {name}"
        declared |= found
    product = set(re.findall(r"\b(avr_[a-z0-9_]+)\s*\(",
                             open(os.path.join(ROOT, "include", "avr_hip.h")).read()))
    assert not any("debug" in name or "history" in name or "counters" in name for name in product)
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    for name in declared:
        assert hasattr(avr_lib, name), name
    assert avr_lib.avr_abi_version() == 2


def test_compute_entry_points_fail_without_device(avr_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    handle = C.c_void_p()
    status = avr_lib.avr_context_create(0, C.byref(handle))
    assert status == _capi.AVR_ERR_NO_DEVICE
    assert b"no HIP device" in avr_lib.avr_last_error()
    with pytest.raises(_capi.AvrNoDevice):
        runtime.Context(0)


@pytest.mark.parametrize("alpha_scale", [1.0, 0.85, 0.03])
@pytest.mark.parametrize("nf", [1.0, 0.5, 0.25, 2.0])
@pytest.mark.parametrize("cmap", [None, LAB_MAP])
@pytest.mark.parametrize("rng", [(0.0, 1.0), (0.1, 0.7), (-2.0, 3.5)])
def test_color_table_bits(O, avr_lib, alpha_scale, nf, cmap, rng):
    got = runtime.build_color_table(alpha_scale, nf, rng, cmap)
    want = O.build_color_table(alpha_scale, nf, rng, cmap)
    assert_bit_equal(got, want, "color table")
    assert np.all(np.isfinite(got))


def test_color_table_default_values(avr_lib):
    # default jet ends (VolumePainter.cpp:471-487): entry 0 = (0,0,0.5625,0.05), 255 = (0.5,0,0,0.5)
    t = runtime.build_color_table(1.0, 1.0)
    assert t[0].tolist() == [0.0, 0.0, 0.5625, np.float32(0.05)]
    assert t[255].tolist() == [0.5, 0.0, 0.0, 0.5]
    # opacity correction 1-(1-a)^nf at nf = 0.5 (computeScaledAlpha, :107-125)
    t2 = runtime.build_color_table(1.0, 0.5)
    assert t2[255, 3] == np.float32(1.0 - (1.0 - np.float32(0.5)) ** 0.5)


def _meta_boxes(spec):
    return [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]


@pytest.mark.parametrize("extent", [1.0, 0.7, 3.3])
def test_sampling_hints_reference_distance(O, avr_lib, extent):
    spec = scenes.make_amr_scene(64, 3, 16, "smooth", extent=extent)
    boxes = _meta_boxes(spec)
    dummy = np.zeros((16, 16, 16))
    oboxes = [O.make_box(dummy, b.min_corner, b.max_corner) for b in boxes]
    cam = scenes.default_camera()
    ocam = O.make_camera(cam.eye, cam.look_at, cam.up, cam.fov_y_degrees, cam.near_plane,
                         cam.far_plane)
    got_ref = runtime.reference_sample_distance(boxes, spec.bounds.min_corner,
                                                spec.bounds.max_corner)
    want_ref = O.reference_sample_distance(oboxes, spec.bounds.min_corner, spec.bounds.max_corner)
    assert np.float32(got_ref).view(np.uint32) == np.float32(want_ref).view(np.uint32)
    params = make_params(64, 64, (0, 1), 0.15, got_ref, spec.bounds)
    oparams = O.make_params(64, 64, (0, 1), 0.15, want_ref, spec.bounds.min_corner,
                            spec.bounds.max_corner)
    factors = set()
    for b, ob in zip(boxes, oboxes):
        got = runtime.box_sampling(b, params)
        want = O.box_sampling(ob, oparams)
        assert_bit_equal(np.array(got), np.array(want), "sampling")
        factors.add(got[1])
        gh = runtime.box_depth_hint(b, cam)
        wh = O.box_depth_hint(ob, ocam)
        assert np.float32(gh).view(np.uint32) == np.float32(wh).view(np.uint32)
    assert sorted(factors) == [0.25, 0.5, 1.0]  # one table per AMR level


def test_degenerate_spacing_fallback(O, avr_lib):
    # zero-extent box: no axis has a positive spacing, so minSpacing keeps its initial
    # numeric_limits<float>::max() -- which is finite and positive, so the bounds fallback of
    # VolumePainter.cpp:593-598 does NOT trigger (it only would for NaN spacing).
    b = AmrBox((0.5, 0.5, 0.5), (0.5, 0.5, 0.5), None, dims=(4, 4, 4))
    ob = O.make_box(np.zeros((4, 4, 4)), b.min_corner, b.max_corner)
    bounds = VolumeBounds((0, 0, 0), (2.0, 1.0, 4.0))
    params = make_params(8, 8, (0, 1), 0.0, 0.0, bounds)
    oparams = O.make_params(8, 8, (0, 1), 0.0, 0.0, bounds.min_corner, bounds.max_corner)
    got = runtime.box_sampling(b, params)
    want = O.box_sampling(ob, oparams)
    assert_bit_equal(np.array(got), np.array(want), "fallback sampling")
    assert got[0] == np.finfo(np.float32).max * np.float32(0.5)
    assert got[1] == 1.0  # referenceDistance <= 0 -> normalizationFactor 1


def test_layer_order_and_runs(O, avr_lib):
    rng = np.random.default_rng(7)
    for n_layers, n_ranks in [(1, 1), (9, 3), (64, 8), (33, 4)]:
        hints = rng.choice(np.linspace(0.5, 3.0, 12).astype(np.float32), n_layers)  # many ties
        owner = rng.integers(0, n_ranks, n_layers).astype(np.int32)
        local = np.zeros(n_layers, dtype=np.int32)
        for r in range(n_ranks):
            idx = np.nonzero(owner == r)[0]
            local[idx] = np.arange(idx.size)
        got_o, got_r = runtime.layer_order(hints, owner, local)
        want_o, want_r = O.layer_order(hints, owner, local)
        assert got_o.tolist() == want_o.tolist()
        assert got_r.tolist() == want_r.tolist()
    o, r = runtime.layer_order([], [], [])
    assert o.size == 0 and r.size == 0


def test_piece_range(O, avr_lib):
    for size, n in [(6144, 4), (10, 3), (7, 8), (0, 2), (2048 * 2048, 8), (110 * 100, 7)]:
        covered = 0
        for k in range(n):
            got = runtime.piece_range(size, k, n)
            assert got == O.piece_range(size, k, n)
            assert got[0] == covered
            covered = got[1]
        assert covered == size
    with pytest.raises(ValueError):
        runtime.piece_range(10, 3, 3)


def test_invalid_arguments_raise(avr_lib):
    with pytest.raises(ValueError):
        runtime.build_color_table(1.0, 1.0, (0, 1), [(0.0, 1.0, 1.0)])  # malformed entry


def test_box_footprint_contains_what_the_oracle_paints(O, avr_lib):
    """avr_box_footprint (host): the conservative screen rectangle the march restricts a box to and
    the per-row extents the tightened exchange layout keeps.  Every pixel the oracle's
    VolumePainter::paint leaves non-empty for the box -- random boxes seen from afar, from close
    up, from inside, along a face -- lies inside both; and the extents are not vacuous."""
    from helpers import oracle_camera, oracle_params, oracle_transform
    from amrvolumerenderer_amd.types import ScalarTransform
    rng = np.random.default_rng(31)
    L = _capi.lib()
    W, H = 96, 72
    bounds = VolumeBounds((-1.0,) * 3, (2.0,) * 3)
    cells = np.full((6, 6, 6), 0.5)
    covered = kept = 0
    for case in range(60):
        lo = rng.uniform(-0.3, 0.6, size=3)
        hi = lo + rng.uniform(0.15, 0.8, size=3)
        centre = 0.5 * (lo + hi)
        kind = case % 4
        if kind == 0:
            d = rng.normal(size=3)
            eye = centre + d / np.linalg.norm(d) * rng.uniform(1.5, 4.0)
            look, fov = centre + rng.uniform(-0.3, 0.3, size=3), rng.uniform(20.0, 70.0)
        elif kind == 1:
            d = rng.normal(size=3)
            eye = centre + d / np.linalg.norm(d) * rng.uniform(0.55, 0.9) * np.linalg.norm(hi - lo)
            look, fov = centre, rng.uniform(60.0, 110.0)
        elif kind == 2:
            eye = lo + rng.uniform(0.1, 0.9, size=3) * (hi - lo)
            look, fov = eye + rng.normal(size=3), rng.uniform(40.0, 100.0)
        else:
            axis = int(rng.integers(0, 3))
            eye = centre.copy()
            eye[axis] = hi[axis] + rng.uniform(0.5, 2.0)
            other = (axis + 1) % 3
            eye[other] = (lo, hi)[int(rng.integers(0, 2))][other] + rng.choice([0.0, 1e-6, -1e-4])
            look, fov = eye.copy(), rng.uniform(20.0, 60.0)
            look[axis] = lo[axis]
        forward = look - eye
        up = (0.0, 1.0, 0.0) if abs(forward[1]) < 0.9 * np.linalg.norm(forward) else (1.0, 0.0, 0.0)
        cam = CameraParameters(tuple(float(v) for v in eye), tuple(float(v) for v in look), up,
                               float(fov), 0.05, 50.0)
        box = AmrBox(tuple(float(v) for v in lo), tuple(float(v) for v in hi), None, dims=(6, 6, 6))
        rect = (C.c_int32 * 4)()
        x0 = (C.c_int32 * H)()
        x1 = (C.c_int32 * H)()
        cbox, ccam = box.to_c(), cam.to_c()
        _capi.check(L.avr_box_footprint(C.byref(cbox), C.byref(ccam), W, H, rect, x0, x1))
        ob = O.make_box(cells, box.min_corner, box.max_corner)
        op = oracle_params(O, W, H, (0.0, 1.0), 0.5, 0.05, bounds)
        layer, n = O.paint_box(ob, oracle_transform(O, ScalarTransform(normalize_to_unit_range=True)),
                               op, oracle_camera(O, cam))
        painted = layer.reshape(H, W, 5)[..., 3] != 0.0
        ys, xs = np.nonzero(painted)
        assert (n > 0) == (len(ys) > 0)
        for y, x in zip(ys, xs):
            assert rect[0] <= x <= rect[2] and rect[1] <= y <= rect[3], (case, x, y, list(rect))
            assert x0[y] <= x <= x1[y], (case, x, y, x0[y], x1[y])
        covered += len(ys)
        kept += sum(max(x1[y] - x0[y] + 1, 0) for y in range(H))
    assert covered > 20000 and kept < 2.2 * covered   # conservative, but not the whole image
