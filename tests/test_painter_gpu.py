"""GPU parity: avr_paint_box / avr_render_runs (HIP, through the C ABI) against the CPU oracle
on the same inputs.  The bar is bit-exact float output and an identical sample count."""
import numpy as np
import pytest
import torch

from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.types import (AmrBox, CameraParameters, ScalarTransform, VolumeBounds,
                                         make_params)

from helpers import (assert_bit_equal, device_box, oracle_camera, oracle_params,
                     oracle_transform, scene_cells)

pytestmark = pytest.mark.gpu

LAB_MAP = [(0.0, 0.0, 0.0, 0.2, 0.0), (0.25, 0.1, 0.3, 0.9, 0.1), (0.5, 0.9, 0.9, 0.2, 0.4),
           (0.8, 1.0, 0.3, 0.0, 0.7), (1.0, 1.0, 1.0, 1.0, 1.0)]
BOUNDS = VolumeBounds((-0.05,) * 3, (1.05,) * 3)
NORM = ScalarTransform(normalize_to_unit_range=True)


def radial(nx, ny, nz):
    x = np.arange(nx, dtype=np.float64) / max(nx - 1, 1)
    y = np.arange(ny, dtype=np.float64) / max(ny - 1, 1)
    z = np.arange(nz, dtype=np.float64) / max(nz - 1, 1)
    return np.ascontiguousarray(
        (x[None, None, :] ** 2 + y[None, :, None] ** 2 + z[:, None, None] ** 2) / 3.0)


def compare_box(O, ctx, cells, minc, maxc, cam, width, height, transform=NORM,
                scalar_range=(0.0, 1.0), transparency=0.0, ref_dist=0.0, color_map=None,
                cells_view=None):
    """cells_view: optional (storage array, slicing) to test strided Array4 views."""
    ob = O.make_box(cells, minc, maxc)
    op = oracle_params(O, width, height, scalar_range, transparency, ref_dist, BOUNDS, color_map)
    want, want_n = O.paint_box(ob, oracle_transform(O, transform), op, oracle_camera(O, cam))
    if cells_view is None:
        box = device_box(ctx, cells, minc, maxc)
    else:
        storage, sl = cells_view
        t = torch.from_numpy(storage).to(ctx.device)
        box = AmrBox(tuple(minc), tuple(maxc), t[sl])
    params = make_params(width, height, scalar_range, transparency, ref_dist, BOUNDS, color_map)
    samples = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    got = ctx.paint_box(box, transform, params, cam, samples=samples)
    ctx.synchronize()
    assert_bit_equal(got.cpu().numpy(), want, "paint_box")
    assert int(samples.item()) == want_n
    return want, want_n


def test_single_box_default_view(O, ctx):
    want, n = compare_box(O, ctx, radial(32, 32, 32), (0, 0, 0), (1, 1, 1),
                          scenes.default_camera(), 128, 128, ref_dist=0.5 / 32)
    assert n > 100000 and np.isfinite(want[..., 4]).sum() > 1000


@pytest.mark.parametrize("size", [(97, 61), (16, 16), (1, 1), (33, 200)])
def test_image_sizes_not_multiple_of_tile(O, ctx, size):
    compare_box(O, ctx, radial(16, 16, 16), (0, 0, 0), (1, 1, 1), scenes.default_camera(),
                size[0], size[1])


def test_non_cubic_box_general_division(O, ctx):
    # spacings 0.7/24, 0.45/20, 1.1/36 are not powers of two: exact IEEE-division indexing
    cells = radial(24, 20, 36)
    compare_box(O, ctx, cells, (0.1, 0.2, -0.3), (0.8, 0.65, 0.8), scenes.default_camera(),
                160, 120, ref_dist=0.01)


def test_strided_view_with_ghost_cells(O, ctx):
    # validBox inside a larger fab: lo offset (2,3,1) and j/k strides of the storage
    storage = np.random.default_rng(3).random((20, 22, 24))
    sl = (slice(1, 17), slice(3, 19), slice(2, 18))
    cells = np.ascontiguousarray(storage[sl])
    compare_box(O, ctx, cells, (0.25, 0.25, 0.25), (0.75, 0.75, 0.75), scenes.default_camera(),
                96, 96, cells_view=(storage, sl))


@pytest.mark.parametrize("sl", [
    (slice(0, 12), slice(0, 15), slice(0, 17)),   # odd nx on 16-byte aligned rows: the classify
    (slice(2, 9), slice(1, 14), slice(4, 21)),    # pass's paired loads end on a single cell
    (slice(0, 5), slice(0, 3), slice(0, 1)),      # one cell per row
    (slice(0, 16), slice(0, 16), slice(1, 17))])  # rows 8-byte aligned only: unpaired loads
def test_classify_row_tails_and_alignment(O, ctx, sl):
    storage = np.random.default_rng(11).random((16, 16, 24))
    cells = np.ascontiguousarray(storage[sl])
    compare_box(O, ctx, cells, (0.2, 0.1, 0.3), (0.9, 0.8, 0.85), scenes.default_camera(),
                80, 64, cells_view=(storage, sl))


def test_classify_chunk_that_starts_on_the_odd_last_cell(O, ctx):
    """nx = 129 on 16-byte aligned rows: the second 128-cell chunk of every row holds one cell,
    which the classify pass reads as the second half of the pair that ends the first chunk (the
    buffer addressing of classify_kernel starts that tile one cell early); nz = 6 leaves the
    last bricklet layer with two of its four planes."""
    storage = np.random.default_rng(19).random((6, 7, 130))
    storage[2, 3, 128] = np.nan
    storage[5, 6, 128] = np.inf
    sl = (slice(0, 6), slice(0, 7), slice(0, 129))
    cells = np.ascontiguousarray(storage[sl])
    compare_box(O, ctx, cells, (0.0, 0.3, 0.3), (1.0, 0.7, 0.65), scenes.default_camera(),
                96, 64, cells_view=(storage, sl))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_classify_random_box_shapes_and_views(O, ctx, seed):
    """Randomised shapes for the classify pass's addressing: nx from 1 to 300 (several 128-cell
    chunks, odd and even row ends, a chunk that starts on the row's last cell), ny / nz that leave
    partial bricklets, sub-boxes at random offsets of a larger fab (16-byte aligned rows or not,
    even or odd row pitch), non-finite cells, the standard and the general scalar transform."""
    rng = np.random.default_rng(4200 + seed)
    total = 0
    for case in range(14):
        nx = int(rng.choice([1, 2, 3, 7, 8, 9, 127, 128, 129, 130, 255, 257, 300,
                             int(rng.integers(1, 301))]))
        ny, nz = int(rng.integers(1, 10)), int(rng.integers(1, 10))
        pad = [int(v) for v in rng.integers(0, 4, size=6)]   # ghost cells: z lo/hi, y lo/hi, x lo/hi
        storage = rng.random((nz + pad[0] + pad[1], ny + pad[2] + pad[3], nx + pad[4] + pad[5]))
        holes = rng.integers(0, storage.size, size=max(storage.size // 200, 1))
        storage.reshape(-1)[holes] = rng.choice([np.nan, np.inf, -np.inf, -3.0, 7.5],
                                                 size=holes.size)
        sl = (slice(pad[0], pad[0] + nz), slice(pad[2], pad[2] + ny), slice(pad[4], pad[4] + nx))
        cells = np.ascontiguousarray(storage[sl])
        lo = (0.1, 0.35, 0.3)
        hi = (0.1 + 0.8 * min(nx, 64) / 64.0, 0.35 + 0.03 * ny, 0.3 + 0.04 * nz)
        general = case % 4 == 3
        _, n = compare_box(O, ctx, cells, lo, hi, scenes.default_camera(), 72, 48,
                           transform=ScalarTransform() if general else NORM,
                           scalar_range=(-0.2, 1.3) if general else (0.0, 1.0),
                           transparency=0.6, cells_view=(storage, sl))
        total += n
    assert total > 20000   # the boxes are in view: the classified cells were sampled


@pytest.mark.parametrize("cam", [
    CameraParameters((0.5, 0.5, 0.5), (0.9, 0.6, 0.1), (0, 1, 0), 60.0),       # eye inside the box
    CameraParameters((0.5, 0.5, 3.0), (0.5, 0.5, 0.5), (0, 1, 0), 30.0),       # axis aligned (dir ~ 0)
    CameraParameters((2.0, 1.0, 0.5), (3.0, 1.0, 0.5), (0, 1, 0), 45.0),       # box behind the eye
    CameraParameters((1.2, 1.0 + 1e-3, 0.5), (0.0, 1.0 + 1e-3, 0.5), (0, 1, 0), 50.0),  # grazing a face
    CameraParameters((0.5, 4.0, 0.5), (0.5, 0.0, 0.5), (0, 1, 0), 45.0),       # forward || up
])
def test_camera_placements(O, ctx, cam):
    compare_box(O, ctx, radial(24, 24, 24), (0, 0, 0), (1, 1, 1), cam, 80, 64)


@pytest.mark.parametrize("transparency", [0.0, 0.15, 0.97, 1.0])
@pytest.mark.parametrize("ref_scale", [1.0, 2.0, 4.0])
def test_transparency_and_level_factor(O, ctx, transparency, ref_scale):
    # ref_scale 2, 4 -> normalizationFactor 0.5, 0.25 (finer AMR levels)
    compare_box(O, ctx, radial(32, 32, 32), (0, 0, 0), (1, 1, 1), scenes.default_camera(), 72, 72,
                transparency=transparency, ref_dist=ref_scale * 0.5 / 32)


def test_lab_colormap(O, ctx):
    compare_box(O, ctx, radial(24, 24, 24), (0, 0, 0), (1, 1, 1), scenes.default_camera(), 64, 64,
                color_map=LAB_MAP, transparency=0.5)


def test_soft_clip_and_scalar_range(O, ctx):
    # scalarRange.second < 1 - 1e-5 activates saturateSoftTail (VolumePainter.cpp:723-724)
    compare_box(O, ctx, radial(24, 24, 24), (0, 0, 0), (1, 1, 1), scenes.default_camera(), 64, 64,
                scalar_range=(0.1, 0.6), transparency=0.3)


def test_unnormalised_transform_and_wide_range(O, ctx):
    cells = radial(20, 20, 20) * 8.0 - 2.0
    compare_box(O, ctx, cells, (0, 0, 0), (1, 1, 1), scenes.default_camera(), 64, 64,
                transform=ScalarTransform(), scalar_range=(-2.0, 6.0), transparency=0.5)


def test_nan_inf_negative_cells(O, ctx):
    cells = radial(16, 16, 16)
    rng = np.random.default_rng(11)
    flat = cells.reshape(-1)
    pick = rng.choice(flat.size, 600, replace=False)
    flat[pick[:200]] = np.nan
    flat[pick[200:300]] = np.inf
    flat[pick[300:400]] = -np.inf
    flat[pick[400:]] = -3.5
    compare_box(O, ctx, cells, (0, 0, 0), (1, 1, 1), scenes.default_camera(), 64, 64,
                transparency=0.6)


def test_log_scale_transform(O, ctx):
    """Log scaling runs std::log in double on the CPU and the device libm's log on the GPU;
    both are within 1 ulp(double) of the true value, and the value then passes through a
    normalise/clamp, a float cast and a floor to one of 256 table entries, so a differing bit
    can only appear when a sample lands within ~1e-16 (relative) of a table-entry boundary.
    Tolerance: at most 0.01 % of pixels may differ, none by more than 0.05 in any channel."""
    cells = np.exp(radial(24, 24, 24) * 6.0 - 3.0)
    cells.reshape(-1)[::37] = -1.0  # below the positive floor
    lo, hi = np.log(0.05), np.log(cells.max())
    tr = ScalarTransform(log_scale_input=True, normalize_to_unit_range=True, positive_floor=0.05,
                         normalization_min=lo, normalization_max=hi,
                         inverse_normalization_span=1.0 / (hi - lo))
    cam = scenes.default_camera()
    ob = O.make_box(cells, (0, 0, 0), (1, 1, 1))
    op = oracle_params(O, 96, 96, (0, 1), 0.5, 0.0, BOUNDS)
    want, want_n = O.paint_box(ob, oracle_transform(O, tr), op, oracle_camera(O, cam))
    samples = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    got = ctx.paint_box(device_box(ctx, cells, (0, 0, 0), (1, 1, 1)), tr,
                        make_params(96, 96, (0, 1), 0.5, 0.0, BOUNDS), cam, samples=samples)
    ctx.synchronize()
    got = got.cpu().numpy()
    assert int(samples.item()) == want_n
    differ = np.any(got.view(np.uint32) != want.view(np.uint32), axis=-1)
    assert differ.mean() <= 1e-4
    both = np.isfinite(want) & np.isfinite(got)
    assert np.array_equal(np.isfinite(want), np.isfinite(got))
    assert np.abs(got[both] - want[both]).max() <= 0.05


def test_empty_box_dimensions(O, ctx):
    # nx <= 0: the reference clears the layer (VolumePainter.cpp:670-673)
    box = AmrBox((0, 0, 0), (1, 1, 1), torch.zeros((4, 4, 0), dtype=torch.float64,
                                                   device=ctx.device))
    got = ctx.paint_box(box, NORM, make_params(32, 24), scenes.default_camera())
    ctx.synchronize()
    got = got.cpu().numpy()
    assert np.all(got[..., :4] == 0.0) and np.all(np.isposinf(got[..., 4]))


def test_invalid_arguments(ctx):
    box = AmrBox((0, 0, 0), (1, 1, 1), torch.zeros((4, 4, 4), dtype=torch.float64,
                                                   device=ctx.device))
    with pytest.raises(ValueError):
        ctx.paint_box(box, NORM, make_params(0, 16), scenes.default_camera(),
                      out=ctx.empty(16, 16, 5))
    with pytest.raises(ValueError):
        bad = AmrBox((0, 0, 0), (1, 1, 1), torch.zeros((4, 4, 4), dtype=torch.float32,
                                                       device=ctx.device))
        ctx.paint_box(bad, NORM, make_params(16, 16), scenes.default_camera())


# ---- fused multi-box path -------------------------------------------------------------------

def layered_oracle(O, spec, cells, cam, width, height, transparency, owners, n_ranks,
                   color_map=None):
    """Reference semantics: one layer per box, then DirectSend layered compose + gather."""
    oboxes = [O.make_box(c, m.min_corner, m.max_corner) for c, m in zip(cells, spec.boxes)]
    ref = O.reference_sample_distance(oboxes, spec.bounds.min_corner, spec.bounds.max_corner)
    op = oracle_params(O, width, height, spec.scalar_range, transparency, ref, spec.bounds,
                       color_map)
    ocam = oracle_camera(O, cam)
    otr = oracle_transform(O, spec.transform)
    layers, hints, total = [], [], 0
    for ob in oboxes:
        img, n = O.paint_box(ob, otr, op, ocam)
        layers.append(img)
        hints.append(O.box_depth_hint(ob, ocam))
        total += n
    local_index = np.zeros(len(layers), dtype=np.int32)
    for r in range(n_ranks):
        idx = [i for i, o in enumerate(owners) if o == r]
        local_index[idx] = np.arange(len(idx))
    out, _, runs = O.compose_layered(layers, hints, owners, local_index, n_ranks)
    return out, total, layers, hints


@pytest.mark.parametrize("field,levels,transparency", [
    ("radial", 1, 0.0), ("smooth", 2, 0.0), ("smooth", 2, 0.9), ("noise", 2, 0.97)])
def test_fused_single_rank_matches_layered_reference(O, ctx, field, levels, transparency):
    spec = scenes.make_amr_scene(32, levels, 8, field)
    cells = scene_cells(spec)
    cam = scenes.default_camera()
    W, H = 112, 80
    want, want_n, _, _ = layered_oracle(O, spec, cells, cam, W, H, transparency,
                                        [0] * len(cells), 1)
    boxes = [device_box(ctx, c, m.min_corner, m.max_corner, m.level)
             for c, m in zip(cells, spec.boxes)]
    scene = ctx.create_scene(boxes, spec.transform)
    ref = runtime.reference_sample_distance(boxes, spec.bounds.min_corner, spec.bounds.max_corner)
    params = make_params(W, H, spec.scalar_range, transparency, ref, spec.bounds)
    hints = [runtime.box_depth_hint(b, cam) for b in boxes]
    order, run_end = runtime.layer_order(hints, [0] * len(boxes), list(range(len(boxes))))
    assert run_end.tolist() == [len(boxes)]
    samples = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    got = scene.render_runs(params, cam, order, run_end, 1, samples=samples)
    ctx.synchronize()
    assert_bit_equal(got.cpu().numpy(), want, "fused frame")
    if transparency >= 0.9:
        # translucent regime: no ray saturates, so no box is skipped and the executed fetches
        # equal the reference's count
        assert int(samples.item()) == want_n
    else:
        assert 0 < int(samples.item()) <= want_n


def test_fused_orbit_views(O, ctx):
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    cells = scene_cells(spec)
    boxes = [device_box(ctx, c, m.min_corner, m.max_corner, m.level)
             for c, m in zip(cells, spec.boxes)]
    scene = ctx.create_scene(boxes, spec.transform)
    ref = runtime.reference_sample_distance(boxes, spec.bounds.min_corner, spec.bounds.max_corner)
    W, H = 64, 64
    params = make_params(W, H, spec.scalar_range, 0.8, ref, spec.bounds)
    for view in (1, 6, 11):
        cam = scenes.orbit_camera(view)
        want, want_n, _, _ = layered_oracle(O, spec, cells, cam, W, H, 0.8, [0] * len(cells), 1)
        hints = [runtime.box_depth_hint(b, cam) for b in boxes]
        order, run_end = runtime.layer_order(hints, [0] * len(boxes), list(range(len(boxes))))
        got = scene.render_runs(params, cam, order, run_end, 1)
        ctx.synchronize()
        assert_bit_equal(got.cpu().numpy(), want, f"orbit view {view}")
