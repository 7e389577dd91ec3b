"""AMReX plotfile ingestion without AMReX -- reader / writer of the
HyperCLaw-V1.1 format, our convexify, the scene geometry of BuildSceneGeometry, the automatic
camera, and render(plotfile=...) end to end against the oracle."""
import os

import numpy as np
import pytest

from amrvolumerenderer_amd import api, plotfile as pf, scenes
from amrvolumerenderer_amd.types import CameraParameters, VolumeBounds


def two_level_scene(rng, n0=16, fine=((4, 4, 4), (11, 11, 11))):
    """Level 0: n0^3 in 8 grids; level 1 (ratio 2) refines `fine` (level-0 indices)."""
    half = n0 // 2
    l0_boxes = [((i * half, j * half, k * half), (i * half + half - 1, j * half + half - 1,
                                                   k * half + half - 1))
                for k in range(2) for j in range(2) for i in range(2)]
    flo = tuple(2 * v for v in fine[0])
    fhi = tuple(2 * v + 1 for v in fine[1])
    mid = (flo[0] + fhi[0] + 1) // 2
    l1_boxes = [(flo, (mid - 1, fhi[1], fhi[2])), ((mid, flo[1], flo[2]), fhi)]

    def data(box, level, n_comp=2):
        lo, hi = box
        n = n0 * (2 ** level)
        z, y, x = np.meshgrid(*[(np.arange(lo[a], hi[a] + 1) + 0.5) / n for a in (2, 1, 0)],
                              indexing="ij")
        density = 0.2 + x * x + 0.5 * y + 0.25 * np.sin(6.0 * z) + 0.05 * level
        other = rng.random(density.shape)
        return np.stack([density, other][:n_comp])

    levels = [{"domain": ((0, 0, 0), (n0 - 1,) * 3), "boxes": l0_boxes,
               "data": [data(b, 0) for b in l0_boxes]},
              {"domain": ((0, 0, 0), (2 * n0 - 1,) * 3), "boxes": l1_boxes,
               "data": [data(b, 1) for b in l1_boxes]}]
    return levels


def test_write_read_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    levels = two_level_scene(rng)
    for real_bytes in (8, 4):
        path = str(tmp_path / f"plt{real_bytes}")
        pf.write_plotfile(path, ["density", "noise"], levels, (0.0, -1.0, 2.0), (4.0, 3.0, 6.0), [2],
                          time=1.5, grids_per_file=3, real_bytes=real_bytes)
        plot = pf.PlotFileData(path)
        assert plot.var_names == ["density", "noise"] and plot.space_dim == 3
        assert plot.finest_level == 1 and plot.ref_ratio == [2] and plot.time == 1.5
        assert plot.prob_lo == (0.0, -1.0, 2.0) and plot.prob_hi == (4.0, 3.0, 6.0)
        assert plot.cell_size[0] == (0.25, 0.25, 0.25) and plot.cell_size[1] == (0.125,) * 3
        for level, lev in enumerate(levels):
            assert plot.boxes(level) == lev["boxes"]
            for comp, name in enumerate(plot.var_names):
                got = plot.get(level, name)
                for g, want in enumerate(lev["data"]):
                    expected = want[comp] if real_bytes == 8 else want[comp].astype(np.float32)
                    assert np.array_equal(got[g], expected.astype(np.float64))
        only = plot.get(0, "noise", grids=[5, 2])
        assert sorted(only) == [2, 5]
        with pytest.raises(RuntimeError):
            plot.get(0, "pressure")
    with pytest.raises(RuntimeError):
        pf.PlotFileData(str(tmp_path / "missing"))


AMREX_STYLE_HEADER = """HyperCLaw-V1.1
1
phi
3
0.25
0
0 0 0 
1 1 1 

((0,0,0) (7,7,7) (0,0,0)) 
0 
0.125 0.125 0.125 
0
0
0 2 0.25
0
0 0.5
0 1
0 1
0.5 1
0 1
0 1
Level_0/Cell
"""

AMREX_STYLE_CELL_H = """1
1
1
(0,0,0)
(2 0
((0,0,0) (3,7,7) (0,0,0))
((4,0,0) (7,7,7) (0,0,0))
)
2
FabOnDisk: Cell_D_00000 0
FabOnDisk: Cell_D_00001 0

2,1
1.0,
2.0,

2,1
3.0,
4.0,

"""


def test_reads_files_laid_out_as_amrex_writes_them(tmp_path):
    """Header with the empty refinement-ratio line of a single-level file, vector ghost spec,
    one FAB per file (the VisMF 'how' = 1 layout)."""
    path = tmp_path / "plt00000"
    (path / "Level_0").mkdir(parents=True)
    (path / "Header").write_text(AMREX_STYLE_HEADER)
    (path / "Level_0" / "Cell_H").write_text(AMREX_STYLE_CELL_H)
    rng = np.random.default_rng(2)
    want = []
    for g, box in enumerate(["((0,0,0) (3,7,7) (0,0,0))", "((4,0,0) (7,7,7) (0,0,0))"]):
        cells = rng.random((8, 8, 4))
        want.append(cells)
        head = f"FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1))){box} 1\n".encode()
        (path / "Level_0" / f"Cell_D_0000{g}").write_bytes(head + cells.astype("<f8").tobytes())
    plot = pf.PlotFileData(str(path))
    assert plot.finest_level == 0 and plot.ref_ratio == [] and plot.var_names == ["phi"]
    assert plot.cell_size == [(0.125, 0.125, 0.125)]
    got = plot.get(0, "phi")
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def box_cells(box):
    lo, hi = box
    return {(i, j, k) for i in range(lo[0], hi[0] + 1) for j in range(lo[1], hi[1] + 1)
            for k in range(lo[2], hi[2] + 1)}


def test_box_diff_is_an_exact_disjoint_cover():
    rng = np.random.default_rng(3)
    for _ in range(200):
        lo1, lo2 = rng.integers(-3, 4, 3), rng.integers(-3, 4, 3)
        b1 = (tuple(int(v) for v in lo1), tuple(int(v) for v in lo1 + rng.integers(0, 5, 3)))
        b2 = (tuple(int(v) for v in lo2), tuple(int(v) for v in lo2 + rng.integers(0, 5, 3)))
        parts = pf.box_diff(b1, b2)
        cells = [box_cells(p) for p in parts]
        assert sum(len(c) for c in cells) == len(set().union(*cells)) if cells else True
        assert (set().union(*cells) if cells else set()) == box_cells(b1) - box_cells(b2)
    # amrex::boxDiff's order for a box strictly inside: z slabs, then y, then x
    parts = pf.box_diff(((0, 0, 0), (7, 7, 7)), ((2, 2, 2), (5, 5, 5)))
    assert parts == [((0, 0, 0), (7, 7, 1)), ((0, 0, 6), (7, 7, 7)), ((0, 0, 2), (7, 1, 5)),
                     ((0, 6, 2), (7, 7, 5)), ((0, 2, 2), (1, 5, 5)), ((6, 2, 2), (7, 5, 5))]


def test_convexify_tiles_the_domain_exactly_once():
    levels = two_level_scene(np.random.default_rng(4))
    convex = pf.convexify([lev["boxes"] for lev in levels], [2])
    fine = set().union(*[box_cells(b) for b in levels[1]["boxes"]])
    coarse_kept = [box_cells(b) for _, b in convex[0]]
    assert sum(len(c) for c in coarse_kept) == len(set().union(*coarse_kept))  # disjoint
    covered = {(i // 2, j // 2, k // 2) for i, j, k in fine}
    assert set().union(*coarse_kept) == box_cells(levels[0]["domain"]) - covered
    assert [b for _, b in convex[1]] == levels[1]["boxes"]  # the finest level is kept whole
    for parent, part in convex[0]:  # every part lies inside its parent grid
        assert box_cells(part) <= box_cells(levels[0]["boxes"][parent])
    # single level: untouched
    assert pf.convexify([levels[0]["boxes"]], []) == [list(enumerate(levels[0]["boxes"]))]


def test_mt19937_known_answer_and_automatic_camera():
    rng = api._Mt19937(5489)  # [rand.predef]: the 10000th value of a default mt19937
    for _ in range(9999):
        rng()
    assert rng() == 4123659995
    bounds = VolumeBounds((-0.05, -0.05, -0.05), (1.05, 1.05, 1.05))
    cam = api.automatic_camera(bounds)
    assert cam == api.automatic_camera(bounds) and cam != api.automatic_camera(bounds, 7)
    assert cam.look_at == (0.5, 0.5, 0.5) and cam.near_plane == float(np.float32(0.1))
    assert abs(cam.fov_y_degrees - 45.0) < 1e-4
    radius = np.sqrt(3 * 0.55 ** 2)
    distance = radius / np.tan(np.pi / 8) + max(0.25 * radius, 0.5)
    offset = np.array(cam.eye) - 0.5
    assert abs(np.linalg.norm(offset) - distance) < 1e-4 * distance
    assert abs(np.arcsin(offset[1] / np.linalg.norm(offset))) <= np.pi / 4 + 1e-6
    assert abs(cam.far_plane - 4.0 * distance) < 1e-3
    # a view along the default up vector switches the up vector (VolumeRenderer.cpp:1006-1013)
    assert cam.up == (0.0, 1.0, 0.0)


def test_render_argument_checks(tmp_path):
    with pytest.raises(RuntimeError):
        api.render(str(tmp_path / "nowhere"))
    with pytest.raises(ValueError):
        api.render("x", camera_eye=(0, 0, 1))  # look-at missing
    with pytest.raises(ValueError):
        api.render("x", camera_eye=(0, 0, 1), camera_look_at=(0, 0, 1))  # not distinct
    with pytest.raises(ValueError):
        api.render("x", camera_eye=(0, 0, 1), camera_look_at=(0, 0, 0), camera_up=(0, 0, 2))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto_camera", "explicit_log_override"])
def test_render_plotfile_end_to_end(O, ctx, tmp_path, mode):
    import torch
    from test_frame_plan import oracle_camera, oracle_params, oracle_transform
    rng = np.random.default_rng(9)
    levels = two_level_scene(rng)
    path = str(tmp_path / "plt00010")
    prob_lo, prob_hi = (0.0, 0.0, 0.0), (2.0, 2.0, 2.0)   # shortest edge 2 -> global scale 0.5
    pf.write_plotfile(path, ["density", "noise"], levels, prob_lo, prob_hi, [2])
    W, H = 80, 56
    out = str(tmp_path / "frame.ppm")
    kwargs = dict(width=W, height=H, box_transparency=0.5, variable="density", output=out)
    log_scale = False
    color_map = None
    if mode == "explicit_log_override":
        log_scale = True
        color_map = [(0.25, 0.0, 0.0, 1.0, 0.1), (0.8, 0.0, 1.0, 0.0, 0.5), (1.6, 1.0, 0.0, 0.0, 0.9)]
        kwargs.update(log_scale=True, scalar_range=(0.3, 1.5), color_map=color_map,
                      camera_eye=(2.4, 1.7, 2.2), camera_look_at=(0.5, 0.5, 0.5),
                      camera_up=(0.0, 2.0, 0.0), camera_fov_y=40.0, camera_near=0.05,
                      camera_far=30.0)
    assert api.render(path, **kwargs) == 0

    # ---- the same frame from the oracle, built independently of the product's scene objects ----
    convex = pf.convexify([lev["boxes"] for lev in levels], [2])
    scale = 0.5
    oboxes, all_cells = [], []
    for level, lev in enumerate(levels):
        dx = 2.0 / (16 * 2 ** level)
        for parent, (lo, hi) in convex[level]:
            glo = lev["boxes"][parent][0]
            grid = lev["data"][parent][0]
            cells = np.ascontiguousarray(
                grid[lo[2] - glo[2]:hi[2] - glo[2] + 1, lo[1] - glo[1]:hi[1] - glo[1] + 1,
                     lo[0] - glo[0]:hi[0] - glo[0] + 1])
            all_cells.append(cells)
            oboxes.append(O.make_box(cells, tuple(lo[a] * dx * scale for a in range(3)),
                                     tuple((hi[a] + 1) * dx * scale for a in range(3))))
    lo_v = min(c.min() for c in all_cells)
    hi_v = max(c.max() for c in all_cells)
    pad = 0.05
    bounds = VolumeBounds((-pad,) * 3, (1.0 + pad,) * 3)
    if mode == "auto_camera":
        cam = api.automatic_camera(bounds)
        transform = scenes.ScalarTransform(normalize_to_unit_range=True, normalization_min=lo_v,
                                           inverse_normalization_span=1.0 / (hi_v - lo_v))
        ocmap = None
    else:
        cam = CameraParameters((2.4, 1.7, 2.2), (0.5, 0.5, 0.5), (0.0, 1.0, 0.0), 40.0, 0.05, 30.0)
        f32 = np.float32
        nmin, nmax = f32(np.log(f32(0.3))), f32(np.log(f32(1.5)))
        transform = scenes.ScalarTransform(
            log_scale_input=True, positive_floor=min(c[c > 0].min() for c in all_cells),
            normalize_to_unit_range=True, normalization_min=float(nmin),
            inverse_normalization_span=1.0 / (float(nmax) - float(nmin)))
        ocmap = [((np.clip(f32(f32(f32(np.log(f32(v))) - nmin) / f32(nmax - nmin)), 0, 1)), r, g, b,
                  a) for v, r, g, b, a in color_map]
    ref = O.reference_sample_distance(oboxes, bounds.min_corner, bounds.max_corner)
    op = oracle_params(O, W, H, (0.0, 1.0), 0.5, ref, bounds, ocmap)
    ocam, otr = oracle_camera(O, cam), oracle_transform(O, transform)
    layers = [O.paint_box(ob, otr, op, ocam)[0] for ob in oboxes]
    hints = [O.box_depth_hint(ob, ocam) for ob in oboxes]
    want, _, _ = O.compose_layered(layers, hints, [0] * len(layers), np.arange(len(layers)), 1)
    tight = O.tight_bounds(oboxes, bounds.min_corner, bounds.max_corner)
    want = O.bbox_overlay(want, W, H, tight[0], tight[1], ocam, 1).reshape(-1, 5)
    data = open(out, "rb").read()
    header = f"P6\n{W} {H}\n255\n".encode()
    assert data.startswith(header)
    got = np.frombuffer(data[len(header):], np.uint8).reshape(H, W, 3)
    want8 = O.quantize_rgb8(want, W, H)
    if log_scale:
        # device log vs glibc log (Tolerances): a cell on a table-bin edge may move
        assert (got != want8).any(axis=2).mean() <= 1e-3
    else:
        assert np.array_equal(got, want8)
    assert want8.any()


@pytest.mark.gpu
def test_example_script_runs(tmp_path):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "examples", "render_plotfile.py")
    out = tmp_path / "frame.png"
    subprocess.run([sys.executable, script, "--size", "64", "--output", str(out)], check=True,
                   cwd=tmp_path)
    assert out.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"
    subprocess.run([sys.executable, script, "--size", "48", "--orbit", "3", "--output",
                    str(tmp_path / "orbit.ppm")], check=True, cwd=tmp_path)
    assert all((tmp_path / f"orbit_{v:03d}.ppm").exists() for v in range(3))


@pytest.mark.gpu
def test_compute_histogram_of_a_plotfile(O, ctx, tmp_path):
    """The python module's second entry: counts of the uncovered cells of both levels in bins of
    the normalised scalar."""
    rng = np.random.default_rng(21)
    levels = two_level_scene(rng)
    path = str(tmp_path / "plt_hist")
    pf.write_plotfile(path, ["density", "noise"], levels, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), [2])
    result = api.compute_histogram(path, variable="noise", bins=64, ctx=ctx)
    convex = pf.convexify([lev["boxes"] for lev in levels], [2])
    values = []
    for level, lev in enumerate(levels):
        for parent, (lo, hi) in convex[level]:
            glo = lev["boxes"][parent][0]
            values.append(lev["data"][parent][1][lo[2] - glo[2]:hi[2] - glo[2] + 1,
                                                 lo[1] - glo[1]:hi[1] - glo[1] + 1,
                                                 lo[0] - glo[0]:hi[0] - glo[0] + 1].reshape(-1))
    values = np.concatenate(values)
    assert result["samples"] == values.size == int(result["counts"].sum())
    lo_v, hi_v = values.min(), values.max()
    assert result["normalized_range"] == (0.0, 1.0)
    assert result["original_range"] == (float(np.float32(lo_v)), float(np.float32(hi_v)))
    oboxes = [O.make_box(np.ascontiguousarray(values.reshape(1, 1, -1)), (0, 0, 0), (1, 1, 1))]
    want = O.histogram(oboxes, O.make_transform(normalize=True, norm_min=lo_v,
                                                inv_norm_span=1.0 / (hi_v - lo_v)), 0.0, 1.0, 64)
    assert np.array_equal(result["counts"], want)
    with pytest.raises(ValueError):
        api.compute_histogram(path, bins=0, ctx=ctx)


@pytest.mark.gpu
def test_package_entry_points_with_an_initialised_runtime(tmp_path):
    """import amrvolumerenderer_amd as avr; avr.initialize_runtime(); avr.render(...);
    avr.compute_histogram(...); avr.finalize_runtime() -- the reference package's surface."""
    import amrvolumerenderer_amd as avr
    levels = two_level_scene(np.random.default_rng(2))
    path = str(tmp_path / "plt_pkg")
    pf.write_plotfile(path, ["density", "noise"], levels, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), [2])
    with pytest.raises(RuntimeError):
        avr.finalize_runtime()  # without a matching initialize_runtime
    avr.initialize_runtime()
    avr.initialize_runtime()    # reference counted
    try:
        out = str(tmp_path / "a.ppm")
        assert avr.render(path, width=48, height=40, output=out) == 0
        assert os.path.getsize(out) == len(b"P6\n48 40\n255\n") + 48 * 40 * 3
        hist = avr.compute_histogram(path, bins=32)
        assert hist["samples"] == int(hist["counts"].sum()) > 0
    finally:
        avr.finalize_runtime()
        avr.finalize_runtime()


@pytest.mark.gpu
def test_render_amr_data_equals_render_of_the_same_plotfile(ctx, tmp_path):
    """api::Render(AmrData): in-memory levels (a multi-component array per grid) give the same
    image as the plotfile holding the same data."""
    rng = np.random.default_rng(4)
    levels = two_level_scene(rng)
    path = str(tmp_path / "plt_mem")
    pf.write_plotfile(path, ["density", "noise"], levels, (0.0, 0.0, 0.0), (2.0, 2.0, 2.0), [2])
    cam = CameraParameters((2.4, 1.7, 2.2), (0.5, 0.5, 0.5), (0.0, 1.0, 0.0), 40.0, 0.05, 30.0)
    a, b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    assert api.render(path, width=72, height=48, box_transparency=0.4, variable="noise", output=a,
                      camera_eye=cam.eye, camera_look_at=cam.look_at, camera_fov_y=40.0,
                      camera_near=0.05, camera_far=30.0) == 0
    data = api.AmrData([lev["boxes"] for lev in levels], [lev["data"] for lev in levels],
                       (0.0, 0.0, 0.0), [(2.0 / 16,) * 3, (2.0 / 32,) * 3], [2])
    options = api.RenderOptions(width=72, height=48, box_transparency=0.4, component=1,
                                camera=cam, output_filename=b)
    assert api.render_amr_data(data, options, ctx=ctx) == 0
    assert open(a, "rb").read() == open(b, "rb").read()
    with pytest.raises(ValueError):
        api.render_amr_data(data, api.RenderOptions(component=5, camera=cam, output_filename=b),
                            ctx=ctx)
    with pytest.raises(ValueError):  # no ratio for the level transition
        api.render_amr_data(api.AmrData(data.level_boxes, data.level_data, data.prob_lo,
                                        data.cell_sizes, []), options, ctx=ctx)
