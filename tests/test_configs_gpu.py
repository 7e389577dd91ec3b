"""Parity at the stated size of every BASELINE.json configuration other than config-4 (which is
tests/test_full_size_gpu.py): the HIP path against the oracle, bit for bit.

  config-1  insitu_example: one level of 64^3 in 8 boxes of 32^3, field x^2+y^2+z^2 with
            x = i/(n-1) (VolumeRenderer/Examples/RenderFromMultiFab.cpp:20-58, scaled per
            SURVEY.md 8d), 256 x 256, through api.render_amr_data -> the PPM's bytes
  config-2  one level of 512^3 in 64 boxes of 128^3, 1024 x 1024 (the painter alone)
  config-3  3-level AMR, 256^3 base, 176 boxes of 64^3, 2048 x 2048: one rank, and eight simulated
            ranks whose group order comes from the visibility ordering
  config-5  4-level AMR, 1024^3 base, 1856 boxes of 128^3 (31.1 GB of f64 cells), 4096 x 4096 with
            antialiasing 4 (marched at 8192^2).  The reference's layer-per-box frame does not fit
            anywhere (1856 x 1.34 GB), so the oracle paints crops of the supersampled frame for
            ALL boxes (orc_paint_box_window) and the crops of march + downsampleImage + 8-bit
            conversion are compared bit for bit, as one rank and as eight simulated ranks, next
            to full-size properties (sample counts).
"""
import os

import numpy as np
import pytest
import torch

from amrvolumerenderer_amd import api, runtime, scenes
from amrvolumerenderer_amd.compositor import FramePlan
from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters
from amrvolumerenderer_amd.types import AmrBox, VolumeBounds, make_params

import plan_helpers as PH
from helpers import oracle_camera, oracle_params, oracle_transform
from test_frame_plan import local_indices

pytestmark = pytest.mark.gpu
THREADS = min(os.cpu_count() or 1, 16)


def device_scene(ctx, spec):
    """Cells generated on the device; the SAME bits are handed to the oracle."""
    device_cells = [scenes.box_cells_torch(spec, i, ctx.device) for i in range(len(spec.boxes))]
    torch.cuda.synchronize()
    host_cells = [c.cpu().numpy() for c in device_cells]
    return device_cells, host_cells


def oracle_layers(O, spec, host_cells, cam, W, H, transparency):
    oboxes = [O.make_box(c, m.min_corner, m.max_corner) for c, m in zip(host_cells, spec.boxes)]
    ref = O.reference_sample_distance(oboxes, spec.bounds.min_corner, spec.bounds.max_corner)
    op = oracle_params(O, W, H, spec.scalar_range, transparency, ref, spec.bounds)
    ocam, otr = oracle_camera(O, cam), oracle_transform(O, spec.transform)
    layers, samples = [], 0
    for ob in oboxes:
        layer, n = O.paint_box(ob, otr, op, ocam, threads=THREADS)
        layers.append(layer)
        samples += n
    hints = [O.box_depth_hint(ob, ocam) for ob in oboxes]
    return layers, hints, ref, samples


def single_rank_frame(ctx, spec, device_cells, cam, W, H, transparency):
    meta = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    local = [AmrBox(m.min_corner, m.max_corner, c, m.level)
             for c, m in zip(device_cells, spec.boxes)]
    renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range)
    counter = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    image, rgb8 = renderer.render(RenderParameters(W, H, transparency, 1, draw_bounds=False), cam,
                                  samples=counter, want_image=True)
    renderer.synchronize()
    return renderer, image.cpu().numpy().reshape(-1, 5), rgb8.cpu().numpy(), int(counter.item())


# ---- config-1 ----------------------------------------------------------------------------------

def test_config1_insitu_example_through_the_public_api(O, ctx, tmp_path):
    n, g, W, H = 64, 32, 256, 256
    axis = np.arange(n, dtype=np.float64) / (n - 1)   # RenderFromMultiFab.cpp:42-44
    field = (axis[None, None, :] ** 2 + axis[None, :, None] ** 2) + axis[:, None, None] ** 2
    boxes, grids = [], []
    for bz in range(n // g):          # BoxArray::maxSize order: x fastest
        for by in range(n // g):
            for bx in range(n // g):
                lo = (bx * g, by * g, bz * g)
                hi = tuple(v + g - 1 for v in lo)
                boxes.append((lo, hi))
                grids.append(np.ascontiguousarray(
                    field[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1]))
    data = api.AmrData([boxes], [grids], (0.0, 0.0, 0.0), [(1.0 / n,) * 3], [])
    out = str(tmp_path / "multifab-render.ppm")
    options = api.RenderOptions(width=W, height=H, output_filename=out)   # the example's defaults
    assert api.render_amr_data(data, options, ctx=ctx) == 0

    # the oracle's chain, built from the same cells without the product's scene objects
    h = 1.0 / n
    oboxes = [O.make_box(c, tuple(v * h for v in lo), tuple((v + 1) * h for v in hi))
              for c, (lo, hi) in zip(grids, boxes)]
    bounds = VolumeBounds((-0.05,) * 3, (1.05,) * 3)
    lo_v, hi_v = float(field.min()), float(field.max())
    otr = O.make_transform(normalize=True, norm_min=lo_v, inv_norm_span=1.0 / (hi_v - lo_v))
    cam = api.automatic_camera(bounds)
    ocam = oracle_camera(O, cam)
    ref = O.reference_sample_distance(oboxes, bounds.min_corner, bounds.max_corner)
    op = oracle_params(O, W, H, (0.0, 1.0), 0.0, ref, bounds)
    layers = [O.paint_box(ob, otr, op, ocam, threads=THREADS)[0] for ob in oboxes]
    hints = [O.box_depth_hint(ob, ocam) for ob in oboxes]
    want, _, _ = O.compose_layered(layers, hints, [0] * 8, np.arange(8), 1)
    tight = O.tight_bounds(oboxes, bounds.min_corner, bounds.max_corner)
    want = O.bbox_overlay(want, W, H, tight[0], tight[1], ocam, 1).reshape(-1, 5)
    blob = open(out, "rb").read()
    header = f"P6\n{W} {H}\n255\n".encode()
    assert blob.startswith(header)
    got = np.frombuffer(blob[len(header):], np.uint8).reshape(H, W, 3)
    want8 = O.quantize_rgb8(want, W, H)
    assert np.array_equal(got, want8)
    assert (want8.reshape(-1, 3).max(axis=1) > 0).mean() > 0.2   # the cube is on screen


# ---- config-2 ----------------------------------------------------------------------------------

def test_config2_uniform_512_at_1024(O, ctx):
    spec = scenes.config2("smooth")
    cam = scenes.default_camera()
    W = H = 1024
    device_cells, host_cells = device_scene(ctx, spec)
    layers, hints, ref, samples = oracle_layers(O, spec, host_cells, cam, W, H, 0.97)
    n = len(layers)
    assert n == 64
    want, _, _ = O.compose_layered(layers, hints, [0] * n, np.arange(n), 1)
    renderer, got, rgb8, counted = single_rank_frame(ctx, spec, device_cells, cam, W, H, 0.97)
    assert np.float32(renderer.reference_sample_distance) == np.float32(ref)
    assert counted == samples and samples > 100_000_000
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(rgb8, O.quantize_rgb8(want, W, H))
    # VolumePainter alone: one box, one full layer, through avr_paint_box
    params = make_params(W, H, spec.scalar_range, 0.97, ref, spec.bounds)
    for index in (0, 21, 63):
        m = spec.boxes[index]
        box = AmrBox(m.min_corner, m.max_corner, device_cells[index], m.level)
        layer = ctx.paint_box(box, spec.transform, params, cam)
        ctx.synchronize()
        assert np.array_equal(layer.cpu().numpy().reshape(-1).view(np.uint32),
                              layers[index].reshape(-1).view(np.uint32))


# ---- config-3 ----------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def config3(O, ctx):
    spec = scenes.config3("smooth")
    cam = scenes.default_camera()
    device_cells, host_cells = device_scene(ctx, spec)
    layers, hints, ref, samples = oracle_layers(O, spec, host_cells, cam, 2048, 2048, 0.97)
    return spec, cam, device_cells, layers, hints, ref, samples


def test_config3_single_rank(O, ctx, config3):
    spec, cam, device_cells, layers, hints, ref, samples = config3
    W = H = 2048
    n = len(layers)
    assert n == 176
    want, _, _ = O.compose_layered(layers, hints, [0] * n, np.arange(n), 1)
    renderer, got, rgb8, counted = single_rank_frame(ctx, spec, device_cells, cam, W, H, 0.97)
    assert counted == samples
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(rgb8, O.quantize_rgb8(want, W, H))


def test_config3_eight_ranks_in_visibility_order(O, ctx, config3):
    spec, cam, device_cells, layers, hints, ref, samples = config3
    W = H = 2048
    n_ranks = 8
    scenes.assign_owners(spec, n_ranks, "morton")
    try:
        owners = [b.owner for b in spec.boxes]
        meta = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
        graph = runtime.VisibilityGraph(meta, n_ranks)
        group = graph.order(cam, 1.0, True)
        assert sorted(group) == list(range(n_ranks))
        want, piece_owner, _ = O.compose_layered(layers, hints, owners,
                                                 local_indices(owners, n_ranks), n_ranks,
                                                 group_order=group)
        params = make_params(W, H, spec.scalar_range, 0.97, ref, spec.bounds)
        plans, sends, counted = [], [], 0
        for r in range(n_ranks):
            plan = FramePlan(meta, params, cam, r, n_ranks, group)
            local = [AmrBox(spec.boxes[i].min_corner, spec.boxes[i].max_corner, device_cells[i],
                            spec.boxes[i].level, owner=r)
                     for i in scenes.local_box_indices(spec, r)]
            scene = ctx.create_scene(local, spec.transform)
            counter = torch.zeros(1, dtype=torch.int64, device=ctx.device)
            send = scene.render_plan(plan, samples=counter)
            ctx.synchronize()
            counted += int(counter.item())
            plans.append(plan)
            sends.append(send[:plan.send_floats].cpu().numpy())
        assert counted == samples     # the ranks' shares add up to the one-rank frame
        got = np.zeros((W * H, 5), np.float32)
        for plan, recv in zip(plans, PH.route(plans, sends)):
            dev = torch.from_numpy(np.ascontiguousarray(recv)).to(ctx.device)
            if dev.numel() == 0:
                dev = torch.zeros(1, device=ctx.device)
            piece, _ = ctx.fold_plan(plan, dev)
            ctx.synchronize()
            got[plan.piece_begin:plan.piece_end] = piece.cpu().numpy()
            # the rank at group position k holds piece k
            assert (piece_owner[plan.piece_begin:plan.piece_end] == plan.rank).all()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    finally:
        scenes.assign_owners(spec, 1, "morton")


# ---- config-5 ----------------------------------------------------------------------------------

# crops of the 4096 x 4096 output: (x0, y0, side) -- the image centre (all four levels behind one
# another), the silhouette of the domain, and two seams between refinement levels
CROPS5 = [(1984, 1984, 128), (1000, 2300, 96), (2450, 1650, 96), (1700, 2500, 96), (3000, 900, 64)]


def test_config5_crops_of_the_antialiased_frame(O, ctx):
    spec = scenes.config5("smooth")
    cam = scenes.default_camera()
    W = H = 4096
    AA, B = 4, 2
    SW, SH = W * B, H * B
    n = len(spec.boxes)
    assert n == 1856
    device_cells = [scenes.box_cells_torch(spec, i, ctx.device) for i in range(n)]
    torch.cuda.synchronize()

    # ---- oracle: crops of the supersampled frame for all boxes -------------------------------
    meta = [scenes.metadata_box(spec, i) for i in range(n)]
    ref = runtime.reference_sample_distance(meta, spec.bounds.min_corner, spec.bounds.max_corner)
    op = oracle_params(O, SW, SH, spec.scalar_range, 0.97, ref, spec.bounds)
    ocam, otr = oracle_camera(O, cam), oracle_transform(O, spec.transform)
    crop_layers = [[None] * n for _ in CROPS5]
    hints = [0.0] * n
    crop_samples = 0
    for i in range(n):
        cells = device_cells[i].cpu().numpy()      # the same bits on both sides
        ob = O.make_box(cells, spec.boxes[i].min_corner, spec.boxes[i].max_corner)
        hints[i] = O.box_depth_hint(ob, ocam)
        for c, (x0, y0, side) in enumerate(CROPS5):
            layer, ns = O.paint_box_window(ob, otr, op, ocam, x0 * B, y0 * B, (x0 + side) * B,
                                           (y0 + side) * B, threads=THREADS)
            crop_layers[c][i] = layer
            crop_samples += ns
    assert crop_samples > 50_000_000

    def oracle_crop(c, owners, n_ranks):
        side = CROPS5[c][2]
        full, _, _ = O.compose_layered(crop_layers[c], hints, owners,
                                       local_indices(owners, n_ranks), n_ranks)
        small = O.downsample(full, side, side, B)
        return small.reshape(side, side, 5), O.quantize_rgb8(small, side, side)

    def check(image, rgb8, owners, n_ranks, what):
        # image [H, W, 5] (origin bottom-left), rgb8 [H, W, 3] rows top-down
        for c, (x0, y0, side) in enumerate(CROPS5):
            want, want8 = oracle_crop(c, owners, n_ranks)
            got = image[y0:y0 + side, x0:x0 + side].cpu().numpy()
            assert np.array_equal(np.ascontiguousarray(got).view(np.uint32),
                                  np.ascontiguousarray(want).view(np.uint32)), (what, c)
            got8 = rgb8[H - (y0 + side):H - y0, x0:x0 + side].cpu().numpy()
            assert np.array_equal(got8, want8), (what, c)
            if c == 0:
                assert want8.any()

    # ---- one rank: the frame bench.py --config config5 times ---------------------------------
    local = [AmrBox(m.min_corner, m.max_corner, cells, m.level)
             for cells, m in zip(device_cells, spec.boxes)]
    renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range)
    assert np.float32(renderer.reference_sample_distance) == np.float32(ref)
    counter = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    image, rgb8 = renderer.render(RenderParameters(W, H, 0.97, AA, draw_bounds=False), cam,
                                  samples=counter, want_image=True)
    renderer.synchronize()
    one_rank_samples = int(counter.item())
    assert one_rank_samples > 20_000_000_000
    check(image, rgb8, [0] * n, 1, "one rank")
    del renderer, image, rgb8

    # ---- eight simulated ranks: exchange routed on the device, fold, gather, downsample -------
    n_ranks = 8
    scenes.assign_owners(spec, n_ranks, "morton")
    try:
        owners = [b.owner for b in spec.boxes]
        meta = [scenes.metadata_box(spec, i) for i in range(n)]
        params = make_params(SW, SH, spec.scalar_range, 0.97, ref, spec.bounds)
        plans, sends, counted = [], [], 0
        for r in range(n_ranks):
            plan = FramePlan(meta, params, cam, r, n_ranks)
            mine = [AmrBox(spec.boxes[i].min_corner, spec.boxes[i].max_corner, device_cells[i],
                           spec.boxes[i].level, owner=r)
                    for i in scenes.local_box_indices(spec, r)]
            scene = ctx.create_scene(mine, spec.transform)
            counter.zero_()
            send = scene.render_plan(plan, samples=counter)
            ctx.synchronize()
            counted += int(counter.item())
            plans.append(plan)
            sends.append(send)
            del scene
        assert counted == one_rank_samples
        full = torch.empty(SW * SH, 5, device=ctx.device)
        for s, plan in enumerate(plans):
            parts = []
            for src in range(n_ranks):
                begin = sum(plans[src].send_splits[:s])
                parts.append(sends[src][begin:begin + plans[src].send_splits[s]])
                assert plan.recv_splits[src] == plans[src].send_splits[s]
            recv = torch.cat(parts) if sum(p.numel() for p in parts) else \
                torch.zeros(1, device=ctx.device)
            piece, _ = ctx.fold_plan(plan, recv)
            ctx.synchronize()
            full[plan.piece_begin:plan.piece_end] = piece
            del recv, piece
        del sends
        small = ctx.downsample(full.reshape(-1), W, H, B)
        bytes8 = ctx.quantize_rgb8(small.reshape(-1), W, H)
        ctx.synchronize()
        check(small.view(H, W, 5), bytes8.view(H, W, 3), owners, n_ranks, "eight ranks")
    finally:
        scenes.assign_owners(spec, 1, "morton")
