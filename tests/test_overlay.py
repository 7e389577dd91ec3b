"""tight bounds + wireframe overlay (VolumeRenderer.cpp:139-335, 791-848)
against the oracle, bit for bit."""
import numpy as np
import pytest

from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.types import AmrBox, CameraParameters

from helpers import assert_bit_equal
from test_frame_plan import oracle_camera


def random_image(rng, w, h):
    img = rng.random((h * w, 5), dtype=np.float32)
    img[:, 4] = rng.random(h * w, dtype=np.float32) * 10.0
    img[rng.random(h * w) < 0.3] = (0.0, 0.0, 0.0, 0.0, np.inf)  # empty pixels
    return img


CAMERAS = {
    "default": scenes.default_camera(),
    "orbit5": scenes.orbit_camera(5),
    # eye inside the box: the corners behind the eye are dropped and with them their edges
    "inside": CameraParameters((0.5, 0.45, 0.55), (0.9, 0.6, 0.1), (0.0, 1.0, 0.0), 60.0, 0.01, 10.0),
    # looking away: nothing is drawn
    "away": CameraParameters((2.0, 2.0, 2.0), (4.0, 4.0, 4.0), (0.0, 1.0, 0.0), 45.0, 0.1, 20.0),
    # close grazing view: long edges leave the screen
    "close": CameraParameters((1.05, 0.5, 1.3), (0.4, 0.5, 0.2), (0.0, 1.0, 0.0), 70.0, 0.05, 10.0),
}


def test_tight_bounds_matches_oracle(O, avr_lib):
    rng = np.random.default_rng(11)
    boxes, oboxes = [], []
    for _ in range(9):
        lo = rng.random(3) * 3.0 - 1.0
        hi = lo + rng.random(3) + 1e-9
        cells = np.zeros((2, 2, 2))
        boxes.append(AmrBox(tuple(lo), tuple(hi), dims=(2, 2, 2)))
        oboxes.append(O.make_box(cells, tuple(lo), tuple(hi)))
    fallback = ((-5.0, -5.0, -5.0), (5.0, 5.0, 5.0))
    got = runtime.tight_bounds(boxes, *fallback)
    assert got == O.tight_bounds(oboxes, *fallback)
    # corners are reduced in float (MPI_FLOAT, VolumeRenderer.cpp:823-838)
    assert all(v == float(np.float32(v)) for v in got[0] + got[1])
    assert runtime.tight_bounds([], *fallback) == fallback == O.tight_bounds([], *fallback)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CAMERAS))
@pytest.mark.parametrize("size,scale", [((96, 64), 1), ((131, 77), 2), ((64, 96), 3)])
def test_overlay_matches_oracle(O, ctx, name, size, scale):
    import torch
    W, H = size
    cam = CAMERAS[name]
    rng = np.random.default_rng(W * 7 + scale)
    img = random_image(rng, W, H)
    bounds = ((0.0, 0.03125, 0.0), (1.0, 0.96875, 1.0))
    want = O.bbox_overlay(img, W, H, bounds[0], bounds[1], oracle_camera(O, cam), scale)
    dev = torch.from_numpy(img.reshape(-1).copy()).to(ctx.device)
    rgb8 = ctx.bbox_overlay(dev, bounds[0], bounds[1], cam, scale, W, H, want_rgb8=True)
    ctx.synchronize()
    assert_bit_equal(dev.cpu().numpy().reshape(-1, 5), want.reshape(-1, 5), f"overlay {name}")
    assert np.array_equal(rgb8.cpu().numpy().reshape(H, W, 3)[::-1],
                          O.quantize_rgb8(want.reshape(-1, 5), W, H))
    touched = (want.reshape(-1, 5)[:, 4] == np.finfo(np.float32).min).sum()
    assert (touched == 0) == (name == "away")


@pytest.mark.gpu
def test_overlay_on_pieces_equals_the_whole_image(O, ctx):
    """DirectSend pieces (DirectSendBase.cpp:59-74) overlaid one by one = the reference's
    overlay of the gathered image: pixels are independent."""
    import torch
    W, H = 75, 43
    cam = scenes.default_camera()
    img = random_image(np.random.default_rng(3), W, H)
    bounds = ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    want = O.bbox_overlay(img, W, H, bounds[0], bounds[1], oracle_camera(O, cam), 1).reshape(-1, 5)
    got = np.zeros_like(want)
    for k in range(5):
        b, e = runtime.piece_range(W * H, k, 5)
        dev = torch.from_numpy(img[b:e].reshape(-1).copy()).to(ctx.device)
        assert ctx.bbox_overlay(dev, bounds[0], bounds[1], cam, 1, W, H, b, e) is None
        ctx.synchronize()
        got[b:e] = dev.cpu().numpy().reshape(-1, 5)
    assert_bit_equal(got, want, "overlay by pieces")
    with pytest.raises(ValueError):
        ctx.bbox_overlay(torch.zeros(10, device=ctx.device), bounds[0], bounds[1], cam, 1, W, H)


@pytest.mark.gpu
def test_overlay_of_a_degenerate_box_draws_single_samples(O, ctx):
    """A zero-extent box projects all corners onto one point: every edge takes the
    single-sample branch (VolumeRenderer.cpp:281-288) and blends full coverage 12 times."""
    import torch
    W, H = 40, 30
    cam = scenes.default_camera()
    img = random_image(np.random.default_rng(8), W, H)
    point = (0.5, 0.5, 0.5)
    want = O.bbox_overlay(img, W, H, point, point, oracle_camera(O, cam), 1).reshape(-1, 5)
    dev = torch.from_numpy(img.reshape(-1).copy()).to(ctx.device)
    ctx.bbox_overlay(dev, point, point, cam, 1, W, H)
    ctx.synchronize()
    assert_bit_equal(dev.cpu().numpy().reshape(-1, 5), want, "degenerate overlay")
    changed = np.flatnonzero((want.view(np.uint32) != img.view(np.uint32)).any(axis=1))
    assert changed.size == 1 and np.array_equal(want[changed[0], :4], np.ones(4, np.float32))
