"""GPU parity of the image algebra (blend variants, regions, encode/decode, fold, frame tail)."""
import numpy as np
import pytest
import torch

from helpers import assert_bit_equal

pytestmark = pytest.mark.gpu


def rand_depthsort(rng, n):
    a = rng.random((n, 5), dtype=np.float32)
    a[:, :3] *= a[:, 3:4]  # premultiplied
    a[:, 4] = rng.random(n, dtype=np.float32) * 4.0
    # specials: empty pixels, equal depths, +inf / -inf depths, opaque, zero alpha
    a[rng.choice(n, n // 8, replace=False)] = (0, 0, 0, 0, np.inf)
    a[rng.choice(n, n // 16, replace=False), 4] = 1.5
    a[rng.choice(n, n // 32, replace=False), 4] = np.inf
    a[rng.choice(n, n // 64, replace=False), 4] = -np.inf
    a[rng.choice(n, n // 32, replace=False), 3] = 1.0
    return a


def to_dev(ctx, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)


def test_blend_depthsort(O, ctx):
    rng = np.random.default_rng(1)
    for n in (1, 63, 4099, 110 * 100):
        top, bottom = rand_depthsort(rng, n), rand_depthsort(rng, n)
        got = ctx.blend("depthsort", to_dev(ctx, top), to_dev(ctx, bottom))
        assert_bit_equal(got.cpu().numpy(), O.blend_depthsort(top, bottom), f"depthsort n={n}")


def test_blend_depthsort_in_place(O, ctx):
    rng = np.random.default_rng(2)
    top, bottom = rand_depthsort(rng, 5000), rand_depthsort(rng, 5000)
    t = to_dev(ctx, top)
    ctx.blend("depthsort", t, to_dev(ctx, bottom), out=t)
    assert_bit_equal(t.cpu().numpy(), O.blend_depthsort(top, bottom), "in place")


def test_blend_rgba_f32(O, ctx):
    rng = np.random.default_rng(3)
    top = rng.random((9001, 4), dtype=np.float32)
    bottom = rng.random((9001, 4), dtype=np.float32)
    got = ctx.blend("rgba_f32", to_dev(ctx, top), to_dev(ctx, bottom))
    assert_bit_equal(got.cpu().numpy(), O.blend_rgba_f32(top, bottom), "rgba f32")


def test_blend_rgba_u8_exhaustive_alpha_and_wraparound(O, ctx):
    # every (top alpha, bottom component) pair, plus top components that force uint8 overflow
    ta, bc = np.meshgrid(np.arange(256, dtype=np.uint32), np.arange(256, dtype=np.uint32))
    ta, bc = ta.reshape(-1), bc.reshape(-1)
    top = (ta << 24) | (np.uint32(250) << 16) | (np.uint32(200) << 8) | (ta ^ 0x5A)
    bottom = (bc << 24) | (bc << 16) | ((255 - bc) << 8) | bc
    want = O.blend_rgba_u8(top, bottom)
    got = ctx.blend("rgba_u8", to_dev(ctx, top.view(np.int32)), to_dev(ctx, bottom.view(np.int32)))
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want)
    # wrap-around really happens (no saturation): 250 + x overflows for x >= 6
    assert np.any(((want >> 16) & 0xFF) < 250)
    rng = np.random.default_rng(4)
    top = rng.integers(0, 2 ** 32, 100003, dtype=np.uint64).astype(np.uint32)
    bottom = rng.integers(0, 2 ** 32, 100003, dtype=np.uint64).astype(np.uint32)
    got = ctx.blend("rgba_u8", to_dev(ctx, top.view(np.int32)), to_dev(ctx, bottom.view(np.int32)))
    assert np.array_equal(got.cpu().numpy().view(np.uint32), O.blend_rgba_u8(top, bottom))


def test_encode_decode_u8(O, ctx):
    rng = np.random.default_rng(5)
    rgba = (rng.random((20000, 4), dtype=np.float32) * 1.4 - 0.2).astype(np.float32)
    rgba[:7] = [[0, 0.5, 1.0, 255.0 / 256.0], [-1, 2, np.nextafter(np.float32(1), 0), 0.00390625],
                [1 / 256.0, 0.999, 0.003, 0.5], [0.25, 0.75, 0.125, 1], [0, 0, 0, 0],
                [1, 1, 1, 1], [0.1, 0.2, 0.3, 0.4]]
    want = O.encode_rgba_u8(rgba)
    got = ctx.encode_rgba_u8(to_dev(ctx, rgba))
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want)
    back = ctx.decode_rgba_u8(got)
    assert_bit_equal(back.cpu().numpy(), O.decode_rgba_u8(want), "decode")


@pytest.mark.parametrize("kind,vec,dtype", [("depthsort", 5, np.float32),
                                            ("rgba_f32", 4, np.float32),
                                            ("rgba_u8", 1, np.uint32)])
def test_blend_regions(O, ctx, kind, vec, dtype):
    """The reference's TestBlend region cases (Common/Testing/ImageFullTest.cpp:379-444):
    aligned, four unaligned overlaps, touching and empty regions."""
    rng = np.random.default_rng(6)
    N = 110 * 100
    M1, M2 = N // 3, N // 2
    cases = [(0, N, 0, N), (0, M2, M1, N), (M1, N, 0, M2), (M1, M2, 0, N), (0, N, M1, M2),
             (0, M1, M1, N), (M1, N, 0, M1), (0, 0, 0, N), (0, N, N, N), (5, 5, 5, 5)]
    for tb, te, bb, be in cases:
        if dtype == np.uint32:
            top = rng.integers(0, 2 ** 32, te - tb, dtype=np.uint64).astype(np.uint32)
            bottom = rng.integers(0, 2 ** 32, be - bb, dtype=np.uint64).astype(np.uint32)
            dev = lambda a: to_dev(ctx, a.view(np.int32))
        elif kind == "depthsort":
            top, bottom = rand_depthsort(rng, te - tb).reshape(-1), rand_depthsort(rng, be - bb).reshape(-1)
            dev = lambda a: to_dev(ctx, a)
        else:
            top = rng.random((te - tb) * vec, dtype=np.float32)
            bottom = rng.random((be - bb) * vec, dtype=np.float32)
            dev = lambda a: to_dev(ctx, a)
        want, wb, we = O.blend_regions(kind, top, tb, te, bottom, bb, be)
        got, gb, ge = ctx.blend_regions(kind, dev(top), tb, te, dev(bottom), bb, be)
        assert (gb, ge) == (wb, we)
        g = got.cpu().numpy()
        if dtype == np.uint32:
            assert np.array_equal(g.view(np.uint32), want)
        else:
            assert_bit_equal(g, want, f"regions {kind} {(tb, te, bb, be)}")
    with pytest.raises(ValueError):
        ctx.blend_regions(kind, dev(np.zeros(0, dtype)), 0, 0, dev(np.zeros(0, dtype)), 10, 10)


def test_fold_runs(O, ctx):
    rng = np.random.default_rng(8)
    n = 7001
    slices = [rand_depthsort(rng, n) for _ in range(6)]
    want = slices[0]
    for s in slices[1:]:
        want = O.blend_depthsort(want, s)
    got = ctx.fold_runs([to_dev(ctx, s).reshape(-1) for s in slices], n)
    assert_bit_equal(got.cpu().numpy(), want, "fold")
    one = ctx.fold_runs([to_dev(ctx, slices[2]).reshape(-1)], n)
    assert_bit_equal(one.cpu().numpy(), slices[2], "fold of one")
    none = ctx.fold_runs([], 16).cpu().numpy()
    assert np.all(none[:, :4] == 0) and np.all(np.isposinf(none[:, 4]))


@pytest.mark.parametrize("block", [2, 3, 4])
def test_downsample_and_quantize(O, ctx, block):
    rng = np.random.default_rng(9)
    tw, th = 37, 23
    src = rand_depthsort(rng, tw * block * th * block)
    want = O.downsample(src, tw, th, block)
    got = ctx.downsample(to_dev(ctx, src).reshape(-1), tw, th, block)
    assert_bit_equal(got.cpu().numpy(), want, "downsample")
    want8 = O.quantize_rgb8(want, tw, th)
    got8 = ctx.quantize_rgb8(got, tw, th)
    assert np.array_equal(got8.cpu().numpy(), want8)
    with pytest.raises(ValueError):
        ctx.downsample(to_dev(ctx, src[: tw * th]).reshape(-1), tw, th, 1)
