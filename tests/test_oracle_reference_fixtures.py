"""Pins the oracle's image algebra with the known-answer fixtures of the reference's own unit
test (Common/Testing/ImageFullTest.cpp): the procedural 110x100 triangle images, the
hand-derived "combined" expectation, and the reference's comparison rule (per-channel
tolerance 0.02 on RGB, at most 2 % bad pixels; :56-85).  The fixture generators below restate
the test DATA (which pixels hold which colour); they are inputs/expected outputs, not code of
the path under test."""
import numpy as np
import pytest

W, H, BORDER = 110, 100, 10          # ImageFullTest.cpp:52-54
N = W * H
MID1, MID2, MID3 = N // 3, N // 2, 2 * N // 3


def _xy(begin, end):
    idx = np.arange(begin, end)
    return idx % W, idx // W


def image1(begin=0, end=N):
    """createColorOnlyImage1 (:115-135): (0.5,0,0,0.5) on a bordered lower triangle."""
    x, y = _xy(begin, end)
    img = np.zeros((end - begin, 4), np.float32)
    m = (x > BORDER) & (x < W - BORDER) & (y > BORDER) & (y < H - BORDER) & (x <= y)
    img[m] = (0.5, 0.0, 0.0, 0.5)
    return img


def image2(begin=0, end=N):
    """createColorOnlyImage2 (:185-201): (0,0,0.5,0.5) where x <= H - y."""
    x, y = _xy(begin, end)
    img = np.zeros((end - begin, 4), np.float32)
    img[x <= (H - y)] = (0.0, 0.0, 0.5, 0.5)
    return img


def combined(begin=0, end=N):
    """createColorOnlyImageCombined (:249-277): the expected blend(image1, image2)."""
    x, y = _xy(begin, end)
    img = np.zeros((end - begin, 4), np.float32)
    tri = (x > BORDER) & (x < W - BORDER) & (y > BORDER) & (y < H - BORDER) & (x <= y)
    low = x <= (H - y)
    img[tri & low] = (0.5, 0.0, 0.25, 0.75)
    img[tri & ~low] = (0.5, 0.0, 0.0, 0.5)
    img[~tri & low] = (0.0, 0.0, 0.5, 0.5)
    return img


def images_match(a, b):
    """compareImages (:56-85)."""
    assert a.shape == b.shape
    bad = np.any(np.abs(a[:, :3] - b[:, :3]) > 0.02, axis=1).sum()
    return bad <= 0.02 * a.shape[0]


class FloatImage:
    kind = "rgba_f32"

    @staticmethod
    def store(O, rgba):
        return np.ascontiguousarray(rgba, np.float32).reshape(-1)

    @staticmethod
    def load(O, buf):
        return np.asarray(buf, np.float32).reshape(-1, 4)


class UByteImage:
    kind = "rgba_u8"

    @staticmethod
    def store(O, rgba):           # setColor -> encodeColor
        return O.encode_rgba_u8(rgba)

    @staticmethod
    def load(O, buf):             # getColor -> decodeColor
        return O.decode_rgba_u8(buf)


IMAGE_TYPES = [FloatImage, UByteImage]


def blend(O, T, top, tb, te, bottom, bb, be):
    out, ob, oe = O.blend_regions(T.kind, T.store(O, top), tb, te, T.store(O, bottom), bb, be)
    return T.load(O, out), ob, oe


@pytest.mark.parametrize("T", IMAGE_TYPES)
def test_blend_non_empty_and_empty(O, T):
    """TestBlend, aligned cases (:379-402)."""
    out, ob, oe = blend(O, T, image1(), 0, N, image2(), 0, N)
    assert (ob, oe) == (0, N)
    assert images_match(out, combined())
    empty = np.zeros((N, 4), np.float32)
    out, _, _ = blend(O, T, empty, 0, N, image2(), 0, N)
    assert images_match(out, T.load(O, T.store(O, image2())))
    out, _, _ = blend(O, T, image1(), 0, N, empty, 0, N)
    assert images_match(out, T.load(O, T.store(O, image1())))
    out, _, _ = blend(O, T, empty, 0, N, empty, 0, N)
    assert images_match(out, empty)


def test_float_blend_known_answer_is_exact(O):
    """All fixture values are dyadic, so the float result is the hand-derived colour exactly:
    (0.5,0,0,0.5) over (0,0,0.5,0.5) = (0.5, 0, 0.25, 0.75)."""
    out, _, _ = blend(O, FloatImage, image1(), 0, N, image2(), 0, N)
    assert np.array_equal(out, combined())


def test_ubyte_blend_known_answer_bytes(O):
    """0.5 encodes to 128; bottom scale 1 - 128/255; 128 * 0.498.. = 63.7 -> 63: the blended
    pixel is (128, 0, 63, 191), i.e. (0.502, 0, 0.247, 0.749) after decoding."""
    enc = O.blend_rgba_u8(O.encode_rgba_u8(np.array([[0.5, 0, 0, 0.5]], np.float32)),
                          O.encode_rgba_u8(np.array([[0, 0, 0.5, 0.5]], np.float32)))
    assert [(int(enc[0]) >> s) & 0xFF for s in (0, 8, 16, 24)] == [128, 0, 63, 191]


@pytest.mark.parametrize("T", IMAGE_TYPES)
def test_blend_unaligned(O, T):
    """TestBlend, 'Blend unaligned 1..4' (:404-444)."""
    exp = combined()
    one, two = image1(), image2()
    # 1: top [0,MID2) over bottom [MID1,N)
    out, ob, oe = blend(O, T, one[:MID2], 0, MID2, two[MID1:], MID1, N)
    assert (ob, oe) == (0, N)
    assert images_match(out[:MID1], T.load(O, T.store(O, one[:MID1])))
    assert images_match(out[MID1:MID2], exp[MID1:MID2])
    assert images_match(out[MID2:], T.load(O, T.store(O, two[MID2:])))
    # 2: top [MID1,N) over bottom [0,MID2)
    out, ob, oe = blend(O, T, one[MID1:], MID1, N, two[:MID2], 0, MID2)
    assert images_match(out[:MID1], T.load(O, T.store(O, two[:MID1])))
    assert images_match(out[MID1:MID2], exp[MID1:MID2])
    assert images_match(out[MID2:], T.load(O, T.store(O, one[MID2:])))
    # 3: top [MID1,MID2) over full bottom
    out, ob, oe = blend(O, T, one[MID1:MID2], MID1, MID2, two, 0, N)
    assert images_match(out[:MID1], T.load(O, T.store(O, two[:MID1])))
    assert images_match(out[MID1:MID2], exp[MID1:MID2])
    assert images_match(out[MID2:], T.load(O, T.store(O, two[MID2:])))
    # 4: full top over bottom [MID1,MID2)
    out, ob, oe = blend(O, T, one, 0, N, two[MID1:MID2], MID1, MID2)
    assert images_match(out[:MID1], T.load(O, T.store(O, one[:MID1])))
    assert images_match(out[MID1:MID2], exp[MID1:MID2])
    assert images_match(out[MID2:], T.load(O, T.store(O, one[MID2:])))


@pytest.mark.parametrize("T", IMAGE_TYPES)
def test_window_blend(O, T):
    """TestWindow, 'Window blend' (:481-484): a window [MID2,MID3) of image 1 blended over
    image 2 created on the same region equals the combined image on that region."""
    out, ob, oe = blend(O, T, image1(MID2, MID3), MID2, MID3, image2(MID2, MID3), MID2, MID3)
    assert (ob, oe) == (MID2, MID3)
    assert images_match(out, combined(MID2, MID3))


@pytest.mark.gpu
@pytest.mark.parametrize("T", IMAGE_TYPES)
def test_hip_blend_passes_the_reference_unit_test(O, ctx, T):
    """The same fixtures through the HIP kernels (C ABI): passes the reference's own criterion
    and agrees with the oracle bit for bit."""
    import torch

    def dev(buf):
        a = np.ascontiguousarray(buf)
        if a.dtype == np.uint32:
            a = a.view(np.int32)
        return torch.from_numpy(a).to(ctx.device)

    exp = combined()
    for (top, tb, te, bottom, bb, be) in [
            (image1(), 0, N, image2(), 0, N),
            (image1()[:MID2], 0, MID2, image2()[MID1:], MID1, N),
            (image1()[MID1:], MID1, N, image2()[:MID2], 0, MID2),
            (image1(MID2, MID3), MID2, MID3, image2(MID2, MID3), MID2, MID3)]:
        got, ob, oe = ctx.blend_regions(T.kind, dev(T.store(O, top)), tb, te,
                                        dev(T.store(O, bottom)), bb, be)
        want, wb, we = O.blend_regions(T.kind, T.store(O, top), tb, te, T.store(O, bottom), bb, be)
        g = got.cpu().numpy()
        if T is UByteImage:
            g = g.view(np.uint32)
        assert np.array_equal(g.view(np.uint32), np.asarray(want).view(np.uint32))
        lo, hi = max(tb, bb), min(te, be)
        assert images_match(T.load(O, g)[lo - ob:hi - ob], exp[lo:hi])
