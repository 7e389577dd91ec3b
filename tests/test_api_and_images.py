"""Reference-shaped Python surface: option validation (CPU) and, on the GPU, the image classes,
VolumePainter.paint, compose_layered and render_scene against the oracle."""
import numpy as np
import pytest

from amrvolumerenderer_amd import api, scenes
from amrvolumerenderer_amd.types import CameraParameters, ColorMapControlPoint

from helpers import assert_bit_equal


def decode_png_rgb8(data: bytes) -> np.ndarray:
    """Minimal PNG reader for 8-bit RGB, non-interlaced files (all five filter types)."""
    import struct
    import zlib
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, width, height = 8, b"", 0, 0
    while at < len(data):
        (length,), tag = struct.unpack(">I", data[at:at + 4]), data[at + 4:at + 8]
        payload = data[at + 8:at + 8 + length]
        (crc,) = struct.unpack(">I", data[at + 8 + length:at + 12 + length])
        assert crc == zlib.crc32(tag + payload) & 0xFFFFFFFF
        if tag == b"IHDR":
            width, height, depth, colour, comp, flt, lace = struct.unpack(">IIBBBBB", payload)
            assert (depth, colour, comp, flt, lace) == (8, 2, 0, 0, 0)
        elif tag == b"IDAT":
            idat += payload
        at += 12 + length
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(height, 1 + width * 3)
    out = np.zeros((height, width * 3), np.int32)
    for y in range(height):
        ftype, line = raw[y, 0], raw[y, 1:].astype(np.int32)
        prev = out[y - 1] if y else np.zeros(width * 3, np.int32)
        if ftype == 0:
            out[y] = line
        elif ftype == 2:
            out[y] = (line + prev) & 255
        else:
            for x in range(width * 3):
                a = out[y, x - 3] if x >= 3 else 0
                b, c = prev[x], (prev[x - 3] if x >= 3 else 0)
                if ftype == 1:
                    pred = a
                elif ftype == 3:
                    pred = (a + b) // 2
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                out[y, x] = (line[x] + pred) & 255
    return out.astype(np.uint8).reshape(height, width, 3)


def test_png_writer_round_trips(tmp_path):
    rng = np.random.default_rng(5)
    rgb8 = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    path = tmp_path / "x.png"
    assert api.save_png(rgb8, str(path))
    assert np.array_equal(decode_png_rgb8(path.read_bytes()), rgb8)
    assert not api.save_png(rgb8, str(tmp_path / "missing-dir" / "x.png"))


def test_option_validation_matches_reference_errors():
    ok = api.RenderOptions(camera=scenes.default_camera())
    api.validate_options(ok)
    with pytest.raises(ValueError):
        api.validate_options(api.RenderOptions(output_filename=""))
    with pytest.raises(ValueError):
        api.validate_options(api.RenderOptions(min_level=-1))
    with pytest.raises(ValueError):
        api.validate_options(api.RenderOptions(max_level=-2))
    with pytest.raises(RuntimeError):  # std::runtime_error in the reference
        api.validate_options(api.RenderOptions(min_level=2, max_level=1))
    with pytest.raises(ValueError):
        api.validate_options(api.RenderOptions(scalar_range=(1.0, 1.0)))
    with pytest.raises(ValueError):
        api.validate_options(api.RenderOptions(up_vector=(0.0, 0.0, 0.0)))
    with pytest.raises(ValueError):
        api.validate_options(api.RenderOptions(color_map=[]))
    with pytest.raises(ValueError):
        api.validate_options(api.RenderOptions(
            camera=CameraParameters((0, 0, 1), (0, 0, 0), (0, 1, 0), 45.0, 1.0, 0.5)))


def test_python_render_kwargs_surface():
    with pytest.raises(RuntimeError):   # the path does not exist (VolumeRenderer.cpp:1464-1467)
        api.render("plt00000", width=64, height=64,
                   color_map=[(0.0, 0, 0, 1, 0.1), (1.0, 1, 0, 0, 0.9)])
    with pytest.raises(ValueError):      # a colour map needs two control points
        api.render("plt00000", color_map=[(0.0, 0, 0, 1, 0.1)])
    with pytest.raises(ValueError):      # strictly increasing values
        api.render("plt00000", color_map=[(0.5, 0, 0, 1, 0.1), (0.5, 1, 0, 0, 0.9)])
    with pytest.raises(ValueError):      # components within [0, 1]
        api.render("plt00000", color_map=[(0.0, 0, 0, 1.5, 0.1), (1.0, 1, 0, 0, 0.9)])
    with pytest.raises(ValueError):
        api.render("plt00000", antialiasing=3)            # not a perfect square
    with pytest.raises(ValueError):
        api.render("plt00000", camera_eye=(0, 0, 1))       # eye without look_at
    with pytest.raises(RuntimeError):   # "plotfile path is required" is a runtime_error
        api.render("", width=64)
    with pytest.raises(ValueError):
        api.render("plt00000", color_map=[(0.0, 1.0)])


@pytest.mark.gpu
def test_image_classes_and_painter(O, ctx):
    import torch
    from amrvolumerenderer_amd.images import (ImageRGBAFloatColorDepthSort,
                                              ImageRGBAFloatColorOnly, ImageRGBAUByteColorOnly)
    from amrvolumerenderer_amd.painter import VolumePainter
    from amrvolumerenderer_amd.types import ScalarTransform, VolumeBounds
    from helpers import device_box, oracle_camera, oracle_params, oracle_transform
    W, H = 60, 40
    N = W * H
    img = ImageRGBAFloatColorDepthSort(ctx, W, H)
    img.clear()
    h = img.to_host()
    assert np.all(h[:, :4] == 0) and np.all(np.isposinf(h[:, 4]))
    # window is a view, copy_subrange a copy
    win = img.window(100, 200)
    win.buffer[:] = 1.0
    assert img.to_host()[100:200].min() == 1.0 and img.to_host()[:100, 0].max() == 0.0
    sub = img.copy_subrange(100, 150)
    sub.buffer[:] = 2.0
    assert img.to_host()[100:150].max() == 1.0
    assert (sub.region_begin, sub.region_end) == (100, 150)
    with pytest.raises(IndexError):
        img.window(0, N + 1)
    # blend with regions: window of a window on top of a partial image
    rng = np.random.default_rng(0)
    a = rng.random((N, 4), dtype=np.float32)
    b = rng.random((N, 4), dtype=np.float32)
    top = ImageRGBAFloatColorOnly(ctx, W, H, 0, N, torch.from_numpy(a.reshape(-1)).to(ctx.device))
    bottom = ImageRGBAFloatColorOnly(ctx, W, H, 0, N, torch.from_numpy(b.reshape(-1)).to(ctx.device))
    out = top.window(500, 1500).window(100, 900).blend(bottom.copy_subrange(300, 1200))
    want, wb, we = O.blend_regions("rgba_f32", a[600:1400], 600, 1400, b[300:1200], 300, 1200)
    assert (out.region_begin, out.region_end) == (wb, we)
    assert_bit_equal(out.to_host(), want, "windowed blend")
    with pytest.raises(ValueError):
        top.window(0, 10).blend(bottom.window(20, 30))
    with pytest.raises(TypeError):
        top.blend(img)
    # ubyte image: encode / blend / decode
    u_top, u_bot = ImageRGBAUByteColorOnly(ctx, W, H), ImageRGBAUByteColorOnly(ctx, W, H)
    u_top.set_colors(torch.from_numpy(a).to(ctx.device))
    u_bot.set_colors(torch.from_numpy(b).to(ctx.device))
    got = u_top.blend(u_bot).get_colors().cpu().numpy()
    want = O.decode_rgba_u8(O.blend_rgba_u8(O.encode_rgba_u8(a), O.encode_rgba_u8(b)))
    assert_bit_equal(got, want, "ubyte image blend")
    # painter with the reference's argument list
    cells = np.ascontiguousarray(np.random.default_rng(1).random((16, 16, 16)))
    box = device_box(ctx, cells, (0, 0, 0), (1, 1, 1))
    bounds = VolumeBounds((-0.05,) * 3, (1.05,) * 3)
    cam = scenes.default_camera()
    tr = ScalarTransform(normalize_to_unit_range=True)
    VolumePainter(ctx).paint(box, bounds, tr, (0.0, 1.0), 0, 1, 0.3, 1, 0.02, img, cam, None)
    want, _ = O.paint_box(O.make_box(cells, (0, 0, 0), (1, 1, 1)), oracle_transform(O, tr),
                          oracle_params(O, W, H, (0, 1), 0.3, 0.02, bounds), oracle_camera(O, cam))
    assert_bit_equal(img.to_host(), want, "VolumePainter.paint")
    with pytest.raises(RuntimeError):
        VolumePainter(ctx).paint(box, bounds, tr, (0.0, 1.0), 0, 1, 0.3, 1, 0.02, top, cam, None)


@pytest.mark.gpu
def test_compose_layered_single_rank_and_render_scene(O, ctx, tmp_path):
    import torch
    from amrvolumerenderer_amd.images import (ImageRGBAFloatColorDepthSort, LayeredVolumeImage,
                                              compose_layered)
    from helpers import device_box
    from test_frame_plan import oracle_overlay, painted_scene
    W, H = 72, 48
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    cam = scenes.default_camera()
    cells, layers, hints, ref = painted_scene(O, spec, cam, W, H, 0.6)
    want, _, _ = O.compose_layered(layers, hints, [0] * len(layers), list(range(len(layers))), 1)
    dev_layers = [ImageRGBAFloatColorDepthSort(
        ctx, W, H, 0, W * H, torch.from_numpy(l.reshape(-1)).to(ctx.device)) for l in layers]
    layered = LayeredVolumeImage(W, H, dev_layers, hints)
    assert layered.get_layer_count() == len(layers)
    piece = compose_layered(ctx, layered)
    assert (piece.region_begin, piece.region_end) == (0, W * H)
    assert_bit_equal(piece.to_host(), want, "compose_layered")

    # render_scene: the whole frame to a PPM file, byte-identical to SavePPM of the oracle frame
    meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
    local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level)
             for c, m in zip(cells, spec.boxes)]
    scene = api.SceneGeometry(meta, local, spec.transform, spec.bounds, spec.scalar_range)
    out = tmp_path / "frame.ppm"
    options = api.RenderOptions(width=W, height=H, box_transparency=0.6, camera=cam,
                                output_filename=str(out))
    assert api.render_scene(ctx, scene, options) == 0
    data = out.read_bytes()
    header = f"P6\n{W} {H}\n255\n".encode()
    assert data.startswith(header)
    want8 = O.quantize_rgb8(oracle_overlay(O, spec, cells, cam, want, W, H), W, H)
    assert data[len(header):] == want8.tobytes()
    assert not np.array_equal(want8, O.quantize_rgb8(want, W, H))  # the wireframe is visible

    # .png: same pixels (SavePNG.cpp:50-73: 8-bit RGB rows, top-down)
    png = tmp_path / "frame.png"
    options.output_filename = str(png)
    assert api.render_scene(ctx, scene, options) == 0
    assert np.array_equal(decode_png_rgb8(png.read_bytes()), want8.reshape(H, W, 3))
