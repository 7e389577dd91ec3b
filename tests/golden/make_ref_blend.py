#!/usr/bin/env python3
"""Golden vectors of the three over-blends, their region logic and the ubyte encode / decode made by
the REFERENCE's own image classes: oracle/_ref/ref_blend (oracle/ref_compose/build.sh: the
reference's Image / ImageRGBAFloatColorDepthSort / ImageRGBAFloatColorOnly / ImageRGBAUByteColorOnly
translation units compiled where they lie + this repository's driver).  Writes
tests/golden/ref_blend.npz: inputs and the reference's outputs per case.

    python tests/golden/make_ref_blend.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BINARY = os.path.join(ROOT, "oracle", "_ref", "ref_blend")
W, H = 40, 30                                   # 1200 pixels
KINDS = {"depthsort": (0, np.float32, 5), "rgba_f32": (1, np.float32, 4), "rgba_u8": (2, np.uint32, 1)}


def pixels(kind, n, rng, special=False):
    _, dtype, vec = KINDS[kind]
    if kind == "rgba_u8":
        return rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    alpha = rng.random(n, dtype=np.float32)
    alpha[rng.random(n) < 0.2] = 0.0
    alpha[rng.random(n) < 0.1] = 1.0
    out = np.zeros((n, vec), np.float32)
    out[:, :3] = rng.random((n, 3), dtype=np.float32) * alpha[:, None]
    out[:, 3] = alpha
    if vec == 5:
        depth = (rng.random(n, dtype=np.float32) * 4.0 - 1.0).astype(np.float32)
        depth[alpha == 0.0] = np.inf
        if special:
            depth[::7] = np.inf
            depth[3::11] = -np.inf
            depth[5::13] = 0.0
        out[:, 4] = depth
    return out


def cases():
    rng = np.random.default_rng(23)
    p = W * H
    out = []
    for kind in KINDS:
        out.append((f"{kind}/aligned", kind, 0, p, 0, p, pixels(kind, p, rng, True), pixels(kind, p, rng, True)))
        # the reference's own test matrix (ImageFullTest.cpp:379-444): unaligned overlaps, empties
        for name, tb, te, bb, be in (("top_first", 100, 700, 400, 1100), ("bottom_first", 500, 1200, 0, 800),
                                     ("top_inside", 300, 600, 100, 1000), ("bottom_inside", 0, 1200, 450, 460),
                                     ("adjacent", 0, 600, 600, 1200), ("disjoint_gap_free", 600, 1200, 0, 600),
                                     ("empty_top", 500, 500, 200, 900), ("empty_bottom", 100, 300, 300, 300)):
            out.append((f"{kind}/{name}", kind, tb, te, bb, be, pixels(kind, te - tb, rng),
                        pixels(kind, be - bb, rng)))
    # depth ties: the top image wins a tie (topDepth <= bottomDepth)
    top, bottom = pixels("depthsort", p, rng), pixels("depthsort", p, rng)
    bottom[:, 4] = top[:, 4]
    out.append(("depthsort/equal_depths", "depthsort", 0, p, 0, p, top, bottom))
    top8, bottom8 = exhaustive_ubyte_inputs()
    out.append(("rgba_u8/every_alpha_and_value", "rgba_u8", 0, 65536, 0, 65536, top8, bottom8))
    return out


def exhaustive_ubyte_inputs():
    """ubyte: every top alpha against every bottom component value (wrap-around included); the top's
    colour bytes from a multiplicative hash.  A formula, so the fixture stores the output only."""
    a, c = np.meshgrid(np.arange(256, dtype=np.uint64), np.arange(256, dtype=np.uint64), indexing="ij")
    index = np.arange(65536, dtype=np.uint64)
    colour = ((index * np.uint64(2654435761)) >> np.uint64(7)) & np.uint64(0x00ffffff)
    top = (colour | (a.reshape(-1) << np.uint64(24))).astype(np.uint32)
    bottom = (c.reshape(-1) * np.uint64(0x01010101)).astype(np.uint32)
    return top, bottom


def encode_inputs():
    grid = np.arange(0, 257, dtype=np.float32) / np.float32(256.0)
    near = np.concatenate([np.nextafter(grid, np.float32(-1)), grid, np.nextafter(grid, np.float32(2))])
    extra = np.array([-1.0, -0.0, 0.0, 0.999, 0.9999999, 1.0, 1.5, 255.0 / 256.0, 3.0e9, -3.0e9, 1e-30],
                     np.float32)
    values = np.concatenate([near, extra]).astype(np.float32)
    rng = np.random.default_rng(5)
    colours = np.stack([values, rng.permutation(values), rng.permutation(values), rng.permutation(values)], 1)
    return np.ascontiguousarray(colours, np.float32)


def main():
    if not os.path.exists(BINARY):
        status = subprocess.run(["bash", os.path.join(ROOT, "oracle", "ref_compose", "build.sh")]).returncode
        if status != 0:
            sys.exit("oracle/_ref/ref_blend cannot be built here (no reference tree / MPICH)")
    out = run_reference(verbose=True)
    np.savez_compressed(os.path.join(HERE, "ref_blend.npz"), **out)
    print("wrote tests/golden/ref_blend.npz")


def run_reference(verbose=False):
    """Every case through oracle/_ref/ref_blend; returns the fixture's arrays."""
    todo = cases()
    colours = encode_inputs()
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(src, "wb") as fh:
            fh.write(struct.pack("<i", len(todo) + 1))
            for _, kind, tb, te, bb, be, top, bottom in todo:
                width, height = (W, H) if max(te, be) <= W * H else (256, 256)
                fh.write(struct.pack("<7i", KINDS[kind][0], width, height, tb, te, bb, be))
                fh.write(np.ascontiguousarray(top).tobytes())
                fh.write(np.ascontiguousarray(bottom).tobytes())
            fh.write(struct.pack("<7i", 3, len(colours), 1, 0, len(colours), 0, 0))
            fh.write(colours.tobytes())
        subprocess.run([BINARY, src, dst], check=True, timeout=300)
        raw = open(dst, "rb").read()
    out, at = {}, 0
    for name, kind, tb, te, bb, be, top, bottom in todo:
        _, dtype, vec = KINDS[kind]
        begin, end = struct.unpack_from("<2i", raw, at)
        at += 8
        count = (end - begin) * vec
        result = np.frombuffer(raw, dtype, count, at).copy()
        at += count * 4
        if name != "rgba_u8/every_alpha_and_value":   # (its inputs are a formula)
            out[name + "/top"], out[name + "/bottom"] = top, bottom
        out[name + "/regions"] = np.asarray([tb, te, bb, be, begin, end], np.int64)
        out[name + "/out"] = result
        if verbose:
            print(f"{name}: [{tb},{te}) over [{bb},{be}) -> [{begin},{end})")
    begin, end = struct.unpack_from("<2i", raw, at)
    at += 8
    n = end - begin
    out["encode/colours"] = colours
    out["encode/encoded"] = np.frombuffer(raw, np.uint32, n, at).copy()
    at += 4 * n
    out["encode/decoded"] = np.frombuffer(raw, np.float32, 4 * n, at).reshape(n, 4).copy()
    at += 16 * n
    assert at == len(raw)
    return out


if __name__ == "__main__":
    main()
