#!/usr/bin/env python3
"""Golden vectors of the layered DirectSend compose made by the REFERENCE's own compositor:
oracle/_ref/ref_compose (oracle/ref_compose/build.sh: the reference's Image / LayeredVolumeImage /
DirectSendBase translation units compiled where they lie + this repository's driver) run under
`mpiexec -n N` on synthetic depth-sort layers -- what VolumeRenderer::renderSingleTrial does between
VolumeRenderer.cpp:1225 and :1294 (LayeredVolumeImage -> DirectSendBase::compose -> Gather).

Writes tests/golden/ref_compose.npz: the input layers (once) and, per case, the gathered image, the
pixel bytes of the reference's SavePPM of it, and the pixel range every rank returned from compose.  Cases: 1 / 2 / 3 / 4 / 8 ranks x round-robin /
block ownership, the reversed group order, hints that tie, a rank without layers, layers whose
pixels hold +inf / equal depths / zero alpha.  Only runs where the reference tree and MPICH are
(this container); the vectors are data and travel with the repository, the binary does not.

    python tests/golden/make_ref_compose.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BINARY = os.path.join(ROOT, "oracle", "_ref", "ref_compose")
MPIEXEC = os.environ.get("AVR_MPIEXEC", "/opt/conda/bin/mpiexec")
W, H, N_LAYERS = 37, 29, 8          # 1073 pixels: every piece boundary splits a row


def layers_and_hints(seed=11):
    """Overlapping translucent discs (premultiplied, per-pixel depth) with the edge cases of the
    depth-sort blend: cleared pixels (0, 0, 0, 0, +inf), equal depths in two layers, an opaque
    patch, a pixel with alpha but +inf depth, negative and zero depths."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    layers, hints = [], []
    for l in range(N_LAYERS):
        cx, cy, r = rng.uniform(6, W - 6), rng.uniform(5, H - 5), rng.uniform(6, 14)
        inside = (xx - cx) ** 2 + (yy - cy) ** 2 < r * r
        alpha = np.where(inside, rng.uniform(0.2, 0.8), 0.0).astype(np.float32)
        alpha *= (0.6 + 0.4 * rng.random((H, W))).astype(np.float32)
        img = np.zeros((H, W, 5), np.float32)
        colour = rng.random(3).astype(np.float32)
        img[..., :3] = colour * alpha[..., None]
        img[..., 3] = alpha
        depth = np.float32(1.0 + 0.37 * l) + (0.05 * rng.random((H, W))).astype(np.float32)
        img[..., 4] = np.where(alpha > 0, depth, np.inf)
        layers.append(img)
        hints.append(np.float32(1.0 + 0.37 * l))
    layers[1][3:9, 4:12, 4] = layers[0][3:9, 4:12, 4]        # equal depths in two layers
    layers[2][10:14, 10:16, :] = [0.3, 0.2, 0.1, 1.0, 1.2]   # an opaque patch
    layers[3][20, 5, :] = [0.1, 0.1, 0.1, 0.4, np.inf]       # alpha, but no depth
    layers[4][0:2, :, 4] = np.where(layers[4][0:2, :, 3] > 0, -0.5, np.inf)   # behind the eye
    layers[5][15, 15, :] = [0.2, 0.0, 0.0, 0.5, 0.0]
    order = rng.permutation(N_LAYERS)                         # layer ids are not in depth order
    return ([layers[i].reshape(-1, 5) for i in order], [hints[i] for i in order])


def ownership(n_ranks, policy):
    if policy == "round_robin":
        return [l % n_ranks for l in range(N_LAYERS)]
    if policy == "block":
        chunk = -(-N_LAYERS // n_ranks)
        return [min(l // chunk, n_ranks - 1) for l in range(N_LAYERS)]
    if policy == "rank1_empty":                               # a rank that owns nothing
        return [0 if l % 2 == 0 else 2 for l in range(N_LAYERS)]
    raise ValueError(policy)


def run_reference(layers, hints, owner, n_ranks, reverse=False):
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "layers.bin"), os.path.join(tmp, "out.bin")
        with open(src, "wb") as fh:
            fh.write(struct.pack("<3i", W, H, len(layers)))
            for layer, hint, who in zip(layers, hints, owner):
                fh.write(struct.pack("<if", int(who), float(hint)))
                fh.write(np.ascontiguousarray(layer, dtype="<f4").tobytes())
        ppm = os.path.join(tmp, "out.ppm")
        cmd = [MPIEXEC, "-n", str(n_ranks), BINARY, src, dst, "reverse" if reverse else "forward", ppm]
        subprocess.run(cmd, check=True, timeout=300)
        raw = open(dst, "rb").read()
        file_bytes = open(ppm, "rb").read()
    n = struct.unpack_from("<i", raw, 0)[0]
    regions = np.frombuffer(raw, "<i4", 2 * n, 4).reshape(n, 2).copy()
    image = np.frombuffer(raw, "<f4", W * H * 5, 4 + 8 * n).reshape(-1, 5).copy()
    header = f"P6\n{W} {H}\n255\n".encode()
    assert file_bytes.startswith(header) and len(file_bytes) == len(header) + W * H * 3
    rgb8 = np.frombuffer(file_bytes, np.uint8, W * H * 3, len(header)).reshape(H, W, 3).copy()
    return image, regions, rgb8


def classic_images(kind, n_ranks, seed=31):
    """One plain image per rank for the classic direct send (kinds as in make_ref_blend.py)."""
    rng = np.random.default_rng(seed + 7 * kind)
    p = W * H
    images = []
    for _ in range(n_ranks):
        if kind == 2:
            images.append(rng.integers(0, 2 ** 32, p, dtype=np.uint64).astype(np.uint32))
            continue
        vec = 5 if kind == 0 else 4
        alpha = rng.random(p, dtype=np.float32)
        alpha[rng.random(p) < 0.25] = 0.0
        img = np.zeros((p, vec), np.float32)
        img[:, :3] = rng.random((p, 3), dtype=np.float32) * alpha[:, None]
        img[:, 3] = alpha
        if vec == 5:
            depth = (1.0 + 3.0 * rng.random(p, dtype=np.float32)).astype(np.float32)
            img[:, 4] = np.where(alpha > 0, depth, np.inf)
        images.append(img)
    return images


def run_classic(kind, images, reverse=False):
    n_ranks = len(images)
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "images.bin"), os.path.join(tmp, "out.bin")
        with open(src, "wb") as fh:
            fh.write(struct.pack("<3i", W, H, n_ranks))
            for image in images:
                fh.write(np.ascontiguousarray(image).tobytes())
        cmd = [MPIEXEC, "-n", str(n_ranks), BINARY, "classic", str(kind), src, dst] + (
            ["reverse"] if reverse else [])
        subprocess.run(cmd, check=True, timeout=300)
        raw = open(dst, "rb").read()
    n = struct.unpack_from("<i", raw, 0)[0]
    regions = np.frombuffer(raw, "<i4", 2 * n, 4).reshape(n, 2).copy()
    dtype = np.uint32 if kind == 2 else np.float32
    image = np.frombuffer(raw, dtype, offset=4 + 8 * n).copy()
    return image, regions


CLASSIC = [(kind, n, reverse) for kind in (0, 1, 2) for n, reverse in ((1, False), (2, False), (2, True))]


def classic_name(kind, n, reverse):
    return f"classic_kind{kind}_n{n}" + ("_reversed" if reverse else "")


CASES = ([(n, policy, False, False) for n in (1, 2, 3, 4, 8) for policy in ("round_robin", "block")]
         + [(4, "block", True, False), (3, "round_robin", True, False), (3, "rank1_empty", False, False),
            (4, "block", False, True), (2, "round_robin", False, True)])


def case_name(n, policy, reverse, ties):
    return f"n{n}_{policy}" + ("_reversed" if reverse else "") + ("_tied_hints" if ties else "")


def main():
    if not os.path.exists(BINARY):
        status = subprocess.run(["bash", os.path.join(ROOT, "oracle", "ref_compose", "build.sh")]).returncode
        if status != 0:
            sys.exit("oracle/_ref/ref_compose cannot be built here (no reference tree / MPICH)")
    layers, hints = layers_and_hints()
    tied = [np.float32(1.5) if i in (1, 4, 6) else h for i, h in enumerate(hints)]
    out = {"width": W, "height": H, "layers": np.stack(layers), "hints": np.asarray(hints, np.float32),
           "tied_hints": np.asarray(tied, np.float32)}
    for n, policy, reverse, ties in CASES:
        owner = ownership(n, policy)
        image, regions, rgb8 = run_reference(layers, tied if ties else hints, owner, n, reverse)
        name = case_name(n, policy, reverse, ties)
        out[name + "/image"] = image
        out[name + "/rgb8"] = rgb8       # the reference's SavePPM of it: rows top-down
        out[name + "/regions"] = regions
        out[name + "/owner"] = np.asarray(owner, np.int32)
        print(f"{name}: pieces {regions.tolist()}")
    # the classic direct send of ONE plain image per rank (DirectSendBase.cpp:257-281): two ranks
    # have one blend and therefore one answer (the first group member on top)
    for kind, n, reverse in CLASSIC:
        images = classic_images(kind, n)
        image, regions = run_classic(kind, images, reverse)
        name = classic_name(kind, n, reverse)
        out[name + "/image"] = image
        out[name + "/regions"] = regions
        print(f"{name}: pieces {regions.tolist()}")
    np.savez_compressed(os.path.join(HERE, "ref_compose.npz"), **out)
    print("wrote tests/golden/ref_compose.npz")


if __name__ == "__main__":
    main()
