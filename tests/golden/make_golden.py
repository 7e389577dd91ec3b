#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.npz from the ORACLE (oracle/avr_oracle.c).

Provenance: these are not reference outputs (the reference cannot be built here, see
oracle/README.md); they freeze the oracle's behaviour at the round it was written so that a
later edit of the oracle, the compiler flags or the scene generators shows up as a diff.  The
GPU tests also compare the HIP path against them, so parity is checked against committed data
and not only against a freshly compiled checker.

    python tests/golden/make_golden.py        # rewrites oracle_vectors.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LAB_MAP = [(0.0, 0.0, 0.0, 0.2, 0.0), (0.25, 0.1, 0.3, 0.9, 0.1), (0.5, 0.9, 0.9, 0.2, 0.4),
           (0.8, 1.0, 0.3, 0.0, 0.7), (1.0, 1.0, 1.0, 1.0, 1.0)]


def cases():
    """name -> (scene kwargs, image size, transparency, color map, n_ranks, policy)"""
    return {
        "amr2_smooth_opaque": (dict(n0=32, levels=2, box_cells=8, field_name="smooth"), (48, 32), 0.0, None, 1, "morton"),
        "amr2_smooth_translucent_4ranks": (dict(n0=32, levels=2, box_cells=8, field_name="smooth"), (45, 31), 0.9, None, 4, "morton"),
        "amr1_noise_lab_nonpow2": (dict(n0=16, levels=1, box_cells=8, field_name="noise", extent=0.7), (40, 40), 0.5, LAB_MAP, 2, "round_robin"),
    }


def compute(O, name):
    from amrvolumerenderer_amd import scenes
    from helpers import oracle_camera, oracle_params, oracle_transform, scene_cells
    kw, (W, H), transparency, cmap, n_ranks, policy = cases()[name]
    spec = scenes.make_amr_scene(**kw)
    cam = scenes.default_camera()
    if kw.get("extent", 1.0) != 1.0:
        e = kw["extent"]
        cam.eye = tuple(c * e for c in cam.eye)
        cam.look_at = tuple(c * e for c in cam.look_at)
    cells = scene_cells(spec)
    oboxes = [O.make_box(c, m.min_corner, m.max_corner) for c, m in zip(cells, spec.boxes)]
    ref = O.reference_sample_distance(oboxes, spec.bounds.min_corner, spec.bounds.max_corner)
    op = oracle_params(O, W, H, spec.scalar_range, transparency, ref, spec.bounds, cmap)
    ocam, otr = oracle_camera(O, cam), oracle_transform(O, spec.transform)
    layers, hints, samples = [], [], 0
    for ob in oboxes:
        img, n = O.paint_box(ob, otr, op, ocam)
        layers.append(img)
        hints.append(O.box_depth_hint(ob, ocam))
        samples += n
    scenes.assign_owners(spec, n_ranks, policy)
    owners = [b.owner for b in spec.boxes]
    local = np.zeros(len(owners), np.int32)
    for r in range(n_ranks):
        idx = [i for i, o in enumerate(owners) if o == r]
        local[idx] = np.arange(len(idx))
    frame, _, runs = O.compose_layered(layers, hints, owners, local, n_ranks)
    return {
        "frame": frame.astype(np.float32),
        "rgb8": O.quantize_rgb8(frame, W, H),
        "first_layer": layers[0].astype(np.float32),
        "hints": np.asarray(hints, np.float32),
        "samples": np.int64(samples),
        "runs": np.int64(runs),
        "reference_sample_distance": np.float32(ref),
    }


def main():
    from oracle import oracle as O
    out = {}
    for name in cases():
        for key, value in compute(O, name).items():
            out[f"{name}/{key}"] = value
    out["table/jet_nf1"] = O.build_color_table(1.0, 1.0)
    out["table/jet_nf0.25_a0.03"] = O.build_color_table(0.03, 0.25)
    out["table/lab_nf0.5"] = O.build_color_table(0.85, 0.5, (0.0, 1.0), LAB_MAP)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
