"""GPU rehearsal of the N > 1 frame: 3 processes share cuda:0 (within the box's limit of 6),
each owning a third of the boxes, run FrameRenderer.render with the HIP kernels and exchange
over gloo through host copies (RCCL cannot place two ranks on one device).  Rank 0's frame must
be bit-identical to the oracle's 3-rank layered compose."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 120, 72


def _worker(rank, world, port, policy, antialiasing, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from amrvolumerenderer_amd import runtime, scenes
        from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters
        from helpers import device_box
        from test_frame_plan import local_indices, oracle_overlay, painted_scene

        root = int(round(antialiasing ** 0.5))
        spec = scenes.make_amr_scene(32, 2, 8, "smooth")
        cam = scenes.orbit_camera(3)
        cells, layers, hints, ref = painted_scene(O, spec, cam, W * root, H * root, 0.85)
        scenes.assign_owners(spec, world, policy)
        ctx = runtime.Context(0)
        meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
        local = [device_box(ctx, cells[i], spec.boxes[i].min_corner, spec.boxes[i].max_corner,
                            spec.boxes[i].level, rank)
                 for i in scenes.local_box_indices(spec, rank)]
        renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds, spec.scalar_range,
                                 rank, world, dist.group.WORLD, stage_through_host=True)
        samples = torch.zeros(1, dtype=torch.int64, device=ctx.device)
        image, rgb8 = renderer.render(RenderParameters(W, H, 0.85, antialiasing), cam,
                                      samples=samples, want_image=True)
        renderer.synchronize()
        if rank == 0:
            owners = [b.owner for b in spec.boxes]
            want, _, _ = O.compose_layered(layers, hints, owners, local_indices(owners, world),
                                           world)
            if root > 1:
                want = O.downsample(want, W, H, root).reshape(-1, 5)
            # RenderParameters.draw_bounds defaults to the reference's behaviour: every rank
            # overlays its own piece (or rank 0 the downsampled image)
            want = oracle_overlay(O, spec, cells, cam, want, W, H)
            ok = np.array_equal(image.cpu().numpy().reshape(-1, 5).view(np.uint32),
                                want.view(np.uint32))
            ok8 = np.array_equal(rgb8.cpu().numpy(), O.quantize_rgb8(want, W, H))
            with open(out_path, "w") as fh:
                fh.write(f"{int(ok)} {int(ok8)}")
        else:
            assert image is None and rgb8 is None
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("policy,antialiasing", [("morton", 1), ("round_robin", 4)])
def test_three_ranks_on_one_gpu(tmp_path, policy, antialiasing):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(3, _free_port(), policy, antialiasing, str(out)), nprocs=3, join=True)
    ok, ok8 = out.read_text().split()
    assert ok == "1", "rank 0's frame differs from the oracle's 3-rank compose"
    assert ok8 == "1", "RGB8 bytes differ"


def _rccl_worker(rank, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from oracle import oracle as O
        from amrvolumerenderer_amd import runtime, scenes
        from amrvolumerenderer_amd.renderer import FrameRenderer, RenderParameters
        from helpers import device_box
        from test_frame_plan import oracle_overlay, painted_scene

        results = []
        for antialiasing, (w, h) in ((1, (120, 72)), (4, (60, 36)), (1, (75, 43))):
            root = int(round(antialiasing ** 0.5))
            spec = scenes.make_amr_scene(32, 2, 8, "smooth")
            cam = scenes.orbit_camera(5)
            cells, layers, hints, ref = painted_scene(O, spec, cam, w * root, h * root, 0.85)
            ctx = runtime.Context(0)
            meta = [scenes.metadata_box(spec, i) for i in range(len(cells))]
            local = [device_box(ctx, c, m.min_corner, m.max_corner, m.level)
                     for c, m in zip(cells, spec.boxes)]
            renderer = FrameRenderer(ctx, meta, local, spec.transform, spec.bounds,
                                     spec.scalar_range, 0, 1, dist.group.WORLD,
                                     force_collectives=True)
            frames = [renderer.render(RenderParameters(w, h, 0.85, antialiasing), cam,
                                      want_image=True) for _ in range(3)]  # pipelined burst
            renderer.synchronize()
            want, _, _ = O.compose_layered(layers, hints, [0] * len(layers),
                                           np.arange(len(layers)), 1)
            if root > 1:
                want = O.downsample(want, w, h, root).reshape(-1, 5)
            want = oracle_overlay(O, spec, cells, cam, want, w, h)
            for image, rgb8 in frames:
                results.append(np.array_equal(image.cpu().numpy().reshape(-1, 5).view(np.uint32),
                                              want.view(np.uint32)))
                results.append(np.array_equal(rgb8.cpu().numpy(), O.quantize_rgb8(want, w, h)))
        t = torch.ones(1, device="cuda:0")
        dist.all_reduce(t)
        dist.barrier()
        with open(out_path, "w") as fh:
            fh.write(" ".join(str(int(r)) for r in results))
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_with_one_rank(tmp_path):
    """The RCCL code path itself (all_to_all_single with split lists, gather of uint8 and float
    pieces into views, three streams, process-group stream ordering) on a one-rank "nccl" group:
    the only RCCL configuration a one-GPU box can run."""
    out = tmp_path / "result.txt"
    mp.spawn(_rccl_worker, args=(_free_port(), str(out)), nprocs=1, join=True)
    flags = out.read_text().split()
    assert flags and all(f == "1" for f in flags), flags
