"""Multi-process CPU test of the N > 1 path: 2 ranks over torch.distributed (gloo) run the
DirectSend exchange (one all_to_all_single), the fold and the gather of
amrvolumerenderer_amd.compositor.DirectSendCompositor.  No HIP kernel can run here, so the
device work (run layers -> send buffer, receiver fold) is supplied by an oracle-backed stand-in;
what is under test is the host logic: plan, split sizes, buffer routing, piece ownership,
gather order."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

from helpers import spawn_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, TRANSPARENCY = 75, 43, 0.8


class OracleOps:
    """Stand-in for runtime.Context in CPU tests (fold on the CPU with the oracle's blend)."""

    def __init__(self, O):
        self.O = O

    def fold_plan(self, plan, recv, want_rgb8=False):
        import plan_helpers as PH
        piece = PH.fold_recv_buffer(self.O, plan, recv.numpy())
        rgb8 = None
        if want_rgb8:
            rgb8 = torch.from_numpy(
                self.O.quantize_rgb8(piece, piece.shape[0], 1)[::-1].reshape(-1, 3).copy())
        return torch.from_numpy(piece), rgb8


def _worker(rank, world, port, policy, group, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from amrvolumerenderer_amd import scenes
        from amrvolumerenderer_amd.compositor import DirectSendCompositor, FramePlan
        from amrvolumerenderer_amd.types import make_params
        import plan_helpers as PH
        from test_frame_plan import local_indices, painted_scene

        spec = scenes.make_amr_scene(32, 2, 8, "smooth")
        cam = scenes.default_camera()
        cells, layers, hints, ref = painted_scene(O, spec, cam, W, H, TRANSPARENCY)
        scenes.assign_owners(spec, world, policy)
        owners = [b.owner for b in spec.boxes]
        boxes = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
        params = make_params(W, H, spec.scalar_range, TRANSPARENCY, ref, spec.bounds)
        plan = FramePlan(boxes, params, cam, rank, world, group)
        # this rank's share of the work: only its own run layers
        run_layers = PH.oracle_run_layers(O, layers, plan)
        for g, run in enumerate(plan.runs()):
            if run.owner != rank:
                run_layers[g] = None
        send = torch.from_numpy(PH.pack_send_buffer(plan, run_layers, fill=0.0))
        compositor = DirectSendCompositor(OracleOps(O), dist.group.WORLD)
        piece, rgb8 = compositor.compose(plan, send, want_rgb8=True)
        full = compositor.gather(plan, piece, dst=0)
        full8 = compositor.gather(plan, rgb8, dst=0)
        if rank == 0:
            want, _, _ = O.compose_layered(layers, hints, owners, local_indices(owners, world),
                                           world, group_order=group)
            ok = np.array_equal(full.numpy().view(np.uint32), want.view(np.uint32))
            want8 = O.quantize_rgb8(want, W, H)[::-1].reshape(-1, 3)
            ok8 = np.array_equal(full8.numpy(), want8)
            with open(out_path, "w") as fh:
                fh.write(f"{int(ok)} {int(ok8)} {plan.n_runs_total}")
        else:
            assert full is None and full8 is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,policy,group", [(2, "morton", None), (2, "round_robin", [1, 0]),
                                                (3, "block", [2, 0, 1])])
def test_directsend_over_gloo(tmp_path, O, avr_lib, world, policy, group):
    out = tmp_path / "result.txt"
    spawn_ranks(_worker, world, lambda port: (world, port, policy, group, str(out)))
    ok, ok8, runs = out.read_text().split()
    assert ok == "1", "gathered float image differs from the oracle's layered compose"
    assert ok8 == "1", "gathered RGB8 bytes differ"
    assert int(runs) >= world


def _control_worker(rank, world, port, name, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        from amrvolumerenderer_amd import _capi, runtime, scenes
        from amrvolumerenderer_amd.compositor import FramePlan
        from amrvolumerenderer_amd.types import make_params

        lib = _capi.lib()
        spec = scenes.make_amr_scene(32, 2, 8, "smooth")
        scenes.assign_owners(spec, world, "level_pairs")
        boxes = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
        params = make_params(W, H, spec.scalar_range, TRANSPARENCY, 0.01, spec.bounds)
        # the communicator of the multi-process rehearsal, its control plane the CALLER's: gloo,
        # reached from inside the C ABI through the callback (what bench.py --gpus N installs)
        comm = runtime.Comm.shared(name, rank, world, 1 << 20)
        comm.set_control(runtime.control_over_process_group(dist.group.WORLD))
        results = []
        everybody = comm.control_allgather(bytes([rank + 1] * 8))
        results.append(everybody == [bytes([r + 1] * 8) for r in range(world)])
        plan = FramePlan(boxes, params, scenes.default_camera(), rank, world)
        _capi.check(lib.avr_frame_plan_agree(plan._handle, comm._handle, None, 3))
        results.append(True)
        # rank 1 is handed another camera: every rank gets the error, over gloo too
        other = FramePlan(boxes, params, scenes.orbit_camera(5) if rank == 1 else scenes.default_camera(),
                          rank, world)
        try:
            _capi.check(lib.avr_frame_plan_agree(other._handle, comm._handle, None, 3))
            results.append(False)
        except _capi.AvrError as error:
            results.append("rank 1's plan differs from rank 0's" in str(error))
        results.append(comm.control_rounds() == 3)
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as fh:
            fh.write(" ".join(str(int(bool(r))) for r in results))
        dist.barrier()
        comm.close()
    finally:
        dist.destroy_process_group()


def test_plan_agreement_over_the_callers_control_plane(tmp_path, avr_lib):
    """Two and three rank PROCESSES on the CPU: the communicator's small host-side agreements
    (avr_comm_control_allgather, avr_frame_plan_agree) travel over the caller's control plane --
    gloo here, MPI_Allgather in the reference's host -- through the C ABI's callback."""
    for world in (2, 3):
        name = f"/avr_control_{os.getpid()}_{world}"
        out_dir = tmp_path / str(world)
        out_dir.mkdir()
        spawn_ranks(_control_worker, world, lambda port: (world, port, name, str(out_dir)))
        for rank in range(world):
            assert (out_dir / f"rank{rank}.txt").read_text() == "1 1 1 1", (world, rank)
