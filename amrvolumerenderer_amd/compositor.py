"""Sort-last DirectSend compositing, one rank per GPU.

Mirrors the reference's Compositor plugin (Common/Compositor.hpp:19-40) and its DirectSend
implementation for layered images (DirectSend/Base/DirectSendBase.cpp:316-458), redesigned for
xGMI: instead of one full direct-send round per run (N(N-1) MPI messages each, almost all of
them empty full-frame layers), every rank sends each peer ONE contiguous block holding, for each
of its runs, only the rows/columns of the run's screen rectangle that fall into the peer's pixel
piece -- a single all-to-all per frame (RCCL over xGMI through torch.distributed; gloo on CPU in
the tests) -- and the receiver folds the runs in global order with a HIP kernel.  Empty layer
pixels (0,0,0,0,+inf) are an exact two-sided identity of the depth-sort blend, so dropping them
leaves every bit of the result unchanged.

The plan itself (layer order, runs, rectangles, block offsets, split sizes) is computed by the
C ABI's avr_frame_plan_* (host only, no GPU needed) identically on every rank.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _capi
from .types import AmrBox, CameraParameters


class FramePlan:
    """avr_frame_plan: one frame's global layer order, runs and sparse exchange layout."""

    def __init__(self, all_boxes: Sequence[AmrBox], params: _capi.PaintParams,
                 camera: CameraParameters, rank: int = 0, n_ranks: int = 1,
                 group_order: Optional[Sequence[int]] = None, _box_array=None, _owner_array=None,
                 piece_layout: int = _capi.PIECES_CONTIGUOUS, band_rows: int = 1):
        n = len(all_boxes)
        # the C arrays can be prepared once per scene and re-used for every frame
        self._boxes = _box_array if _box_array is not None else make_box_array(all_boxes)
        self._owner = _owner_array if _owner_array is not None else make_owner_array(all_boxes)
        group = None
        if group_order is not None:
            group = (C.c_int32 * n_ranks)(*[int(g) for g in group_order])
        ccam = camera.to_c()
        handle = C.c_void_p()
        _capi.check(_capi.lib().avr_frame_plan_create_pieces(
            self._boxes, self._owner, n, int(n_ranks), int(rank), group, C.byref(params),
            C.byref(ccam), int(piece_layout), int(band_rows), C.byref(handle)))
        self._handle = handle
        self._params = params  # keeps the colour map alive
        info = _capi.FramePlanInfo()
        _capi.check(_capi.lib().avr_frame_plan_get_info(self._handle, C.byref(info)))
        self.n_ranks = info.n_ranks
        self.rank = info.rank
        self.n_runs_total = info.n_runs_total
        self.n_local_runs = info.n_local_runs
        self.n_local_boxes = info.n_local_boxes
        self.n_pixels = info.n_pixels
        self.piece_begin = info.piece_begin
        self.piece_end = info.piece_end
        self.send_floats = info.send_floats
        self.recv_floats = info.recv_floats
        self.piece_layout = info.piece_layout
        self.band_rows = info.band_rows
        self.width = int(params.width)
        self.height = int(params.height)
        self._read_splits()
        order = list(group_order) if group_order is not None else list(range(self.n_ranks))
        self.group_order = [int(g) for g in order]
        self.piece_of_rank = [self.group_order.index(r) for r in range(self.n_ranks)]

    def _read_splits(self) -> None:
        send = (C.c_int64 * self.n_ranks)()
        recv = (C.c_int64 * self.n_ranks)()
        _capi.check(_capi.lib().avr_frame_plan_splits(self._handle, send, recv))
        self.send_splits = [int(v) for v in send]
        self.recv_splits = [int(v) for v in recv]

    def tighten(self) -> None:
        """avr_frame_plan_tighten: per-row extents instead of the runs' rectangles (every rank of
        the frame, or none); send_floats / recv_floats and the splits shrink."""
        _capi.check(_capi.lib().avr_frame_plan_tighten(self._handle, self._boxes,
                                                       len(self._boxes)))
        info = _capi.FramePlanInfo()
        _capi.check(_capi.lib().avr_frame_plan_get_info(self._handle, C.byref(info)))
        self.send_floats = info.send_floats
        self.recv_floats = info.recv_floats
        self._read_splits()

    def close(self) -> None:
        if getattr(self, "_handle", None):
            _capi.lib().avr_frame_plan_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inspection (tests, tools) ----------------------------------------------------------
    def layers(self) -> np.ndarray:
        """Global layer order: indices into all_boxes."""
        n = len(self._boxes) if self.n_runs_total else 0
        out = np.zeros(max(n, 1), dtype=np.int32)
        _capi.check(_capi.lib().avr_frame_plan_layers(
            self._handle, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out[:n] if self.n_runs_total else out[:0]

    def runs(self) -> List[_capi.RunInfo]:
        arr = (_capi.RunInfo * max(self.n_runs_total, 1))()
        _capi.check(_capi.lib().avr_frame_plan_runs(self._handle, arr))
        return list(arr)[:self.n_runs_total]

    def send_block(self, peer: int, local_run: int) -> Tuple[int, int, int]:
        off, first, rows = C.c_int64(), C.c_int32(), C.c_int32()
        _capi.check(_capi.lib().avr_frame_plan_send_block(self._handle, peer, local_run,
                                                          C.byref(off), C.byref(first),
                                                          C.byref(rows)))
        return off.value, first.value, rows.value

    def recv_block(self, global_run: int) -> Tuple[int, int, int]:
        off, first, rows = C.c_int64(), C.c_int32(), C.c_int32()
        _capi.check(_capi.lib().avr_frame_plan_recv_block(self._handle, global_run, C.byref(off),
                                                          C.byref(first), C.byref(rows)))
        return off.value, first.value, rows.value


def make_box_array(all_boxes: Sequence[AmrBox]):
    return (_capi.Box * max(len(all_boxes), 1))(*[b.to_c() for b in all_boxes])


def make_owner_array(all_boxes: Sequence[AmrBox]):
    return (C.c_int32 * max(len(all_boxes), 1))(*[int(b.owner) for b in all_boxes])


class DirectSendCompositor:
    """The exchange + fold + gather of one frame.  `ops` supplies the device work
    (runtime.Context on a GPU; the CPU tests plug in a stand-in built on the oracle);
    `process_group` is a torch.distributed group (RCCL "nccl" backend on GPUs)."""

    def __init__(self, ops, process_group=None, stage_through_host: bool = False,
                 force_collectives: bool = False):
        self.ops = ops
        self.process_group = process_group
        # issue the all-to-all and the gather even for a single rank (a one-rank RCCL group on
        # one GPU exercises the real collective path: tests/test_distributed_gpu.py)
        self.force_collectives = force_collectives
        # rehearsal mode: a gloo group with device tensors (several ranks sharing one GPU);
        # collectives then run on host copies.  Never used with the RCCL backend.
        self.stage_through_host = stage_through_host

    def exchange(self, plan: FramePlan, send_buffer):
        """One all-to-all: block for peer s -> rank s."""
        import torch
        import torch.distributed as dist
        if plan.n_ranks == 1 and not self.force_collectives:
            return send_buffer
        if self.stage_through_host:
            host_recv = torch.empty(max(plan.recv_floats, 1), dtype=send_buffer.dtype)
            dist.all_to_all_single(host_recv[:plan.recv_floats],
                                   send_buffer[:plan.send_floats].cpu(), plan.recv_splits,
                                   plan.send_splits, group=self.process_group)
            return host_recv.to(send_buffer.device)
        recv = torch.empty(max(plan.recv_floats, 1), dtype=send_buffer.dtype,
                           device=send_buffer.device)
        dist.all_to_all_single(recv[:plan.recv_floats], send_buffer[:plan.send_floats],
                               plan.recv_splits, plan.send_splits, group=self.process_group)
        return recv

    def compose(self, plan: FramePlan, send_buffer, want_rgb8: bool = False,
                on_ops_stream: bool = False, want_piece: bool = True):
        """Returns (piece [piece_len, 5] or None, rgb8 [piece_len, 3] or None).  on_ops_stream:
        the caller already made the ops' stream current (no extra stream ordering needed)."""
        recv = self.exchange(plan, send_buffer)
        if not want_piece:  # only the bytes are wanted (the ops skip the 20 B/pixel float store)
            return self.ops.fold_plan(plan, recv, want_rgb8, sync_streams=not on_ops_stream,
                                      want_piece=False)
        if on_ops_stream:
            return self.ops.fold_plan(plan, recv, want_rgb8, sync_streams=False)
        return self.ops.fold_plan(plan, recv, want_rgb8)

    def gather(self, plan: FramePlan, piece, dst: int = 0):
        """ImageFull::Gather (Common/ImageColorOnly.hpp:220-270): pieces concatenated by
        region begin on rank `dst`; other ranks return None.  Works for any per-pixel tensor
        [piece_len, C] (the float image or its RGB8 bytes)."""
        import torch
        import torch.distributed as dist
        if plan.n_ranks == 1 and not self.force_collectives:
            return piece
        if self.stage_through_host and piece.device.type != "cpu":
            device = piece.device
            self.stage_through_host = False
            try:
                full = self.gather(plan, piece.cpu(), dst)
            finally:
                self.stage_through_host = True
            return full.to(device) if full is not None else None
        from . import runtime
        ranges = [runtime.piece_range(plan.n_pixels, k, plan.n_ranks) for k in range(plan.n_ranks)]
        max_len = max(e - b for b, e in ranges)
        if all(e - b == max_len for b, e in ranges):
            # equal pieces (the image size is a multiple of the rank count): every rank's piece
            # is received straight into its place in the full image
            if plan.rank != dst:
                dist.gather(piece, None, dst=dst, group=self.process_group)
                return None
            full = torch.empty((plan.n_pixels,) + tuple(piece.shape[1:]), dtype=piece.dtype,
                               device=piece.device)
            parts = []
            for r in range(plan.n_ranks):
                b, e = ranges[plan.piece_of_rank[r]]
                parts.append(full[b:e])
            dist.gather(piece, parts, dst=dst, group=self.process_group)
            return full
        padded = piece
        if piece.shape[0] != max_len:  # equal-size gather; only the last piece can be longer
            padded = torch.zeros((max_len,) + tuple(piece.shape[1:]), dtype=piece.dtype,
                                 device=piece.device)
            padded[:piece.shape[0]] = piece
        if plan.rank == dst:
            parts = [torch.empty_like(padded) for _ in range(plan.n_ranks)]
            dist.gather(padded, parts, dst=dst, group=self.process_group)
            full = torch.empty((plan.n_pixels,) + tuple(piece.shape[1:]), dtype=piece.dtype,
                               device=piece.device)
            # rank r holds the piece of its group position
            info_order = plan_piece_of_rank(plan)
            for r in range(plan.n_ranks):
                b, e = ranges[info_order[r]]
                full[b:e] = parts[r][:e - b]
            return full
        dist.gather(padded, None, dst=dst, group=self.process_group)
        return None


def plan_piece_of_rank(plan: FramePlan) -> List[int]:
    """Piece index held by each rank (its position in the ordered group)."""
    return plan.piece_of_rank
