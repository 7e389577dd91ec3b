"""Sort-last DirectSend compositing, one rank per GPU.

Mirrors the reference's Compositor plugin (Common/Compositor.hpp:19-40) and its DirectSend
implementation for layered images (DirectSend/Base/DirectSendBase.cpp:316-458), redesigned for
xGMI: instead of one full direct-send round per run (N(N-1) MPI messages each, almost all of
them empty layers), every rank sends each peer ONE contiguous block holding the peer's pixel
piece of all of its run layers -- a single all-to-all per frame (RCCL over xGMI through
torch.distributed; gloo on CPU in the tests) -- and the receiver folds the runs in global order
with the HIP fold kernel.  Empty layer pixels (0,0,0,0,+inf) are an exact two-sided identity of
the depth-sort blend, so dropping the non-owners' empty contributions leaves every bit of the
result unchanged.

The pure planning part (layer order, runs, piece ranges, split sizes) has no device dependency
and is what the multi-process CPU tests exercise.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import runtime


@dataclass
class ExchangePlan:
    """Everything a rank needs to paint its runs and take part in the exchange."""
    n_ranks: int
    rank: int
    n_pixels: int
    # global layer order: parallel arrays over all layers, sorted by (hint, owner, local index)
    layer_owner: np.ndarray
    layer_local_index: np.ndarray
    # global runs, in order: owner rank and the index of the run among that owner's runs
    run_owner: List[int]
    run_local_id: List[int]
    runs_per_rank: List[int]
    # this rank's boxes in global order + run ends (input of avr_render_runs)
    local_order: np.ndarray
    local_run_end: np.ndarray
    # position of each rank in the ordered group (piece k belongs to group_order[k])
    group_order: List[int]

    @property
    def n_local_runs(self) -> int:
        return int(self.local_run_end.size)

    def piece_of_rank(self, rank: int) -> int:
        return self.group_order.index(rank)

    def piece_range(self, piece: int) -> Tuple[int, int]:
        return runtime.piece_range(self.n_pixels, piece, self.n_ranks)

    def my_piece_range(self) -> Tuple[int, int]:
        return self.piece_range(self.piece_of_rank(self.rank))

    # all_to_all_single split sizes, in floats, indexed by PEER RANK
    def send_splits(self) -> List[int]:
        sizes = []
        for peer in range(self.n_ranks):
            b, e = self.piece_range(self.piece_of_rank(peer))
            sizes.append(self.n_local_runs * (e - b) * 5)
        return sizes

    def recv_splits(self) -> List[int]:
        b, e = self.my_piece_range()
        return [self.runs_per_rank[peer] * (e - b) * 5 for peer in range(self.n_ranks)]


def plan_exchange(hints_by_rank: Sequence[Sequence[float]], rank: int, n_pixels: int,
                  group_order: Optional[Sequence[int]] = None) -> ExchangePlan:
    """Global order, run grouping and this rank's share (DirectSendBase.cpp:329-410).

    hints_by_rank[r][i] is the depth hint of local layer i of rank r (what the reference
    all-gathers).  group_order is the visibility-ordered group (identity when None); it only
    decides which pixel piece each rank ends up holding, never pixel values."""
    n_ranks = len(hints_by_rank)
    hints, owner, local_index = [], [], []
    for r, rank_hints in enumerate(hints_by_rank):
        for i, h in enumerate(rank_hints):
            hints.append(h)
            owner.append(r)
            local_index.append(i)
    order, run_end = runtime.layer_order(hints, owner, local_index)
    owner_a = np.asarray(owner, dtype=np.int32)
    local_a = np.asarray(local_index, dtype=np.int32)
    layer_owner = owner_a[order] if order.size else np.zeros(0, np.int32)
    layer_local = local_a[order] if order.size else np.zeros(0, np.int32)

    run_owner: List[int] = []
    run_local_id: List[int] = []
    runs_per_rank = [0] * n_ranks
    local_order: List[int] = []
    local_run_end: List[int] = []
    start = 0
    for end in run_end.tolist():
        o = int(layer_owner[start])
        run_owner.append(o)
        run_local_id.append(runs_per_rank[o])
        runs_per_rank[o] += 1
        if o == rank:
            local_order.extend(int(v) for v in layer_local[start:end])
            local_run_end.append(len(local_order))
        start = end
    group = list(group_order) if group_order is not None else list(range(n_ranks))
    if sorted(group) != list(range(n_ranks)):
        raise ValueError("group_order must be a permutation of the ranks")
    return ExchangePlan(n_ranks=n_ranks, rank=rank, n_pixels=int(n_pixels),
                        layer_owner=layer_owner, layer_local_index=layer_local,
                        run_owner=run_owner, run_local_id=run_local_id,
                        runs_per_rank=runs_per_rank,
                        local_order=np.asarray(local_order, dtype=np.int32),
                        local_run_end=np.asarray(local_run_end, dtype=np.int32),
                        group_order=group)


def send_block_offset(plan: ExchangePlan, peer: int) -> int:
    """Offset (floats) of the block for `peer` inside a rank's send-layout buffer written by
    avr_render_runs with n_pieces = n_ranks: 5 * n_local_runs * piece_begin(piece of peer)."""
    b, _ = plan.piece_range(plan.piece_of_rank(peer))
    return 5 * plan.n_local_runs * b


def exchange_permutation(plan: ExchangePlan) -> Optional[List[int]]:
    """avr_render_runs lays blocks out by PIECE index; all_to_all_single wants them by PEER
    rank.  With the identity group order both coincide (None)."""
    if plan.group_order == list(range(plan.n_ranks)):
        return None
    return [plan.piece_of_rank(peer) for peer in range(plan.n_ranks)]


def slices_in_run_order(plan: ExchangePlan, recv_buffer, piece_len: int):
    """Views of the received buffer ([src rank][run of src][piece pixel][5]) in global run
    order -- the operand list of the receiver-side fold (DirectSendBase.cpp:441-445)."""
    offsets = [0] * plan.n_ranks
    total = 0
    for peer in range(plan.n_ranks):
        offsets[peer] = total
        total += plan.runs_per_rank[peer] * piece_len * 5
    out = []
    for owner, local_id in zip(plan.run_owner, plan.run_local_id):
        begin = offsets[owner] + local_id * piece_len * 5
        out.append(recv_buffer[begin:begin + piece_len * 5])
    return out


class DirectSendCompositor:
    """compose(): run layers (send layout) -> this rank's fully composited pixel piece.

    `process_group` is a torch.distributed group (RCCL "nccl" backend on GPUs); None with a
    single rank skips the collective."""

    def __init__(self, ctx: "runtime.Context", process_group=None):
        self.ctx = ctx
        self.process_group = process_group

    def exchange(self, plan: ExchangePlan, send_buffer):
        import torch
        import torch.distributed as dist
        b, e = plan.my_piece_range()
        piece_len = e - b
        if plan.n_ranks == 1:
            return send_buffer, piece_len
        send_splits = plan.send_splits()
        recv_splits = plan.recv_splits()
        perm = exchange_permutation(plan)
        if perm is not None:  # reorder blocks from piece order to peer order
            chunks = []
            for peer in range(plan.n_ranks):
                off = send_block_offset(plan, peer)
                chunks.append(send_buffer[off:off + send_splits[peer]])
            send_buffer = torch.cat(chunks)
        recv = torch.empty(sum(recv_splits), dtype=send_buffer.dtype, device=send_buffer.device)
        dist.all_to_all_single(recv, send_buffer[:sum(send_splits)], recv_splits, send_splits,
                               group=self.process_group)
        return recv, piece_len

    def compose(self, plan: ExchangePlan, send_buffer):
        """Returns (piece tensor [piece_len, 5], piece_begin, piece_end)."""
        recv, piece_len = self.exchange(plan, send_buffer)
        slices = slices_in_run_order(plan, recv, piece_len)
        b, e = plan.my_piece_range()
        piece = self.ctx.fold_runs(slices, piece_len)
        return piece, b, e

    def gather(self, plan: ExchangePlan, piece, dst: int = 0):
        """ImageFull::Gather (Common/ImageColorOnly.hpp:220-270): pieces concatenated by
        region begin on rank `dst`; other ranks return None."""
        import torch
        import torch.distributed as dist
        if plan.n_ranks == 1:
            return piece
        lens = []
        for r in range(plan.n_ranks):
            b, e = plan.piece_range(plan.piece_of_rank(r))
            lens.append(e - b)
        max_len = max(lens)
        padded = piece
        if piece.shape[0] != max_len:  # equal-size gather; the last piece is the long one
            padded = torch.zeros(max_len, 5, dtype=piece.dtype, device=piece.device)
            padded[:piece.shape[0]] = piece
        if plan.rank == dst:
            parts = [torch.empty_like(padded) for _ in range(plan.n_ranks)]
            dist.gather(padded, parts, dst=dst, group=self.process_group)
            full = torch.empty(plan.n_pixels, 5, dtype=piece.dtype, device=piece.device)
            for r in range(plan.n_ranks):
                b, e = plan.piece_range(plan.piece_of_rank(r))
                full[b:e] = parts[r][:e - b]
            return full
        dist.gather(padded, None, dst=dst, group=self.process_group)
        return None
