"""Device-resident mirrors of the reference's image classes on the hot path.

  ImageRGBAFloatColorDepthSort   Common/ImageRGBAFloatColorDepthSort.hpp  (5 floats / pixel)
  ImageRGBAFloatColorOnly        Common/ImageRGBAFloatColorOnly.hpp       (4 floats / pixel)
  ImageRGBAUByteColorOnly        Common/ImageRGBAUByteColorOnly.hpp       (packed RGBA8)
  LayeredVolumeImage             Common/LayeredVolumeImage.hpp

Same semantics as Image / ImageFull / ImageColorOnly (Common/Image.hpp:134-266,
Common/ImageColorOnly.hpp:46-327): an image covers the pixel range [region_begin, region_end) of
a width x height frame; `window` is a zero-copy view, `copy_subrange` a copy, `blend` returns a
new image covering the union of the two regions.  Buffers are torch tensors in HBM; all
arithmetic runs in the HIP kernels behind the C ABI.

`compose_layered` is the reference-shaped entry of the compositor
(Compositor::compose on a LayeredImageInterface, DirectSend/Base/DirectSendBase.cpp:285-458) for
callers that already hold one layer per box; the frame driver uses the fused plan path instead.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from . import runtime


class _ImageColorOnly:
    KIND = ""
    VEC = 0
    DTYPE = torch.float32

    def __init__(self, ctx: runtime.Context, width: int, height: int,
                 region_begin: int = 0, region_end: Optional[int] = None,
                 buffer: Optional[torch.Tensor] = None):
        self.ctx = ctx
        self.width = int(width)
        self.height = int(height)
        self.region_begin = int(region_begin)
        self.region_end = self.width * self.height if region_end is None else int(region_end)
        if not (0 <= self.region_begin <= self.region_end <= self.width * self.height):
            raise ValueError("invalid image region")
        n = self.number_of_pixels * self.VEC
        if buffer is None:
            buffer = torch.zeros(n, dtype=self.DTYPE, device=ctx.device)
        if buffer.numel() != n:
            raise ValueError("buffer does not match the region")
        self.buffer = buffer

    # -- Image ---------------------------------------------------------------------------------
    @property
    def number_of_pixels(self) -> int:
        return self.region_end - self.region_begin

    def create_new(self, region_begin: Optional[int] = None, region_end: Optional[int] = None):
        rb = self.region_begin if region_begin is None else region_begin
        re = self.region_end if region_end is None else region_end
        return type(self)(self.ctx, self.width, self.height, rb, re)

    def shallow_copy(self):
        return type(self)(self.ctx, self.width, self.height, self.region_begin, self.region_end,
                          self.buffer)

    def deep_copy(self):
        return type(self)(self.ctx, self.width, self.height, self.region_begin, self.region_end,
                          self.buffer.clone())

    def window(self, subregion_begin: int, subregion_end: int):
        """Zero-copy view of pixels [begin, end) relative to this image (ImageFull.hpp:66-80)."""
        if not (0 <= subregion_begin <= subregion_end <= self.number_of_pixels):
            raise IndexError("window outside the image region")
        view = self.buffer[subregion_begin * self.VEC:subregion_end * self.VEC]
        return type(self)(self.ctx, self.width, self.height, self.region_begin + subregion_begin,
                          self.region_begin + subregion_end, view)

    def copy_subrange(self, subregion_begin: int, subregion_end: int):
        w = self.window(subregion_begin, subregion_end)
        w.buffer = w.buffer.clone()
        return w

    def blend(self, other):
        """this on top of `other` (ImageColorOnly.hpp:119-199)."""
        if type(other) is not type(self):
            raise TypeError("attempting to blend images of different types")
        if self.region_begin > other.region_end or other.region_begin > self.region_end:
            raise ValueError("regions neither overlap nor touch")
        out, ob, oe = self.ctx.blend_regions(
            self.KIND, self.buffer.contiguous(), self.region_begin, self.region_end,
            other.buffer.contiguous(), other.region_begin, other.region_end)
        return type(self)(self.ctx, self.width, self.height, ob, oe, out)

    def blend_is_order_dependent(self) -> bool:
        return True

    def to_host(self) -> np.ndarray:
        self.ctx.synchronize()
        return self.buffer.cpu().numpy().reshape(self.number_of_pixels, -1)


class ImageRGBAFloatColorDepthSort(_ImageColorOnly):
    KIND, VEC, DTYPE = "depthsort", 5, torch.float32

    def clear(self, color=(0.0, 0.0, 0.0, 0.0)) -> None:
        """encodeColor sets the depth to +inf (ImageRGBAFloatColorDepthSort.hpp:29-36)."""
        px = torch.tensor(list(color) + [float("inf")], dtype=torch.float32, device=self.ctx.device)
        self.buffer.view(-1, 5)[:] = px


class ImageRGBAFloatColorOnly(_ImageColorOnly):
    KIND, VEC, DTYPE = "rgba_f32", 4, torch.float32

    def clear(self, color=(0.0, 0.0, 0.0, 0.0)) -> None:
        self.buffer.view(-1, 4)[:] = torch.tensor(list(color), dtype=torch.float32,
                                                  device=self.ctx.device)


class ImageRGBAUByteColorOnly(_ImageColorOnly):
    KIND, VEC, DTYPE = "rgba_u8", 1, torch.int32

    def clear(self, color=(0.0, 0.0, 0.0, 0.0)) -> None:
        rgba = torch.tensor([list(color)], dtype=torch.float32, device=self.ctx.device)
        self.buffer[:] = self.ctx.encode_rgba_u8(rgba.reshape(-1))[0]

    def set_colors(self, rgba: torch.Tensor) -> None:
        """setColor for every pixel: Color::GetComponentAsByte (Color.hpp:86-90)."""
        self.buffer = self.ctx.encode_rgba_u8(rgba.contiguous().reshape(-1))

    def get_colors(self) -> torch.Tensor:
        return self.ctx.decode_rgba_u8(self.buffer.contiguous())


class LayeredVolumeImage:
    """LayeredImageInterface (Common/LayeredImageInterface.hpp:9-28): N local layers with one
    depth hint each."""

    def __init__(self, width: int, height: int,
                 layers: Sequence[ImageRGBAFloatColorDepthSort], depth_hints: Sequence[float]):
        if len(layers) != len(depth_hints):
            raise ValueError("one depth hint per layer")
        self.width, self.height = int(width), int(height)
        self.layers = list(layers)
        self.depth_hints = [float(h) for h in depth_hints]

    def get_layer_count(self) -> int:
        return len(self.layers)

    def get_layer(self, index: int) -> ImageRGBAFloatColorDepthSort:
        return self.layers[index]

    def get_layer_depth_hint(self, index: int) -> float:
        return self.depth_hints[index]

    def create_empty_layer(self, ctx: runtime.Context, region_begin: int, region_end: int):
        layer = ImageRGBAFloatColorDepthSort(ctx, self.width, self.height, region_begin, region_end)
        layer.clear()
        return layer


def compose_layered(ctx: runtime.Context, layered: LayeredVolumeImage, rank: int = 0,
                    n_ranks: int = 1, process_group=None, group_order: Optional[Sequence[int]] = None,
                    stage_through_host: bool = False) -> ImageRGBAFloatColorDepthSort:
    """Compositor::compose for a layered image (DirectSendBase.cpp:285-458): returns this
    rank's fully composited pixel piece.  Layer counts and hints are all-gathered as in the
    reference (:329-361); each run is folded on its owner (:413-426); the pieces of all local
    runs travel in one all-to-all instead of one direct-send round per run; the receiver folds
    the runs in global order (:441-445)."""
    import torch.distributed as dist
    n_pixels = layered.width * layered.height
    hints_by_rank: List[List[float]] = [list(layered.depth_hints)]
    if n_ranks > 1:
        gathered: List[Optional[List[float]]] = [None] * n_ranks
        dist.all_gather_object(gathered, list(layered.depth_hints), group=process_group)
        hints_by_rank = [list(h) for h in gathered]
    hints, owner, local = [], [], []
    for r, hs in enumerate(hints_by_rank):
        for i, h in enumerate(hs):
            hints.append(h)
            owner.append(r)
            local.append(i)
    order, run_end = runtime.layer_order(hints, owner, local)
    group = list(group_order) if group_order is not None else list(range(n_ranks))
    my_piece = group.index(rank)

    # owner-side fold of every local run, then the dense piece slices, grouped by peer rank
    runs, start = [], 0
    for end in run_end.tolist():
        runs.append((owner[order[start]], [local[i] for i in order[start:end]]))
        start = end
    local_run_layers = []
    for run_owner, members in runs:
        if run_owner != rank:
            continue
        acc = layered.get_layer(members[0])
        for m in members[1:]:
            acc = acc.blend(layered.get_layer(m))
        local_run_layers.append(acc.buffer.contiguous())
    ranges = [runtime.piece_range(n_pixels, k, n_ranks) for k in range(n_ranks)]
    runs_per_rank = [sum(1 for o, _ in runs if o == r) for r in range(n_ranks)]
    b, e = ranges[my_piece]
    piece_len = e - b
    if n_ranks == 1:
        slices = [t[b * 5:e * 5] for t in local_run_layers]
    else:
        chunks, send_splits = [], []
        for peer in range(n_ranks):
            pb, pe = ranges[group.index(peer)]
            for t in local_run_layers:
                chunks.append(t[pb * 5:pe * 5])
            send_splits.append(len(local_run_layers) * (pe - pb) * 5)
        send = torch.cat(chunks) if chunks else torch.zeros(0, device=ctx.device)
        recv_splits = [runs_per_rank[s] * piece_len * 5 for s in range(n_ranks)]
        if stage_through_host:
            host = torch.empty(sum(recv_splits), dtype=torch.float32)
            dist.all_to_all_single(host, send.cpu(), recv_splits, send_splits, group=process_group)
            recv = host.to(ctx.device)
        else:
            recv = torch.empty(sum(recv_splits), dtype=torch.float32, device=ctx.device)
            dist.all_to_all_single(recv, send, recv_splits, send_splits, group=process_group)
        offsets, total = [], 0
        for s in range(n_ranks):
            offsets.append(total)
            total += recv_splits[s]
        seen = [0] * n_ranks
        slices = []
        for run_owner, _ in runs:  # global order
            at = offsets[run_owner] + seen[run_owner] * piece_len * 5
            seen[run_owner] += 1
            slices.append(recv[at:at + piece_len * 5])
    piece = ctx.fold_runs([s.contiguous() for s in slices], piece_len)
    return ImageRGBAFloatColorDepthSort(ctx, layered.width, layered.height, b, e, piece.reshape(-1))
