"""Frame driver: what VolumeRenderer::renderSingleTrial does between "per-box rendering" and
the saved image (VolumeRenderer/VolumeRenderer.cpp:1103-1339), re-cut for one rank per GPU.

reference stage                                  here
-----------------------------------------------  ------------------------------------------------
referenceSampleDistance + MPI_Allreduce (:1138)  host, from replicated box metadata
per-box paint loop -> one W*H*5 layer per box    ONE fused HIP launch: paint + owner-side run fold,
  (:1201-1219) + owner-side run fold               written straight into DirectSend send layout
  (DirectSendBase.cpp:413-426)
allgather of layer counts / depth hints (:329)   host, from replicated box metadata (hints depend
                                                   only on box corners and the camera)
one direct-send round per run (:400-446)         one all-to-all per frame (RCCL over xGMI)
receiver blend chain                             HIP fold kernel over runs in global order
Gather to rank 0 (:1293)                         dist.gather of the pieces
AA downsample, 8-bit conversion (:479, SavePPM)  HIP kernels on rank 0
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import runtime, scenes
from .compositor import DirectSendCompositor, ExchangePlan, plan_exchange
from .types import AmrBox, CameraParameters, ScalarTransform, VolumeBounds, make_params


@dataclass
class RenderParameters:
    """VolumeRenderer::RenderParameters (VolumeRenderer/VolumeRenderer.hpp:33-44), the fields
    the hot path reads."""
    width: int = 512
    height: int = 512
    box_transparency: float = 0.0
    antialiasing: int = 1
    use_visibility_graph: bool = True


def validate_render_parameters(p: RenderParameters) -> int:
    """validateRenderParameters (VolumeRenderer.cpp:562-579); returns sqrt(antialiasing)."""
    if p.width <= 0 or p.height <= 0:
        raise ValueError("image dimensions must be positive")
    if not (0.0 <= p.box_transparency <= 1.0):
        raise ValueError("box_transparency must be in [0, 1]")
    if p.antialiasing < 1:
        raise ValueError("antialiasing must be >= 1")
    root = int(round(math.sqrt(p.antialiasing)))
    if root * root != p.antialiasing:
        raise ValueError("antialiasing must be a perfect square")
    return root


class FrameRenderer:
    """Renders frames of one scene on one rank.

    all_boxes: metadata of EVERY box of the scene (replicated on all ranks), with .owner set;
    local_boxes: this rank's boxes with cell data in HBM, in localBoxes order (level-major).
    """

    def __init__(self, ctx: runtime.Context, all_boxes: Sequence[AmrBox],
                 local_boxes: Sequence[AmrBox], transform: ScalarTransform,
                 bounds: VolumeBounds, scalar_range=(0.0, 1.0), rank: int = 0,
                 n_ranks: int = 1, process_group=None, color_map=None):
        self.ctx = ctx
        self.rank = rank
        self.n_ranks = n_ranks
        self.all_boxes = list(all_boxes)
        self.local_boxes = list(local_boxes)
        self.transform = transform
        self.bounds = bounds
        self.scalar_range = tuple(scalar_range)
        self.color_map = color_map
        self.scene = ctx.create_scene(self.local_boxes, transform)
        self.compositor = DirectSendCompositor(ctx, process_group)
        # localIndex of every box on its owner (position in that rank's localBoxes)
        self._by_rank: List[List[int]] = [[] for _ in range(n_ranks)]
        for i, b in enumerate(self.all_boxes):
            self._by_rank[b.owner].append(i)
        if len(self._by_rank[rank]) != len(self.local_boxes):
            raise ValueError("local_boxes does not match the ownership of all_boxes")
        # coarsest min spacing over all ranks == MPI_Allreduce(MAX) of VolumeRenderer.cpp:1166
        self.reference_sample_distance = runtime.reference_sample_distance(
            self.all_boxes, bounds.min_corner, bounds.max_corner)
        self._send: Optional[torch.Tensor] = None
        self.last_plan: Optional[ExchangePlan] = None

    # -- planning (host) -------------------------------------------------------------------------
    def plan(self, camera: CameraParameters, n_pixels: int,
             group_order: Optional[Sequence[int]] = None) -> ExchangePlan:
        hints_by_rank = [[runtime.box_depth_hint(self.all_boxes[i], camera) for i in idx]
                         for idx in self._by_rank]
        return plan_exchange(hints_by_rank, self.rank, n_pixels, group_order)

    # -- one frame ------------------------------------------------------------------------------
    def paint(self, plan: ExchangePlan, params, camera: CameraParameters,
              samples: Optional[torch.Tensor] = None, sync_streams: bool = True) -> torch.Tensor:
        need = max(plan.n_local_runs, 1) * plan.n_pixels * 5
        if self._send is None or self._send.numel() < need:
            self._send = self.ctx.empty(need)
        return self.scene.render_runs(params, camera, plan.local_order, plan.local_run_end,
                                      plan.n_ranks, out=self._send, samples=samples,
                                      sync_streams=sync_streams)

    def make_params(self, p: RenderParameters):
        root = validate_render_parameters(p)
        return make_params(p.width * root, p.height * root, self.scalar_range, p.box_transparency,
                           self.reference_sample_distance, self.bounds, self.color_map), root

    def render(self, p: RenderParameters, camera: CameraParameters,
               samples: Optional[torch.Tensor] = None, quantize: bool = True,
               group_order: Optional[Sequence[int]] = None):
        """Returns (image, rgb8) on rank 0 -- the gathered (downsampled) depth-sort image
        [H, W, 5] and its RGB8 bytes [H, W, 3] in file row order -- and (None, None) elsewhere."""
        params, root = self.make_params(p)
        n_pixels = params.width * params.height
        plan = self.plan(camera, n_pixels, group_order)
        self.last_plan = plan
        self.ctx.join()  # cell data / earlier torch work on the caller's stream
        with torch.cuda.stream(self.ctx.stream):
            send = self.paint(plan, params, camera, samples, sync_streams=False)
            if plan.n_ranks == 1 and plan.n_local_runs == 1:
                full = send[:n_pixels * 5].view(n_pixels, 5)  # one run on one rank: already final
            else:
                piece, _, _ = self.compositor.compose(plan, send)
                full = self.compositor.gather(plan, piece, dst=0)
            if full is None:
                return None, None
            image = full.view(params.height, params.width, 5)
            if root > 1:
                image = self.ctx.downsample(full.reshape(-1), p.width, p.height, root)
            rgb8 = self.ctx.quantize_rgb8(image.reshape(-1), p.width, p.height) if quantize else None
        return image, rgb8


def build_scene_on_device(ctx: runtime.Context, spec: scenes.SceneSpec, rank: int = 0):
    """Materialises this rank's boxes of a synthetic scene in HBM (torch, float64) and returns
    (all_boxes metadata, local_boxes)."""
    all_boxes = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    local = []
    for i in scenes.local_box_indices(spec, rank):
        local.append(scenes.amr_box(spec, i, scenes.box_cells_torch(spec, i, ctx.device)))
    return all_boxes, local
