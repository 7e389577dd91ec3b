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
AA downsample (:479)                             HIP kernel on rank 0
wireframe of the tight bounds (:139, :1311)      HIP kernel; without antialiasing each rank
                                                   overlays its own piece before the gather
8-bit conversion (SavePPM)                       fused into the fold / overlay kernel
"""
from __future__ import annotations

import dataclasses
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import runtime, scenes
from .compositor import DirectSendCompositor, FramePlan, make_box_array, make_owner_array
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
    # The reference always draws the wireframe of the tight bounds over the final image
    # (VolumeRenderer.cpp:1311-1314).  SURVEY.md 8(d)'s frames/s metric times paint + composite +
    # gather + downsample + quantise, so the bench switches it off.
    draw_bounds: bool = True
    write_visibility_graph: bool = False


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
                 n_ranks: int = 1, process_group=None, color_map=None,
                 stage_through_host: bool = False, force_collectives: bool = False,
                 march_workgroups_per_cu: Optional[int] = None,
                 stream_priorities: Sequence[int] = (-1, -1, 0),
                 cache_classification: bool = False, native: Optional[bool] = None,
                 comm: Optional["runtime.Comm"] = None):
        """native (default: whenever possible): frames are driven by the C++ frame driver behind
        the C ABI (avr_renderer: three streams, RCCL exchange and gather) and this class only
        forwards camera, parameters and output tensors.  The Python pipeline below is the same
        frame expressed with torch streams and torch.distributed collectives; it remains for
        rehearsals of N ranks on one GPU over gloo (stage_through_host), for the one-rank
        collective check (force_collectives) and for the tools that time single stages."""
        self.ctx = ctx
        self.rank = rank
        self.n_ranks = n_ranks
        self.all_boxes = list(all_boxes)
        self.local_boxes = list(local_boxes)
        self.transform = transform
        self.bounds = bounds
        self.native = None
        self._scalar_range = tuple(scalar_range)
        self.color_map = color_map
        self.scene = ctx.create_scene(self.local_boxes, transform)
        # Off by default: every frame re-reads the f64 cells like the reference.  On: a camera
        # moving over static data only pays for the march (call scene.invalidate() after changing
        # cells in place).
        self.scene.set_classification_cache(cache_classification)
        self._cache_classification = cache_classification
        # Three HIP streams, frames are independent: frame i+1 is classified on
        # classify_ctx.stream while frame i is marched on march_ctx.stream (the march's tail leaves CUs
        # idle that the bandwidth-bound classify pass fills) and frame i-1 is exchanged, folded
        # and gathered on comm_ctx.stream.  Two classified volumes and two send buffers rotate.
        # The march and the compositing streams are high-priority, the classify stream is not.
        # What matters is that the two classes never share a hardware queue: HIP multiplexes
        # its streams onto a few HSA queues in creation order, and when the classify and the
        # march stream landed on the same queue the two kernels ran strictly one after the other
        # (config-4 frame 1.40 ms instead of 1.12; measured with all three at default priority).
        self.march_ctx = runtime.Context(ctx.device_index, priority=stream_priorities[0])
        self.comm_ctx = runtime.Context(ctx.device_index, priority=stream_priorities[1])
        self.classify_ctx = runtime.Context(ctx.device_index, priority=stream_priorities[2])
        # With one rank the march leaves 3 of a CU's 8 workgroup slots (and some LDS) to the
        # classify pass of the next frame, so that the VALU-bound and the HBM-bound kernel really
        # share the CUs: config-4 frame 1.25 -> 1.12 ms.  A rank's share of an N-rank frame is many
        # short runs whose tail, not their throughput, sets the time; there the cap costs
        # (N = 4: 0.355 -> 0.387 ms, N = 8: 0.223 -> 0.237 ms; N = 2 neutral), so it is off.
        if march_workgroups_per_cu is None:
            march_workgroups_per_cu = 0   # uncapped: best for the present march (6 per CU fit)
        self.march_ctx.set_march_occupancy(march_workgroups_per_cu)
        self.march_workgroups_per_cu = march_workgroups_per_cu
        self.compositor = DirectSendCompositor(self.comm_ctx, process_group, stage_through_host,
                                               force_collectives)
        n_local = sum(1 for b in self.all_boxes if b.owner == rank)
        if n_local != len(self.local_boxes):
            raise ValueError("local_boxes does not match the ownership of all_boxes")
        # replicated metadata in C form, built once per scene
        self._box_array = make_box_array(self.all_boxes)
        self._owner_array = make_owner_array(self.all_boxes)
        # coarsest min spacing over all ranks == MPI_Allreduce(MAX) of VolumeRenderer.cpp:1166
        self.reference_sample_distance = runtime.reference_sample_distance(
            self.all_boxes, bounds.min_corner, bounds.max_corner)
        # rank order of the compositing group (VolumeRenderer.cpp:1235-1241); trivial for one rank
        self.visibility = runtime.VisibilityGraph(self.all_boxes, n_ranks) if n_ranks > 1 else None
        # computeTightBounds (:791-848): the MPI min/max over all ranks' boxes == over all_boxes
        self.tight_bounds = runtime.tight_bounds(self.all_boxes, bounds.min_corner,
                                                 bounds.max_corner)
        self._send: List[Optional[torch.Tensor]] = [None, None]   # double-buffered send layout
        self._send_free: List[Optional[torch.cuda.Event]] = [None, None]
        self._classified_free: List[Optional[torch.cuda.Event]] = [None, None]
        # when a list, paint() appends (classify_begin, classify_end, march_begin, march_end)
        # timing events of every frame (bench.py's live kernel durations)
        self.kernel_events: Optional[list] = None
        self._frame = 0
        self._last_plan = None
        # ---- the native frame driver -------------------------------------------------------------
        # The process group is the CONTROL plane only (any backend; bench.py uses gloo): it carries
        # RCCL's 128-byte id from rank 0 to the others and the agreement below.  The data plane is
        # the C++ driver's own RCCL communicator -- the only one in the process.
        if native is None:
            native = not stage_through_host and not force_collectives
            if native and n_ranks > 1 and comm is None:
                import torch.distributed as dist
                native = process_group is not None or dist.is_initialized()
        self.native_error = None
        if native and n_ranks > 1 and comm is None:
            import torch.distributed as dist
            try:
                comm = runtime.Comm.from_process_group(ctx.device_index, process_group)
                if dist.get_backend(process_group) != "nccl":
                    # the caller's host-side control plane (gloo; MPI in the reference's host)
                    # carries the communicator's small agreements -- a new plan's check, the
                    # co-run search's window decisions -- instead of tiny RCCL rounds in band
                    comm.set_control(runtime.control_over_process_group(process_group))
            except Exception as error:   # e.g. RCCL not loadable: say so, keep the frame on the GPU
                # NOT a CPU fallback: the same HIP kernels, with torch.distributed's RCCL
                # collectives (all_to_all_single / gather) instead of the C++ driver's; every rank
                # must take the same path, so the decision is agreed on below
                self.native_error = f"{type(error).__name__}: {error}"
            on_host = dist.get_backend(process_group) != "nccl"
            failed = torch.tensor([1 if self.native_error else 0],
                                  device="cpu" if on_host else ctx.device)
            dist.all_reduce(failed, op=dist.ReduceOp.MAX, group=process_group)
            if int(failed.item()):
                import warnings
                warnings.warn("native RCCL communicator unavailable on some rank "
                              f"({self.native_error}); using the torch.distributed frame loop")
                native, comm = False, None
                if on_host:
                    # the fallback's collectives need an RCCL group of torch's own: created here,
                    # by every rank alike, and only now (normally no second communicator exists)
                    fallback_group = dist.new_group(backend="nccl")
                    self.compositor = DirectSendCompositor(self.comm_ctx, fallback_group, False,
                                                           force_collectives)
        if native:
            merged = []   # replicated metadata, this rank's boxes with their cells
            mine = iter(self.local_boxes)
            for b in self.all_boxes:
                merged.append(dataclasses.replace(next(mine), owner=b.owner)
                              if b.owner == rank else b)
            self.native = runtime.NativeRenderer(ctx.device_index, merged, transform, bounds,
                                                 self.scalar_range, rank, n_ranks, comm, color_map)
            self.native.set_options(-1 if march_workgroups_per_cu is None
                                    else march_workgroups_per_cu, cache_classification)
        # The last frame's host plan (visibility order, layer order, runs, exchange layout, per-box
        # prologue) is kept: a frame with the same camera and parameters re-uses it.  Host work
        # only (~60 us of the ~160 us a frame costs the host at N = 8) -- every frame still
        # classifies, marches, exchanges and folds.
        self._plan_cache = None

    @property
    def last_plan(self):
        """The plan of the last frame rendered (runs, piece, exchange volume)."""
        if self.native is not None:
            return self.native.plan_info()
        return self._last_plan

    @property
    def scalar_range(self):
        """geometry.scalarRange: the normalised scalar interval the colour map spans."""
        return self._scalar_range

    @scalar_range.setter
    def scalar_range(self, value) -> None:
        self._scalar_range = tuple(value)
        if self.native is not None:
            self.native.set_scalar_range(self._scalar_range)

    def invalidate(self) -> None:
        """Cell data was changed in place: cached classifications are stale."""
        self.scene.invalidate()
        if self.native is not None:
            self.native.invalidate()

    # -- planning (host) -------------------------------------------------------------------------
    def make_params(self, p: RenderParameters):
        root = validate_render_parameters(p)
        return make_params(p.width * root, p.height * root, self.scalar_range, p.box_transparency,
                           self.reference_sample_distance, self.bounds, self.color_map), root

    def plan(self, params, camera: CameraParameters,
             group_order: Optional[Sequence[int]] = None) -> FramePlan:
        """Layer order, runs and exchange layout of one frame (what composeLayered derives from
        its allgathers, DirectSendBase.cpp:329-410), from the replicated box metadata."""
        return FramePlan(self.all_boxes, params, camera, self.rank, self.n_ranks, group_order,
                         _box_array=self._box_array, _owner_array=self._owner_array)

    # -- one frame ------------------------------------------------------------------------------
    def paint(self, plan: FramePlan, samples: Optional[torch.Tensor] = None,
              slot: int = 0) -> torch.Tensor:
        """Classify (classify_ctx.stream) + march (march_ctx.stream) of this rank's runs into the sparse
        send buffer `slot`; classified volume `slot` carries the table indices between them."""
        need = max(plan.send_floats, 1)
        if self._send[slot] is None or self._send[slot].numel() < need:
            # the old block may still be read by the exchange / fold of the frame before last on
            # the compositing stream (and written by nothing else): wait for those users on the
            # host before the caching allocator may hand the block to someone else, and allocate
            # with headroom so that an orbit's growing plans rarely get here
            if self._send_free[slot] is not None:
                self._send_free[slot].synchronize()
            if self._classified_free[slot] is not None:
                self._classified_free[slot].synchronize()
            with torch.cuda.stream(self.march_ctx.stream):
                self._send[slot] = torch.empty(need + need // 4, dtype=torch.float32,
                                               device=self.ctx.device)
            self._send[slot].record_stream(self.comm_ctx.stream)
        ctx, cls = self.march_ctx, self.classify_ctx
        timed = self.kernel_events is not None
        if self._classified_free[slot] is not None:
            cls.stream.wait_event(self._classified_free[slot])  # the frame before last read it
        if timed:
            c0 = torch.cuda.Event(enable_timing=True)
            c0.record(cls.stream)
        self.scene.classify_plan(cls, plan, slot)
        classified = torch.cuda.Event(enable_timing=timed)
        classified.record(cls.stream)
        ctx.stream.wait_event(classified)
        if timed:
            m0 = torch.cuda.Event(enable_timing=True)
            m0.record(ctx.stream)
        self.scene.march_plan(ctx, plan, slot, self._send[slot], samples)
        marched = torch.cuda.Event(enable_timing=timed)
        marched.record(ctx.stream)
        self._classified_free[slot] = marched
        if timed:
            self.kernel_events.append((c0, classified, m0, marched))
        return self._send[slot]

    def autotune(self, p: RenderParameters, camera: CameraParameters, frames: int = 5,
                 candidates: Sequence[int] = (0, 5)) -> int:
        """Picks the march occupancy cap (avr_context_set_march_occupancy) for this workload by
        timing a few pipelined frames with each candidate.  Leaving CU slots to the classify
        pass pays when the two kernels take comparable time (config-4: 1.25 -> 1.08 ms) and
        costs when the march dominates (config-5, marched at 8192^2: 35.6 -> 42.3 ms).  Every rank
        of a multi-rank renderer must call it (the frames are real, collective frames).
        Returns the chosen cap."""
        import time
        best, best_time = candidates[0], float("inf")
        for cap in candidates:
            self.march_ctx.set_march_occupancy(cap)
            if self.native is not None:
                self.native.set_options(cap, self._cache_classification)
            for _ in range(2):
                self.render(p, camera)
            self.synchronize()
            t0 = time.perf_counter()
            for _ in range(frames):
                self.render(p, camera)
            self.synchronize()
            elapsed = time.perf_counter() - t0
            if elapsed < best_time:
                best, best_time = cap, elapsed
        self.march_ctx.set_march_occupancy(best)
        if self.native is not None:
            self.native.set_options(best, self._cache_classification)
        self.march_workgroups_per_cu = best
        return best

    def synchronize(self) -> None:
        if self.native is not None:
            self.native.synchronize()
        self.classify_ctx.synchronize()
        self.march_ctx.synchronize()
        self.comm_ctx.synchronize()

    def render(self, p: RenderParameters, camera: CameraParameters,
               samples: Optional[torch.Tensor] = None, want_image: bool = False,
               group_order: Optional[Sequence[int]] = None):
        """One frame.  On rank 0 returns (image, rgb8): rgb8 = the output file's pixel bytes
        [H, W, 3] (rows top-down, SavePPM.cpp:25), image = the gathered (downsampled) depth-sort
        image [H, W, 5] if want_image (or antialiasing > 1), else None.  Other ranks get
        (None, None).  The results are produced on the compositing stream: call synchronize()
        before reading them.  (Ordering a stream of your own after the compositing stream is
        enough for one rank; for ranks of several the native driver delivers a frame's bytes with
        the NEXT frame's round -- NativeRenderer.render / outputs_complete -- and synchronize() is
        a collective that every rank calls after the same frame.)"""
        if self.native is not None:
            validate_render_parameters(p)
            out = self.native.render(p.width, p.height, p.box_transparency, p.antialiasing, camera,
                                     p.use_visibility_graph, p.draw_bounds,
                                     p.write_visibility_graph, group_order, samples, want_image)
            return out    # (last_plan asks the driver when somebody wants to know)
        key = (p.width, p.height, p.box_transparency, p.antialiasing, p.use_visibility_graph,
               tuple(camera.eye), tuple(camera.look_at), tuple(camera.up), camera.fov_y_degrees,
               camera.near_plane, camera.far_plane, tuple(self.scalar_range), id(self.color_map),
               None if group_order is None else tuple(group_order))
        cached = self._plan_cache
        if cached is not None and cached[0] == key and not p.write_visibility_graph:
            _, params, root, plan = cached
        else:
            params, root = self.make_params(p)
            if group_order is None and self.visibility is not None:
                # aspect as VolumeRenderer.cpp:1114 computes it (float division of the image size)
                aspect = float(np.float32(p.width) / np.float32(max(p.height, 1)))
                group_order = self.visibility.order(
                    camera, aspect, p.use_visibility_graph,
                    "visibility_graph_" if (p.write_visibility_graph and self.rank == 0) else None)
            plan = self.plan(params, camera, group_order)
            self._plan_cache = (key, params, root, plan)
        self._last_plan = plan
        ctx, comm = self.march_ctx, self.comm_ctx
        slot = self._frame & 1
        self._frame += 1
        # cell data / earlier torch work on the caller's stream is read by the classify pass only
        # (nothing to order against when that stream is idle)
        if not torch.cuda.current_stream(self.ctx.device).query():
            self.classify_ctx.join()
        if self._send_free[slot] is not None:
            # the frame before last read this send buffer on the other stream
            ctx.stream.wait_event(self._send_free[slot])
        send = self.paint(plan, samples, slot)
        painted = self._classified_free[slot]  # recorded on ctx.stream after the march
        with torch.cuda.stream(comm.stream):
            comm.stream.wait_event(painted)
            # 8-bit conversion is per pixel, so without antialiasing it is done on each rank's
            # piece before the gather (3 bytes per pixel on the wire instead of 20)
            early_rgb8 = (root == 1)
            overlay_piece = early_rgb8 and p.draw_bounds
            bytes_only = early_rgb8 and not overlay_piece and not want_image
            piece, piece_rgb8 = self.compositor.compose(
                plan, send, want_rgb8=early_rgb8 and not overlay_piece, on_ops_stream=True,
                want_piece=not bytes_only)
            if overlay_piece:
                # pixels are independent: each rank overlays its own piece (and converts it)
                piece_rgb8 = comm.bbox_overlay(
                    piece.reshape(-1), *self.tight_bounds, camera, 1, p.width, p.height,
                    plan.piece_begin, plan.piece_end, want_rgb8=True, sync_streams=False)
            released = torch.cuda.Event()
            released.record(comm.stream)
            self._send_free[slot] = released
            image = None
            rgb8 = None
            if early_rgb8:
                flat = self.compositor.gather(plan, piece_rgb8, dst=0)
                if flat is not None:
                    rgb8 = torch.flip(flat.view(p.height, p.width, 3), dims=[0])
                if want_image:
                    full = self.compositor.gather(plan, piece, dst=0)
                    if full is not None:
                        image = full.view(p.height, p.width, 5)
            else:
                full = self.compositor.gather(plan, piece, dst=0)
                if full is not None:
                    image = comm.downsample(full.reshape(-1), p.width, p.height, root)
                    if p.draw_bounds:
                        comm.bbox_overlay(image.reshape(-1), *self.tight_bounds, camera, 1,
                                          p.width, p.height, sync_streams=False)
                    rgb8 = comm.quantize_rgb8(image.reshape(-1), p.width, p.height)
        return image, rgb8


def build_scene_on_device(ctx: runtime.Context, spec: scenes.SceneSpec, rank: int = 0):
    """Materialises this rank's boxes of a synthetic scene in HBM (torch, float64) and returns
    (all_boxes metadata, local_boxes)."""
    all_boxes = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    local = []
    for i in scenes.local_box_indices(spec, rank):
        local.append(scenes.amr_box(spec, i, scenes.box_cells_torch(spec, i, ctx.device)))
    return all_boxes, local
