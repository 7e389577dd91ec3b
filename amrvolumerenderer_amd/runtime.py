"""Context (one per rank = per GPU) and thin tensor-level wrappers over the C ABI.

PyTorch is plumbing here: it owns device memory (HBM tensors) and the HIP stream; every
computation goes through libavr_hip.so.  Host-only helpers (colour table, sampling constants,
depth hints, layer order, piece ranges) work without a GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi
from .types import AmrBox, CameraParameters, ScalarTransform

INF = float("inf")


# ---------------------------------------------------------------------------------------------
# host-only helpers
# ---------------------------------------------------------------------------------------------

def build_color_table(alpha_scale: float, normalization_factor: float,
                      scalar_range=(0.0, 1.0), color_map=None) -> np.ndarray:
    """buildColorTable (Common/VolumePainter.cpp:442-516) -> [256, 4] float32."""
    from .types import colormap_to_c
    out = np.empty(1024, dtype=np.float32)
    rng = (C.c_float * 2)(float(scalar_range[0]), float(scalar_range[1]))
    arr, n = colormap_to_c(color_map)
    _capi.check(_capi.lib().avr_build_color_table(
        float(alpha_scale), float(normalization_factor), rng,
        C.cast(arr, C.POINTER(_capi.ColormapPoint)) if n else None, n,
        out.ctypes.data_as(C.POINTER(C.c_float))))
    return out.reshape(256, 4)


def box_sampling(box: AmrBox, params: _capi.PaintParams) -> Tuple[float, float, float]:
    """(sampleDistance, normalizationFactor, alphaScale), VolumePainter.cpp:571-613."""
    sd, nf, als = C.c_float(), C.c_float(), C.c_float()
    cbox = box.to_c()
    _capi.check(_capi.lib().avr_box_sampling(C.byref(cbox), C.byref(params), C.byref(sd),
                                             C.byref(nf), C.byref(als)))
    return sd.value, nf.value, als.value


def box_depth_hint(box: AmrBox, camera: CameraParameters) -> float:
    """computeBoxDepthHint (VolumeRenderer/VolumeRenderer.cpp:541-553)."""
    out = C.c_float()
    cbox, ccam = box.to_c(), camera.to_c()
    _capi.check(_capi.lib().avr_box_depth_hint(C.byref(cbox), C.byref(ccam), C.byref(out)))
    return out.value


def reference_sample_distance(boxes: Sequence[AmrBox], bounds_min, bounds_max) -> float:
    """VolumeRenderer/VolumeRenderer.cpp:1138-1190 over the given boxes."""
    arr = (_capi.Box * max(len(boxes), 1))(*[b.to_c() for b in boxes])
    bmin = (C.c_double * 3)(*map(float, bounds_min))
    bmax = (C.c_double * 3)(*map(float, bounds_max))
    out = C.c_float()
    _capi.check(_capi.lib().avr_reference_sample_distance(arr, len(boxes), bmin, bmax,
                                                          C.byref(out)))
    return out.value


class VisibilityGraph:
    """avr_visibility_graph: the rank order of the compositing group for a camera
    (BuildVisibilityOrderedGroup, Common/VisibilityOrdering.cpp:63-632) over the replicated box
    metadata.  Host only."""

    def __init__(self, all_boxes: Sequence[AmrBox], n_ranks: int):
        self.n_ranks = int(n_ranks)
        arr = (_capi.Box * max(len(all_boxes), 1))(*[b.to_c() for b in all_boxes])
        owners = (C.c_int32 * max(len(all_boxes), 1))(*[int(b.owner) for b in all_boxes])
        handle = C.c_void_p()
        _capi.check(_capi.lib().avr_visibility_graph_create(arr, owners, len(all_boxes),
                                                            self.n_ranks, C.byref(handle)))
        self._handle = handle
        self.last_succeeded = True
        self.last_splits = 0

    def close(self) -> None:
        if getattr(self, "_handle", None):
            _capi.lib().avr_visibility_graph_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def order(self, camera: CameraParameters, aspect: float, use_visibility_graph: bool = True,
              dot_prefix: Optional[str] = None) -> List[int]:
        """The ranks in group order.  A failed ordering returns the default order and clears
        last_succeeded (the reference warns once and continues)."""
        out = (C.c_int32 * self.n_ranks)()
        ok, splits = C.c_int(1), C.c_int(0)
        ccam = camera.to_c()
        _capi.check(_capi.lib().avr_visibility_order(
            self._handle, C.byref(ccam), C.c_float(aspect), int(bool(use_visibility_graph)),
            dot_prefix.encode() if dot_prefix else None, out, C.byref(ok), C.byref(splits)))
        self.last_succeeded = bool(ok.value)
        self.last_splits = int(splits.value)
        return list(out)


def tight_bounds(boxes: Sequence[AmrBox], fallback_min, fallback_max):
    """computeTightBounds (VolumeRenderer/VolumeRenderer.cpp:791-848) over replicated metadata."""
    arr = (_capi.Box * max(len(boxes), 1))(*[b.to_c() for b in boxes])
    fmin = (C.c_double * 3)(*map(float, fallback_min))
    fmax = (C.c_double * 3)(*map(float, fallback_max))
    omin, omax = (C.c_double * 3)(), (C.c_double * 3)()
    _capi.check(_capi.lib().avr_tight_bounds(arr, len(boxes), fmin, fmax, omin, omax))
    return tuple(omin), tuple(omax)


def layer_order(hints, owner, local_index) -> Tuple[np.ndarray, np.ndarray]:
    """Global layer order and run ends (DirectSend/Base/DirectSendBase.cpp:363-410)."""
    hints = np.ascontiguousarray(hints, dtype=np.float32)
    owner = np.ascontiguousarray(owner, dtype=np.int32)
    local_index = np.ascontiguousarray(local_index, dtype=np.int32)
    n = int(hints.size)
    order = np.empty(max(n, 1), dtype=np.int32)
    run_end = np.empty(max(n, 1), dtype=np.int32)
    n_runs = C.c_int()
    ip = C.POINTER(C.c_int32)
    _capi.check(_capi.lib().avr_layer_order(
        hints.ctypes.data_as(C.POINTER(C.c_float)), owner.ctypes.data_as(ip),
        local_index.ctypes.data_as(ip), n, order.ctypes.data_as(ip), run_end.ctypes.data_as(ip),
        C.byref(n_runs)))
    return order[:n].copy(), run_end[:n_runs.value].copy()


def scene_transform_from_stats(stats: Sequence[float], finite_count: int, log_scale: bool = False,
                               normalize_to_data_range: bool = True):
    """The scalar-transform part of BuildSceneGeometry (SceneBuilder.cpp:315-443) from reduced
    statistics (min, max, min positive).  Returns (ScalarTransform, processed_range,
    scalar_range); raises RuntimeError where the reference throws std::runtime_error."""
    st = (C.c_double * 3)(*[float(v) for v in stats])
    ctr = _capi.ScalarTransform()
    processed = (C.c_double * 2)()
    pr, sr = (C.c_float * 2)(), (C.c_float * 2)()
    _capi.check(_capi.lib().avr_scene_transform_from_stats(
        st, int(finite_count), int(bool(log_scale)), int(bool(normalize_to_data_range)),
        C.byref(ctr), processed, pr, sr))
    span = processed[1] - processed[0]
    transform = ScalarTransform(
        log_scale_input=bool(ctr.log_scale_input),
        normalize_to_unit_range=bool(ctr.normalize_to_unit_range),
        positive_floor=ctr.positive_floor, processed_min=processed[0], processed_max=processed[1],
        inverse_processed_span=1.0 / span, normalization_min=ctr.normalization_min,
        normalization_max=processed[1],
        inverse_normalization_span=ctr.inverse_normalization_span)
    return transform, (pr[0], pr[1]), (sr[0], sr[1])


def piece_range(image_size: int, piece_index: int, num_pieces: int) -> Tuple[int, int]:
    """getPieceRange (DirectSend/Base/DirectSendBase.cpp:59-74)."""
    b, e = C.c_int64(), C.c_int64()
    _capi.check(_capi.lib().avr_piece_range(int(image_size), int(piece_index), int(num_pieces),
                                            C.byref(b), C.byref(e)))
    return b.value, e.value


# ---------------------------------------------------------------------------------------------
# device context
# ---------------------------------------------------------------------------------------------

class Context:
    """One rendering context per rank (= per GPU).  Owns a HIP stream (a torch stream, so
    torch allocations / collectives can be ordered against the kernels)."""

    def __init__(self, device: Optional[int] = None, priority: int = 0):
        """priority: 0 = default, -1 = high (torch.cuda.Stream's convention): work queued on a
        high-priority stream is dispatched before work of default streams."""
        if not torch.cuda.is_available():
            raise _capi.AvrNoDevice("no HIP device visible to PyTorch; the renderer has no CPU "
                                    "fallback")
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        handle = C.c_void_p()
        _capi.check(_capi.lib().avr_context_create(self.device_index, C.byref(handle)))
        self._handle = handle
        self.stream = torch.cuda.Stream(device=self.device, priority=priority)
        _capi.check(_capi.lib().avr_context_set_stream(self._handle,
                                                       C.c_void_p(self.stream.cuda_stream)))

    def close(self) -> None:
        if getattr(self, "_handle", None):
            _capi.lib().avr_context_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- stream ordering -----------------------------------------------------------------------
    def join(self) -> None:
        """Make the context stream wait for work already queued on torch's current stream."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def publish(self) -> None:
        """Make torch's current stream wait for the context stream."""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def synchronize(self) -> None:
        _capi.check(_capi.lib().avr_context_synchronize(self._handle))

    def set_march_occupancy(self, workgroups_per_cu: int) -> None:
        """avr_context_set_march_occupancy: resident march workgroups per CU (0 = uncapped)."""
        _capi.check(_capi.lib().avr_context_set_march_occupancy(self._handle,
                                                                int(workgroups_per_cu)))

    def set_march_counters(self, counters: Optional[torch.Tensor]) -> None:
        """avr_context_set_march_counters: diagnostics for the parity tests (5 x int64 on the
        device, or None to switch them off)."""
        if counters is not None:
            self._check_tensor(counters, torch.int64, "counters")
            if counters.numel() < 5:
                raise ValueError("counters needs 5 entries")
        self._march_counters = counters  # keep alive while set
        _capi.check(_capi.lib().avr_context_set_march_counters(
            self._handle, C.c_void_p(counters.data_ptr()) if counters is not None else None))

    # -- helpers ----------------------------------------------------------------------------
    def _check_tensor(self, t: torch.Tensor, dtype, what: str) -> None:
        if not isinstance(t, torch.Tensor) or t.device != self.device:
            raise ValueError(f"{what} must be a tensor on {self.device}")
        if t.dtype != dtype or not t.is_contiguous():
            raise ValueError(f"{what} must be a contiguous {dtype} tensor")

    def empty(self, *shape, dtype=torch.float32) -> torch.Tensor:
        return torch.empty(*shape, dtype=dtype, device=self.device)

    # -- painter ----------------------------------------------------------------------------
    def paint_box(self, box: AmrBox, transform: ScalarTransform, params: _capi.PaintParams,
                  camera: CameraParameters, out: Optional[torch.Tensor] = None,
                  samples: Optional[torch.Tensor] = None) -> torch.Tensor:
        """avr_paint_box: one box -> [H, W, 5] depth-sort layer."""
        if out is None:
            out = self.empty(params.height, params.width, 5)
        self._check_tensor(out, torch.float32, "out")
        if out.numel() != params.width * params.height * 5:
            raise ValueError("out has the wrong size")
        if samples is not None:
            self._check_tensor(samples, torch.int64, "samples")
        cbox, ctr, ccam = box.to_c(), transform.to_c(), camera.to_c()
        self.join()
        _capi.check(_capi.lib().avr_paint_box(
            self._handle, C.byref(cbox), C.byref(ctr), C.byref(params), C.byref(ccam),
            C.c_void_p(out.data_ptr()),
            C.c_void_p(samples.data_ptr()) if samples is not None else None))
        self.publish()
        return out

    def create_scene(self, boxes: Sequence[AmrBox], transform: ScalarTransform) -> "Scene":
        return Scene(self, boxes, transform)

    # -- image algebra -------------------------------------------------------------------------
    _BLEND = {"depthsort": ("avr_blend_depthsort_f32x5", torch.float32, 5),
              "rgba_f32": ("avr_blend_rgba_f32x4", torch.float32, 4),
              "rgba_u8": ("avr_blend_rgba_u8x4", torch.int32, 1)}
    _KIND = {"depthsort": 0, "rgba_f32": 1, "rgba_u8": 2}

    def blend(self, kind: str, top: torch.Tensor, bottom: torch.Tensor,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
        name, dtype, vec = self._BLEND[kind]
        self._check_tensor(top, dtype, "top")
        self._check_tensor(bottom, dtype, "bottom")
        if top.numel() != bottom.numel():
            raise ValueError("top and bottom differ in size")
        if out is None:
            out = torch.empty_like(top)
        self._check_tensor(out, dtype, "out")
        self.join()
        _capi.check(getattr(_capi.lib(), name)(
            self._handle, C.c_void_p(top.data_ptr()), C.c_void_p(bottom.data_ptr()),
            C.c_void_p(out.data_ptr()), top.numel() // vec))
        self.publish()
        return out

    def blend_regions(self, kind: str, top: torch.Tensor, tb: int, te: int, bottom: torch.Tensor,
                      bb: int, be: int) -> Tuple[torch.Tensor, int, int]:
        _, dtype, vec = self._BLEND[kind]
        self._check_tensor(top, dtype, "top")
        self._check_tensor(bottom, dtype, "bottom")
        if top.numel() != (te - tb) * vec or bottom.numel() != (be - bb) * vec:
            raise ValueError("buffer sizes do not match the regions")
        ob, oe = min(tb, bb), max(te, be)
        out = self.empty((oe - ob) * vec, dtype=dtype)
        self.join()
        _capi.check(_capi.lib().avr_blend_regions(
            self._handle, self._KIND[kind], C.c_void_p(top.data_ptr()), tb, te,
            C.c_void_p(bottom.data_ptr()), bb, be, C.c_void_p(out.data_ptr())))
        self.publish()
        return out, ob, oe

    def encode_rgba_u8(self, rgba: torch.Tensor) -> torch.Tensor:
        self._check_tensor(rgba, torch.float32, "rgba")
        out = self.empty(rgba.numel() // 4, dtype=torch.int32)
        self.join()
        _capi.check(_capi.lib().avr_encode_rgba_u8(self._handle, C.c_void_p(rgba.data_ptr()),
                                                   C.c_void_p(out.data_ptr()), out.numel()))
        self.publish()
        return out

    def decode_rgba_u8(self, encoded: torch.Tensor) -> torch.Tensor:
        self._check_tensor(encoded, torch.int32, "encoded")
        out = self.empty(encoded.numel(), 4)
        self.join()
        _capi.check(_capi.lib().avr_decode_rgba_u8(self._handle, C.c_void_p(encoded.data_ptr()),
                                                   C.c_void_p(out.data_ptr()), encoded.numel()))
        self.publish()
        return out

    def fold_runs(self, slices: Sequence[torch.Tensor], n_pixels: int,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """avr_fold_runs_depthsort: left fold of run slices (each n_pixels x 5) in order."""
        for s in slices:
            self._check_tensor(s, torch.float32, "slice")
            if s.numel() != n_pixels * 5:
                raise ValueError("slice has the wrong size")
        if out is None:
            out = self.empty(n_pixels, 5)
        self._check_tensor(out, torch.float32, "out")
        ptrs = (C.c_void_p * max(len(slices), 1))(*[s.data_ptr() for s in slices])
        self.join()
        _capi.check(_capi.lib().avr_fold_runs_depthsort(self._handle, ptrs, len(slices),
                                                        C.c_void_p(out.data_ptr()), n_pixels))
        self.publish()
        return out

    def fold_plan(self, plan, recv: torch.Tensor, want_rgb8: bool = False,
                  sync_streams: bool = True, want_piece: bool = True,
                  own_send: Optional[torch.Tensor] = None):
        """avr_fold_plan: receiver-side fold of this rank's piece.  Returns
        (piece [piece_len, 5] or None, rgb8 [piece_len, 3] or None).  sync_streams=False: the
        caller already runs on this context's stream.  own_send (avr_fold_plan_own): the rank's
        send buffer, from which the blocks of its own runs are read instead of from recv."""
        if not (want_piece or want_rgb8):
            raise ValueError("nothing to produce")
        self._check_tensor(recv, torch.float32, "recv")
        if own_send is not None:
            self._check_tensor(own_send, torch.float32, "own_send")
            if own_send.numel() < plan.send_floats:
                raise ValueError("send buffer is too small")
        if recv.numel() < plan.recv_floats:
            raise ValueError("receive buffer is too small")
        n = plan.piece_end - plan.piece_begin
        piece = self.empty(max(n, 0), 5) if want_piece else None
        rgb8 = self.empty(max(n, 0), 3, dtype=torch.uint8) if want_rgb8 else None
        if sync_streams:
            self.join()
        _capi.check(_capi.lib().avr_fold_plan_own(
            self._handle, plan._handle, C.c_void_p(recv.data_ptr()),
            C.c_void_p(own_send.data_ptr()) if own_send is not None else None,
            C.c_void_p(piece.data_ptr()) if piece is not None else None,
            C.c_void_p(rgb8.data_ptr()) if rgb8 is not None else None))
        if sync_streams:
            self.publish()
        return piece, rgb8

    def fold_plan_image(self, plan, recv: torch.Tensor) -> torch.Tensor:
        """avr_fold_plan_image (one rank): the fold with its bytes written as the output file's
        rows, top-down.  Returns rgb8 [height, width, 3]."""
        self._check_tensor(recv, torch.float32, "recv")
        if recv.numel() < plan.recv_floats:
            raise ValueError("receive buffer is too small")
        rgb8 = self.empty(plan.height, plan.width, 3, dtype=torch.uint8)
        self.join()
        _capi.check(_capi.lib().avr_fold_plan_image(self._handle, plan._handle,
                                                    C.c_void_p(recv.data_ptr()), None,
                                                    C.c_void_p(rgb8.data_ptr())))
        self.publish()
        return rgb8

    def downsample(self, src: torch.Tensor, target_w: int, target_h: int, block: int
                   ) -> torch.Tensor:
        self._check_tensor(src, torch.float32, "src")
        if src.numel() != target_w * block * target_h * block * 5:
            raise ValueError("src has the wrong size")
        out = self.empty(target_h, target_w, 5)
        self.join()
        _capi.check(_capi.lib().avr_downsample_depthsort(
            self._handle, C.c_void_p(src.data_ptr()), target_w, target_h, block,
            C.c_void_p(out.data_ptr())))
        self.publish()
        return out

    def bbox_overlay(self, image: torch.Tensor, bounds_min, bounds_max,
                     camera: CameraParameters, sqrt_antialiasing: int, width: int, height: int,
                     pixel_begin: int = 0, pixel_end: Optional[int] = None,
                     want_rgb8: bool = False, sync_streams: bool = True) -> Optional[torch.Tensor]:
        """avr_bbox_overlay: the wireframe of the bounds blended in place over the pixels
        [pixel_begin, pixel_end) held by `image` (renderBoundingBoxLayer,
        VolumeRenderer.cpp:139-335); optionally also returns those pixels as RGB8."""
        self._check_tensor(image, torch.float32, "image")
        pixel_end = width * height if pixel_end is None else pixel_end
        if image.numel() != (pixel_end - pixel_begin) * 5:
            raise ValueError("image does not hold the pixel range")
        bmin = (C.c_double * 3)(*map(float, bounds_min))
        bmax = (C.c_double * 3)(*map(float, bounds_max))
        ccam = camera.to_c()
        rgb8 = self.empty(pixel_end - pixel_begin, 3, dtype=torch.uint8) if want_rgb8 else None
        if sync_streams:
            self.join()
        _capi.check(_capi.lib().avr_bbox_overlay(
            self._handle, bmin, bmax, C.byref(ccam), int(sqrt_antialiasing), width, height,
            pixel_begin, pixel_end, C.c_void_p(image.data_ptr()),
            C.c_void_p(rgb8.data_ptr()) if rgb8 is not None else None))
        if sync_streams:
            self.publish()
        return rgb8

    def quantize_rgb8(self, src: torch.Tensor, w: int, h: int) -> torch.Tensor:
        self._check_tensor(src, torch.float32, "src")
        stride = src.numel() // (w * h)
        out = self.empty(h, w, 3, dtype=torch.uint8)
        self.join()
        _capi.check(_capi.lib().avr_quantize_rgb8(self._handle, C.c_void_p(src.data_ptr()), w, h,
                                                  stride, C.c_void_p(out.data_ptr())))
        self.publish()
        return out


class Scene:
    """The rank's local boxes + scalar transform (avr_scene).  Keeps the cell tensors alive."""

    def scalar_stats(self) -> Tuple[float, float, float, int]:
        """avr_scene_scalar_stats: (min, max, min positive, finite count) of this rank's cells
        (reduceLocalScalarStats, SceneBuilder.cpp:53-97).  Synchronises."""
        stats = (C.c_double * 3)()
        count = C.c_int64()
        self.ctx.join()
        _capi.check(_capi.lib().avr_scene_scalar_stats(self.ctx._handle, self._handle, stats,
                                                       C.byref(count)))
        return stats[0], stats[1], stats[2], int(count.value)

    def histogram(self, transform: ScalarTransform, range_min: float, range_max: float,
                  bin_count: int, counts: Optional[torch.Tensor] = None) -> torch.Tensor:
        """avr_scene_histogram: adds this rank's cells to counts[bin_count] (int64 on device)."""
        if bin_count <= 0:
            raise ValueError("binCount must be positive")
        if counts is None:
            counts = torch.zeros(bin_count, dtype=torch.int64, device=self.ctx.device)
        self.ctx._check_tensor(counts, torch.int64, "counts")
        if counts.numel() != bin_count:
            raise ValueError("counts has the wrong size")
        ctr = transform.to_c()
        self.ctx.join()
        _capi.check(_capi.lib().avr_scene_histogram(
            self.ctx._handle, self._handle, C.byref(ctr), float(range_min), float(range_max),
            int(bin_count), C.c_void_p(counts.data_ptr())))
        self.ctx.publish()
        return counts

    def set_classification_cache(self, enabled: bool) -> None:
        """avr_scene_set_classification_cache: keep classified volumes across frames while the
        boxes, the scalar transform and the scalar range are unchanged (off by default)."""
        _capi.check(_capi.lib().avr_scene_set_classification_cache(self._handle, int(bool(enabled))))

    def invalidate(self) -> None:
        """avr_scene_invalidate: the cells were changed in place."""
        _capi.check(_capi.lib().avr_scene_invalidate(self._handle))

    def classify_plan(self, ctx: "Context", plan, slot: int) -> None:
        """avr_classify_plan on `ctx`'s stream: cells -> table indices into classified volume
        `slot`.  No stream ordering is added here; the caller orders it against the march."""
        _capi.check(_capi.lib().avr_classify_plan(ctx._handle, self._handle, plan._handle,
                                                  int(slot)))

    def march_plan(self, ctx: "Context", plan, slot: int, out: torch.Tensor,
                   samples: Optional[torch.Tensor] = None) -> torch.Tensor:
        """avr_march_plan on `ctx`'s stream: ray march of classified volume `slot` into the
        sparse send buffer."""
        ctx._check_tensor(out, torch.float32, "out")
        if out.numel() < plan.send_floats:
            raise ValueError("send buffer is too small")
        if samples is not None:
            ctx._check_tensor(samples, torch.int64, "samples")
        _capi.check(_capi.lib().avr_march_plan(
            ctx._handle, self._handle, plan._handle, int(slot), C.c_void_p(out.data_ptr()),
            C.c_void_p(samples.data_ptr()) if samples is not None else None))
        return out

    def classify_plan_chunked(self, ctx: "Context", plan, slot: int, n_chunks: int,
                              events=None, first_alone: bool = False) -> None:
        """avr_classify_plan_chunked: the frame's boxes in n_chunks depth-ordered chunks, one
        classify launch each; events (torch.cuda.Event list or None) are recorded behind them."""
        handles = None
        if events is not None:
            handles = (C.c_void_p * n_chunks)(*[C.c_void_p(e.cuda_event) for e in events])
        _capi.check(_capi.lib().avr_classify_plan_chunked(
            ctx._handle, self._handle, plan._handle, int(slot), int(n_chunks), handles,
            int(bool(first_alone))))

    def march_plan_chunked(self, ctx: "Context", plan, slot: int, out: torch.Tensor, n_chunks: int,
                           samples: Optional[torch.Tensor] = None, events=None) -> torch.Tensor:
        """avr_march_plan_chunked: one march launch per chunk, each resuming the run accumulators
        the launch before stored (waits for events[k] before launch k when given)."""
        ctx._check_tensor(out, torch.float32, "out")
        if out.numel() < plan.send_floats:
            raise ValueError("send buffer is too small")
        if samples is not None:
            ctx._check_tensor(samples, torch.int64, "samples")
        handles = None
        if events is not None:
            handles = (C.c_void_p * n_chunks)(*[C.c_void_p(e.cuda_event) for e in events])
        _capi.check(_capi.lib().avr_march_plan_chunked(
            ctx._handle, self._handle, plan._handle, int(slot), C.c_void_p(out.data_ptr()),
            C.c_void_p(samples.data_ptr()) if samples is not None else None, int(n_chunks),
            handles))
        return out

    def render_plan_culled(self, ctx: "Context", plan, slot: int, out: torch.Tensor, n_chunks: int,
                           samples: Optional[torch.Tensor] = None,
                           visibility: Optional[torch.Tensor] = None) -> torch.Tensor:
        """avr_render_plan_culled: the frame in n_chunks depth-ordered chunks, boxes no ray can
        still sample left out of the classify pass.  Returns the visibility flags (uint8,
        [n_chunks, n_local_boxes]: row k = what march k found visible behind chunk k)."""
        ctx._check_tensor(out, torch.float32, "out")
        if out.numel() < plan.send_floats:
            raise ValueError("send buffer is too small")
        if samples is not None:
            ctx._check_tensor(samples, torch.int64, "samples")
        n = max(len(self.boxes), 1)
        if visibility is None:
            visibility = torch.empty((n_chunks, n), dtype=torch.uint8, device=ctx.device)
        _capi.check(_capi.lib().avr_render_plan_culled(
            ctx._handle, self._handle, plan._handle, int(slot), C.c_void_p(out.data_ptr()),
            C.c_void_p(samples.data_ptr()) if samples is not None else None, int(n_chunks),
            C.c_void_p(visibility.data_ptr())))
        return visibility

    def classify_plan_flagged(self, ctx: "Context", plan, slot: int, flags: torch.Tensor,
                              gate: Optional[torch.Tensor] = None) -> None:
        """avr_classify_plan_flagged: the classify pass of the boxes whose flag (uint8, by position
        in the rank's layer order) is set; with a gate (int32[1]) nothing at all unless it is != 0."""
        ctx._check_tensor(flags, torch.uint8, "flags")
        if flags.numel() < len(self.boxes):
            raise ValueError("flags needs one entry per local box")
        if gate is not None:
            ctx._check_tensor(gate, torch.int32, "gate")
        _capi.check(_capi.lib().avr_classify_plan_flagged(
            ctx._handle, self._handle, plan._handle, int(slot), C.c_void_p(flags.data_ptr()),
            C.c_void_p(gate.data_ptr()) if gate is not None else None))

    @staticmethod
    def march_plan_workgroups(plan) -> int:
        """avr_march_plan_workgroups: an upper bound of the march launch's workgroups."""
        count = C.c_int64(0)
        _capi.check(_capi.lib().avr_march_plan_workgroups(plan._handle, C.byref(count)))
        return int(count.value)

    def classify_plan_positions(self, ctx: "Context", plan, slot: int, positions) -> None:
        """avr_classify_plan_positions: the classify pass of the boxes at these (ascending)
        positions of the rank's layer order -- a launch of exactly their tiles."""
        values = [int(v) for v in positions]
        array = (C.c_int32 * max(len(values), 1))(*values)
        _capi.check(_capi.lib().avr_classify_plan_positions(
            ctx._handle, self._handle, plan._handle, int(slot), array, len(values)))

    def march_plan_speculative(self, ctx: "Context", plan, slot: int, out: torch.Tensor,
                               classified: Optional[torch.Tensor] = None,
                               visited: Optional[torch.Tensor] = None,
                               missed: Optional[torch.Tensor] = None,
                               miss_count: Optional[torch.Tensor] = None,
                               gate: Optional[torch.Tensor] = None,
                               dirty_workgroups: Optional[torch.Tensor] = None) -> torch.Tensor:
        """avr_march_plan_speculative: the march that checks `classified` (the flags this frame's
        classify pass was given), records the boxes it samples in `visited` and the ones it needs
        but finds unclassified in `missed` / `miss_count` (uint8 per local box / int32[1])."""
        ctx._check_tensor(out, torch.float32, "out")
        if out.numel() < plan.send_floats:
            raise ValueError("send buffer is too small")
        n = len(self.boxes)
        for name, tensor, dtype in (("classified", classified, torch.uint8), ("visited", visited, torch.uint8),
                                    ("missed", missed, torch.uint8)):
            if tensor is not None:
                if name == "classified" and not tensor.is_cuda:   # (host flags are allowed here)
                    if tensor.dtype != dtype:
                        raise ValueError("classified must be uint8")
                else:
                    ctx._check_tensor(tensor, dtype, name)
                if tensor.numel() < n:
                    raise ValueError(f"{name} needs one entry per local box")
        for name, tensor in (("miss_count", miss_count), ("gate", gate)):
            if tensor is not None:
                ctx._check_tensor(tensor, torch.int32, name)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        host = None
        if classified is not None and not classified.is_cuda:   # (host flags: staged with the launch)
            host, classified = classified.contiguous(), None
        if dirty_workgroups is not None:
            ctx._check_tensor(dirty_workgroups, torch.uint8, "dirty_workgroups")
            if dirty_workgroups.numel() < self.march_plan_workgroups(plan):
                raise ValueError("dirty_workgroups is smaller than avr_march_plan_workgroups says")
        spec = _capi.Speculation(ptr(classified), ptr(visited), ptr(missed), ptr(miss_count), None,
                                 ptr(gate), ptr(host), ptr(dirty_workgroups))
        _capi.check(_capi.lib().avr_march_plan_speculative(
            ctx._handle, self._handle, plan._handle, int(slot), C.c_void_p(out.data_ptr()),
            C.byref(spec)))
        return out

    def render_plan(self, plan, out: Optional[torch.Tensor] = None,
                    samples: Optional[torch.Tensor] = None,
                    sync_streams: bool = True) -> torch.Tensor:
        """avr_render_plan: classify + march of this rank's runs into the sparse send buffer."""
        ctx = self.ctx
        if out is None:
            out = ctx.empty(max(plan.send_floats, 1))
        ctx._check_tensor(out, torch.float32, "out")
        if out.numel() < plan.send_floats:
            raise ValueError("send buffer is too small")
        if samples is not None:
            ctx._check_tensor(samples, torch.int64, "samples")
        if sync_streams:
            ctx.join()
        _capi.check(_capi.lib().avr_render_plan(
            ctx._handle, self._handle, plan._handle, C.c_void_p(out.data_ptr()),
            C.c_void_p(samples.data_ptr()) if samples is not None else None))
        if sync_streams:
            ctx.publish()
        return out

    def __init__(self, ctx: Context, boxes: Sequence[AmrBox], transform: ScalarTransform):
        self.ctx = ctx
        self.boxes = list(boxes)
        self.transform = transform
        for b in self.boxes:
            if b.values is None or b.values.device != ctx.device:
                raise ValueError("scene boxes need cell data on the context's device")
        arr = (_capi.Box * max(len(self.boxes), 1))(*[b.to_c() for b in self.boxes])
        ctr = transform.to_c()
        handle = C.c_void_p()
        _capi.check(_capi.lib().avr_scene_create(ctx._handle, arr, len(self.boxes), C.byref(ctr),
                                                 C.byref(handle)))
        self._handle = handle

    def close(self) -> None:
        if getattr(self, "_handle", None):
            _capi.lib().avr_scene_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_runs(self, params: _capi.PaintParams, camera: CameraParameters,
                    box_order: Sequence[int], run_end: Sequence[int], n_pieces: int = 1,
                    out: Optional[torch.Tensor] = None,
                    samples: Optional[torch.Tensor] = None, sync_streams: bool = True
                    ) -> torch.Tensor:
        """avr_render_runs: fused paint + owner-side run fold; returns the send-layout buffer
        (n_runs * H * W * 5 floats)."""
        ctx = self.ctx
        order = np.ascontiguousarray(box_order, dtype=np.int32)
        ends = np.ascontiguousarray(run_end, dtype=np.int32)
        n_runs = int(ends.size)
        n_px = int(params.width) * int(params.height)
        if out is None:
            out = ctx.empty(max(n_runs, 1) * n_px * 5)
        ctx._check_tensor(out, torch.float32, "out")
        if out.numel() < n_runs * n_px * 5:
            raise ValueError("out is too small")
        if samples is not None:
            ctx._check_tensor(samples, torch.int64, "samples")
        ccam = camera.to_c()
        ip = C.POINTER(C.c_int32)
        if sync_streams:
            ctx.join()
        _capi.check(_capi.lib().avr_render_runs(
            ctx._handle, self._handle, C.byref(params), C.byref(ccam),
            order.ctypes.data_as(ip), int(order.size), ends.ctypes.data_as(ip), n_runs,
            int(n_pieces), C.c_void_p(out.data_ptr()),
            C.c_void_p(samples.data_ptr()) if samples is not None else None))
        if sync_streams:
            ctx.publish()
        return out


class Comm:
    """avr_comm: this rank's communicator over RCCL / xGMI (one process per GPU).  The 128-byte
    unique id is created on rank 0 and handed to the other ranks through `broadcast_bytes`, a
    callable (bytes or None) -> bytes of the caller's own control plane: torch.distributed's
    store / object broadcast in this package, MPI_Bcast in the reference's host."""

    def __init__(self, device_index: int, rank: int, n_ranks: int, broadcast_bytes):
        self.rank, self.n_ranks = int(rank), int(n_ranks)
        ident, failure = None, None
        if self.rank == 0:
            # Whatever happens here, the broadcast below still takes place: the other ranks are
            # waiting in it, and a rank 0 that raised first would leave them there (an empty id
            # tells them that rank 0 failed).
            try:
                buf = C.create_string_buffer(_capi.COMM_ID_BYTES)
                _capi.check(_capi.lib().avr_comm_unique_id(buf))
                ident = buf.raw
            except Exception as error:
                ident, failure = b"", error
        ident = broadcast_bytes(ident)
        if failure is not None:
            raise failure
        if not ident:
            raise RuntimeError("rank 0 could not create the communicator id (RCCL not loadable?)")
        if len(ident) != _capi.COMM_ID_BYTES:
            raise ValueError("the communicator id did not survive the broadcast")
        handle = C.c_void_p()
        _capi.check(_capi.lib().avr_comm_create(int(device_index), ident, self.rank, self.n_ranks,
                                                C.byref(handle)))
        self._handle = handle

    @classmethod
    def solo(cls, rank: int, n_ranks: int) -> "Comm":
        """avr_comm_create_solo: one rank of an N-rank frame played alone (timing studies)."""
        self = cls.__new__(cls)
        self.rank, self.n_ranks = int(rank), int(n_ranks)
        handle = C.c_void_p()
        _capi.check(_capi.lib().avr_comm_create_solo(self.rank, self.n_ranks, C.byref(handle)))
        self._handle = handle
        return self

    @classmethod
    def solo_rccl(cls, device_index: int, rank: int, n_ranks: int, percent: int = 100) -> "Comm":
        """avr_comm_create_solo_rccl: as solo(), the rank's collectives played through a one-rank
        RCCL communicator to itself (timing studies: RCCL's kernels beside the paint kernels)."""
        self = cls.__new__(cls)
        self.rank, self.n_ranks = int(rank), int(n_ranks)
        handle = C.c_void_p()
        _capi.check(_capi.lib().avr_comm_create_solo_rccl(int(device_index), self.rank,
                                                          self.n_ranks, int(percent),
                                                          C.byref(handle)))
        self._handle = handle
        return self

    @classmethod
    def shared(cls, name: str, rank: int, n_ranks: int, capacity_bytes: int = 256 << 20) -> "Comm":
        """avr_comm_create_shared: rank processes sharing ONE GPU meet in a POSIX shared-memory
        segment (collective).  Rehearsal of the multi-process flow, not a performance path."""
        self = cls.__new__(cls)
        self.rank, self.n_ranks = int(rank), int(n_ranks)
        handle = C.c_void_p()
        _capi.check(_capi.lib().avr_comm_create_shared(name.encode(), self.rank, self.n_ranks,
                                                       int(capacity_bytes), C.byref(handle)))
        self._handle = handle
        return self

    @classmethod
    def from_process_group(cls, device_index: int, group=None) -> "Comm":
        """Bootstraps over an initialised torch.distributed group (any backend)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def broadcast(ident):
            box = [ident]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group else 0,
                                       group=group)
            return box[0]

        return cls(device_index, rank, world, broadcast)

    def set_control(self, allgather_bytes) -> None:
        """avr_comm_set_control: the caller's control plane for the small host-side agreements of
        an N-rank frame (a new plan's agreement check, the co-run search's window decisions) --
        `allgather_bytes(mine: bytes) -> list of n_ranks bytes objects`, e.g. over gloo
        (control_over_process_group), MPI_Allgather in the reference's host.  None: back to the
        communicator's own (RCCL flavour: a tiny grouped round in band)."""
        if allgather_bytes is None:
            _capi.check(_capi.lib().avr_comm_set_control(self._handle, None, None))
            self._control = None
            return
        n = self.n_ranks

        def trampoline(_user, mine, out, size):
            try:
                parts = allgather_bytes(C.string_at(mine, size))
                if len(parts) != n or any(len(part) != size for part in parts):
                    return 1
                C.memmove(out, b"".join(parts), n * size)
                return 0
            except Exception:  # noqa: BLE001 -- no exception may cross the C ABI
                import traceback
                traceback.print_exc()
                return 1

        callback = _capi.CONTROL_ALLGATHER_FN(trampoline)
        _capi.check(_capi.lib().avr_comm_set_control(self._handle, callback, None))
        self._control = callback   # kept alive for as long as the communicator may call it

    def control_allgather(self, mine: bytes, ctx: Optional["Context"] = None):
        """avr_comm_control_allgather (collective, host-blocking): every rank's bytes."""
        size = len(mine)
        out = C.create_string_buffer(size * self.n_ranks)
        _capi.check(_capi.lib().avr_comm_control_allgather(
            self._handle, ctx._handle if ctx is not None else None, mine, out, size))
        return [out.raw[i * size:(i + 1) * size] for i in range(self.n_ranks)]

    def control_rounds(self) -> int:
        return int(_capi.lib().avr_comm_control_rounds(self._handle))

    def close(self) -> None:
        if getattr(self, "_handle", None):
            _capi.lib().avr_comm_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def control_over_process_group(group=None):
    """A control plane for Comm.set_control over an initialised torch.distributed group (gloo in
    bench.py): all_gather of a byte tensor on the CPU."""
    import torch.distributed as dist

    def allgather(mine: bytes):
        world = dist.get_world_size(group)
        local = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local, group=group)
        return [bytes(part.numpy().tobytes()) for part in parts]

    return allgather


def set_frame_timeout_ms(milliseconds: int) -> None:
    """avr_set_frame_timeout_ms: deadline of every host wait on device work (0: none, < 0: default
    = AVR_FRAME_TIMEOUT_MS or 30 s)."""
    _capi.check(_capi.lib().avr_set_frame_timeout_ms(int(milliseconds)))


class NativeRenderer:
    """avr_renderer: the C++ frame driver of one rank (three HIP streams, rotating
    classified volumes and send layouts, RCCL exchange and gather).  Python only hands over the
    scene once and, per frame, the camera, the parameters and the output tensors."""

    def __init__(self, device_index: int, all_boxes: Sequence[AmrBox], transform: ScalarTransform,
                 bounds, scalar_range=(0.0, 1.0), rank: int = 0, n_ranks: int = 1,
                 comm: Optional[Comm] = None, color_map=None):
        from .types import make_params
        self.device_index = int(device_index)
        self.device = torch.device("cuda", self.device_index)
        self.rank, self.n_ranks = int(rank), int(n_ranks)
        self._comm = comm
        self._boxes = list(all_boxes)   # keeps the cell tensors alive
        n = len(self._boxes)
        arr = (_capi.Box * max(n, 1))(*[b.to_c() for b in self._boxes])
        owners = (C.c_int32 * max(n, 1))(*[int(b.owner) for b in self._boxes])
        ctr = transform.to_c()
        bmin = (C.c_double * 3)(*map(float, bounds.min_corner))
        bmax = (C.c_double * 3)(*map(float, bounds.max_corner))
        rng = (C.c_float * 2)(float(scalar_range[0]), float(scalar_range[1]))
        # make_params owns the conversion of a colour map to C points
        params = make_params(1, 1, scalar_range, 0.0, 0.0, bounds, color_map)
        handle = C.c_void_p()
        _capi.check(_capi.lib().avr_renderer_create(
            self.device_index, self.rank, self.n_ranks, comm._handle if comm is not None else None,
            arr, owners, n, C.byref(ctr), bmin, bmax, rng, params.colormap, params.colormap_count,
            C.byref(handle)))
        self._handle = handle
        self._keep = (arr, owners, params)
        out = C.c_float()
        _capi.check(_capi.lib().avr_renderer_reference_sample_distance(self._handle, C.byref(out)))
        self.reference_sample_distance = out.value
        # torch views of the driver's streams: outputs are allocated and consumed on stream X
        self.streams = [torch.cuda.ExternalStream(int(_capi.lib().avr_renderer_stream(self._handle, w)),
                                                  device=self.device) for w in range(3)]

    def close(self) -> None:
        helper = getattr(self, "_plan_ahead", None)
        if helper is not None:   # no plan may be in the making when the renderer goes
            self._plan_ahead = None
            try:
                helper.close()
            except Exception:  # noqa: BLE001 -- its error belongs to whoever submitted the job
                pass
        if getattr(self, "_handle", None):
            _capi.lib().avr_renderer_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_options(self, march_workgroups_per_cu: int = -1, cache_classification: bool = False):
        _capi.check(_capi.lib().avr_renderer_set_options(self._handle, int(march_workgroups_per_cu),
                                                         int(bool(cache_classification))))

    def synchronize(self) -> None:
        """avr_renderer_synchronize: everything queued so far has finished.  For ranks of several
        also sends the last frame's RGB8 pieces to rank 0 (collective: every rank calls it after
        the same frame)."""
        _capi.check(_capi.lib().avr_renderer_synchronize(self._handle))
        self._held_outputs = None

    def set_frame_chunks(self, chunks: int = -1) -> None:
        """avr_renderer_set_frame_chunks: -1 / 1 one launch per kernel (default), k > 1 every
        frame classified and marched in k depth-ordered chunks (measured not to pay:
        profiles/r5_latency/)."""
        _capi.check(_capi.lib().avr_renderer_set_frame_chunks(self._handle, int(chunks)))

    def set_occlusion_culling(self, chunks: int = -1) -> None:
        """avr_renderer_set_occlusion_culling: k >= 2 every frame in k depth-ordered chunks whose
        classify launches leave out the boxes no ray can still sample; -1 / 0 never (default:
        exact but not faster, profiles/r5_opaque/)."""
        _capi.check(_capi.lib().avr_renderer_set_occlusion_culling(self._handle, int(chunks)))

    def set_corun_balance(self, mode: int = -1) -> None:
        """avr_renderer_set_corun_balance: -1 / 1 one rank balances its two kernels by their
        durations (a bisection of ~70 frames), 0 always the full search of every candidate."""
        _capi.check(_capi.lib().avr_renderer_set_corun_balance(self._handle, int(mode)))

    def set_visibility_speculation(self, mode: int = -1) -> None:
        """avr_renderer_set_visibility_speculation: -1 / 1 (default) a frame whose plan repeats
        classifies only the boxes the frame two before sampled, checks and, if need be, repairs
        (exact); 0 never."""
        _capi.check(_capi.lib().avr_renderer_set_visibility_speculation(self._handle, int(mode)))

    def debug_set_speculation_threshold(self, sampled_fraction: float) -> None:
        """avr_renderer_debug_set_speculation_threshold (include/avr_hip_debug.h): tests only."""
        _capi.check(_capi.lib().avr_renderer_debug_set_speculation_threshold(
            self._handle, float(sampled_fraction)))

    _SPECULATION_STATES = {-1: "off", 0: "observing", 1: "waiting for an observation",
                           2: "speculating", 3: "not worth it", 4: "suspended after repairs"}

    def speculation_state(self) -> dict:
        """avr_renderer_speculation_state."""
        state = C.c_int(0)
        frames, repaired = C.c_int64(0), C.c_int64(0)
        fraction = C.c_float(0.0)
        _capi.check(_capi.lib().avr_renderer_speculation_state(
            self._handle, C.byref(state), C.byref(frames), C.byref(repaired), C.byref(fraction)))
        return {"state": self._SPECULATION_STATES.get(state.value, str(state.value)),
                "speculative_frames": int(frames.value), "repaired_frames": int(repaired.value),
                "sampled_fraction": None if fraction.value < 0 else round(float(fraction.value), 4)}

    def last_frame_chunks(self) -> int:
        return int(_capi.lib().avr_renderer_last_frame_chunks(self._handle))

    def outputs_complete(self):
        """avr_renderer_outputs_complete: (frames whose outputs are written once stream X has
        passed what is queued now, frames rendered so far)."""
        done, frames = C.c_uint64(), C.c_uint64()
        _capi.check(_capi.lib().avr_renderer_outputs_complete(self._handle, C.byref(done),
                                                              C.byref(frames)))
        return done.value, frames.value

    def set_overlap(self, overlap_classify: int) -> None:
        """avr_renderer_set_overlap (-1 default: measured, 0 back to back, 1 classify beside the
        march, 2 paired: frames alternate between two streams)."""
        _capi.check(_capi.lib().avr_renderer_set_overlap(self._handle, int(overlap_classify)))

    def set_host_backpressure(self, mode: int = -1) -> None:
        """avr_renderer_set_host_backpressure: -1 default (host-side for a rank of several), 0 the
        streams wait for the frame three back, 1 the host does."""
        _capi.check(_capi.lib().avr_renderer_set_host_backpressure(self._handle, int(mode)))

    def set_tighten(self, enabled: bool = True) -> None:
        """avr_renderer_set_tighten: per-row exchange layout of every plan the driver makes."""
        _capi.check(_capi.lib().avr_renderer_set_tighten(self._handle, int(bool(enabled))))

    def set_piece_layout(self, layout: int = _capi.PIECES_ROW_BANDS, band_rows: int = 8) -> None:
        """avr_renderer_set_piece_layout: how the image is dealt to the ranks' pieces (bands of
        rows dealt round-robin by default; _capi.PIECES_CONTIGUOUS = the reference's ranges)."""
        _capi.check(_capi.lib().avr_renderer_set_piece_layout(self._handle, int(layout),
                                                              int(band_rows)))

    def set_classify_share(self, lds_reserve_bytes: int = -1) -> None:
        """avr_renderer_set_classify_share: -1 = measured by the driver (default), >= 0 = fixed
        LDS reserve per classify workgroup while the pass runs beside the march."""
        _capi.check(_capi.lib().avr_renderer_set_classify_share(self._handle,
                                                                int(lds_reserve_bytes)))

    def corun_state(self) -> dict:
        """avr_renderer_corun_state: how the last frame placed the classify pass and the march."""
        overlap, reserve, settled, windows = C.c_int(0), C.c_int(0), C.c_int(0), C.c_long(0)
        _capi.check(_capi.lib().avr_renderer_corun_state(self._handle, C.byref(overlap),
                                                         C.byref(reserve), C.byref(settled),
                                                         C.byref(windows)))
        layout = {0: "before the march", 1: "beside the march",
                  2: "before its march, the frames alternating between two streams"}
        return {"classify": layout.get(overlap.value, str(overlap.value)),
                "lds_reserve_bytes": reserve.value, "settled": bool(settled.value),
                "timed_windows": windows.value}

    def set_deferred_gather(self, mode: int = -1) -> None:
        """avr_renderer_set_deferred_gather: -1 default (ranks of several: a frame's RGB8 pieces
        travel to rank 0 with the NEXT frame's grouped round; synchronize() sends the last
        frame's), 0 every frame gathers at once, 1 on."""
        _capi.check(_capi.lib().avr_renderer_set_deferred_gather(self._handle, int(mode)))

    def set_plan_check(self, enabled: bool = True) -> None:
        """avr_renderer_set_plan_check: whether a new plan of a rank of several is agreed on over
        the control plane before its first frame is queued (default on)."""
        _capi.check(_capi.lib().avr_renderer_set_plan_check(self._handle, int(bool(enabled))))

    def set_corun_coordination(self, mode: int = -1) -> None:
        """avr_renderer_set_corun_coordination: -1 default (ranks of several search as one
        system), 0 every rank on its own (round 3's behaviour), 1 on."""
        _capi.check(_capi.lib().avr_renderer_set_corun_coordination(self._handle, int(mode)))

    def set_corun_history(self, frames: int) -> None:
        """Record the co-run candidate of each of the next `frames` frames (diagnostics)."""
        _capi.check(_capi.lib().avr_renderer_set_corun_history(self._handle, int(frames)))

    def corun_history(self):
        """The recorded candidates: -1 back to back, k side by side with reserve k * 2 KiB,
        29 + k paired."""
        n = C.c_int(0)
        _capi.check(_capi.lib().avr_renderer_corun_history(self._handle, None, 0, C.byref(n)))
        out = (C.c_int16 * max(n.value, 1))()
        _capi.check(_capi.lib().avr_renderer_corun_history(self._handle, out, n.value, C.byref(n)))
        return list(out[:n.value])

    def failure(self) -> Optional[str]:
        """avr_renderer_failure: what did not finish within the deadline, or None."""
        text = _capi.lib().avr_renderer_failure(self._handle)
        return text.decode("utf-8", "replace") if text else None

    def set_scalar_range(self, scalar_range) -> None:
        rng = (C.c_float * 2)(float(scalar_range[0]), float(scalar_range[1]))
        _capi.check(_capi.lib().avr_renderer_set_scalar_range(self._handle, rng))

    def invalidate(self) -> None:
        _capi.check(_capi.lib().avr_renderer_invalidate(self._handle))

    def plan_info(self) -> "_capi.FramePlanInfo":
        info = _capi.FramePlanInfo()
        _capi.check(_capi.lib().avr_renderer_plan_info(self._handle, C.byref(info)))
        return info

    def host_profile(self, reset: bool = True):
        """({section: microseconds per frame}, frames) spent inside avr_renderer_render."""
        sec = (C.c_double * 6)()
        n = C.c_long()
        _capi.check(_capi.lib().avr_renderer_host_profile(self._handle, sec, C.byref(n), int(reset)))
        names = ("plan", "classify", "march", "exchange", "fold", "gather+tail")
        frames = max(n.value, 1)
        return {k: 1e6 * v / frames for k, v in zip(names, sec)}, n.value

    def set_timing(self, enabled: bool) -> None:
        _capi.check(_capi.lib().avr_renderer_set_timing(self._handle, int(bool(enabled))))

    def timings(self):
        """(classify_ms, march_ms, busy_ms, frames) averaged per frame since set_timing(True)."""
        c, m, b, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        _capi.check(_capi.lib().avr_renderer_timings(self._handle, C.byref(c), C.byref(m),
                                                     C.byref(b), C.byref(n)))
        return c.value, m.value, b.value, n.value

    def prepare(self, width: int, height: int, box_transparency: float, antialiasing: int,
                camera: CameraParameters, use_visibility_graph: bool = True,
                draw_bounds: bool = True, write_visibility_graph: bool = False,
                group_order: Optional[Sequence[int]] = None) -> None:
        """avr_renderer_prepare: makes and keeps the frame plan render() will want for these
        arguments -- host geometry only.  May run on another thread while render() queues a frame
        (ctypes releases the GIL for the call): see PlanAhead."""
        rp = _capi.RenderParams(int(width), int(height), float(box_transparency),
                                int(antialiasing), int(bool(use_visibility_graph)),
                                int(bool(draw_bounds)), int(bool(write_visibility_graph)))
        ccam = camera.to_c()
        group = None
        if group_order is not None:
            group = (C.c_int32 * self.n_ranks)(*[int(g) for g in group_order])
        _capi.check(_capi.lib().avr_renderer_prepare(self._handle, C.byref(rp), C.byref(ccam),
                                                     group))

    def render(self, width: int, height: int, box_transparency: float, antialiasing: int,
               camera: CameraParameters, use_visibility_graph: bool = True,
               draw_bounds: bool = True, write_visibility_graph: bool = False,
               group_order: Optional[Sequence[int]] = None,
               samples: Optional[torch.Tensor] = None, want_image: bool = False):
        """One frame (asynchronous).  Rank 0 returns (image [H, W, 5] or None, rgb8 [H, W, 3] with
        rows top-down); other ranks (None, None).  One rank: the tensors are complete on stream X
        (self.streams[2]) -- synchronize(), or order your stream after it.  Ranks of several: the
        RGB8 bytes of a frame without want_image travel to rank 0 inside the NEXT frame's round
        (avr_renderer_set_deferred_gather, the default), so after ordering a stream behind stream X
        only the frames outputs_complete() counts are written -- the last frame's tensor is still
        uninitialised memory until the next render() has passed stream X or synchronize() (a
        collective then: every rank calls it after the same frame) has returned."""
        # (a camera and parameters that repeat are converted once: per frame the Python layer costs
        # the host what matters beside a rank's 0.17 ms frame)
        key = (width, height, box_transparency, antialiasing, use_visibility_graph, draw_bounds,
               write_visibility_graph, tuple(map(float, camera.eye)),
               tuple(map(float, camera.look_at)), tuple(map(float, camera.up)),
               float(camera.fov_y_degrees), float(camera.near_plane), float(camera.far_plane))
        if getattr(self, "_converted_key", None) != key:
            self._converted = (
                _capi.RenderParams(int(width), int(height), float(box_transparency),
                                   int(antialiasing), int(bool(use_visibility_graph)),
                                   int(bool(draw_bounds)), int(bool(write_visibility_graph))),
                camera.to_c())
            self._converted_key = key
        rp, ccam = self._converted
        group = None
        if group_order is not None:
            group = (C.c_int32 * self.n_ranks)(*[int(g) for g in group_order])
        rgb8 = image = None
        caller = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.streams[2]):
            if self.rank == 0:
                rgb8 = torch.empty((height, width, 3), dtype=torch.uint8, device=self.device)
                if want_image:
                    image = torch.empty((height, width, 5), dtype=torch.float32, device=self.device)
        if samples is not None and (samples.dtype != torch.int64 or samples.device != self.device):
            raise ValueError("samples must be an int64 tensor on the renderer's device")
        # Cell data / the samples counter may have been written on the caller's stream: the driver
        # orders this frame's classify pass (and with it the march) after that stream.  torch's
        # default stream has handle 0, which the C ABI reads as "nothing to wait for": it is named
        # by AVR_DEFAULT_STREAM (-1) instead -- the driver's streams are non-blocking and never
        # order themselves after the null stream implicitly.
        wait = None if caller.query() else C.c_void_p(caller.cuda_stream or _capi.DEFAULT_STREAM)
        _capi.check(_capi.lib().avr_renderer_render(
            self._handle, C.byref(rp), C.byref(ccam), group, wait,
            C.c_void_p(samples.data_ptr()) if samples is not None else None, int(bool(want_image)),
            C.c_void_p(rgb8.data_ptr()) if rgb8 is not None else None,
            C.c_void_p(image.data_ptr()) if image is not None else None))
        # (ranks of several: this frame's bytes are written while the NEXT frame is queued, or by
        # synchronize(): the tensor must outlive the caller's interest in it until then)
        self._held_outputs = (image, rgb8)
        return image, rgb8


class PlanAhead:
    """Plans the NEXT frame on a helper thread while the caller queues this one: for a camera
    that never repeats the host's geometry of a new camera (visibility order, frame plan, tightened
    exchange layout: ~0.1 ms for 176 boxes at N = 8) otherwise adds to every frame's queueing, and
    a rank of eight has 0.15 ms per frame.  submit() the arguments of the render() call that will
    follow the next one; render() finds the plan (or, if the helper is not through, waits for it
    and goes on).  Results never depend on it."""

    def __init__(self, native: "NativeRenderer"):
        import queue
        import threading
        self._native = native
        native._plan_ahead = self   # NativeRenderer.close() closes the helper first
        self._jobs = queue.SimpleQueue()
        self._error = None
        self._thread = threading.Thread(target=self._run, name="avr-plan-ahead", daemon=True)
        self._thread.start()

    def _run(self) -> None:
        while True:
            job = self._jobs.get()
            if job is None:
                return
            try:
                self._native.prepare(*job[0], **job[1])
            except Exception as error:  # noqa: BLE001 -- handed to the submitting thread
                self._error = error

    def submit(self, *args, **kwargs) -> None:
        if self._error is not None:
            error, self._error = self._error, None
            raise error
        self._jobs.put((args, kwargs))

    def close(self) -> None:
        if self._thread is not None:
            self._jobs.put(None)
            self._thread.join()
            self._thread = None
            if getattr(self._native, "_plan_ahead", None) is self:
                self._native._plan_ahead = None
        if self._error is not None:
            error, self._error = self._error, None
            raise error
