"""VolumePainter with the reference's call shape (Common/VolumePainter.hpp:15-32)."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import runtime
from .images import ImageRGBAFloatColorDepthSort
from .types import AmrBox, CameraParameters, ScalarTransform, VolumeBounds, make_params


class VolumePainter:
    """Renders one AMR box into an ImageRGBAFloatColorDepthSort (every pixel is written)."""

    def __init__(self, ctx: runtime.Context):
        self.ctx = ctx

    def paint(self, box: AmrBox, bounds: VolumeBounds, scalar_transform: ScalarTransform,
              scalar_range: Tuple[float, float], rank: int, num_procs: int,
              box_transparency: float, antialiasing: int, reference_sample_distance: float,
              image: ImageRGBAFloatColorDepthSort, camera: CameraParameters,
              color_map: Optional[Sequence] = None,
              samples: Optional[torch.Tensor] = None) -> None:
        """rank, num_procs and antialiasing are accepted and ignored, as in the reference
        (VolumePainter.cpp:561-562)."""
        if not isinstance(image, ImageRGBAFloatColorDepthSort):
            # VolumePainter.cpp:564-568
            raise RuntimeError("VolumePainter expects ImageRGBAFloatColorDepthSort images.")
        if image.region_begin != 0 or image.number_of_pixels != image.width * image.height:
            raise RuntimeError("VolumePainter paints full-frame images")
        if image.width <= 0 or image.height <= 0:
            return
        params = make_params(image.width, image.height, scalar_range, box_transparency,
                             reference_sample_distance, bounds, color_map)
        self.ctx.paint_box(box, scalar_transform, params, camera, out=image.buffer,
                           samples=samples)
