"""MI355X-native (gfx950 / CDNA4) hot path of the AMR volume renderer.

A from-scratch implementation of the one data-parallel path of BenWibking/amrVolumeRenderer:
VolumePainter's ray march over AMR bricks, the depth-sorted / float / ubyte over-blend and the
DirectSend sort-last compositor, as hand-written HIP kernels behind a C ABI
(include/avr_hip.h, built to amrvolumerenderer_amd/libavr_hip.so).  See DESIGN.md.
"""
from . import _capi  # noqa: F401
from .types import (AmrBox, CameraParameters, ColorMapControlPoint, ScalarTransform,  # noqa: F401
                    VolumeBounds, make_params)

# the reference python package's four entry points (python/amrVolumeRenderer/__init__.py)
from .api import compute_histogram, finalize_runtime, initialize_runtime, render  # noqa: F401

__all__ = ["render", "initialize_runtime", "finalize_runtime", "compute_histogram",
           "AmrBox", "CameraParameters", "ColorMapControlPoint", "ScalarTransform",
           "VolumeBounds", "make_params"]
