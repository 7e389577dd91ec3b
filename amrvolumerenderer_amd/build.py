"""In-tree build of libavr_hip.so (hipcc, gfx950).  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libavr_hip.so")
SOURCES = ["avr_kernels.hip", "avr_scene_stats.hip", "avr_overlay.hip", "avr_device.h", "avr_host.cpp", "avr_visibility.cpp", "avr_plan.cpp", "avr_plan.h", "avr_corun.h", "avr_capi.cpp", "avr_comm.cpp", "avr_renderer.cpp",
           "avr_internal.h", "Makefile"]


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES]
    deps.append(os.path.join(_HERE, "..", "include", "avr_hip.h"))
    return any(os.path.getmtime(d) > built for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source into amrvolumerenderer_amd/libavr_hip.so."""
    if os.environ.get("AVR_HIP_LIBRARY"):
        return os.environ["AVR_HIP_LIBRARY"]   # an explicitly chosen build (A/B tools)
    if force or is_stale():
        cmd = ["make", "-C", CSRC] + (["-B"] if force else [])
        subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    if not os.path.exists(LIB):
        raise RuntimeError("hipcc build did not produce libavr_hip.so")
    return LIB
