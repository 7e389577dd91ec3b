#!/bin/bash
# Builds oracle/_ref/ref_compose and oracle/_ref/ref_blend: the REFERENCE's compositor / image classes (its own translation units, compiled
# where they lie under /root/reference -- nothing is copied) + oracle/ref_compose/driver.cpp.
#   Common/Image.cpp  Common/ImageRGBAFloatColorDepthSort.cpp  Common/ImageSparse.cpp
#   Common/LayeredVolumeImage.cpp  Common/SavePPM.cpp  DirectSend/Base/DirectSendBase.cpp
# g++ directly, no CMake, no AMReX (none of these files includes it).  Needs from the image: MPICH
# (/opt/conda: mpi.h, libmpi.so, mpiexec).  The ONE header the reference generates at configure time
# (CMake/amrVolumeRendererConfig.h.in: a single @VAR@, the application's name in a string that none
# of these files reads) is produced from the reference's own template by CMake's configure_file()
# (cmake -P configure_header.cmake: the same command the reference's build runs, without running
# its build system) into oracle/_ref/include.
# Exit 77: no reference tree / MPI here (the GPU box): skipped.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_ref"
REF="${AVR_REFERENCE:-/root/reference}"
MPI="${AVR_MPI_PREFIX:-/opt/conda}"
[ -f "$REF/DirectSend/Base/DirectSendBase.cpp" ] && [ -f "$MPI/include/mpi.h" ] && [ -f "$MPI/lib/libmpi.so" ] \
  || { echo "skipped: no reference tree / MPICH here"; exit 77; }
mkdir -p "$OUT/include" "$OUT/mpi"
# configure_file() itself, by the image's cmake in script mode, on the reference's own template
cmake -DAMRVOLUMERENDERER_APP_NAME=ref_compose \
      -DTEMPLATE="$REF/CMake/amrVolumeRendererConfig.h.in" -DOUTPUT="$OUT/include/amrVolumeRendererConfig.h" \
      -P "$HERE/configure_header.cmake"
# only MPI's own headers from the prefix (it holds unrelated packages' headers as well)
ln -sf "$MPI"/include/mpi*.h "$OUT/mpi/"
# (x86-64 baseline: no -march=native / -mfma / -ffast-math -- IEEE arithmetic without contraction;
#  static libstdc++: conda's older one must not be picked up through the MPI rpath)
g++ -std=c++20 -O2 -I"$OUT/include" -I"$REF" -I"$OUT/mpi" \
  "$REF/Common/Image.cpp" "$REF/Common/ImageRGBAFloatColorDepthSort.cpp" "$REF/Common/ImageSparse.cpp" \
  "$REF/Common/ImageRGBAFloatColorOnly.cpp" "$REF/Common/ImageRGBAUByteColorOnly.cpp" \
  "$REF/Common/LayeredVolumeImage.cpp" "$REF/Common/SavePPM.cpp" "$REF/DirectSend/Base/DirectSendBase.cpp" \
  "$HERE/driver.cpp" \
  -static-libstdc++ -static-libgcc "$MPI/lib/libmpi.so" -Wl,-rpath,"$MPI/lib" -o "$OUT/ref_compose"
# the reference's image classes alone (blend / regions / ubyte encode-decode): oracle/_ref/ref_blend
g++ -std=c++20 -O2 -I"$OUT/include" -I"$REF" -I"$OUT/mpi" \
  "$REF/Common/Image.cpp" "$REF/Common/ImageRGBAFloatColorDepthSort.cpp" "$REF/Common/ImageRGBAFloatColorOnly.cpp" \
  "$REF/Common/ImageRGBAUByteColorOnly.cpp" "$REF/Common/ImageSparse.cpp" \
  "$HERE/blend_driver.cpp" \
  -static-libstdc++ -static-libgcc "$MPI/lib/libmpi.so" -Wl,-rpath,"$MPI/lib" -o "$OUT/ref_blend"
echo "built $OUT/ref_compose and $OUT/ref_blend"
