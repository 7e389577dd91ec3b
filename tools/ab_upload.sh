#!/bin/bash
# A/B: descriptor copies on a stream of their own (AVR_UPLOAD_STREAM=1 default class, =2 high
# priority class) against copies on the consumers' streams (=0, the default), for a rank of eight
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/ab_upload
mkdir -p $out
export TMPDIR=/tmp
for u in 0 1 2; do
  echo "== AVR_UPLOAD_STREAM=$u, rank 0 of 8 (level_pairs)"
  AVR_UPLOAD_STREAM=$u timeout -k 10 200 python3 tools/rccl_resident.py comm_used --n-ranks 8 --rank 0 --ownership level_pairs --reserve 43008 2>/dev/null | grep -v back_to_back || exit 1
  AVR_UPLOAD_STREAM=$u timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/tl$u -o t -- python3 tools/rccl_resident.py comm_used --n-ranks 8 --rank 0 --ownership level_pairs --reserve 43008 --trace-frames 80 > $out/tl$u.log 2>&1 || exit 1
  python3 tools/share_timeline.py $out/tl$u/t_kernel_trace.csv | tail -14
done
