#!/bin/bash
# usage: single_frame_trace.sh <out dir under gpurun_out> <chunks...>   (kernel timeline of single, synchronised frames)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$1; shift
mkdir -p $out
export TMPDIR=/tmp; cd /tmp
for k in "$@"; do
  rm -rf /tmp/sft
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sft -o t -- \
    python3 $R/tools/single_frame_trace.py --chunks $k $SFT_ARGS > $out/run_$k.log 2> $out/run_$k.err || exit 1
  f=$(find /tmp/sft -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/single_frame_trace.py --read $f > $out/timeline_$k.txt || exit 1
  # and without the profiler
  python3 $R/tools/single_frame_trace.py --chunks $k --frames 12 $SFT_ARGS >> $out/run_$k.log 2>> $out/run_$k.err || exit 1
done
tail -n 3 $out/run_*.log
