#!/bin/bash
# Kernel trace of bench.py (the program itself after `--`): the last frames' timeline of all kernels
# (tools/stream_timeline.py) and the pipeline's phases per block of frames (tools/phase_timeline.py).
# usage: trace_bench.sh <out dir under gpurun_out> [bench args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$1; shift
mkdir -p $out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/trb
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trb -o t -- \
  python3 $R/bench.py "$@" --no-cpu-baseline > $out/bench.log 2> $out/bench.err || exit 1
grep '"metric"' $out/bench.log > $out/bench_line.json
f=$(find /tmp/trb -name '*kernel_trace.csv' | head -1)
python3 $R/tools/stream_timeline.py $f 3 > $out/timeline.txt || exit 1
python3 $R/tools/phase_timeline.py $f 200 > $out/phases.txt || exit 1
python3 -c "
import json
d=json.load(open('$out/bench_line.json')); c=d['config']['corun']
print(d['ms_per_step'], c)"
