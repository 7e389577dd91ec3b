#!/bin/bash
# Kernel trace (rocprofv3 --kernel-trace, RCCL's kernels included) of ONE share process of
# tools/rank_share.py -- the program itself after `--`, no launcher in between -- and its timeline.
# usage: trace_worker.sh <out_dir under gpurun_out> <n_ranks> <rank> [rank_share args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$1; n=$2; rank=$3; shift 3
mkdir -p $out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/trw
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trw -o t -- \
  python3 $R/tools/rank_share.py --ranks $n "$@" --worker $n $rank > $out/worker.json 2> $out/worker.err || exit 1
f=$(find /tmp/trw -name '*kernel_trace.csv' | head -1)
python3 $R/tools/stream_timeline.py $f 3 > $out/timeline.txt || exit 1
grep '^{' $out/worker.json | tail -1 > $out/share.json
cat $out/timeline.txt
