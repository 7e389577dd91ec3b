#!/bin/bash
# The same bench command N times, each in a process of its own, on ONE box: ms per frame and what
# the co-run search held.  usage: repeat_bench.sh <out file under gpurun_out> <n> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$1; n=$2; shift 2
echo "== bench.py $* (x$n)" >> $out
for i in $(seq $n); do
  timeout -k 10 300 python3 $R/bench.py "$@" --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
c=d['config']['corun']; r=d['roofline']
print(f\"  {d['ms_per_step']:.4f} ms  classify {r['classify_ms']:.3f} march {r['march_ms']:.3f} union {r['kernel_ms']:.3f}  {c['classify']}, reserve {c['lds_reserve_bytes']}, settled {c['settled']}, windows {c['timed_windows']}, settle frames {d['config']['untimed_frames']['settle']}\")" >> $out || exit 1
done
