#!/bin/bash
# tools/rccl_resident.py in every mode, each in a process of its own, plus kernel traces (queue ids)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/rccl
mkdir -p $out
export TMPDIR=/tmp
: > $out/modes.txt
for mode in none comm_first comm_last comm_used; do
  timeout -k 10 240 python3 tools/rccl_resident.py $mode >> $out/modes.txt 2> $out/$mode.err || exit 1
done
GPU_MAX_HW_QUEUES=8 timeout -k 10 240 python3 tools/rccl_resident.py comm_first >> $out/modes.txt 2> $out/q8.err || exit 1
GPU_MAX_HW_QUEUES=8 timeout -k 10 240 python3 tools/rccl_resident.py none >> $out/modes.txt 2> $out/q8n.err || exit 1
GPU_MAX_HW_QUEUES=2 timeout -k 10 240 python3 tools/rccl_resident.py none >> $out/modes.txt 2> $out/q2n.err || exit 1
for mode in none comm_first; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$mode -o t -- \
    python3 tools/rccl_resident.py $mode --trace-frames 40 > $out/trace_$mode.log 2>&1 || exit 1
done
cat $out/modes.txt
