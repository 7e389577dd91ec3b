#!/bin/bash
# A/B of builds of libavr_hip.so kept under tools/_variants/<name>.so ("tree": the in-tree library):
# the march alone (classification cached), the pipelined frame, optionally other workloads.
# usage: tools/ab_variants.sh name [name ...]     (AB_EXTRA="--config config5 --steps 10" adds a third leg)
cd "$(dirname "$0")/.."
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-10s %-10s frame %.4f ms  classify %.4f  march %.4f  reserve %s' % ('$1', '$2', d['ms_per_step'], r['classify_ms'], r['march_ms'], d['config'].get('corun',{}).get('lds_reserve_bytes')))"; }
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = tree ]; then unset AVR_HIP_LIBRARY; else export AVR_HIP_LIBRARY=$PWD/tools/_variants/$v.so; fi
  python3 bench.py --no-cpu-baseline --no-latency --cache-classification --march-occupancy 0 --steps 200 2>/dev/null | line $v alone
  python3 bench.py --no-cpu-baseline --no-latency --steps 200 --warmup 20 2>/dev/null | line $v pipelined
  if [ -n "$AB_EXTRA" ]; then python3 bench.py --no-cpu-baseline --no-latency $AB_EXTRA 2>/dev/null | line $v extra; fi
done
done
