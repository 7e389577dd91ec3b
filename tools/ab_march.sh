#!/bin/bash
# A/B timing of builds of libavr_hip.so (build/variants/*.so, or the in-tree library as "tree"):
# the march alone (classification cached, uncapped) and the pipelined frame of the default bench.
# usage: tools/ab_march.sh [variant ...]        e.g. tools/ab_march.sh base tree
cd "$(dirname "$0")/.."
for v in "$@"; do
  if [ "$v" = tree ]; then unset AVR_HIP_LIBRARY; else export AVR_HIP_LIBRARY=$PWD/build/variants/$v.so; fi
  for rep in 1 2; do
  python3 bench.py --no-cpu-baseline --cache-classification --march-occupancy 0 --steps 300 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v march-only  frame %.4f ms  march kernel %.4f ms' % (d['ms_per_step'], d['roofline']['march_ms']))"
  done
  for rep in 1 2; do
  python3 bench.py --no-cpu-baseline --steps 300 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v pipelined   frame %.4f ms  classify %.4f march %.4f share %s' % (d['ms_per_step'], d['roofline']['classify_ms'], d['roofline']['march_ms'], d['config'].get('corun')))"
  done
done
