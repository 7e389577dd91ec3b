#!/bin/bash
# march alone (classification cached) for a library variant at several occupancy caps
cd "$(dirname "$0")/.."
v=$1; shift
if [ "$v" = tree ]; then unset AVR_HIP_LIBRARY; else export AVR_HIP_LIBRARY=$PWD/build/variants/$v.so; fi
for cap in "$@"; do
  python3 bench.py --no-cpu-baseline --cache-classification --steps 200 --march-occupancy $cap 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v cap $cap march-only frame %.4f ms  march %.4f' % (d['ms_per_step'], d['roofline']['march_ms']))"
done
