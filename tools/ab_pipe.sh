#!/bin/bash
# pipelined default frame for a library variant at several march occupancy caps
# usage: tools/ab_pipe.sh variant cap...
cd "$(dirname "$0")/.."
v=$1; shift
if [ "$v" = tree ]; then unset AVR_HIP_LIBRARY; else export AVR_HIP_LIBRARY=$PWD/build/variants/$v.so; fi
for cap in "$@"; do
  python3 bench.py --no-cpu-baseline --steps 300 --march-occupancy $cap 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v cap $cap pipelined frame %.4f ms  classify %.4f march %.4f' % (d['ms_per_step'], d['roofline']['classify_ms'], d['roofline']['march_ms']))"
done
