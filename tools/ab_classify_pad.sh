#!/bin/bash
# The classify pass's share of the GPU beside the march (avr_renderer_set_classify_share) on the
# default bench: the driver's balance (-1) against fixed LDS reserves.
# usage: tools/ab_classify_pad.sh -1 0 16000 26000 ...
cd "$(dirname "$0")/.."
for share in "$@"; do
  for rep in 1 2; do
  python3 bench.py --no-cpu-baseline --steps 300 --classify-share $share 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('share $share  frame %.4f ms  classify %.4f march %.4f  %s' % (d['ms_per_step'], d['roofline']['classify_ms'], d['roofline']['march_ms'], d['config'].get('corun')))"
  done
done
