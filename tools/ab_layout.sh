#!/bin/bash
# A/B of the layouts the driver searches (bench.py --layout: -1 all, 1 side by side only, 2 paired only)
cd "$(dirname "$0")/.." || exit 1
for cfg in "$@"; do
  for lay in -1 1 2 -1 1; do
    line=$(timeout -k 10 200 python3 bench.py --config $cfg --layout=$lay --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1)
    python3 -c "import sys,json; d=json.loads(sys.argv[1]); c=d['config']['corun']; print('$cfg layout $lay: %.4f ms  %s, reserve %d, %d windows' % (d['ms_per_step'], c['classify'][:22], c['lds_reserve_bytes'], c['timed_windows']))" "$line" || echo "$cfg layout $lay: failed"
  done
done
