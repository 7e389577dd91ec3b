#!/bin/bash
# Other workloads through bench.py (one GPU): ms per frame, G samples/s, algorithmic fraction.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/regimes
mkdir -p $out
: > $out/regimes.txt
run() {
  name=$1; shift
  timeout -k 10 600 python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 "$@" > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; return 1; }
  python3 - "$name" $out/$name.json >> $out/regimes.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[1]:34s} {d['ms_per_step']:8.4f} ms/frame  {d['value'] / 1e3:8.1f} G samples/s  "
      f"algorithmic {r['frac']:.3f}  compulsory {r['compulsory_frac']:.3f}  "
      f"LDS reserve {d['config']['corun']['lds_reserve_bytes']}")
PY
  tail -1 $out/regimes.txt
}
run config4_translucent_default || exit 1
run config4_opaque --transparency 0.0 || exit 1
run config4_noise_field --field noise || exit 1
run config4_orbit16 --orbit 16 || exit 1
run config4_fly_through --fly-through || exit 1
run config3_256base_2048 --config config3 || exit 1
run config2_uniform512_1024 --config config2 || exit 1
run config5_1024base_4096_aa4 --config config5 --antialiasing 4 --steps 20 --warmup 3 || exit 1
