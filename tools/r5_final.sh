#!/bin/bash
# Round 5's evidence at HEAD.  usage: r5_final.sh kernels | pmc | regimes | latency
#   kernels  GPU parity suite, bench lines (20 steps x 3, 200 steps), kernel stats
#   pmc      PMC counters (separate --pmc passes) for every workload the sheet quotes, one summary
#            per workload, headed by the digest of the kernel sources: profiles/r5_final/pmc_<key>.txt
#   regimes  the other workloads with all three roofline fractions (needs the pmc summaries in the tree)
#   latency  single synchronised frame / first 100 frames / frames to settle per configuration
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
R=$PWD
export GRAFT_REPO_ROOT=$R
out=$R/gpurun_out/r5_final
mkdir -p $out
part=${1:-kernels}
if [ "$part" = kernels ]; then
  timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1 || { tail -40 $out/pytest_gpu.txt; exit 1; }
  tail -2 $out/pytest_gpu.txt
  for i in 1 2 3; do
    timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $out/bench_20_$i.json 2> $out/bench_20.err || { tail -20 $out/bench_20.err; exit 1; }
  done
  timeout -k 10 400 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/bench_200.json 2> $out/bench_200.err || exit 1
  python3 -c "
import json
for f in ('bench_20_1','bench_20_2','bench_20_3','bench_200'):
    d=json.load(open('$out/'+f+'.json')); print(f, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('traffic_frac'), d['roofline']['compulsory_frac'], d['config']['untimed_frames'], d.get('latency'))"
  timeout -k 10 600 bash tools/kernel_stats.sh $out/stats --steps 200 --warmup 20 --no-latency > $out/kernel_stats.log 2>&1 || { tail -20 $out/kernel_stats.log; exit 1; }
  tail -12 $out/kernel_stats.log
fi
if [ "$part" = pmc ]; then
  run_pmc() {
    key=$1; shift
    cd $R
    timeout -k 10 1000 bash tools/pmc_passes.sh $out/pmc_$key --no-latency "$@" > $out/pmc_$key.log 2>&1 || { tail -20 $out/pmc_$key.log; return 1; }
    cd $R
    { head -1 $out/pmc_$key/summary.txt; echo "# workload: $key  (bench.py --no-latency $*)"; tail -n +2 $out/pmc_$key/summary.txt; } > $out/pmc_$key.txt
    grep -E "^[a-z]|FETCH_SIZE|WRITE_SIZE" $out/pmc_$key.txt
  }
  run_pmc config4_translucent --steps 20 --warmup 3 || exit 1
  run_pmc config4_opaque --transparency 0.0 --steps 20 --warmup 3 || exit 1
  run_pmc config2_translucent --config config2 --steps 20 --warmup 3 || exit 1
  run_pmc config3_translucent --config config3 --steps 20 --warmup 3 || exit 1
  run_pmc config5_translucent --config config5 --antialiasing 4 --steps 6 --warmup 2 || exit 1
fi
if [ "$part" = regimes ]; then
  : > $out/regimes.txt
  run() {
    name=$1; shift
    timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-latency --steps 100 --warmup 10 "$@" > $out/regime_$name.json 2> $out/regime_$name.err || { tail -5 $out/regime_$name.err; return 1; }
    python3 - "$name" $out/regime_$name.json >> $out/regimes.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
traffic = f"{r['traffic_frac']:.3f}" if r.get("traffic_frac") else "  -  "
print(f"{sys.argv[1]:30s} {d['ms_per_step']:8.4f} ms/frame {d['value'] / 1e3:8.1f} G samples/s  "
      f"algorithmic {r['frac']:.3f}  traffic {traffic}  compulsory {r['compulsory_frac']:.3f}  "
      f"reserve {d['config']['corun']['lds_reserve_bytes']:6d}  settle {d['config']['untimed_frames']['settle']}")
PY
    tail -1 $out/regimes.txt
  }
  run config4_translucent || exit 1
  run config4_opaque --transparency 0.0 || exit 1
  run config4_opaque_every_box --transparency 0.0 --no-speculation || exit 1
  run config4_opaque_fly_through --transparency 0.0 --fly-through || exit 1
  run config4_opaque_orbit16 --transparency 0.0 --orbit 16 || exit 1
  run config4_noise_field --field noise || exit 1
  run config4_orbit16 --orbit 16 || exit 1
  run config4_fly_through --fly-through || exit 1
  run config3_translucent --config config3 || exit 1
  run config2_translucent --config config2 || exit 1
  run config2_opaque --config config2 --transparency 0.0 || exit 1
  run config5_translucent --config config5 --antialiasing 4 --steps 20 --warmup 3 || exit 1
  run config5_opaque --config config5 --antialiasing 4 --transparency 0.0 --steps 20 --warmup 3 || exit 1
fi
if [ "$part" = latency ]; then
  : > $out/latency.txt
  for cfg in "config4 0.97 1" "config4 0 1" "config2 0.97 1" "config3 0.97 1" "config5 0.97 4"; do set -- $cfg
    timeout -k 10 600 python3 tools/latency_probe.py --config $1 --transparency $2 --antialiasing $3 >> $out/latency.txt 2>> $out/latency.err || { tail -5 $out/latency.err; exit 1; }
    timeout -k 10 600 python3 tools/latency_probe.py --config $1 --transparency $2 --antialiasing $3 --repeat-camera --burst 8 >> $out/latency.txt 2>> $out/latency.err || exit 1
  done
  python3 -c "
import json
for line in open('$out/latency.txt'):
    d=json.loads(line); s=d['single_frame_ms']; f=d['first_frames']
    print(d['config'], d['transparency'], s['camera'][:12], 'single', s['median'], s['min'], s['max'], '| first', f['frames'], 'mean', f['mean_period_ms'], 'settle', f['frames_to_settle'], f['corun']['lds_reserve_bytes'])"
fi
