#!/bin/bash
# Round 4's evidence at HEAD: parity suite, bench lines, kernel stats, PMC passes (with the digest
# of the kernel sources), the other regimes, the N-rank share model.  usage: r4_final.sh [kernels|regimes|shares]
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
R=$PWD
export GRAFT_REPO_ROOT=$R
out=$R/gpurun_out/r4_final
mkdir -p $out
part=${1:-all}
if [ "$part" = kernels ] || [ "$part" = all ]; then
  timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
  tail -2 $out/pytest.txt
  for i in 1 2 3; do
    timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $out/bench_20_$i.json 2> $out/bench_20.err || { tail -20 $out/bench_20.err; exit 1; }
  done
  timeout -k 10 400 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/bench_200.json 2> $out/bench_200.err || exit 1
  python3 -c "
import json
for f in ('bench_20_1','bench_20_2','bench_20_3','bench_200'):
    d=json.load(open('$out/'+f+'.json')); print(f, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('traffic_frac'), d['roofline']['compulsory_frac'], d['config']['untimed_frames'])"
  timeout -k 10 600 bash tools/kernel_stats.sh $out/stats --steps 200 --warmup 20 > $out/kernel_stats.log 2>&1 || { tail -20 $out/kernel_stats.log; exit 1; }
  tail -12 $out/kernel_stats.log
  cd $R
  timeout -k 10 1500 bash tools/pmc_passes.sh $out/pmc --steps 20 --warmup 3 > $out/pmc.log 2>&1 || { tail -20 $out/pmc.log; exit 1; }
  cd $R
  head -3 $out/pmc/summary.txt; grep -E "FETCH_SIZE|WRITE_SIZE" $out/pmc/summary.txt
fi
if [ "$part" = regimes ] || [ "$part" = all ]; then
  bash tools/regimes.sh || exit 1
  cp gpurun_out/regimes/regimes.txt $out/regimes.txt
  rm -f $out/repeat_config2.txt
  bash tools/repeat_bench.sh r4_final/repeat_config2.txt 5 --config config2 --steps 200 || exit 1
fi
if [ "$part" = shares ] || [ "$part" = all ]; then
  timeout -k 10 1000 python3 tools/rank_share.py --ownership level_pairs > $out/rank_share_level_pairs.txt 2> $out/rank_share.err || { tail -20 $out/rank_share.err; exit 1; }
  tail -5 $out/rank_share_level_pairs.txt
  timeout -k 10 900 python3 tools/rank_share.py --ownership level_pairs --ranks 8 --through-rccl -1 > $out/rank_share_n8_one_link.txt 2>> $out/rank_share.err || exit 1
  tail -3 $out/rank_share_n8_one_link.txt
fi
