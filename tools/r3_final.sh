#!/bin/bash
# Round 3's evidence at HEAD: parity suite, kernel stats, PMC passes, bench lines, N-rank share model.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
R=$PWD
out=$R/gpurun_out/r3_final
mkdir -p $out
part=${1:-all}   # "kernels" (tests, bench lines, kernel stats, PMC), "shares" (the N-rank model) or all
if [ "$part" != shares ]; then
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -2 $out/pytest.txt
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $out/bench_20.json 2> $out/bench_20.err || { tail -20 $out/bench_20.err; exit 1; }
cat $out/bench_20.json
GRAFT_REPO_ROOT=$R timeout -k 10 600 bash tools/kernel_stats.sh $out/stats --steps 200 --warmup 20 > $out/kernel_stats.log 2>&1 || { tail -20 $out/kernel_stats.log; exit 1; }
tail -12 $out/kernel_stats.log
cd $R
GRAFT_REPO_ROOT=$R timeout -k 10 1500 bash tools/pmc_passes.sh $out/pmc --steps 20 --warmup 3 > $out/pmc.log 2>&1 || { tail -20 $out/pmc.log; exit 1; }
cd $R
tail -60 $out/pmc.log
fi
[ "$part" = kernels ] && exit 0
for policy in level_pairs morton; do
  timeout -k 10 900 python3 tools/rank_share.py --ownership $policy > $out/rank_share_$policy.txt 2> $out/rank_share_$policy.err || { tail -20 $out/rank_share_$policy.err; exit 1; }
  tail -5 $out/rank_share_$policy.txt
done
timeout -k 10 900 python3 tools/rank_share.py --ownership level_pairs --fly-through --ranks 1 2 8 > $out/rank_share_fly_through.txt 2> $out/rank_share_fly.err || { tail -20 $out/rank_share_fly.err; exit 1; }
tail -4 $out/rank_share_fly_through.txt
