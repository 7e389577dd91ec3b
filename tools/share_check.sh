#!/bin/bash
# fly-through bench lines + the N-rank share model under two ownership policies (RCCL resident)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/shares
mkdir -p $out
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --fly-through --no-cpu-baseline > $out/fly20.json 2> $out/fly20.err || { tail -20 $out/fly20.err; exit 1; }
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --fly-through --no-cpu-baseline > $out/fly200.json 2> $out/fly200.err || { tail -20 $out/fly200.err; exit 1; }
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/static200.json 2> $out/static200.err || { tail -20 $out/static200.err; exit 1; }
cat $out/fly20.json $out/fly200.json $out/static200.json
for policy in level_pairs morton; do
  timeout -k 10 1500 python3 tools/rank_share.py --ownership $policy > $out/rank_share_$policy.txt 2> $out/rank_share_$policy.err || { tail -20 $out/rank_share_$policy.err; exit 1; }
  tail -6 $out/rank_share_$policy.txt
done
