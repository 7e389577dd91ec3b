#!/bin/bash
# bench.py under different environments, one after the other on ONE box, n calls each.
# usage: ab_env_bench.sh <out file under gpurun_out> <n> "<bench args>" "ENV=.. ENV=.." ...
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; n=$2; args=$3; shift 3
rm -f $R/gpurun_out/$out
for envs in "$@"; do
  echo "## $envs" >> $R/gpurun_out/$out
  env $envs bash $R/tools/repeat_bench.sh $out $n $args || exit 1
done
cat $R/gpurun_out/$out
