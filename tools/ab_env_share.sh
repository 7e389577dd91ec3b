#!/bin/bash
# A rank's share of the N-rank frame (tools/rank_share.py) under different environments, one after
# the other on ONE box.  usage: ab_env_share.sh <out file under gpurun_out> "<rank_share args>" "ENV=.. ENV=.." ...
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$1; shift
args=$1; shift
: > $out
for envs in "$@"; do
  echo "== $envs" >> $out
  env $envs timeout -k 10 300 python3 $R/tools/rank_share.py $args 2>&1 | grep -E "N=8 rank|classify |slowest" >> $out || exit 1
done
cat $out
