#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py; prints our kernels' rows.  usage: kernel_stats.sh <outdir> <bench args>
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -- python3 $R/bench.py "$@" --no-cpu-baseline > "$OUT/bench.log" 2>&1
grep '"metric"' "$OUT/bench.log" > "$OUT/bench_line.json"
STATS=$(find "$OUT/raw" -name "*kernel_stats.csv" | head -1)
head -1 "$STATS" > "$OUT/kernel_stats.csv"
grep -E "avr::" "$STATS" >> "$OUT/kernel_stats.csv"
rm -rf "$OUT/raw"
cat "$OUT/kernel_stats.csv" | sed 's/(avr::FrameConsts[^"]*"/(...)"/' | cut -c1-200
