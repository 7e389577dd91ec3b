#!/bin/bash
# rocprofv3 kernel stats of one simulated rank (tools/rank_share.py).
# usage: share_stats.sh <outdir> --worker <n_ranks> <rank> [rank_share args]
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -- python3 $R/tools/rank_share.py "$@" > "$OUT/run.log" 2>&1
STATS=$(find "$OUT/raw" -name "*kernel_stats.csv" | head -1)
head -1 "$STATS" > "$OUT/kernel_stats.csv"
grep -E "avr::" "$STATS" >> "$OUT/kernel_stats.csv"
rm -rf "$OUT/raw"
tail -2 "$OUT/run.log"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    print(row["Name"][:60], row["Calls"], round(float(row["AverageNs"]) / 1e3, 1), "us")
PY
