#!/bin/bash
# usage: burst_trace.sh <outdir> [frames]   -- prints the kernels of the last burst, times relative to its first kernel
set -u
OUT=$1; N=${2:-10}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -- python3 $R/tools/burst_trace.py $N > "$OUT/run.log" 2>&1
tail -1 "$OUT/run.log"
python3 - "$OUT" $N <<'PY'
import csv, glob, sys
out, n = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        short = ("classify" if "classify_kernel" in name else "march" if "render_runs" in name else
                 "fold" if "fold_plan" in name else "flip" if "flip_rows" in name else
                 "upload" if "upload_kernel" in name else None)
        if short:
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short))
rows.sort()
marches = [i for i, r in enumerate(rows) if r[2] == "march"]
first_march = marches[-n]
# the burst starts with the classify before its first march
start = max(i for i in range(first_march) if rows[i][2] == "classify" and
            (i == 0 or rows[i][0] - rows[i - 1][1] > 200000 or True) and i <= first_march)
# walk back to the classify that follows the idle gap
i = first_march
while i > 0 and rows[i][0] - max(r[1] for r in rows[:i]) < 100000:
    i -= 1
t0 = rows[i][0]
for s, e, k in rows[i:]:
    if k in ("classify", "march"):
        print("%-8s %8.3f -> %8.3f  (%.3f ms)" % (k, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
PY
rm -rf "$OUT/raw"
