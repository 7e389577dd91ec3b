#!/bin/bash
# PMC counters of the march kernel for one simulated rank.
# usage: share_pmc.sh <outdir> --worker <n_ranks> <rank> [rank_share args]   (the worker itself: under
# rocprofv3 the process must not start children)
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-include-regex "render_runs" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d "$OUT/raw" -- python3 $R/tools/rank_share.py "$@" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/raw/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_runs_kernel<false" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-24s n=%d mean=%.5g" % (k, len(v), sum(v) / len(v)))
PY
rm -rf "$OUT/raw"
