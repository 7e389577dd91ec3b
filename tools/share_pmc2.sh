#!/bin/bash
# PMC counters of the march of one simulated rank (both kernels back to back: the march alone).
# usage: share_pmc2.sh <outdir> <n_ranks> <rank> [ownership]
set -u
OUT=$1; N=$2; RANK=$3; OWN=${4:-level_pairs}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rm -rf "$OUT"; mkdir -p "$OUT"
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"
 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES"
)
i=0
for P in "${PASSES[@]}"; do
  timeout -k 10 300 rocprofv3 --kernel-include-regex "render_runs" --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 $R/tools/rank_share.py --worker $N $RANK --ownership $OWN --overlap 0 --frames 40 --no-rccl > "$OUT/run$i.log" 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_runs_kernel<false" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-24s n=%d mean=%.5g" % (k, len(v), sum(v) / len(v)))
PY
rm -rf "$OUT"/pass*
