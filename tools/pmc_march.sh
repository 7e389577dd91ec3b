#!/bin/bash
# PMC counters of the march alone (classification cached, uncapped), separate passes.
# usage: tools/pmc_march.sh <outdir> [bench args]
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
mkdir -p "$OUT"
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD"
 "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_IFETCH"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_SMEM"
 "GRBM_GUI_ACTIVE FETCH_SIZE"
)
i=0
for P in "${PASSES[@]}"; do
  timeout -k 10 300 rocprofv3 --kernel-include-regex "render_runs" --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 $R/bench.py --no-cpu-baseline --cache-classification --march-occupancy 0 --steps 20 --warmup 3 "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "render_runs_kernel" in k and "<true" not in k:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for c, v in sorted(agg.items()):
        fh.write("  %-36s n=%-3d mean=%.6g\n" % (c, len(v), sum(v) / len(v)))
print(open(out + "/summary.txt").read())
import shutil
for d in glob.glob(out + "/pass*"):
    if not d.endswith(".log"): shutil.rmtree(d)
PY
