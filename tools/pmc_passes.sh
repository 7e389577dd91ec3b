#!/bin/bash
# Collects PMC counters for bench.py in separate rocprofv3 passes (never combined with traces).
# usage: tools/pmc_passes.sh <outdir> <bench args...>
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
mkdir -p "$OUT"
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
 "TA_BUSY_sum TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
 "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_READ_sum"
)
i=0
for P in "${PASSES[@]}"; do
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 $R/bench.py "$@" --no-cpu-baseline > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "render_runs_kernel" not in k: continue
        name = "render_runs<%s>" % ("stats" if "<true" in k else "timed")
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, d in agg.items():
        fh.write(k + "\n")
        for c, v in sorted(d.items()):
            fh.write("  %-36s n=%-3d mean=%.6g\n" % (c, len(v), sum(v) / len(v)))
print(open(out + "/summary.txt").read())
PY
