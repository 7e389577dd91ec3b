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
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE"
 "TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
 "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_READ_sum"
 "TCC_HIT_sum TCC_MISS_sum"
)
i=0
for P in "${PASSES[@]}"; do
  timeout -k 10 300 rocprofv3 --kernel-include-regex "render_runs|classify" --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 $R/bench.py "$@" --no-cpu-baseline > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "render_runs_kernel" in k:
            if "<true" in k: continue
            name = "render_runs_kernel"
        elif "classify_kernel" in k:
            name = "classify_kernel"
        elif "render_runs_repair_kernel" in k or "classify_gated_kernel" in k:
            # (a speculative frame's two repair launches: idle unless its march missed a box)
            name = "render_runs_repair_kernel" if "render_runs" in k else "classify_gated_kernel"
        else:
            continue
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
import hashlib, os
digest = hashlib.sha256()
sources = ("avr_kernels.hip", "avr_device.h", "avr_renderer.cpp")   # bench.py: PMC_SOURCES
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for name in sources:
    digest.update(open(os.path.join(root, "amrvolumerenderer_amd", "csrc", name), "rb").read())
with open(out + "/summary.txt", "w") as fh:
    # what the counters were taken on: bench.py quotes them only for a tree with the same digest
    fh.write("# sources sha256: %s  (%s)\n" % (digest.hexdigest(), " ".join(sources)))
    for k, d in agg.items():
        fh.write(k + "\n")
        for c, v in sorted(d.items()):
            fh.write("  %-36s n=%-3d mean=%.6g\n" % (c, len(v), sum(v) / len(v)))
print(open(out + "/summary.txt").read())
import shutil
for d in glob.glob(out + "/pass*"):
    shutil.rmtree(d) if not d.endswith(".log") else None
PY
