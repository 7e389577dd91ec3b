#!/bin/bash
# quick: GPU tests + per-kernel average durations for config4
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
bash tools/kernel_stats.sh $R/gpurun_out/ks --config ${1:-config4} --steps 20 --warmup 3 >/dev/null
python3 - <<'PY'
import csv, json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for row in csv.DictReader(open(R + "/gpurun_out/ks/kernel_stats.csv")):
    print(row["Name"][33:75], row["Calls"], round(float(row["AverageNs"]) / 1e3, 1), "us")
line = json.loads(open(R + "/gpurun_out/ks/bench_line.json").read())
print("ms_per_step", line["ms_per_step"], "frac", line["roofline"]["frac"])
PY
