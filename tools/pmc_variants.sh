#!/bin/bash
# usage: tools/pmc_variants.sh "<counters>" variant...   (march alone, classification cached)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
COUNTERS=$1; shift
export TMPDIR=/tmp
cd /tmp
for v in "$@"; do
  if [ "$v" = tree ]; then unset AVR_HIP_LIBRARY; else export AVR_HIP_LIBRARY=$R/build/variants/$v.so; fi
  rm -rf /tmp/pmcv
  timeout -k 10 300 rocprofv3 --kernel-include-regex "render_runs" --pmc $COUNTERS --output-format csv -d /tmp/pmcv -- python3 $R/bench.py --no-cpu-baseline --cache-classification --march-occupancy 0 --steps 20 --warmup 3 > /tmp/pmcv.log 2>&1 || echo "$v failed"
  python3 - "$v" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
dur = []
for f in glob.glob("/tmp/pmcv/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_runs_kernel" in row["Kernel_Name"] and "<true" not in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
            if "Start_Timestamp" in row and row.get("End_Timestamp"):
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
line = sys.argv[1] + ": " + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(agg.items()))
if dur:
    line += "  duration_us=%.1f" % (sum(dur) / len(dur))
print(line)
PY
done
