#!/bin/bash
# usage: rocprof_script.sh <python script> [args]: per-kernel average durations of our kernels
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/rp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -- python3 "$@" > /tmp/rp.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/rp/**/*kernel_stats.csv", recursive=True)[0]
for row in csv.DictReader(open(f)):
    if "avr::" in row["Name"]:
        print(row["Name"][33:90], row["Calls"], round(float(row["AverageNs"]) / 1e3, 1), "us")
PY
