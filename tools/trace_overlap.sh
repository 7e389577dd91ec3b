#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/tr; 
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $R/bench.py --config config4 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/tr.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'avr::' in r['Kernel_Name']]
t0 = min(int(r['Start_Timestamp']) for r in rows)
for r in rows[-16:]:
    n = r['Kernel_Name']
    name = 'classify' if 'classify' in n else 'march' if 'render_runs' in n else 'fold' if 'fold' in n else n[:20]
    print(f"{name:9s} q={r['Queue_Id']} start={(int(r['Start_Timestamp'])-t0)/1e3:10.1f} end={(int(r['End_Timestamp'])-t0)/1e3:10.1f} dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}")
PY
