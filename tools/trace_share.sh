#!/bin/bash
# Kernel timeline (rocprofv3 --kernel-trace) of the last pipelined frames of one simulated rank.
# usage: trace_share.sh <rank_share args>
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $R/tools/rank_share.py "$@" > /tmp/tr.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'avr::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-28:]
t0 = min(int(r['Start_Timestamp']) for r in rows)
for r in rows:
    n = r['Kernel_Name']
    name = ('classify' if 'classify' in n else 'march' if 'render_runs' in n else 'fold' if 'fold' in n
            else 'upload' if 'upload' in n else n[:20])
    print(f"{name:9s} q={r['Queue_Id']} start={(int(r['Start_Timestamp'])-t0)/1e3:9.1f} end={(int(r['End_Timestamp'])-t0)/1e3:9.1f} dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}")
PY
