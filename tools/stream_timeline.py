#!/usr/bin/env python3
"""Every kernel of a rank's frames from a rocprofv3 kernel trace (CSV), RCCL's included: for a few
steady-state frames, what ran on which hardware queue when (us relative to the end of a march),
and per queue the busy time per frame.  usage: stream_timeline.py trace.csv [n_frames]"""
import csv
import re
import sys

n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    m = re.search(r"avr::(?:\(anonymous namespace\)::)?(\w+)", name)
    short = m.group(1) if m else ("rccl:" + name[:40] if "nccl" in name.lower() or "rccl" in name.lower()
                                   else name[:40])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r["Queue_Id"])))
rows.sort()
marches = [r for r in rows if r[2].startswith("render_runs_kernel")]
if len(marches) < 2 * n_show + 4:
    raise SystemExit("too few frames in the trace")
first = len(marches) * 3 // 4
window = marches[first:first + n_show + 1]
t0 = window[0][1]
print(f"{n_show} frames from march #{first} (us relative to that march's end):")
for prev, cur in zip(window, window[1:]):
    print(f"  -- frame period {(cur[1] - prev[1]) / 1e3:7.1f} us")
    for s, e, name, q in rows:
        if prev[1] <= s < cur[1]:
            print(f"    q{q} {name:44s} {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f})")
# per queue: busy time per frame over the second half of the trace
half = marches[len(marches) // 2:]
lo, hi, frames = half[0][0], half[-1][1], len(half) - 1
per_queue = {}
for s, e, name, q in rows:
    if lo <= s < hi:
        per_queue.setdefault(q, {}).setdefault(name, [0, 0])
        per_queue[q][name][0] += e - s
        per_queue[q][name][1] += 1
print(f"second half of the trace: {frames} frames, period {(hi - lo) / 1e3 / frames:.1f} us; per queue, us per frame (launches per frame):")
for q in sorted(per_queue):
    total = sum(v[0] for v in per_queue[q].values())
    parts = ", ".join(f"{n} {v[0] / 1e3 / frames:.1f} ({v[1] / frames:.2f})" for n, v in
                      sorted(per_queue[q].items(), key=lambda kv: -kv[1][0]))
    print(f"  q{q}: {total / 1e3 / frames:6.1f}  {parts}")
