#!/usr/bin/env python3
"""Timeline of a rank's frames from a rocprofv3 kernel trace (CSV): per frame, when each kernel
ran relative to the march before it, and the gaps in which neither paint kernel was running."""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
    if not m or "avr::" not in r["Kernel_Name"]:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), int(r["Queue_Id"])))
rows.sort()
marches = [r for r in rows if r[2] == "render_runs_kernel"]
skip = len(marches) // 2
print("last frames (us relative to the end of the previous march):")
for prev, cur in zip(marches[skip:skip + 4], marches[skip + 1:skip + 5]):
    t0 = prev[1]
    print(f"  frame period {(cur[1] - prev[1]) / 1e3:7.1f} us")
    for s, e, name, q in rows:
        if s >= prev[0] and s <= cur[1]:
            print(f"    q{q} {name:22s} {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f})")
# idle time between paint kernels over the second half
paint = sorted((s, e) for s, e, n, q in rows if n in ("render_runs_kernel", "classify_kernel") and s >= marches[skip][0])
busy, cursor = 0, paint[0][0]
for s, e in paint:
    if e > cursor:
        busy += e - max(s, cursor)
        cursor = e
span = paint[-1][1] - paint[0][0]
n = len(marches) - skip
print(f"paint kernels busy {busy / 1e3 / n:.1f} us per frame of {span / 1e3 / n:.1f} us: idle {100 * (1 - busy / span):.1f} %")
