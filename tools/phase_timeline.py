#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (CSV) of a running pipeline: per block of frames, the frame
period, the duration of the march and of the classify pass, and where the classify pass starts
relative to the march it runs beside (diagnostics: what differs between two periods the same
schedule is seen to settle at).  python tools/phase_timeline.py kernel_trace.csv [block]"""
import bisect
import csv
import re
import sys

block = int(sys.argv[2]) if len(sys.argv) > 2 else 200
march, classify = [], []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "avr::" not in name:
        continue
    m = re.search(r"(\w+_kernel)", name)
    if not m:
        continue
    row = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]))
    if m.group(1) == "render_runs_kernel":
        march.append(row)
    elif m.group(1) == "classify_kernel":
        classify.append(row)
march.sort()
classify.sort()
starts = [m[0] for m in march]
print("frames   period  march  classify  classify starts after the march's start (us, of the march it begins in; -: in a gap)  gap after march  queues")
for at in range(0, len(march) - block, block):
    ms = march[at:at + block + 1]
    period = (ms[-1][1] - ms[0][1]) / block / 1e3
    dur_m = sum(e - s for s, e, _ in ms[:-1]) / block / 1e3
    gap = sum(max(0, b[0] - a[1]) for a, b in zip(ms[:-1], ms[1:])) / block / 1e3
    lo = bisect.bisect_left(classify, (ms[0][0], 0, 0))
    hi = bisect.bisect_left(classify, (ms[-1][0], 0, 0))
    cs = classify[lo:hi]
    dur_c = sum(e - s for s, e, _ in cs) / max(1, len(cs)) / 1e3
    offs, in_gap = [], 0
    for s, e, _ in cs:
        k = bisect.bisect_right(starts, s) - 1
        if k >= 0 and s < march[k][1]:
            offs.append((s - march[k][0]) / 1e3)
        else:
            in_gap += 1
    off = sum(offs) / len(offs) if offs else float("nan")
    queues = sorted({q for _, _, q in ms} | {q for _, _, q in cs})
    print(f"{at:6d} {period:8.1f} {dur_m:6.1f} {dur_c:9.1f} {off:10.1f} ({in_gap} in a gap) {gap:8.1f}   {queues}")
