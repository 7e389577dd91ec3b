#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: what happens where two pairs of paint kernels meet
(steady state), the gaps on each queue, and the first frames after the drain before the timed
region.  python tools/boundary_report.py kernel_trace.csv [STEPS [WARMUP]]  (bench.py's --steps, --warmup)"""
import csv
import statistics
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    if "avr::" not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].replace("void ", "").replace("avr::(anonymous namespace)::", "")[:24]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r["Queue_Id"])))
rows.sort()
classify = [r for r in rows if r[2].startswith("classify_kernel")]
march = [r for r in rows if r[2].startswith("render_runs_kernel<false")]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 5
at = len(march) - steps                      # the timed frames are the last `steps` marches
pause = march[at][0] - march[at - 1][1]
steady = march[at + steps // 2:]            # the timed region's second half: the pipeline has settled
c_steady = [c for c in classify if steady[0][0] <= c[0] <= steady[-1][0]]
print("settled (the second half of the timed frames):")
print("  march    duration %.1f us, gap to the next march    %.1f us (medians)" % (
    statistics.median((e - s) / 1e3 for s, e, _, _ in steady),
    statistics.median((b[0] - a[1]) / 1e3 for a, b in zip(steady, steady[1:]))))
print("  classify duration %.1f us, gap to the next classify %.1f us" % (
    statistics.median((e - s) / 1e3 for s, e, _, _ in c_steady),
    statistics.median((b[0] - a[1]) / 1e3 for a, b in zip(c_steady, c_steady[1:]))))
print("  period %.1f us" % ((steady[-1][1] - steady[0][1]) / 1e3 / (len(steady) - 1)))
mid = rows.index(steady[len(steady) // 2])
t0 = rows[mid][0]
print("  two frames, us from the start of a march:")
for s, e, n, q in rows[mid: mid + 8]:
    print("    q%d %-24s %9.1f .. %9.1f  (%6.1f)" % (q, n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
print("after the drain before the timed region (pause %.0f us), frame by frame:" % (pause / 1e3))
first = march[at]
ahead = [c for c in classify if c[0] >= first[0] - 2_000_000]
for k, m in enumerate(march[at: at + 14]):
    running = [c for c in ahead if c[0] < m[1] and c[1] > m[0]]
    print("  march %2d: %7.1f us, starts %8.1f us after the first; %d classify passes run beside it" % (
        k + 1, (m[1] - m[0]) / 1e3, (m[0] - first[0]) / 1e3, len(running)))
