#!/usr/bin/env python3
"""Per-frame counter values of the windows tools/corun_counters logged inside a steady phase.
  python3 tools/corun_counters/summarise.py phase.json counters.log"""
import collections, json, sys
phase = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
period = (phase["end"] - phase["begin"]) / phase["frames"]
sums = collections.defaultdict(lambda: [0.0, 0.0, 0])
for line in open(sys.argv[2]):
    parts = line.split()
    if len(parts) != 4:
        continue
    begin, length, name, value = float(parts[0]), float(parts[1]), parts[2], float(parts[3])
    if begin < phase["begin"] + 0.2 or begin + length > phase["end"] - 0.1:
        continue
    entry = sums[name]
    entry[0] += value
    entry[1] += length
    entry[2] += 1
print(f"# {phase['mode']}: {phase['ms_per_frame']:.4f} ms per frame, {phase['frames']} frames, {phase['corun']}")
for name in sorted(sums):
    value, seconds, windows = sums[name]
    print(f"{name:28s} per frame {value / seconds * period:14.6g}   per second {value / seconds:14.6g}   ({windows} windows)")
