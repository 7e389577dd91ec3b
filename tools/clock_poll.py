#!/usr/bin/env python3
"""Polls the GPU's shader clock and power from sysfs / rocm-smi beside a running bench (diagnostics:
is a change of the frame period a change of the clock?).  python tools/clock_poll.py OUT SECONDS"""
import glob
import subprocess
import sys
import time

out, seconds = sys.argv[1], float(sys.argv[2])
files = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
power = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average"))
freq = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
with open(out, "w") as f:
    f.write(f"# files {files} {power} {freq}\n")
    if not files and not freq:
        try:
            f.write(subprocess.run(["rocm-smi", "-c", "-P"], capture_output=True, text=True, timeout=20).stdout)
        except Exception as e:  # noqa: BLE001
            f.write(f"# rocm-smi: {e}\n")
    end = time.monotonic() + seconds
    while time.monotonic() < end:
        row = [f"{time.monotonic():.4f}"]
        for p in freq:
            try:
                row.append(str(int(open(p).read()) // 1000000))
            except OSError:
                row.append("-")
        for p in files:
            try:
                cur = [l for l in open(p).read().splitlines() if l.endswith("*")]
                row.append(cur[0].split()[1] if cur else "-")
            except OSError:
                row.append("-")
        for p in power:
            try:
                row.append(str(int(open(p).read()) // 1000000))
            except OSError:
                row.append("-")
        f.write(" ".join(row) + "\n")
        time.sleep(0.02)
