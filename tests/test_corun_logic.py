"""CPU test of the frame driver's co-run search (csrc/avr_corun.h, no GPU call in it): driven by
synthetic frame periods in tests/cxx/corun_test.cpp -- the one-rank curve (flat, dip at 24-26
KiB, cliff), a rank of eight (back to back beats the first stretch, the dip lies far out), a
process where side by side never wins, fixed choices, bursts too short for a window, occasional
drains, long frames, drift of the held candidate."""
import os
import subprocess

CXX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cxx")


def test_corun_search_logic():
    subprocess.run(["make", "-C", CXX, "corun_test"], check=True, stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CXX, "corun_test")], capture_output=True, text=True,
                         timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
