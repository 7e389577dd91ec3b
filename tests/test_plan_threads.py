"""avr_renderer_prepare's premise, under ThreadSanitizer on the CPU: frame plans (create + tighten)
made concurrently on four threads equal the serial ones and no data race is reported
(tests/cxx/plan_threads_test.cpp, built from the library's host sources, no HIP)."""
import os
import subprocess

import pytest

CXX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cxx")


def test_plans_made_concurrently_are_race_free():
    build = subprocess.run(["make", "-C", CXX, "plan_threads_test"], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in (build.stderr or ""):
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    out = subprocess.run([os.path.join(CXX, "plan_threads_test")], capture_output=True, text=True,
                         timeout=300, env=env)
    if "unexpected memory mapping" in (out.stderr or ""):   # ThreadSanitizer against this kernel's ASLR
        pytest.skip("ThreadSanitizer cannot map its shadow memory here")
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    assert out.stdout.startswith("ok"), out.stdout
