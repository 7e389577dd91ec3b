#!/bin/bash
# The round's standard GPU check: parity suite, then the driver-style bench and the fly-through.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/check
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -3 $out/pytest.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/bench20.json 2> $out/bench20.err || { tail -20 $out/bench20.err; exit 1; }
cat $out/bench20.json
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --fly-through --no-cpu-baseline > $out/fly20.json 2> $out/fly20.err || { tail -20 $out/fly20.err; exit 1; }
cat $out/fly20.json
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --fly-through --no-cpu-baseline > $out/fly200.json 2> $out/fly200.err || { tail -20 $out/fly200.err; exit 1; }
cat $out/fly200.json
