#!/bin/bash
# The two bounded one-GPU experiments of round 3: classify order, wave footprint.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3_experiments
mkdir -p $out
timeout -k 10 600 python3 tools/ab_classify_order.py > $out/classify_order.txt 2> $out/classify_order.err || { tail $out/classify_order.err; exit 1; }
cat $out/classify_order.txt
timeout -k 10 900 bash tools/ab_march.sh tree wave1 wave2 tree > $out/wave_shape.txt 2> $out/wave_shape.err || { tail $out/wave_shape.err; exit 1; }
cat $out/wave_shape.txt
