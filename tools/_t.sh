line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-30s frame %.4f ms  classify %.4f  march %.4f  %s  settle %s  spec %s' % ('$1', d['ms_per_step'], r['classify_ms'], r['march_ms'], d['config']['corun'], d['config']['untimed_frames']['settle'], d['config']['visibility_speculation']))"; }
B="python3 bench.py --no-cpu-baseline --no-latency"
$B --config config2 --transparency 0.0 --steps 200 --warmup 20 2>/dev/null | line "config2 opaque"
$B --config config2 --transparency 0.0 --steps 200 --warmup 20 --no-speculation 2>/dev/null | line "config2 opaque plain"
$B --config config3 --transparency 0.0 --steps 200 --warmup 20 2>/dev/null | line "config3 opaque"
$B --config config3 --transparency 0.0 --steps 200 --warmup 20 --no-speculation 2>/dev/null | line "config3 opaque plain"
timeout -k 10 300 python -m pytest tests/test_speculative_gpu.py -x -q 2>&1 | tail -2
