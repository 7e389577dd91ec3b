cd $GRAFT_REPO_ROOT
for args in "--transparency 0.0" "--field noise" "--orbit 16" "--config config3" "--config config2"; do
  echo "== $args"
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 --warmup 10 $args | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['frames_per_s'], d['value'], d['roofline']['frac'], d['roofline']['classify_ms'], d['roofline']['march_ms'], d['config'].get('corun'))" || exit 1
done
echo "== config5"
timeout -k 10 500 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --config config5 --antialiasing 4 > gpurun_out/config5.log 2>&1; tail -3 gpurun_out/config5.log
