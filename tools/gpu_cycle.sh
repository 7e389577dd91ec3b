#!/bin/bash
# quick GPU cycle: parity tests of the march + timing of march alone and of the pipelined frame
cd /root/repo
python -m pytest tests/test_painter_gpu.py tests/test_index_modes_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > gpurun_out/cycle_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/cycle_tests.log
tail -3 gpurun_out/cycle_tests.log
python3 bench.py --no-cpu-baseline --cache-classification --march-occupancy 0 --steps 200 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('march-only frame ms', d['ms_per_step'], 'march_ms', d['roofline']['march_ms'])"
python3 bench.py --no-cpu-baseline --steps 200 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipelined frame ms', d['ms_per_step'], 'classify', d['roofline']['classify_ms'], 'march', d['roofline']['march_ms'], 'cap', d['config']['march_workgroups_per_cu'])"
