#!/bin/bash
# round 6 call 13: deferred mid-conditioning prep: native tests (wavefront / farm / transitions), timeline, bench
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_native_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu > gpurun_out/r06_native_tests13.txt 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r06_native_tests13.txt
timeout 600 python tools/transition_timeline.py > gpurun_out/r06_transition_timeline13.txt 2>&1
echo "rc=$?"; tail -20 gpurun_out/r06_transition_timeline13.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/r06_bench_call13.json 2> gpurun_out/r06_bench_call13.err
echo "bench rc=$?"; python -c "
import json
d=json.loads(open('gpurun_out/r06_bench_call13.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')})"
