#!/bin/bash
# round 6 call 8: the whole GPU suite at the current tree (with durations), then a bench line
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r06_gpu_suite_call8.txt 2>&1
echo "suite rc=$?"; tail -25 gpurun_out/r06_gpu_suite_call8.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_bench_call8.json 2> gpurun_out/r06_bench_call8.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_call8.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d.get('roofline'))
print([ (s.get('name') or s.get('what'), s.get('value')) for s in d.get('secondary',[])])
print(d.get('phases_per_transition'))
PY
