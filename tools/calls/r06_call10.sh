#!/bin/bash
# round 6 call 10: B = 17 step as one program vs two half-batch programs on two streams; bench with the prior-speculation line
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 600 python tools/dual_stream_ab.py > gpurun_out/r06_dual_stream_ab.txt 2>&1
echo "dual rc=$?"; cat gpurun_out/r06_dual_stream_ab.txt | tail -5
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r06_bench_call10.json 2> gpurun_out/r06_bench_call10.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_call10.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')})
for s in d.get('secondary',[]): print(round(s.get('value',0),2), s.get('frontier_rounds'), (s.get('name') or '')[:110])
PY
