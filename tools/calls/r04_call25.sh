#!/bin/bash
# host-side frontier change (no kernel change): native tests that exercise it, then the bench line again (secondary lines move)
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py tests/test_native_gpu.py -q -m gpu -k "cfg or frontier or two_stage or skew or farm or recycled or wavefront or specul or elid" > $OUT/r04_tests_call25a.txt 2>&1
echo "native tests rc=$?"; tail -n 4 $OUT/r04_tests_call25a.txt
timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "cfg2 or multilevel" > $OUT/r04_tests_call25b.txt 2>&1
echo "fullsize tests rc=$?"; tail -n 3 $OUT/r04_tests_call25b.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/r04_bench_1gpu_b.json 2> $OUT/r04_bench_1gpu_b.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_1gpu_b.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for s in d['secondary']: print({k:(round(v,2) if isinstance(v,float) else v) for k,v in s.items() if k in ('name','value','ms_per_step','frontier_rounds','speculation_hit_rate','sequential')})
PY
