#!/bin/bash
# round-4 GPU call 13: full bench line (new CPU baseline, secondary lines incl. two-stage speculation) + the changed tests
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 1200 python bench.py --steps 10 --warmup 3 > $OUT/r04_bench_call13.json 2> $OUT/r04_bench_call13.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$OUT/r04_bench_call13.json").read().strip().splitlines()[-1])
print("value",d["value"],"ms",d["ms_per_step"])
print("phases",d.get("phases_per_transition"))
print("roofline frac",d["roofline"]["frac"],d["roofline"]["achieved"])
for s in d.get("secondary",[]): print("  sec:",{k:(round(v,2) if isinstance(v,float) else v) for k,v in s.items() if k!="config"})
c=d.get("cpu_baseline",{}); print("cpu:",c.get("value"),c.get("cores"),c.get("measured"),c.get("host"),c.get("seconds"),c.get("cfg1_tree",{}).get("value"))
print(c.get("sample","")[:400])
PY
tail -5 $OUT/r04_bench_call13.err
timeout 1200 python -m pytest tests/test_native_gpu.py -x -q -m gpu -k "two_stage or frontier or wavefront or recycled or farm" > $OUT/r04_tests_call13.txt 2>&1
echo "pytest rc=$?"; tail -n 8 $OUT/r04_tests_call13.txt
