#!/bin/bash
# round 6 call 26: bench.py default path after the --torch-baseline addition (short run, no secondary / cpu baseline)
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > gpurun_out/r06_bench_short_call26.json 2> gpurun_out/r06_bench_short_call26.err
echo "rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_bench_short_call26.json"))
print({k: d[k] for k in ("metric","value","unit","n_gpus","steps","ms_per_step")})
print(d["roofline"])
print(d.get("phases_per_transition"))
PY
