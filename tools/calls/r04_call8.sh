#!/bin/bash
# round-4 GPU call 8: end-to-end A/B of the ping-pong routing (same box, same process order), then the kernel tests
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
LB_GEMM_PP_AUTO=0 timeout 600 python bench.py $ARGS > $OUT/r04_ab_pp0.json 2> $OUT/r04_ab_pp0.err; echo "pp0 rc=$?"
LB_GEMM_PP_AUTO=1 timeout 600 python bench.py $ARGS > $OUT/r04_ab_pp1.json 2> $OUT/r04_ab_pp1.err; echo "pp1 rc=$?"
LB_GEMM_PP_AUTO=0 timeout 600 python bench.py $ARGS > $OUT/r04_ab_pp0b.json 2> $OUT/r04_ab_pp0b.err; echo "pp0b rc=$?"
for f in pp0 pp1 pp0b; do python - <<PY
import json
t=open("$OUT/r04_ab_$f.json").read().strip().splitlines()[-1]
d=json.loads(t)
print("$f", d["value"], d["ms_per_step"], d.get("phases_per_transition"), d.get("roofline"))
PY
done
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_native_gpu.py -x -q -m gpu -k "gemm or unet_tiny or transition_tree" > $OUT/r04_tests_call8.txt 2>&1
echo "pytest rc=$?"; tail -n 5 $OUT/r04_tests_call8.txt
