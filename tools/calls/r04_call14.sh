#!/bin/bash
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -40
ARGS="--steps 40 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline"
LB_GEMM_PP_AUTO=0 python tools/probes/power_trace.py r04_pp0 -- python bench.py $ARGS > $OUT/r04_power_pp0.log 2>&1; tail -2 $OUT/r04_power_pp0.log | cut -c1-1500
LB_GEMM_PP_AUTO=1 python tools/probes/power_trace.py r04_pp1 -- python bench.py $ARGS > $OUT/r04_power_pp1.log 2>&1; tail -2 $OUT/r04_power_pp1.log | cut -c1-1500
LB_GEMM_PP_AUTO=0 python tools/probes/power_trace.py r04_pp0b -- python bench.py $ARGS > $OUT/r04_power_pp0b.log 2>&1; tail -2 $OUT/r04_power_pp0b.log | cut -c1-1500
timeout 600 python -m pytest tests/test_native_gpu.py -x -q -m gpu -k "two_stage or recycled" > $OUT/r04_tests_call14.txt 2>&1
echo "pytest rc=$?"; tail -n 4 $OUT/r04_tests_call14.txt
