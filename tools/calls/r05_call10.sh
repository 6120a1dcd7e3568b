#!/bin/bash
# round-5 GPU call 10: the oracle-heavy new tests with the host threads bounded (suite duration)
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py tests/test_native_gpu.py -q -m gpu -k "frontier64 or ddim" --durations=5 > $OUT/r05_tests_call10.txt 2>&1
echo "rc=$?"; tail -n 12 $OUT/r05_tests_call10.txt
