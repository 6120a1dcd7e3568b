#!/bin/bash
# round-5 GPU call 5: kernel + native test files after the DDIM rounding fix (call 4 stopped at it)
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_native_gpu.py -q -m gpu --durations=6 > $OUT/r05_tests_call5.txt 2>&1
echo "pytest rc=$?"; tail -n 16 $OUT/r05_tests_call5.txt
