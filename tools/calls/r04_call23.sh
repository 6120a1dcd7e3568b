#!/bin/bash
# full GPU suite at the round's final kernel code; keeps the parity metrics under their own name
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=10 > $OUT/r04_gpu_suite.txt 2>&1
echo "pytest rc=$?"; tail -n 16 $OUT/r04_gpu_suite.txt
cp $OUT/parity_metrics.json $OUT/r04_parity_metrics_full.json
