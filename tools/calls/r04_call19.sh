#!/bin/bash
# full GPU suite at the (near-)final commit
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > $OUT/r04_gpu_suite.txt 2>&1
echo "pytest rc=$?"; tail -n 30 $OUT/r04_gpu_suite.txt
cp parity_metrics.json $OUT/r04_parity_metrics.json 2>/dev/null
