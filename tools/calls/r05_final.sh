#!/bin/bash
# round-5 final GPU call: the whole GPU suite at HEAD, then tools/final_measure.sh r05 <commit> (kernel statistics + PMC traffic passes +
# the bench line + rocprof-reported mixing figures)
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=10 > $OUT/r05_gpu_suite.txt 2>&1
echo "pytest rc=$?"; tail -n 18 $OUT/r05_gpu_suite.txt
bash tools/final_measure.sh r05 $1 > $OUT/r05_final_measure.log 2>&1
echo "final_measure rc=$?"; tail -n 30 $OUT/r05_final_measure.log
