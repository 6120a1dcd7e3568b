#!/bin/bash
# round 6 final (second session, after the instruction-diet changes, at 8903191): the whole GPU suite, then tools/final_measure.sh r06 8903191 (kernel stats, PMC passes, bench line, mixing rocprof)
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=10 > gpurun_out/r06_gpu_suite.txt 2>&1
echo "suite rc=$?"; tail -6 gpurun_out/r06_gpu_suite.txt
bash tools/final_measure.sh r06 8903191
