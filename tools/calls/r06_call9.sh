#!/bin/bash
# round 6 call 9: the whole GPU suite again (noise-view fix), durations
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r06_gpu_suite_call9.txt 2>&1
echo "suite rc=$?"; tail -30 gpurun_out/r06_gpu_suite_call9.txt
