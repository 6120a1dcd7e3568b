#!/bin/bash
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 600 python tools/transition_timeline.py > gpurun_out/r06_transition_timeline.txt 2>&1
echo "rc=$?"; tail -45 gpurun_out/r06_transition_timeline.txt
