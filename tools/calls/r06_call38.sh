#!/bin/bash
# round 6 call 38: cProfile of the host side of 10 cfg-2 transitions
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 900 python tools/host_profile.py > gpurun_out/r06_host_profile.txt 2>&1
echo "rc=$?"; grep -v "amdgpu.ids\|set_dim" gpurun_out/r06_host_profile.txt | head -75 | cut -c1-170
