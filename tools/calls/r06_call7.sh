#!/bin/bash
# round 6 call 7: one-launch GroupNorm (tests + in-program A/B), whole kernel test file for regressions
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/r06_kernel_tests7.txt 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r06_kernel_tests7.txt
timeout 900 python tools/unet_knob_ab.py > gpurun_out/r06_unet_knob_ab7.txt 2>&1
echo "knob rc=$?"; grep -E "^B=" gpurun_out/r06_unet_knob_ab7.txt
