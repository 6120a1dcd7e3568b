#!/bin/bash
# round 6 call 15: small-batch GEGLU on the 8-wave 192x128 tile: full-size UNet parity (B = 2 / B = 17) + in-program timing
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "full_unet_batched or cfg2_transition" > gpurun_out/r06_fullsize_tests15.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r06_fullsize_tests15.txt
timeout 900 python tools/unet_knob_ab.py > gpurun_out/r06_unet_knob_ab15.txt 2>&1
echo "knob rc=$?"; grep -E "^B=" gpurun_out/r06_unet_knob_ab15.txt
