#!/bin/bash
# round 6 call 3: streaming attention kernel v2 (tests + A/B against the rounds 1-5 kernel), tile-10 GEMM tests, UNet knob A/B
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention or 192x128 or glds_variant or lean_epilogue_gemm" > gpurun_out/r06_attn_tests.txt 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r06_attn_tests.txt
timeout 300 python tools/attn_ab.py > gpurun_out/r06_attn_ab.txt 2>&1
echo "attn_ab rc=$?"; cat gpurun_out/r06_attn_ab.txt
timeout 900 python tools/unet_knob_ab.py > gpurun_out/r06_unet_knob_ab.txt 2>&1
echo "knob rc=$?"; grep -E "^B=" gpurun_out/r06_unet_knob_ab.txt
