#!/bin/bash
# round 6 call 4: ping-pong attention kernel (tests + A/B), branch-free Q loads in every attention kernel
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > gpurun_out/r06_attn_tests4.txt 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r06_attn_tests4.txt
timeout 300 python tools/attn_ab.py > gpurun_out/r06_attn_ab4.txt 2>&1
echo "attn_ab rc=$?"; cat gpurun_out/r06_attn_ab4.txt
