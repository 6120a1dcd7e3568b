#!/bin/bash
# round 6 call 1: guidance-chain parity tests on the device + refreshed per-op tables (B=2, B=17 UNet / VAE)
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_baseline_configs_gpu.py -x -q -m gpu -k "guidance_chain" > gpurun_out/r06_guidance_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r06_guidance_tests.txt
timeout 900 python tools/profile_programs.py 2 17 > gpurun_out/r06_profile.log 2>&1
echo "profile rc=$?"
cp gpurun_out/program_profile.txt gpurun_out/r06_program_op_breakdown.txt
head -60 gpurun_out/r06_program_op_breakdown.txt
