#!/bin/bash
# round 6 call 20: per-op tables at B = 1 (does a VAE level whose activations fit the 256 MB Infinity Cache run its GroupNorm / conv
# passes faster per sample than the B = 17 batch? - sizing a sample-major order of the decoder's top levels)
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 900 python tools/profile_programs.py 1 > gpurun_out/r06_profile_b1.log 2>&1
echo "profile rc=$?"
cp gpurun_out/program_profile.txt gpurun_out/r06_program_op_breakdown_b1.txt
grep -n "VAE decode program" -A 40 gpurun_out/r06_program_op_breakdown_b1.txt | head -60
