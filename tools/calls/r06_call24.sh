#!/bin/bash
# round 6 call 24: the programs' attention calls - hand-written kernels vs torch SDPA (ROCm flash / mem-efficient) on the same operands
mkdir -p gpurun_out
( timeout 900 python tools/attn_vs_sdpa.py ) > gpurun_out/r06_attn_vs_sdpa.txt 2>&1
echo "rc=$?"; grep -v "amdgpu.ids" gpurun_out/r06_attn_vs_sdpa.txt | cut -c1-300 | tail -20
