#!/bin/bash
# round 6 call 27: GroupNorm + SiLU / LayerNorm of the programs' shapes against torch's own kernels on the same GPU
mkdir -p gpurun_out
( timeout 600 python tools/norms_vs_torch.py ) > gpurun_out/r06_norms_vs_torch.txt 2>&1
echo "rc=$?"; grep -v "amdgpu.ids" gpurun_out/r06_norms_vs_torch.txt | cut -c1-250 | tail -14
