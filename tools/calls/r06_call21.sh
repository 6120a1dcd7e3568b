#!/bin/bash
# round 6 call 21: halo-tile conv with the A fragments of a step's first K-half read before the barrier - conv tests, then the
# in-process A/B over the VAE decode and UNet step programs
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "halo or conv or upconv or stats" > gpurun_out/r06_call21_tests.txt 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r06_call21_tests.txt
timeout 1200 python tools/halo_preread_ab.py > gpurun_out/r06_halo_preread_ab.txt 2>&1
echo "ab rc=$?"; grep -v "^set_dim\|amdgpu.ids" gpurun_out/r06_halo_preread_ab.txt | tail -30
