#!/bin/bash
# round 6 call 22: halo-tile conv, A fragments of a step's first K-half read before the barrier (af0 persistent across steps, compile-time
# switch LB_HALO_PREREAD): conv tests on the shipped library, then the timing tool under the =0 library and the shipped one, alternating
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "halo or conv or upconv or stats" > gpurun_out/r06_call22_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r06_call22_tests.txt
: > gpurun_out/r06_halo_preread_ab.txt
for round in 1 2; do
  LB_HIP_LIBRARY=$PWD/latentblending_amd/hip/liblbhip_ab0.so timeout 900 python tools/halo_preread_ab.py >> gpurun_out/r06_halo_preread_ab.txt 2>&1
  timeout 900 python tools/halo_preread_ab.py >> gpurun_out/r06_halo_preread_ab.txt 2>&1
done
grep "best" gpurun_out/r06_halo_preread_ab.txt
