#!/bin/bash
# round 6 call 29 (call 28 + the wave index through readfirstlane): halo-tile conv with leaner address arithmetic (LB_HALO_LEAN_ADDR: weight requests as a 32-bit lane offset on a uniform base
# without zero-page selects; for TW = 32 one fragment address per pixel PAIR): conv tests on the shipped library, then the timing tool under the
# =0 library and the shipped one, alternating
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "halo or conv or upconv or stats" > gpurun_out/r06_call29_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r06_call29_tests.txt
: > gpurun_out/r06_halo_lean_addr_ab.txt
for round in 1 2; do
  LB_HIP_LIBRARY=$PWD/latentblending_amd/hip/liblbhip_ab0.so timeout 900 python tools/programs_lib_ab.py >> gpurun_out/r06_halo_lean_addr_ab.txt 2>&1
  timeout 900 python tools/programs_lib_ab.py >> gpurun_out/r06_halo_lean_addr_ab.txt 2>&1
done
grep "best" gpurun_out/r06_halo_lean_addr_ab.txt
