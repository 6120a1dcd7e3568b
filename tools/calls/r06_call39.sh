#!/bin/bash
# round 6 call 39: GroupNorm apply pass as ONE resident round of blocks walking small chunks (LB_GN_STRIDED) against the 2.0005-round grid: norm tests, then the
# programs under the =0 library and the shipped one, alternating
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "groupnorm or norm or stats or gn" > gpurun_out/r06_call39_tests.txt 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r06_call39_tests.txt
: > gpurun_out/r06_gn_strided_ab.txt
for round in 1 2; do
  LB_HIP_LIBRARY=$PWD/latentblending_amd/hip/liblbhip_ab0.so timeout 900 python tools/programs_lib_ab.py >> gpurun_out/r06_gn_strided_ab.txt 2>&1
  timeout 900 python tools/programs_lib_ab.py >> gpurun_out/r06_gn_strided_ab.txt 2>&1
done
grep "best" gpurun_out/r06_gn_strided_ab.txt
