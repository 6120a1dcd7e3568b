#!/bin/bash
# round 6 call 34: implicit-GEMM convolutions with lean requests (LB_GLDS_CLEAN: block-uniform tap per K-tile, per-request 32-bit pixel offset + tap mask, scalar
# tap offset / bit): kernel tests (conv, gemm, lpips, upsamplers), model-level tests, then the programs under the =0 library and the shipped one
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/r06_call34_tests_kernels.txt 2>&1
echo "kernel tests rc=$?"; tail -2 gpurun_out/r06_call34_tests_kernels.txt
timeout 1500 python -m pytest tests/test_native_gpu.py -x -q -m gpu -k "not farm and not full_size" > gpurun_out/r06_call34_tests_native.txt 2>&1
echo "native tests rc=$?"; tail -2 gpurun_out/r06_call34_tests_native.txt
: > gpurun_out/r06_glds_conv_lean_ab.txt
for round in 1 2; do
  LB_HIP_LIBRARY=$PWD/latentblending_amd/hip/liblbhip_ab0.so timeout 900 python tools/programs_lib_ab.py >> gpurun_out/r06_glds_conv_lean_ab.txt 2>&1
  timeout 900 python tools/programs_lib_ab.py >> gpurun_out/r06_glds_conv_lean_ab.txt 2>&1
done
grep "best" gpurun_out/r06_glds_conv_lean_ab.txt
