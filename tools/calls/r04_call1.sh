#!/bin/bash
# round-4 GPU call 1: ping-pong GEMM correctness + timing (torch-free), then the new parity tests
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 tools/build/gemm_bench b17 5; timeout 300 tools/build/gemm_bench big 3 ) > gpurun_out/r04_gemm_bench_call1.txt 2>&1
echo "gemm_bench rc=$?"
tail -n 120 gpurun_out/r04_gemm_bench_call1.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_baseline_configs_gpu.py tests/test_native_gpu.py -x -q -m gpu -k "pingpong or stated or chain_native or state_round or duck_type" > gpurun_out/r04_tests_call1.txt 2>&1
echo "pytest rc=$?"
tail -n 30 gpurun_out/r04_tests_call1.txt
