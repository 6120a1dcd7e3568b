#!/bin/bash
# round 6 call 5: XCD-aware block order of the attention kernels (tests + A/B), in-block K-groups for the 64x64 GEMM tile (gemm_bench b2)
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > gpurun_out/r06_attn_tests5.txt 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r06_attn_tests5.txt
timeout 300 python tools/attn_ab.py > gpurun_out/r06_attn_ab5.txt 2>&1
echo "attn_ab rc=$?"; cat gpurun_out/r06_attn_ab5.txt
export GB_VARIANTS=auto,t3,t11,t3-s1,t11-s1
( timeout 400 tools/build/gemm_bench b2 5 ) > gpurun_out/r06_gemm_bench_call5.txt 2>&1
echo "gemm_bench rc=$?"; grep -v "BIT-IDENTICAL" gpurun_out/r06_gemm_bench_call5.txt
