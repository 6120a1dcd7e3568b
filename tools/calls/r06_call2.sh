#!/bin/bash
# round 6 call 2: 8-wave 192x128 GEMM tile (t10) vs the 6-wave form (t7) / auto / rocBLAS; new LayerNorm kernel (bit-identity test + in-situ time)
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm" > gpurun_out/r06_ln_tests.txt 2>&1
echo "ln tests rc=$?"; tail -3 gpurun_out/r06_ln_tests.txt
export GB_VARIANTS=auto,t7,t10,t4
( timeout 400 tools/build/gemm_bench b17 5 ) > gpurun_out/r06_gemm_bench_call2.txt 2>&1
echo "gemm_bench rc=$?"; grep -v "BIT-IDENTICAL" gpurun_out/r06_gemm_bench_call2.txt
unset GB_VARIANTS
timeout 900 python tools/profile_programs.py 17 > gpurun_out/r06_profile2.log 2>&1
echo "profile rc=$?"
cp gpurun_out/program_profile.txt gpurun_out/r06_program_op_breakdown_call2.txt
grep -E "layernorm|hipGraph|ops," gpurun_out/r06_program_op_breakdown_call2.txt
