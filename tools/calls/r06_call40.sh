#!/bin/bash
# round 6 call 40: the vendor-library comparisons again at the final kernels (after the instruction diet): convs vs MIOpen, GEMMs vs rocBLAS
mkdir -p gpurun_out
export MIOPEN_USER_DB_PATH=/tmp/miopen_db MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen_cache
( timeout 900 python tools/conv_vs_miopen.py ) > gpurun_out/r06_conv_vs_miopen_final.txt 2>&1
echo "conv rc=$?"; grep -v "amdgpu.ids" gpurun_out/r06_conv_vs_miopen_final.txt | cut -c1-230 | tail -13
export GB_VARIANTS=auto,auto-noepi GB_NOCHECK=1
( timeout 300 tools/build/gemm_bench b17 5 ) > gpurun_out/r06_gemm_bench_final.txt 2>&1
echo "gemm rc=$?"; grep -v "^#" gpurun_out/r06_gemm_bench_final.txt | cut -c1-150
