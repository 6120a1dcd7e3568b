#!/bin/bash
# round 6 call 14: the B = 2 GEMM shapes on every tile family (is the small-M policy leaving time on the table?)
mkdir -p gpurun_out
export GB_VARIANTS=auto,t1,t4,t10,t3,t11 GB_NOCHECK=1
( timeout 400 tools/build/gemm_bench b2 5 ) > gpurun_out/r06_gemm_bench_call14.txt 2>&1
echo "gemm_bench rc=$?"; cat gpurun_out/r06_gemm_bench_call14.txt
