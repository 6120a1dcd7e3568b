#!/bin/bash
# round 6 call 16: split-K / tile alternatives for the B = 2 shapes
mkdir -p gpurun_out
export GB_VARIANTS=auto,t11-s2,t3-s2,t3-s8,t2,t2-s2,t2-s4,t2-s8,t1-s4,t1-s8,t10-s2 GB_NOCHECK=1 GB_NOROCBLAS=1
( timeout 400 tools/build/gemm_bench b2 5 ) > gpurun_out/r06_gemm_bench_call16.txt 2>&1
echo "gemm_bench rc=$?"; cat gpurun_out/r06_gemm_bench_call16.txt
