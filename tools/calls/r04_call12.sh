#!/bin/bash
mkdir -p gpurun_out
export GB_VARIANTS=auto,pp-m1,pp-sk2,pp-sk3,pp-sk4,t4-sk2 GB_NOROCBLAS=1
timeout 600 tools/build/gemm_bench b17 5 > gpurun_out/r04_gemm_bench_call12.txt 2>&1
echo rc=$?
grep -v "BIT-IDENTICAL" gpurun_out/r04_gemm_bench_call12.txt
