#!/bin/bash
# round-4 GPU call 4: does the ping-pong loop's rate follow the bytes in flight?  ring 8 slots / 4 ahead, 8 / 6, 10 / 8
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export GB_VARIANTS=auto,pp-m1,pp-r10,pp-d4 GB_NOROCBLAS=1
( timeout 600 tools/build/gemm_bench b17 5; timeout 300 tools/build/gemm_bench big 3 ) > $OUT/r04_gemm_bench_call4.txt 2>&1
echo "gemm_bench rc=$?"
grep -v "check" $OUT/r04_gemm_bench_call4.txt; grep "check" $OUT/r04_gemm_bench_call4.txt | grep -v "BIT-IDENTICAL" | head
unset GB_VARIANTS GB_NOROCBLAS
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "pingpong" > $OUT/r04_tests_call4.txt 2>&1
echo "pytest rc=$?"
tail -n 5 $OUT/r04_tests_call4.txt
