#!/bin/bash
# round-5 GPU call 11: K sweep of M 4352 x N 1280 (forced 192x128 tile, forced 256x128 tile, automatic choice, rocBLAS): fixed cost and
# per-K-tile slope of a launch - the numbers behind the stream-K sizing in DESIGN.md section 4
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export GB_VARIANTS=auto,t7,t4 GB_NOCHECK=1
timeout 300 tools/build/gemm_bench ksweep 7 > $OUT/r05_gemm_ksweep.txt 2>&1
echo "rc=$?"; cat $OUT/r05_gemm_ksweep.txt
