#!/bin/bash
# round-5 GPU call 12: the 6-wave 192x128 tile against the automatic choice on every B = 17 program shape (with their epilogues)
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export GB_VARIANTS=auto,t7,t4 GB_NOROCBLAS=1
timeout 300 tools/build/gemm_bench b17 7 > $OUT/r05_gemm_t7.txt 2>&1
echo "rc=$?"; grep -v "BIT-IDENT" $OUT/r05_gemm_t7.txt
