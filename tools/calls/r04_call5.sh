#!/bin/bash
# round-4 GPU call 5: ping-pong with the LDS-DMA requests inside the compute segments (ring 3)
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export GB_VARIANTS=auto,pp-m1,pp-sic GB_NOROCBLAS=1
( timeout 600 tools/build/gemm_bench b17 5; timeout 300 tools/build/gemm_bench big 3 ) > $OUT/r04_gemm_bench_call5.txt 2>&1
echo "gemm_bench rc=$?"
grep -v "check" $OUT/r04_gemm_bench_call5.txt; grep "check" $OUT/r04_gemm_bench_call5.txt | grep -v "BIT-IDENTICAL" | head
cd /tmp
export GB_VARIANTS=pp-m1,pp-sic GB_NOCHECK=1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
    --kernel-trace --output-format csv -d $OUT/r04_pmc5 -- $R/tools/build/gemm_bench big 1 > $OUT/r04_pmc5.log 2>&1
echo "pmc rc=$?"
unset GB_VARIANTS GB_NOROCBLAS GB_NOCHECK
cd $R
python tools/pmc_fold.py $OUT/r04_pmc5 $OUT/r04_pmc5.json > /dev/null 2>&1
find $OUT/r04_pmc5 -type f -size +512k -delete 2>/dev/null
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "pingpong" > $OUT/r04_tests_call5.txt 2>&1
echo "pytest rc=$?"
tail -n 5 $OUT/r04_tests_call5.txt
