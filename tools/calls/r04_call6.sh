#!/bin/bash
# round-4 GPU call 6: tile order inside an XCD's run (grouped GM x 32/GM patches vs strips): timing + fabric traffic (FETCH_SIZE)
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export GB_VARIANTS=auto,pp-m1,pp-g2,pp-g4,pp-g8 GB_NOROCBLAS=1
( timeout 600 tools/build/gemm_bench b17 5; timeout 300 tools/build/gemm_bench big 3 ) > $OUT/r04_gemm_bench_call6.txt 2>&1
echo "gemm_bench rc=$?"
grep -v "check" $OUT/r04_gemm_bench_call6.txt; grep "check" $OUT/r04_gemm_bench_call6.txt | grep -v "BIT-IDENTICAL" | head
cd /tmp
export GB_VARIANTS=pp-m1,pp-g4 GB_NOCHECK=1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace --output-format csv -d $OUT/r04_pmc6 -- $R/tools/build/gemm_bench big 1 > $OUT/r04_pmc6.log 2>&1
echo "pmc rc=$?"
cd $R
python tools/pmc_fold.py $OUT/r04_pmc6 $OUT/r04_pmc6.json > /dev/null 2>&1
find $OUT/r04_pmc6 -type f -size +512k -delete 2>/dev/null
grep -A2 "gemm_f16_pp" $OUT/r04_pmc6.json | head -40
