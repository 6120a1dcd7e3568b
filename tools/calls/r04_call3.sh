#!/bin/bash
# round-4 GPU call 3: one-wave-per-SIMD GEMM (gemm_w4.hip): bit-identity + timing vs ping-pong / lock-step / rocBLAS, PMC, the vendor kernels' names
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export GB_VARIANTS=auto,w4,pp-m1,t4
( timeout 600 tools/build/gemm_bench b17 5; timeout 300 tools/build/gemm_bench big 3 ) > $OUT/r04_gemm_bench_call3.txt 2>&1
echo "gemm_bench rc=$?"
grep -v "check" $OUT/r04_gemm_bench_call3.txt; grep "check" $OUT/r04_gemm_bench_call3.txt | grep -v "BIT-IDENTICAL" | head
cd /tmp
export GB_VARIANTS=auto,w4,pp-m1 GB_NOCHECK=1 GB_NOROCBLAS=1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $OUT/r04_pmc3_sq -- $R/tools/build/gemm_bench big 1 > $OUT/r04_pmc3_sq.log 2>&1
echo "pmc sq rc=$?"
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC \
    --kernel-trace --output-format csv -d $OUT/r04_pmc3_b -- $R/tools/build/gemm_bench big 1 > $OUT/r04_pmc3_b.log 2>&1
echo "pmc b rc=$?"
unset GB_NOROCBLAS
export GB_VARIANTS=auto
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_vendor_names -- $R/tools/build/gemm_bench b17 1 > $OUT/r04_vendor_names.log 2>&1
echo "vendor names rc=$?"
unset GB_VARIANTS GB_NOCHECK
cd $R
python tools/pmc_fold.py $OUT/r04_pmc3_sq $OUT/r04_pmc3_sq.json > /dev/null 2>&1
python tools/pmc_fold.py $OUT/r04_pmc3_b $OUT/r04_pmc3_b.json > /dev/null 2>&1
for f in $(find $OUT/r04_vendor_names -name "*kernel_stats.csv"); do cp $f $OUT/r04_vendor_kernel_stats.csv; done
find $OUT/r04_pmc3_sq $OUT/r04_pmc3_b $OUT/r04_vendor_names -type f -size +512k -delete 2>/dev/null
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "w4 or pingpong" > $OUT/r04_tests_call3.txt 2>&1
echo "pytest rc=$?"
tail -n 15 $OUT/r04_tests_call3.txt
