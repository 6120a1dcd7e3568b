#!/bin/bash
# round-4 GPU call 2: ping-pong GEMM modes (timing + bit-identity), PMC of the main loops, the parity tests that did not run in call 1
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 600 tools/build/gemm_bench b17 5; timeout 300 tools/build/gemm_bench big 3 ) > $OUT/r04_gemm_bench_call2.txt 2>&1
echo "gemm_bench rc=$?"
cat $OUT/r04_gemm_bench_call2.txt
cd /tmp
export GB_VARIANTS=auto,pp-m1,pp-m0 GB_NOCHECK=1 GB_NOROCBLAS=1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $OUT/r04_pmc_sq -- $R/tools/build/gemm_bench big 1 > $OUT/r04_pmc_sq.log 2>&1
echo "pmc sq rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $OUT/r04_pmc_sq_b17 -- $R/tools/build/gemm_bench b17 1 > $OUT/r04_pmc_sq_b17.log 2>&1
echo "pmc sq b17 rc=$?"
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace --output-format csv -d $OUT/r04_pmc_fetch -- $R/tools/build/gemm_bench big 1 > $OUT/r04_pmc_fetch.log 2>&1
echo "pmc fetch rc=$?"
unset GB_VARIANTS GB_NOCHECK GB_NOROCBLAS
cd $R
python tools/pmc_fold.py $OUT/r04_pmc_sq $OUT/r04_pmc_sq.json > /dev/null 2>&1
python tools/pmc_fold.py $OUT/r04_pmc_sq_b17 $OUT/r04_pmc_sq_b17.json > /dev/null 2>&1
python tools/pmc_fold.py $OUT/r04_pmc_fetch $OUT/r04_pmc_fetch.json > /dev/null 2>&1
find $OUT/r04_pmc_sq $OUT/r04_pmc_sq_b17 $OUT/r04_pmc_fetch -type f -size +1M -delete 2>/dev/null
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu -k "pingpong or cfg4 or chain_native or state_round" > $OUT/r04_tests_call2.txt 2>&1
echo "pytest rc=$?"
tail -n 15 $OUT/r04_tests_call2.txt
