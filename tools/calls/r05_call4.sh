#!/bin/bash
# round-5 GPU call 4: DDIM (oracle restated with the device's fp16 semantics), the rest of the native / kernel files, like-for-like GEMM
# rows (no epilogue operands, as rocBLAS), PMC of the halo conv with the per-row and the lean epilogue
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_native_gpu.py -x -q -m gpu --durations=5 > $OUT/r05_tests_call4.txt 2>&1
echo "pytest rc=$?"; tail -n 14 $OUT/r05_tests_call4.txt
export GB_VARIANTS=auto,auto-noepi
( timeout 300 tools/build/gemm_bench b17 5 ) > $OUT/r05_gemm_bench_call4.txt 2>&1
echo "gemm_bench rc=$?"; grep -v "BIT-IDENTICAL" $OUT/r05_gemm_bench_call4.txt
unset GB_VARIANTS
cd /tmp
for LEAN in 0 1; do
  export LB_GEMM_LEAN_EPILOGUE=$LEAN
  timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/r05_halo_pmc/lean$LEAN/a -- python $R/tools/halo_pmc.py > $OUT/r05_halo_pmc_a$LEAN.log 2>&1
  echo "pmc a lean=$LEAN rc=$?"
  timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/r05_halo_pmc/lean$LEAN/b -- python $R/tools/halo_pmc.py > $OUT/r05_halo_pmc_b$LEAN.log 2>&1
  echo "pmc b lean=$LEAN rc=$?"
done
unset LB_GEMM_LEAN_EPILOGUE
cd $R
for LEAN in 0 1; do python tools/pmc_fold.py $OUT/r05_halo_pmc/lean$LEAN $OUT/r05_halo_pmc_lean$LEAN.json > /dev/null 2>&1; done
find $OUT/r05_halo_pmc -type f -size +1M -delete 2>/dev/null
head -c 3000 $OUT/r05_halo_pmc_lean1.json
