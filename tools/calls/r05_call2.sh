#!/bin/bash
# round-5 GPU call 2: the lean epilogue with the fp32 result kept out of v_fma_mix (call 1: rounding-level differences) - the whole
# kernel test file, per-shape timing incl. the forced 256x128 tile, in-program A/B with the identity check
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $OUT/r05_tests_call2.txt 2>&1
echo "pytest rc=$?"; tail -n 8 $OUT/r05_tests_call2.txt
export GB_VARIANTS=auto,auto-rowepi,t4
( timeout 400 tools/build/gemm_bench b17 5 ) > $OUT/r05_gemm_bench_call2.txt 2>&1
echo "gemm_bench rc=$?"; grep -v "BIT-IDENTICAL" $OUT/r05_gemm_bench_call2.txt
unset GB_VARIANTS
timeout 900 python tools/epilogue_ab.py --unet > $OUT/r05_epilogue_ab.txt 2>&1
echo "epilogue_ab rc=$?"; grep -v Warning $OUT/r05_epilogue_ab.txt | tail -n 40
