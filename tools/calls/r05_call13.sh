#!/bin/bash
# round-5 GPU call 13: after the tile-7 rule lost its K limit - GEMM tests, the UNet step in situ (lean epilogue A/B doubles as the timing)
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or lean" > $OUT/r05_tests_call13.txt 2>&1
echo "pytest rc=$?"; tail -n 4 $OUT/r05_tests_call13.txt
timeout 600 python tools/epilogue_ab.py --unet > $OUT/r05_epilogue_ab_call13.txt 2>&1
echo "ab rc=$?"; grep -v Warning $OUT/r05_epilogue_ab_call13.txt | tail -n 16
