#!/bin/bash
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_native_gpu.py -x -q -m gpu -k "attention or attn or clip or unet" > $OUT/r04_tests_call21.txt 2>&1
echo "attention tests rc=$?"; tail -n 3 $OUT/r04_tests_call21.txt
timeout 300 python tools/attn_ab.py > $OUT/r04_attn_ab2.txt 2>&1
cat $OUT/r04_attn_ab2.txt | grep -v amdgpu.ids
