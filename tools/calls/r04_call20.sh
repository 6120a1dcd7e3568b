#!/bin/bash
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > $OUT/r04_tests_call20.txt 2>&1
echo "attention tests rc=$?"; tail -n 3 $OUT/r04_tests_call20.txt
timeout 300 python tools/attn_ab.py > $OUT/r04_attn_ab.txt 2>&1
cat $OUT/r04_attn_ab.txt | grep -v amdgpu.ids
