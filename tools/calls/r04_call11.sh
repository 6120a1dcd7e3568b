#!/bin/bash
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/halo_pp_ab.py > $OUT/r04_halo_pp_ab.txt 2>&1; echo "ab rc=$?"
cat $OUT/r04_halo_pp_ab.txt | grep -v amdgpu.ids
LB_CONV_HALO_PP=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "halo or conv or groupnorm_from" > $OUT/r04_tests_call11.txt 2>&1
echo "pytest rc=$?"; tail -n 5 $OUT/r04_tests_call11.txt
