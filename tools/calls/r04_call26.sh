#!/bin/bash
# sanity of the clean-rebuilt library: kernel tests + smoke
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu > $OUT/r04_tests_call26.txt 2>&1
echo "kernel tests rc=$?"; tail -n 3 $OUT/r04_tests_call26.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
