#!/bin/bash
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 600 python -m pytest tests/test_baseline_configs_gpu.py -q -m gpu > $OUT/r04_tests_call24.txt 2>&1
echo "baseline-config tests rc=$?"; tail -n 12 $OUT/r04_tests_call24.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
