#!/bin/bash
# round-5 last GPU call: the whole GPU suite at HEAD (after the host-thread bounds of the new tests, the lighter DDIM transition, the
# pipe clean-up and the bench's new secondary lines), and the smoke entry point
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $OUT/r05_gpu_suite_head.txt 2>&1
echo "pytest rc=$?"; tail -n 16 $OUT/r05_gpu_suite_head.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
