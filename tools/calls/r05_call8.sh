#!/bin/bash
# (record: ran at commit dc2e3d2, whose in-launch split-K reduction was then reverted - the --knob option of tools/epilogue_ab.py existed there)
# round-5 GPU call 8: in-launch split-K reduction - kernel tests, the model-level tests that run the B = 2 programs, in-program A/B
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $OUT/r05_tests_call8a.txt 2>&1
echo "kernels rc=$?"; tail -n 6 $OUT/r05_tests_call8a.txt
timeout 1200 python -m pytest tests/test_native_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "unet_tiny or transition_tree or frontier_equals or wavefront or dead_step or full_unet_batched or ddim or duck_type" > $OUT/r05_tests_call8b.txt 2>&1
echo "native rc=$?"; tail -n 6 $OUT/r05_tests_call8b.txt
timeout 600 python tools/epilogue_ab.py --unet --no-vae --knob fused_splitk > $OUT/r05_fused_splitk_ab.txt 2>&1
echo "ab rc=$?"; grep -v Warning $OUT/r05_fused_splitk_ab.txt | tail -n 14
