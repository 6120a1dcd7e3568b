#!/bin/bash
# round 6 call 19: epilogue operands requested in front of the first K-tile - bit-identity tests, like-for-like GEMM timings (off / on),
# in-program A/B over the full UNet step (B = 2 / B = 17)
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "epilogue or gemm" > gpurun_out/r06_call19_tests.txt 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r06_call19_tests.txt
export GB_VARIANTS=auto,auto-noepi GB_NOCHECK=1
for pf in 0 1; do
  ( GB_EPI_PREFETCH=$pf timeout 300 tools/build/gemm_bench all 5 ) > gpurun_out/r06_gemm_bench_call19_prefetch$pf.txt 2>&1
  echo "gemm_bench prefetch=$pf rc=$?"
done
paste -d'\n' gpurun_out/r06_gemm_bench_call19_prefetch0.txt gpurun_out/r06_gemm_bench_call19_prefetch1.txt | grep -v "^#" | awk 'NR%2==1 {a=$0} NR%2==0 {if (a ~ /^M=/) print a; else print "  off: " a "   | on: " $0}' | grep -v "BIG\|8192\|4096" | head -80
LB_KNOB_FILTER=epilogue timeout 900 python tools/unet_knob_ab.py > gpurun_out/r06_unet_knob_ab_call19.txt 2>&1
echo "knob rc=$?"; grep -v "^set_dim\|amdgpu.ids" gpurun_out/r06_unet_knob_ab_call19.txt | tail -24
