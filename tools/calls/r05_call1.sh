#!/bin/bash
# round-5 GPU call 1: the one-round-trip tile epilogue - bit-identity tests, per-shape GEMM timing (lean vs per-row epilogue vs rocBLAS),
# in-program A/B (VAE B=17, UNet B=17 / B=2), a short bench line
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "lean_epilogue or gemm_epilogues or conv_resnet or halo or channel_stats or pingpong or subpixel or gemm_plain or glds_variant or layernorm_fused" > $OUT/r05_tests_call1.txt 2>&1
echo "pytest rc=$?"; tail -n 8 $OUT/r05_tests_call1.txt
export GB_VARIANTS=auto,auto-rowepi
( timeout 400 tools/build/gemm_bench b17 5; timeout 200 tools/build/gemm_bench b2 5 ) > $OUT/r05_gemm_bench_call1.txt 2>&1
echo "gemm_bench rc=$?"; grep -v "check" $OUT/r05_gemm_bench_call1.txt
unset GB_VARIANTS
timeout 900 python tools/epilogue_ab.py --unet > $OUT/r05_epilogue_ab.txt 2>&1
echo "epilogue_ab rc=$?"; grep -v Warning $OUT/r05_epilogue_ab.txt | tail -n 40
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/r05_bench_call1.json 2> $OUT/r05_bench_call1.err
echo "bench rc=$?"; tail -c 3000 $OUT/r05_bench_call1.json
