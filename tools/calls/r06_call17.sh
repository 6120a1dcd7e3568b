#!/bin/bash
# round 6 call 17: (a) per-node floor of a dependent hipGraph chain, (b) B = 2 GEMM shapes with WARM weights and with the next
# weight copy touched from a second stream - is there anything for a weight prefetcher to win?
mkdir -p gpurun_out
( timeout 200 tools/build/launch_floor 700 ) > gpurun_out/r06_launch_floor.txt 2>&1
echo "launch_floor rc=$?"; cat gpurun_out/r06_launch_floor.txt
export GB_VARIANTS=auto GB_NOCHECK=1
( timeout 300 tools/build/gemm_bench b2 5 ) > gpurun_out/r06_gemm_bench_call17_cold.txt 2>&1
echo "cold rc=$?"; cat gpurun_out/r06_gemm_bench_call17_cold.txt
( GB_WARM=1 timeout 300 tools/build/gemm_bench b2 5 ) > gpurun_out/r06_gemm_bench_call17_warm.txt 2>&1
echo "warm rc=$?"; cat gpurun_out/r06_gemm_bench_call17_warm.txt
( GB_PREFETCH=32 GB_NOROCBLAS=1 timeout 300 tools/build/gemm_bench b2 5 ) > gpurun_out/r06_gemm_bench_call17_prefetch32.txt 2>&1
echo "prefetch32 rc=$?"; cat gpurun_out/r06_gemm_bench_call17_prefetch32.txt
( GB_PREFETCH=96 GB_NOROCBLAS=1 timeout 300 tools/build/gemm_bench b2 5 ) > gpurun_out/r06_gemm_bench_call17_prefetch96.txt 2>&1
echo "prefetch96 rc=$?"; cat gpurun_out/r06_gemm_bench_call17_prefetch96.txt
