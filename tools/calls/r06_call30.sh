#!/bin/bash
# round 6 call 30: direct-to-LDS GEMM with lean request addressing (LB_GLDS_LEAN: 32-bit lane offsets on a uniform base that carries K, no zero-page
# selects, scalar M0; K loop instantiated per request form; VGPR-form MFMAs): GEMM tests, like-for-like timings under the =0 library and the
# shipped one, then the full programs under both
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or epilogue or layernorm or geglu or splitk or kgroup or lna" > gpurun_out/r06_call30_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r06_call30_tests.txt
export GB_VARIANTS=auto GB_NOCHECK=1 GB_NOROCBLAS=1
( timeout 300 tools/build/gemm_bench_ab0 all 5 ) > gpurun_out/r06_gemm_bench_call30_lean0.txt 2>&1; echo "bench0 rc=$?"
( timeout 300 tools/build/gemm_bench all 5 ) > gpurun_out/r06_gemm_bench_call30_lean1.txt 2>&1; echo "bench1 rc=$?"
paste -d'\n' gpurun_out/r06_gemm_bench_call30_lean0.txt gpurun_out/r06_gemm_bench_call30_lean1.txt | grep -v "^#" | awk 'NR%2==1 {a=$0} NR%2==0 {if (a ~ /^M=/) print a; else print "  old: " a "   | lean: " $0}' | head -60
: > gpurun_out/r06_glds_lean_ab.txt
for round in 1 2; do
  LB_HIP_LIBRARY=$PWD/latentblending_amd/hip/liblbhip_ab0.so timeout 900 python tools/programs_lib_ab.py >> gpurun_out/r06_glds_lean_ab.txt 2>&1
  timeout 900 python tools/programs_lib_ab.py >> gpurun_out/r06_glds_lean_ab.txt 2>&1
done
grep "best" gpurun_out/r06_glds_lean_ab.txt
