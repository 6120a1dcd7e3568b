#!/bin/bash
# round 6 call 31: GroupNorm apply pass with non-temporal stores (nt1) / loads + stores (nt3) against the shipped library: VAE decode + UNet programs
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
: > gpurun_out/r06_gn_nt_ab.txt
for round in 1 2; do
  timeout 900 python tools/programs_lib_ab.py >> gpurun_out/r06_gn_nt_ab.txt 2>&1
  LB_HIP_LIBRARY=$PWD/latentblending_amd/hip/liblbhip_nt1.so timeout 900 python tools/programs_lib_ab.py >> gpurun_out/r06_gn_nt_ab.txt 2>&1
  LB_HIP_LIBRARY=$PWD/latentblending_amd/hip/liblbhip_nt3.so timeout 900 python tools/programs_lib_ab.py >> gpurun_out/r06_gn_nt_ab.txt 2>&1
done
grep "best" gpurun_out/r06_gn_nt_ab.txt
