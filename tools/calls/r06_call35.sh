#!/bin/bash
# round 6 call 35: does HIP_FORCE_DEV_KERNARG=1 (kernel arguments staged in device memory) lower the per-node floor of a dependent hipGraph chain and the
# launch-bound B = 2 UNet step?
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
( echo "== default"; timeout 120 tools/build/launch_floor 700; echo "== HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 timeout 120 tools/build/launch_floor 700 ) > gpurun_out/r06_kernarg_floor.txt 2>&1
cat gpurun_out/r06_kernarg_floor.txt | head -40
: > gpurun_out/r06_kernarg_programs.txt
for round in 1 2; do
  timeout 900 python tools/programs_lib_ab.py >> gpurun_out/r06_kernarg_programs.txt 2>&1
  HIP_FORCE_DEV_KERNARG=1 LB_HIP_LIBRARY=$PWD/latentblending_amd/hip/liblbhip.so timeout 900 python tools/programs_lib_ab.py 2>&1 | sed 's/\[liblbhip.so\]/[DEV_KERNARG=1]/' >> gpurun_out/r06_kernarg_programs.txt
done
grep best gpurun_out/r06_kernarg_programs.txt
