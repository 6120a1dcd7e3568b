#!/bin/bash
# round 6 call 18: host frames - copy + unpack at the end of the transition vs started at decode time (1 / 2 / 3 decode chunks),
# one process, interleaved; then the per-op tables of the programs at this commit ("after" of r06_program_op_breakdown_before.txt)
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
( timeout 900 python tools/host_frames_ab.py 10 ) > gpurun_out/r06_host_frames_ab.txt 2>&1
echo "host_frames_ab rc=$?"; grep -v "^set_dim" gpurun_out/r06_host_frames_ab.txt | tail -40
timeout 900 python tools/profile_programs.py 2 17 > gpurun_out/r06_profile_after.log 2>&1
echo "profile rc=$?"
cp gpurun_out/program_profile.txt gpurun_out/r06_program_op_breakdown_after.txt
head -12 gpurun_out/r06_program_op_breakdown_after.txt
