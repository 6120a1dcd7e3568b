#!/bin/bash
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 330 python tools/unet_knob_ab.py > $OUT/r04_unet_knob_ab.txt 2>&1
echo "rc=$?"; grep -v amdgpu.ids $OUT/r04_unet_knob_ab.txt | tail -30
