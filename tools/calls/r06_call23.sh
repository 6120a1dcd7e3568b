#!/bin/bash
# round 6 call 23: the programs' 3x3 convolutions - halo-tile kernel vs MIOpen (torch conv2d, channels_last, find mode) on the same operands
mkdir -p gpurun_out
export MIOPEN_USER_DB_PATH=/tmp/miopen_db MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen_cache
( timeout 1500 python tools/conv_vs_miopen.py ) > gpurun_out/r06_conv_vs_miopen.txt 2>&1
echo "rc=$?"; grep -v "amdgpu.ids" gpurun_out/r06_conv_vs_miopen.txt | cut -c1-260 | tail -20
