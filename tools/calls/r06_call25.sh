#!/bin/bash
# round 6 call 25: bench.py --torch-baseline - the oracle's PyTorch graph (UNet step, VAE decode) on the SAME GPU through the vendor libraries
# (torch eager, autocast fp16) beside this repo's launch programs
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
export MIOPEN_USER_DB_PATH=/tmp/miopen_db MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen_cache
timeout 1500 python bench.py --torch-baseline > gpurun_out/r06_torch_baseline.json 2> gpurun_out/r06_torch_baseline.err
echo "rc=$?"; cat gpurun_out/r06_torch_baseline.json; tail -5 gpurun_out/r06_torch_baseline.err | cut -c1-300
