#!/bin/bash
# round 6 call 33: sanity at HEAD after the eligibility guard (host-side only): conv / VAE tests, smoke(), a short bench
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "halo or conv or upconv or stats" > gpurun_out/r06_call33_tests.txt 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r06_call33_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','n_gpus')})"
