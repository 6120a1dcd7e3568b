#!/bin/bash
# round-5 GPU call 3: GPU suite without the full-size file (new: strict cfg-4 fixture, frontier 64 vs the oracle engine, DDIM, sessions),
# the residency A/B (UNet step alone vs with the VAE resident), kernel statistics of the bench command after the lean epilogue
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 1500 python -m pytest tests/test_baseline_configs_gpu.py tests/test_kernels_gpu.py tests/test_native_gpu.py -x -q -m gpu --durations=8 > $OUT/r05_tests_call3.txt 2>&1
echo "pytest rc=$?"; tail -n 22 $OUT/r05_tests_call3.txt
timeout 600 python tools/residency_ab.py > $OUT/r05_residency_ab.txt 2>&1
echo "residency rc=$?"; grep -v Warning $OUT/r05_residency_ab.txt | tail -n 20
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05c3_stats -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $OUT/r05c3_stats.log 2>&1
echo "stats rc=$?"; tail -c 1500 $OUT/r05c3_stats.log
cd $R
for f in $(find $OUT/r05c3_stats -name "*kernel_stats.csv"); do cp $f $OUT/r05c3_kernel_stats.csv; done
find $OUT/r05c3_stats -type f -size +1M -delete 2>/dev/null
