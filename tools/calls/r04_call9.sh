#!/bin/bash
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_stats9 -- python $R/bench.py $ARGS > $OUT/r04_stats9.log 2>&1
echo "rc=$?"
for f in $(find $OUT/r04_stats9 -name "*kernel_stats.csv"); do cp $f $OUT/r04_stats9_kernel_stats.csv; done
find $OUT/r04_stats9 -type f -size +256k -delete 2>/dev/null
cut -c1-140 $OUT/r04_stats9_kernel_stats.csv | head -40
