#!/bin/bash
# same-box A/B with per-kernel durations: does routing the GEGLU / qkv GEMMs to the ping-pong kernel slow the OTHER kernels (power / clock)?
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline"
for v in 0 1 0 1; do
  tag=pp${v}_$RANDOM
  LB_GEMM_PP_AUTO=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_ab10_$tag -- python $R/bench.py $ARGS > $OUT/r04_ab10_$tag.log 2>&1
  echo "$tag rc=$?"
  for f in $(find $OUT/r04_ab10_$tag -name "*kernel_stats.csv"); do cp $f $OUT/r04_ab10_${tag}_kernel_stats.csv; done
  rm -rf $OUT/r04_ab10_$tag
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $OUT/r04_ab10_$tag.log | tr '\n' ' '; echo
done
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | head -30
