#!/bin/bash
# Round-end measurement on one MI355X (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the driver's bench command            -> gpurun_out/<tag>_stats/
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, eager)   -> gpurun_out/<tag>_pmc_*/
#   3. tools/pmc_summary.py folds them into profiles/<tag>_rocprof_summary.json  (bench.py reads roofline.traffic from it)
#   4. the bench line itself, cfg 2 (+ the secondary lines: skewed metric, cfg 3)
# Raw traces are dropped after summarising (gpurun merges at most 64 MiB back).
TAG=${1:-r06}
COMMIT=${2:-unknown}        # git rev-parse --short HEAD of the tree being measured (passed in: the GPU box has no .git)
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export LB_SYNTH_CACHE=/tmp          # seeded synthetic weights: generated once, the later processes load them
ARGS="--steps 20 --warmup 5"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python $R/bench.py $ARGS --no-cpu-baseline --no-roofline --no-secondary > $OUT/${TAG}_stats.log 2>&1
echo "stats rc=$?"
PARGS="--steps 4 --warmup 1"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$C -- python $R/bench.py $PARGS --no-graphs --no-cpu-baseline --no-roofline --no-secondary > $OUT/${TAG}_pmc_$C.log 2>&1
  echo "pmc $C rc=$?"
done
cd $R
python tools/pmc_summary.py $TAG gpurun_out "$ARGS (kernel stats) / $PARGS (PMC passes)" $COMMIT > $OUT/${TAG}_pmc_summary.log 2>&1
cp profiles/${TAG}_rocprof_summary.json $OUT/ 2>/dev/null
# keep the per-kernel summaries, drop the raw traces
for f in $(find $OUT/${TAG}_stats -name "*kernel_stats.csv" -o -name "*domain_stats.csv"); do cp $f $OUT/${TAG}_bench_$(basename $f | sed 's/^[0-9]*_//'); done
find $OUT/${TAG}_stats $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE -type f -size +2M -delete 2>/dev/null
timeout 1200 python bench.py $ARGS > $OUT/${TAG}_bench_1gpu.json 2> $OUT/${TAG}_bench_1gpu.err
echo "bench rc=$?"; tail -c 600 $OUT/${TAG}_bench_1gpu.json | head -c 600; echo
# (the skewed-metric and cfg-3 lines are part of the bench line itself since round 3: "secondary")
# rocprof-reported GB/s of the mixing / scheduler kernels on >= 1 GiB batches
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_mixing -- python $R/tools/mixing_rocprof.py run > $OUT/${TAG}_mixing.log 2>&1
cd $R
python tools/mixing_rocprof.py fold $OUT/${TAG}_mixing $OUT/${TAG}_mixing_rocprof.json > /dev/null 2>&1
find $OUT/${TAG}_mixing -type f -size +2M -delete 2>/dev/null
ls -la $OUT | head -40
du -sh $OUT
