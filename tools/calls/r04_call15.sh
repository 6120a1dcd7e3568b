#!/bin/bash
# LayerNorm fold inside the ping-pong GEMM: kernel tests, full-size B = 17 parity, A/B of the transition (kernel sums under rocprofv3 + wall clock)
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm_fused or pingpong" > $OUT/r04_tests_call15a.txt 2>&1
echo "kernel tests rc=$?"; tail -n 3 $OUT/r04_tests_call15a.txt
true

ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline"
for V in 0 1 0 1; do
  LB_UNET_LN_PP=$V python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LN_PP=$V', d['ms_per_step'], d['value'])"
done
cd /tmp
for V in 0 1; do
  LB_UNET_LN_PP=$V timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_lnpp$V -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r04_lnpp$V.log 2>&1
  f=$(find $OUT/r04_lnpp$V -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/r04_lnpp${V}_kernel_stats.csv
  find $OUT/r04_lnpp$V -type f -size +1M -delete
done
cd $R
python - <<'PY'
import csv
for v in (0,1):
    rows=list(csv.DictReader(open(f"gpurun_out/r04_lnpp{v}_kernel_stats.csv")))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"LN_PP={v}: total kernel ms {tot/1e6:.1f}")
    for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:14]:
        print(f"   {r['Name'][:70]:70s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:9.2f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
