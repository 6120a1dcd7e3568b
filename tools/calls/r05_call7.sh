#!/bin/bash
# round-5 GPU call 7: the driver's smoke entry point at HEAD and a second bench line on another box (box-to-box spread)
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r05_smoke.txt 2>&1
echo "smoke rc=$?"; tail -n 3 $OUT/r05_smoke.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/r05_bench_box2.json 2> $OUT/r05_bench_box2.err
echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$OUT/r05_bench_box2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['phases_per_transition'])"
