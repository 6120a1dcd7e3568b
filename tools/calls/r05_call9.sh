#!/bin/bash
# round-5 GPU call 9: the bench line with the new secondary lines (BASELINE configs[3] and [4] at full width on one GPU)
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 1200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r05_bench_call9.json 2> $OUT/r05_bench_call9.err
echo "bench rc=$?"; tail -n 5 $OUT/r05_bench_call9.err; python -c "
import json; d=json.loads(open('$OUT/r05_bench_call9.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])
for s in d['secondary']: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in s.items()})"
