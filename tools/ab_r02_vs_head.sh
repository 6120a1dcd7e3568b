#!/bin/bash
# Same-box A/B of the round-2 tree (git archive 183f997 unpacked into _ab_r02/, not committed) against HEAD: fresh gpurun boxes
# differ by 2-3 % in sustained clocks, so numbers of different calls cannot be compared.  Usage (repo root, GPU box):
#   bash tools/ab_r02_vs_head.sh > gpurun_out/ab_r02_vs_head.txt
# Prepare the comparison tree HERE first (the snapshot gpurun ships has no .git):  mkdir _ab_r02 && git archive 183f997 | tar -x -C _ab_r02
R=$PWD
export LB_SYNTH_CACHE=/tmp
(cd $R/_ab_r02 && python __graft_entry__.py > /tmp/ab_build.log 2>&1; tail -1 /tmp/ab_build.log)
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%.2f frames/s  %.2f ms' % (d['value'], d['ms_per_step']))" $1; }
for rep in $(seq 1 ${LB_AB_REPS:-2}); do
  (cd $R/_ab_r02 && timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/ab_r02.json 2>/dev/null)
  echo "rep $rep  round-2 tree (frames left in HBM)            : $(val /tmp/ab_r02.json)"
  (cd $R && timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary --no-materialise > /tmp/ab_head_nomat.json 2>/dev/null)
  echo "rep $rep  HEAD, frames left in HBM (round-2 semantics) : $(val /tmp/ab_head_nomat.json)"
  (cd $R && timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > /tmp/ab_head.json 2>/dev/null)
  echo "rep $rep  HEAD, frames materialised (the metric)        : $(val /tmp/ab_head.json)"
done
