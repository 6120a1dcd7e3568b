#!/bin/bash
# round 6 call 37: RGBX frame cores built on a persistent thread pool (LB_FRAMES_THREADS = 0 / 3 / 6) - alternating short bench runs
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
: > gpurun_out/r06_frames_threads_ab.txt
for round in 1 2; do
  for v in 0 3 6; do
    LB_FRAMES_THREADS=$v timeout 900 python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('LB_FRAMES_THREADS=$v', round(d['value'],2), 'frames/s', round(d['ms_per_step'],3), 'ms')" >> gpurun_out/r06_frames_threads_ab.txt
  done
done
cat gpurun_out/r06_frames_threads_ab.txt
