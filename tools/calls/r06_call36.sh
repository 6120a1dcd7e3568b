#!/bin/bash
# round 6 call 36: host frames copied as RGBX (pad byte added on the device, PIL core filled by its raw "RGBX" decoder = row memcpy) against the RGB form:
# the host-frames test, then short bench runs alternating the two forms (LB_FRAMES_RGBX=0 / 1)
export LB_SYNTH_CACHE=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_native_gpu.py -x -q -m gpu -k "host_frames or reference_script or example" > gpurun_out/r06_call36_tests.txt 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r06_call36_tests.txt
: > gpurun_out/r06_frames_rgbx_ab.txt
for round in 1 2 3; do
  for v in 0 1; do
    LB_FRAMES_RGBX=$v timeout 900 python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('LB_FRAMES_RGBX=$v', round(d['value'],2), 'frames/s', round(d['ms_per_step'],3), 'ms')" >> gpurun_out/r06_frames_rgbx_ab.txt
  done
done
cat gpurun_out/r06_frames_rgbx_ab.txt
