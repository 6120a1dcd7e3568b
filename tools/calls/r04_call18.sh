#!/bin/bash
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp LB_SYNTH_CACHE=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_native_gpu.py -x -q -m gpu -k "slerp or mixing or lerp" > $OUT/r04_tests_call18.txt 2>&1
echo "slerp tests rc=$?"; tail -n 3 $OUT/r04_tests_call18.txt
cd /tmp
for i in 1 2; do
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_mixing -- python $R/tools/mixing_rocprof.py run > $OUT/r04_mixing.log 2>&1
cd $R
python tools/mixing_rocprof.py fold $OUT/r04_mixing $OUT/r04_mixing_rocprof.json > /dev/null 2>&1
rm -rf $OUT/r04_mixing
python -c "
import json; d=json.load(open('gpurun_out/r04_mixing_rocprof.json'))
for k in d['kernels']: print(k['name'][:60], round(k['avg_us_of_the_big_launches'],1), 'us', round(k['GB_per_s']), 'GB/s', round(k['frac_of_8TBs'],3))
"
cd /tmp
done
