#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/build/feed_rate > gpurun_out/r04_feed_rate.txt 2>&1
echo rc=$?
cat gpurun_out/r04_feed_rate.txt
