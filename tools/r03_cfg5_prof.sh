#!/bin/bash
set -x
cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/prof_cfg5
timeout -k 10 800 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg5 -- python bench.py --no-cpu-baseline --mesh bob --style 5 --target both --steps 200 > gpurun_out/prof_cfg5.log 2>&1 || { tail -20 gpurun_out/prof_cfg5.log; exit 1; }
f=$(find gpurun_out/prof_cfg5 -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -c1-200
rm -f $(find gpurun_out/prof_cfg5 -name "*kernel_trace.csv")
