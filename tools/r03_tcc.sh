#!/bin/bash
# L2 (TCC) hit / miss counts per dispatch of the last bench step: two counters per pass (more exceed the hardware's capacity)
cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_tcc
timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_tcc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_tcc.log 2>&1 || { grep -m3 -i "error" gpurun_out/pmc_tcc.log; exit 1; }
python tools/pmc_sq.py gpurun_out/r3_pmc_tcc.csv $(find gpurun_out/pmc_tcc -name "*counter_collection.csv")
