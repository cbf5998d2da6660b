#!/bin/bash
set -x
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_kernels.py -x -q -k "graph or stale_workspace or near_plane or raster" > gpurun_out/r3_misc_test.log 2>&1 || { tail -30 gpurun_out/r3_misc_test.log; exit 1; }
tail -3 gpurun_out/r3_misc_test.log
timeout -k 10 600 python bench.py --no-cpu-baseline --mesh bob --style 5 --target both --steps 200 > gpurun_out/r3_cfg5_200.json 2> gpurun_out/r3_cfg5_200.err || { tail -20 gpurun_out/r3_cfg5_200.err; exit 1; }
timeout -k 10 600 python bench.py --no-cpu-baseline --mesh bob --style 5 --target both --steps 50 > gpurun_out/r3_cfg5_50.json 2> gpurun_out/r3_cfg5_50.err || { tail -20 gpurun_out/r3_cfg5_50.err; exit 1; }
python - <<'P'
import json
for n in ("50", "200"):
    d=json.load(open("gpurun_out/r3_cfg5_%s.json" % n))
    print(n, d["ms_per_step"], d["median_ms_per_step"], d["value"], d["first_step_loss"], d["final_loss"])
P
tail -5 gpurun_out/r3_cfg5_200.err
