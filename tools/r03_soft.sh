#!/bin/bash
set -x
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py -x -q -k "soft or near_plane or clip or renderer" > gpurun_out/r3_soft_test.log 2>&1 || { tail -40 gpurun_out/r3_soft_test.log; exit 1; }
tail -3 gpurun_out/r3_soft_test.log
timeout -k 10 600 python bench.py --no-cpu-baseline --mesh bob --style 5 --target both --steps 200 > gpurun_out/r3_cfg5b_200.json 2> gpurun_out/r3_cfg5b_200.err || { tail -20 gpurun_out/r3_cfg5b_200.err; exit 1; }
python - <<'P'
import json
d=json.load(open("gpurun_out/r3_cfg5b_200.json"))
print(200, d["ms_per_step"], d["median_ms_per_step"], d["value"], d["first_step_loss"], d["final_loss"])
P
