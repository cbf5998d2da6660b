#!/bin/bash
set -x
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gram" > gpurun_out/r3_gram_test.log 2>&1 || { tail -30 gpurun_out/r3_gram_test.log; exit 1; }
tail -3 gpurun_out/r3_gram_test.log
run() { timeout -k 10 300 python bench.py --steps 60 --no-cpu-baseline > gpurun_out/r3g_$1.json 2>gpurun_out/r3g_$1.err || { tail -20 gpurun_out/r3g_$1.err; exit 1; }
python - $1 <<'P'
import json,sys
d=json.load(open("gpurun_out/r3g_%s.json"%sys.argv[1])); k=d["kernels"]
print(sys.argv[1], d["ms_per_step"], d["roofline"]["frac"], "gram_fwd", k["gram_fwd"]["ms_per_step"], "final", d["final_loss"], flush=True)
P
}
ST3D_GRAM_MULTI_SCALE=1 run s1
ST3D_GRAM_MULTI_SCALE=2 run s2
ST3D_GRAM_MULTI_SCALE=4 run s4
ST3D_GRAM_MULTI_SCALE=8 run s8
ST3D_GRAM_MULTI_SCALE=2 ST3D_GRAM_MULTI_DEAL=1 run s2d
ST3D_GRAM_MULTI_SCALE=4 ST3D_GRAM_MULTI_DEAL=1 run s4d
ST3D_GRAM_MULTI_SCALE=3 run s3
