#!/bin/bash
cd /root/repo
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_all4.log 2>&1; rc=$?
tail -4 gpurun_out/r3_gpu_all4.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --no-cpu-baseline --layers > gpurun_out/r3_w43_line.json 2> gpurun_out/r3_w43_line.err || { tail -20 gpurun_out/r3_w43_line.err; exit 1; }
python - <<'P'
import json; d=json.load(open("gpurun_out/r3_w43_line.json")); k=d["kernels"]
print(d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"], "alg_equiv", d["roofline"]["alg_equiv_tflops"], d["first_step_loss"], d["final_loss"])
print({f: (k[f]["ms_per_step"], k[f]["launches_per_step"], k[f].get("mfma_frac")) for f in k if f.startswith("conv")})
for l in d["layers"]:
    if l["family"].startswith("conv"): print(l)
P
