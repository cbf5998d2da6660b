#!/bin/bash
cd /root/repo
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_all3.log 2>&1; rc=$?
tail -8 gpurun_out/r3_gpu_all3.log
[ $rc -eq 0 ] || exit $rc
run() { n=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline --steps 100 --layers "$@" > gpurun_out/r3w_$n.json 2>gpurun_out/r3w_$n.err || { tail -20 gpurun_out/r3w_$n.err; exit 1; }
python - $n <<'P'
import json,sys
d=json.load(open("gpurun_out/r3w_%s.json"%sys.argv[1])); k=d["kernels"]
print(sys.argv[1], d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"], "alg_equiv", d["roofline"]["alg_equiv_tflops"], "step frac", d["step_roofline"]["frac"], "first", d["first_step_loss"], "final", d["final_loss"], flush=True)
print("   ", {f: (k[f]["ms_per_step"], k[f]["launches_per_step"], k[f].get("mfma_frac")) for f in k if f.startswith("conv")}, flush=True)
P
}
ST3D_WINO43=0 run off
run on
ST3D_WINO43_MINK=64 run on64
timeout -k 10 300 python tools/wino43_layers.py 2>&1 | grep -v amdgpu.ids
