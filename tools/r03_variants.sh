#!/bin/bash
set -x
cd /root/repo
run() { n=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > gpurun_out/r3v_$n.json 2>gpurun_out/r3v_$n.err || { tail -20 gpurun_out/r3v_$n.err; exit 1; }
python - $n <<'P'
import json,sys
d=json.load(open("gpurun_out/r3v_%s.json"%sys.argv[1]))
print(sys.argv[1], d["metric"], d["value"], d["ms_per_step"], "roofline", d["roofline"].get("kernel"), d["roofline"]["frac"], "step", d["step_roofline"]["frac"], d["step_roofline"].get("host_or_launch_bound_share"), flush=True)
print("   ", {k: (v.get("ms_per_step", v.get("ms_per_call"))) for k, v in d.get("kernels", {}).items()}, flush=True)
P
}
run second --steps 100
run noise --steps 100 --background noise
run first_a --steps 100 --approach first_a
run first_b --steps 300 --approach first_b
run first_b_both --steps 300 --approach first_b --target both
python -c "
import sys; sys.path.insert(0,'2d-to-3d-style-transfer_amd')
import torch, utils as U
v=U.get_vgg(seed=0); p=v.plan(8,512); print('plan bytes 8x512', p.bytes()/1e9)
p2=v.plan(16,1024); print('plan bytes 16x1024', p2.bytes()/1e9)
"
