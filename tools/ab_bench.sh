#!/bin/bash
# on the GPU box: bench.py under each of the given library builds (lib/diag/libst3d_<name>.so, copied over lib/libst3d.so
# on the box's scratch copy), twice, interleaved: ms/step and the per-family times
cd /root/repo
L=2d-to-3d-style-transfer_amd/lib
cp $L/libst3d.so $L/diag/_shipped.so
for rep in 1 2; do
  for n in "$@"; do
    cp $L/diag/libst3d_$n.so $L/libst3d.so
    timeout -k 10 300 python bench.py --steps ${STEPS:-80} --no-cpu-baseline --layers > gpurun_out/ab_$n.json 2>/dev/null || exit 1
    python - $n <<P
import json,sys
d=json.load(open("gpurun_out/ab_%s.json"%sys.argv[1])); k=d["kernels"]
print(sys.argv[1], d["ms_per_step"], {f: k[f]["ms_per_step"] for f in k if isinstance(k[f], dict) and "ms_per_step" in k[f]}, flush=True)
P
  done
done
cp $L/diag/_shipped.so $L/libst3d.so
