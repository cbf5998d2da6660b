#!/bin/bash
# A/B of the wino4 stage-loop schedules (ST3D_WINO_SCHED=n, compile-time): rebuild wino.o per variant, one bench line each.
# Run on the GPU box from the repo root:  bash tools/wino_sched_ab.sh "0 1 2 3 0"
set -e
for v in ${1:-0 1}; do
  touch 2d-to-3d-style-transfer_amd/csrc/wino.hip
  ST3D_WINO_SCHED=$v python 2d-to-3d-style-transfer_amd/build.py --jobs 16 > /dev/null
  python bench.py --steps 60 --no-cpu-baseline --layers > gpurun_out/sched_$v.json 2> /dev/null
  python - "$v" <<'P'
import json, sys
d = json.load(open("gpurun_out/sched_%s.json" % sys.argv[1]))
k = d["kernels"]
f = [l["ms"] for l in d["layers"] if l["family"] == "conv_fwd"]
g = [l["ms"] for l in d["layers"] if l["family"] == "conv_dgrad"]
print("sched", sys.argv[1], "step %.3f ms" % d["ms_per_step"], "frac", d["roofline"]["frac"], "fwd %.3f dgrad %.3f" % (k["conv_fwd"]["ms_per_step"], k["conv_dgrad"]["ms_per_step"]),
      "| fwd c1_2 %.3f c2_1 %.3f c2_2 %.3f c3_2 %.3f c4_2 %.3f | dgrad c1_2 %.3f c2_2 %.3f c3_2 %.3f c4_2 %.3f" % (f[0], f[1], f[2], f[4], f[8], g[0], g[2], g[4], g[8]), flush=True)
P
done
