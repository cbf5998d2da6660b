#!/bin/bash
# A/B of the wino4 stage-loop variants (compile-time switches of csrc/wino.hip): rebuild wino.o per variant, one bench line
# each.  Run on the GPU box from the repo root, variants as VAR=value tokens ("-" = defaults):
#   bash tools/wino_sched_ab.sh "- ST3D_WINO_SCHED=1 ST3D_WINO_BARRIER_MID=1 -"
set -e
for v in ${1:-- ST3D_WINO_SCHED=1}; do
  touch 2d-to-3d-style-transfer_amd/csrc/wino.hip
  if [ "$v" = "-" ]; then python 2d-to-3d-style-transfer_amd/build.py --jobs 16 > /dev/null; else env "$v" python 2d-to-3d-style-transfer_amd/build.py --jobs 16 > /dev/null; fi
  python bench.py --steps ${STEPS:-60} --no-cpu-baseline --layers > gpurun_out/sched.json 2> /dev/null
  python - "$v" <<'P'
import json, sys
d = json.load(open("gpurun_out/sched.json"))
k = d["kernels"]
f = [l["ms"] for l in d["layers"] if l["family"] == "conv_fwd"]
g = [l["ms"] for l in d["layers"] if l["family"] == "conv_dgrad"]
print("%-26s" % sys.argv[1], "step %.3f ms" % d["ms_per_step"], "frac", d["roofline"]["frac"], "fwd %.3f dgrad %.3f" % (k["conv_fwd"]["ms_per_step"], k["conv_dgrad"]["ms_per_step"]),
      "| fwd c1_2 %.3f c2_1 %.3f c2_2 %.3f c3_2 %.3f c4_2 %.3f | dgrad c1_2 %.3f c2_2 %.3f c3_2 %.3f c4_2 %.3f" % (f[0], f[1], f[2], f[4], f[8], g[0], g[2], g[4], g[8]), flush=True)
P
done
