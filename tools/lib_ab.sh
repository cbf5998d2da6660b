#!/bin/bash
# Same-box A/B of two builds of libst3d.so: lib/libst3d.so (new) against lib/libst3d_prev.so (built from the previous
# commit with `git stash; build; cp; git stash pop; build`).  Alternates new/prev/new/prev, one bench line each.
L=2d-to-3d-style-transfer_amd/lib
run() { python bench.py --steps ${STEPS:-80} --no-cpu-baseline --layers > gpurun_out/ab_$1.json 2>/dev/null; python - $1 <<P
import json,sys
d=json.load(open("gpurun_out/ab_%s.json"%sys.argv[1])); k=d["kernels"]
f=[l["ms"] for l in d["layers"] if l["family"]=="conv_fwd"]; g=[l["ms"] for l in d["layers"] if l["family"]=="conv_dgrad"]
print(sys.argv[1], d["ms_per_step"], d["roofline"]["frac"], "fwd", k["conv_fwd"]["ms_per_step"], "dgrad", k["conv_dgrad"]["ms_per_step"], "gram", k["gram_fwd"]["ms_per_step"], k["gram_bwd"]["ms_per_step"], "| dgrad c1_2 %.3f c2_1 %.3f c2_2 %.3f c3_2 %.3f c3_4 %.3f c4_4 %.3f" % (g[0], g[1], g[2], g[4], g[6], g[10]), flush=True)
P
}
cp $L/libst3d.so $L/new.so
run new1; cp $L/libst3d_prev.so $L/libst3d.so; run prev1; cp $L/new.so $L/libst3d.so; run new2; cp $L/libst3d_prev.so $L/libst3d.so; run prev2; cp $L/new.so $L/libst3d.so
