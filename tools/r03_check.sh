#!/bin/bash
cd /root/repo
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_all2.log 2>&1; rc=$?
tail -6 gpurun_out/r3_gpu_all2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r3_default_line.json 2> gpurun_out/r3_default_line.err || { tail -20 gpurun_out/r3_default_line.err; exit 1; }
python - <<'P'
import json; d=json.load(open("gpurun_out/r3_default_line.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic_source"][:60], d["cpu_baseline"])
P
python -c "import __graft_entry__ as g; g.smoke()"
