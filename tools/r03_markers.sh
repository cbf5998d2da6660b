#!/bin/bash
# roctx ranges of a few steps as rocprofv3 sees them (ST3D_ROCTX=1): the marker trace must name the step's phases
cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/markers
ST3D_ROCTX=1 timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --output-format csv -d gpurun_out/markers -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/markers.log 2>&1 || { tail -5 gpurun_out/markers.log; exit 1; }
f=$(find gpurun_out/markers -name "*marker_api_trace.csv" | head -1)
python - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
c = collections.Counter(r.get("Function", r.get("Name", "?")) for r in rows)
print(dict(c))
P
