#!/usr/bin/env python3
"""Fold rocprofv3 --pmc SQ counter passes (counter_collection.csv files) into one per-dispatch table of the last optimiser
step: python tools/pmc_sq.py out.csv pass1.csv [pass2.csv ...].  Columns are whatever counters the passes hold."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:60]


def last_step(path):
    per, meta = defaultdict(dict), {}
    for r in csv.DictReader(open(path)):
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        meta[d] = (r["Kernel_Name"], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
    ids = sorted(per)
    adam = [i for i, d in enumerate(ids) if "adam_kernel" in meta[d][0]]
    for a, b in reversed(list(zip(adam[:-1], adam[1:]))):
        if any("wino" in meta[ids[k]][0] for k in range(a + 1, b)):
            return [(short(meta[d][0]), meta[d][1], per[d]) for d in ids[a + 1:b + 1]]
    raise SystemExit("no step in " + path)


passes = [last_step(p) for p in sys.argv[2:]]
n = len(passes[0])
assert all(len(p) == n and all(a[0] == b[0] for a, b in zip(p, passes[0])) for p in passes)
cols = []
for p in passes:
    for k in p[0][2]:
        if k not in cols:
            cols.append(k)
with open(sys.argv[1], "w") as fh:
    fh.write("kernel,workgroups," + ",".join(cols) + "\n")
    for i in range(n):
        vals = {}
        for p in passes:
            vals.update(p[i][2])
        fh.write("%s,%d,%s\n" % (passes[0][i][0].replace(",", ";"), passes[0][i][1], ",".join("%.0f" % vals.get(c, 0) for c in cols)))
print("wrote", sys.argv[1], n, "dispatches", cols)
