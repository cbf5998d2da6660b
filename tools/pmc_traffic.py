#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into the per-step
HBM traffic table bench.py reads:  python tools/pmc_traffic.py <fetch counter_collection.csv> <write ...csv> <out prefix>
Takes the dispatches of the LAST optimiser step (between the last two adam_kernel launches)."""
import csv
import json
import re
import sys


def last_step(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    # the last pair of Adam launches with a whole optimiser step (conv launches) between them: bench.py times a few
    # kernels on their own after the steps, Adam among them
    for a, b in reversed(list(zip(adam[:-1], adam[1:]))):
        if any("wino" in r["Kernel_Name"] or "conv3x3_kernel" in r["Kernel_Name"] for r in rows[a + 1:b]):
            return rows[a + 1:b + 1]
    raise SystemExit("no optimiser step found in " + path)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:80]


fetch = last_step(sys.argv[1], "FETCH_SIZE")
write = last_step(sys.argv[2], "WRITE_SIZE")
assert len(fetch) == len(write) and all(short(a["Kernel_Name"]) == short(b["Kernel_Name"]) for a, b in zip(fetch, write))
prefix = sys.argv[3]
tot_f = tot_w = wino_f = wino_w = 0.0
nwino = 0
with open(prefix + "_per_dispatch.csv", "w") as fh:
    fh.write("kernel,grid_threads,FETCH_SIZE_KB_raw,WRITE_SIZE_KB\n")
    for a, b in zip(fetch, write):
        f, w = float(a["Counter_Value"]), float(b["Counter_Value"])
        fh.write("%s,%s,%.0f,%.0f\n" % (short(a["Kernel_Name"]).replace(",", ";"), a.get("Grid_Size", ""), f, w))
        tot_f += f * 1024; tot_w += w * 1024
        if "wino" in a["Kernel_Name"] and "pack" not in a["Kernel_Name"]:
            wino_f += f * 1024; wino_w += w * 1024; nwino += 1
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 "
              "--no-cpu-baseline; dispatches of the last step",
    "fetch_bytes_raw_per_step": tot_f, "write_bytes_per_step": tot_w,
    "fetch_bytes_corrected_upper_per_step": 2 * tot_f,
    "note": "gfx950: FETCH_SIZE reports exactly half of a 16-B/lane coalesced stream (MI355X_MICROARCH.md HBM); these "
            "kernels mix 4/8/16-B loads, so the true read volume lies between the raw value and twice it; WRITE_SIZE is exact.",
    "wino_kernel_launches_per_step": nwino, "wino_fetch_bytes_raw_per_step": wino_f, "wino_write_bytes_per_step": wino_w,
    "dispatches_per_step": len(fetch),
}
json.dump(out, open(prefix + ".json", "w"), indent=1)
print(json.dumps(out, indent=1))
