#!/usr/bin/env python3
"""Fold one rocprofv3 --pmc pass of MFMA counters into the per-dispatch table that stands under bench.py's roofline:

    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
        --output-format csv -d gpurun_out/pmc_mfma -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_mfma.py <...counter_collection.csv> <out prefix> [bench line json of an un-profiled run]

Takes the dispatches of the LAST optimiser step (between the last two adam_kernel launches that have conv launches
between them).  Per dispatch:
  mfma_flops   = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512            (the counter is "add or mul ops / 512", summed over SEs/XCCs)
  busy_cycles  = SQ_VALU_MFMA_BUSY_CYCLES                       (summed over all SIMDs of the chip)
  gui_cycles   = GRBM_GUI_ACTIVE / n_xcc                        (the rocprofv3 row is the SUM over the 8 XCCs)
  mfma_util    = busy_cycles / (gui_cycles * 1024 SIMDs)        (rocprofv3's MfmaUtil expression with SIMD_NUM = 256 CUs x 4)
The counted flops are compared with the ANALYTIC issued count bench.py prices (16/36 of the direct convolution for the
F(2x2,3x3) launches, 36/144 for the F(4x4,3x3) ones, padded tiles included in what is counted), and -- given an un-profiled bench line with --layers --
with the HIP-event time of the same launches: counted flops / un-profiled time / 157.3 TF/s must tell the same story as
`roofline.frac`.
"""
import csv
import json
import re
import sys
from collections import defaultdict

N_XCC, N_SIMD, PEAK = 8, 1024, 157.3e12


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:80]


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    line = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
    per = defaultdict(dict)
    meta = {}
    for r in csv.DictReader(open(path)):
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        meta[d] = (r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    ids = sorted(per)
    adam = [i for i, d in enumerate(ids) if "adam_kernel" in meta[d][0]]
    step = None
    for a, b in reversed(list(zip(adam[:-1], adam[1:]))):
        if any("wino" in meta[ids[k]][0] for k in range(a + 1, b)):
            step = ids[a + 1:b + 1]
            break
    if step is None:
        raise SystemExit("no optimiser step found in " + path)
    rows, fam = [], defaultdict(lambda: [0.0, 0.0, 0.0, 0, 0.0])
    for d in step:
        name, grid, wg, ns = meta[d]
        c = per[d]
        flops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / N_XCC
        util = busy / (gui * N_SIMD) if gui else 0.0
        k = short(name)
        rows.append((k, grid // wg, flops, busy, gui, util, ns))
        f = fam["wino43_kernel" if "wino43_kernel" in k else "wino4_kernel" if "wino4_kernel" in k else k.split("<")[0]]
        f[0] += flops; f[1] += busy; f[2] += gui; f[3] += 1; f[4] += ns
    with open(prefix + "_per_dispatch.csv", "w") as fh:
        fh.write("kernel,workgroups,mfma_flops_counted,mfma_busy_cycles,gui_cycles_per_xcc,mfma_util,duration_ns_under_pmc\n")
        for k, wgs, flops, busy, gui, util, ns in rows:
            fh.write("%s,%d,%.0f,%.0f,%.0f,%.4f,%d\n" % (k.replace(",", ";"), wgs, flops, busy, gui, util, ns))
    out = {"source": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python "
                     "bench.py --steps 2 --warmup 1 --no-cpu-baseline; dispatches of the last optimiser step",
           "dispatches_per_step": len(rows), "families": {}}
    for k, (flops, busy, gui, n, ns) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        if flops == 0 and busy == 0:
            continue
        out["families"][k] = {"launches": n, "mfma_flops_counted": flops, "mfma_busy_cycles": busy,
                              "gui_cycles_per_xcc": gui, "mfma_util": round(busy / (gui * N_SIMD), 4) if gui else None,
                              # a v_mfma_f32_32x32x2_f32 is 4096 flops and occupies its SIMD's matrix pipe for 64 cycles (16 passes);
                              # a v_mfma_f32_16x16x4_f32 (wino43_kernel) is 2048 flops in 32 cycles: the same 64 per 4096
                              "busy_cycles_per_4096_flops": round(busy / (flops / 4096.0), 2) if flops else None,
                              "duration_ms_under_pmc": round(ns * 1e-6, 4)}
    w4, w6 = out["families"].get("wino4_kernel"), out["families"].get("wino43_kernel")
    if w4 or w6:
        # the default plan at 8 x 512^2: F(4x4,3x3) (36/144 of the direct count) where Cin >= 64 and W % 32 == 0 -- conv1_2 ..
        # conv5_1 -- and F(2x2,3x3) (16/36) for whatever is left; forward + input gradient each.  Without the wino43 family in
        # the trace (ST3D_WINO43=0) every layer is priced as F(2x2,3x3).
        S, B = 512, 8
        convs = [(64, 64, 1), (64, 128, 2), (128, 128, 2), (128, 256, 4), (256, 256, 4), (256, 256, 4), (256, 256, 4), (256, 512, 8),
                 (512, 512, 8), (512, 512, 8), (512, 512, 8), (512, 512, 16)]
        direct = lambda ci, co, d: 2.0 * 9 * ci * co * (S // d) ** 2 * B
        on43 = lambda ci, co, d: w6 is not None and ci >= 64 and ((S // d) % 64 == 0 or (S // d) % 32 == 0)
        a4 = sum(direct(*c) for c in convs if not on43(*c)) * 2 * 16.0 / 36.0
        a6 = sum(direct(*c) for c in convs if on43(*c)) * 2 * 36.0 / 144.0
        for w, an in ((w4, a4), (w6, a6)):
            if w:
                w["mfma_flops_analytic_issued"] = an
                w["counted_over_analytic"] = round(w["mfma_flops_counted"] / an, 4)
                w["avg_flops_per_launch_counted"] = w["mfma_flops_counted"] / w["launches"]
        both = {"launches": sum(w["launches"] for w in (w4, w6) if w),
                "mfma_flops_counted": sum(w["mfma_flops_counted"] for w in (w4, w6) if w),
                "mfma_flops_analytic_issued": a4 + a6,
                "mfma_busy_cycles": sum(w["mfma_busy_cycles"] for w in (w4, w6) if w),
                "gui_cycles_per_xcc": sum(w["gui_cycles_per_xcc"] for w in (w4, w6) if w),
                "duration_ms_under_pmc": round(sum(w["duration_ms_under_pmc"] for w in (w4, w6) if w), 4)}
        both["counted_over_analytic"] = round(both["mfma_flops_counted"] / both["mfma_flops_analytic_issued"], 4)
        both["mfma_util"] = round(both["mfma_busy_cycles"] / (both["gui_cycles_per_xcc"] * N_SIMD), 4)
        if line and line.get("roofline", {}).get("ms_per_step"):
            ms = line["roofline"]["ms_per_step"]
            both["unprofiled_ms_per_step_hip_events"] = ms
            both["counted_tflops_over_unprofiled_time"] = round(both["mfma_flops_counted"] / (ms * 1e-3) / 1e12, 2)
            both["counted_frac_of_fp32_mfma_peak"] = round(both["mfma_flops_counted"] / (ms * 1e-3) / PEAK, 4)
            both["bench_line_roofline_frac"] = line["roofline"]["frac"]
        out["winograd_convs"] = both
    tot = sum(v["mfma_flops_counted"] for v in out["families"].values())
    out["mfma_flops_counted_per_step"] = tot
    json.dump(out, open(prefix + ".json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
