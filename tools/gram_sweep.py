#!/usr/bin/env python3
"""Sweep the split-K width of the Gram forward (ST3D_GRAM_TARGET_WGS) per style layer of config 2."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "2d-to-3d-style-transfer_amd")]
    import torch
    from st3d import ops
    dev = torch.device("cuda:0")
    out = {}
    for C, H in ((64, 512), (128, 256), (256, 128), (512, 64), (512, 32)):
        f = torch.rand(8, C, H, H, device=dev)
        for _ in range(3):
            ops.gram_fwd(f)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gram_fwd(f)
        e1.record(); torch.cuda.synchronize()
        out[f"{C}x{H}"] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    print(json.dumps(out))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "nsplit":       # the split count itself (ST3D_GRAM_NSPLIT), for grid-quantisation effects
    for t in (0, 4, 5, 6, 8, 10, 12, 13, 16, 19, 21, 24, 26, 32, 42, 43, 48, 64):
        env = dict(os.environ)
        if t:
            env["ST3D_GRAM_NSPLIT"] = str(t)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print("nsplit", t or "default", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
    sys.exit(0)
for t in (0, 256, 512, 1024, 2048, 4096):
    env = dict(os.environ)
    if t:
        env["ST3D_GRAM_TARGET_WGS"] = str(t)
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    print("target_wgs", t or "default(1024)", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
