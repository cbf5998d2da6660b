#!/usr/bin/env python3
"""Gram backward per style layer of config 2 with 128-row (default) vs 64-row tiles (ST3D_GRAM_BWD_MT=1)."""
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
        D = torch.randn(8, C, C, device=dev)
        acc = torch.zeros_like(f)
        for _ in range(3):
            ops.gram_bwd(D, f, 0.5, out=acc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gram_bwd(D, f, 0.5, out=acc)
        e1.record(); torch.cuda.synchronize()
        out[f"{C}x{H}"] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
        if os.environ.get("SWEEP_EXTRA"):
            e0.record()
            for _ in range(20):
                ops.gram_bwd(D, f, 0.5)          # no accumulate (allocates the output: includes torch.empty)
            e1.record(); torch.cuda.synchronize()
            out[f"{C}x{H}_store"] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
            e0.record()
            for _ in range(20):
                acc.add_(f)
            e1.record(); torch.cuda.synchronize()
            out[f"{C}x{H}_axpy"] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    print(json.dumps(out))
    sys.exit(0)
for mt in ("", "1", "2", "3", "4"):
    env = dict(os.environ)
    if mt:
        env["ST3D_GRAM_BWD_MT"] = mt
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    print("MT", mt or "default", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
