#!/usr/bin/env python3
"""Per-layer timing of the Winograd conv launches of one config-2 step (8 views, 512^2): forward (with the fused pool
where the plan fuses it), gated input-gradient, input-gradient with fused unpool.  Run once per kernel variant:
    ST3D_WINO_VARIANT=8 python tools/wino_layers.py ; ST3D_WINO_VARIANT=4 python tools/wino_layers.py
Prints ms and the fraction of the fp32-MFMA peak (issued flops = 16/36 of the direct count)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "2d-to-3d-style-transfer_amd"))
import torch
from st3d import ops
B = int(os.environ.get("B", "8")); S = int(os.environ.get("S", "512"))
dev = torch.device("cuda:0")
# (name, Cin, Cout, divisor, pooled_after, input_is_pool_output)
LAYERS = [("conv1_2", 64, 64, 1, True, False), ("conv2_1", 64, 128, 2, False, True), ("conv2_2", 128, 128, 2, True, False),
          ("conv3_1", 128, 256, 4, False, True), ("conv3_2", 256, 256, 4, False, False), ("conv3_4", 256, 256, 4, True, False),
          ("conv4_1", 256, 512, 8, False, True), ("conv4_2", 512, 512, 8, False, False), ("conv4_4", 512, 512, 8, True, False),
          ("conv5_1", 512, 512, 16, False, True)]
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
rows = []
tot = 0.0
for name, Cin, Cout, d, pooled, in_pool in LAYERS:
    H = S // d
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, H, device=dev).relu_()
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.randn(Cout, device=dev) * 0.1
    uf, ud = ops.wino_pack(w)
    gf = 2 * 9 * Cin * Cout * H * H * B * (16 / 36) / 1e9
    if pooled:
        tf = timeit(lambda: ops.wino_fwd(x, uf, b, Cout, relu=True, pool=True, keep_full=False))
    else:
        tf = timeit(lambda: ops.wino_fwd(x, uf, b, Cout, relu=True))
    gy = torch.randn(B, Cout, H, H, device=dev)        # gradient w.r.t. this layer's output ...
    y = torch.randn(B, Cout, H, H, device=dev)         # ... gated by its (post-ReLU) activation
    td = timeit(lambda: ops.wino_dgrad(gy, y, ud, Cin))
    rows.append((name, tf, gf / tf / 157.3, td, gf / td / 157.3))
    tot += tf + td
    print(f"{name:8s} fwd {tf:7.4f} ms {gf/tf/157.3:6.3f} | dgrad {td:7.4f} ms {gf/td/157.3:6.3f}", flush=True)
# input-gradients with the fused unpool: the layer BEFORE each pool (conv1_2, conv2_2, conv3_4, conv4_4) receives the
# gradient at pooled resolution H and writes its input-gradient at 2H
for Ci, Co, d in ((64, 64, 2), (128, 128, 4), (256, 256, 8), (512, 512, 16)):
    H = S // d
    gp = torch.randn(B, Co, H, H, device=dev)
    pd = torch.randn(B, Co, H, H, device=dev)
    idx = torch.randint(0, 4, (B, Co, H, H), device=dev, dtype=torch.uint8)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * (2.0 / (Ci * 9)) ** 0.5
    uf, ud = ops.wino_pack(w)
    gf = 2 * 9 * Ci * Co * (2 * H) ** 2 * B * (16 / 36) / 1e9
    t = timeit(lambda: ops.wino_dgrad_unpool(gp, idx, pd, ud, Ci))
    print(f"dgrad_unpool {Co}->{Ci} at {2*H}: {t:7.4f} ms {gf/t/157.3:6.3f}", flush=True)
print("variant", os.environ.get("ST3D_WINO_VARIANT", "default"), "sum fwd+dgrad of listed layers: %.3f ms" % tot)
