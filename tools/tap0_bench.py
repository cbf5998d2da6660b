#!/usr/bin/env python3
"""conv1_1 backward fused with the relu1_1 style gradient (st3d_conv1_bwd) at config 2's shape, against the two launches
it replaces.  ST3D_TAP0_J=1|2 selects the pixels per lane."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2d-to-3d-style-transfer_amd")]
import torch
from st3d import ops
dev = torch.device("cuda:0")
N, S = 8, 512
g = torch.Generator().manual_seed(0)
act = torch.relu(torch.randn(N, 64, S, S, generator=g)).to(dev)
gy = torch.randn(N, 64, S, S, generator=g).to(dev)
D = torch.randn(N, 64, 64, generator=g).to(dev)
w = torch.randn(64, 3, 3, 3, generator=g).to(dev)
_, wd = ops.conv3x3_pack(w)


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


acc = gy.clone()
print("J", os.environ.get("ST3D_TAP0_J", "2"), "fused %.3f ms" % t(lambda: ops.conv1_bwd(gy, act, D, 0.3, wd)),
      "| gram_bwd(acc) %.3f + dgrad %.3f ms" % (t(lambda: ops.gram_bwd(D, act, 0.3, out=acc)), t(lambda: ops.conv3x3_dgrad(acc, act, wd, 3))))
