#!/usr/bin/env python3
"""Gram forward / backward per style layer of config 2 (8 views, 512^2), one line per run: use with an env switch
(ST3D_GRAM_FAST=0|1, ST3D_GRAM_BWD_MT=...) for same-box A/B comparisons."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2d-to-3d-style-transfer_amd")]
import torch
from st3d import ops, _lib
dev = torch.device("cuda:0")
out = []
for C, H in ((64, 512), (128, 256), (256, 128), (512, 64), (512, 32)):
    f = torch.rand(8, C, H, H, device=dev).sub_(0.5).relu_()
    D = torch.randn(8, C, C, device=dev)
    acc = torch.zeros_like(f)
    ws = torch.empty((_lib.load().st3d_gram_workspace_bytes(8, C, H * H) // 4,), device=dev)
    G = torch.empty((8, C, C), device=dev)
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    fw = t(lambda: _lib.call("st3d_gram_fwd", _lib.dptr(f), 8, C, H * H, _lib.dptr(ws), ws.numel() * 4, _lib.dptr(G), _lib.stream_ptr()))
    bw = t(lambda: ops.gram_bwd(D, f, 0.5, out=acc, gated=C % 32 == 0))
    out.append(f"{C}x{H}: fwd {fw:6.1f} bwd {bw:6.1f}")
print(" | ".join(out))
