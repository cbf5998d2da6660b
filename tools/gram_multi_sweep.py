#!/usr/bin/env python3
"""Time st3d_gram_fwd_multi on the five style-layer shapes of config 2 (8 views, 512^2) for split-scale choices
(ST3D_GRAM_MULTI_SCALES = scale per C in {64,128,256,512}; read per call by gram.hip)."""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "2d-to-3d-style-transfer_amd"))
import torch  # noqa: E402
from st3d import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
g = torch.Generator().manual_seed(0)
feats = [torch.randn((B, C, S // d, S // d), generator=g).clamp_min(0).to(dev) for C, d in [(64, 1), (128, 2), (256, 4), (512, 8), (512, 16)]]


def t(reps=30):
    for _ in range(3):
        ops.gram_fwd_multi(feats)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.gram_fwd_multi(feats)
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


res = []
for combo in itertools.product([1, 2, 4], [1, 2, 4], [1, 2, 4], [1, 2, 4]):
    os.environ["ST3D_GRAM_MULTI_SCALES"] = ",".join(str(c) for c in combo)
    res.append((t(), combo))
res.sort()
for us, combo in res[:12]:
    print("%7.1f us  scales %s" % (us, combo))
print("...")
for us, combo in res[-3:]:
    print("%7.1f us  scales %s" % (us, combo))
for deal in (0, 1):
    os.environ["ST3D_GRAM_MULTI_DEAL"] = str(deal)
os.environ.pop("ST3D_GRAM_MULTI_SCALES")
print("default: %.1f us" % t())
