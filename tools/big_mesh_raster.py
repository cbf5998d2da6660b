#!/usr/bin/env python3
"""Rasteriser / scatter timings on a high-polygon mesh: the cow subdivided twice (93 696 faces, the stand-in for the
bunny of BASELINE config 3), 16 views at 1024^2, next to the cow itself."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2d-to-3d-style-transfer_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import _scenes as SC
from st3d import ops
import utils as U
dev = torch.device("cuda:0"); U.device = dev
cow = SC.load_asset("cow")
v, f, uv, fuv = cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"]
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for level in range(3):
    for B, S in ((8, 512), (16, 1024)):
        cams = U.build_random_cameras(B, generator=torch.Generator().manual_seed(0))
        vd = torch.from_numpy(v).to(dev); fd = torch.from_numpy(f.astype(np.int32)).to(dev)
        ndc = ops.project_verts(vd, cams.R.to(dev), cams.T.to(dev))
        ms = t(lambda: ops.raster_fwd(ndc, fd, S))
        print(f"faces {f.shape[0]:6d}  {B:2d} views {S:4d}^2: raster_fwd {ms:7.3f} ms", flush=True)
    v, f, uv, fuv = SC.subdivide(v, f, uv, fuv)
