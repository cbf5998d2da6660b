#!/usr/bin/env python3
"""Time the render kernels of config 2 (8 views, 512^2, cow) in isolation: shade backward with/without d/d(bary)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2d-to-3d-style-transfer_amd")]
import numpy as np, torch
from st3d import ops
from st3d.render import look_at_view_transform
cow = np.load(os.path.join(ROOT, "tests/golden/assets_cow_mesh.npz"))
dev = torch.device("cuda:0")
B, S = 8, 512
g = torch.Generator().manual_seed(0)
elev = torch.acos(torch.rand(B, generator=g) * 2 - 1) * 180 / torch.pi - 90
azim = torch.rand(B, generator=g) * 360 - 180
Rt, Tt = look_at_view_transform(dist=2.10, elev=elev, azim=azim, at=((0, 0.10, 0.25),))
R, T = Rt.numpy(), Tt.numpy()
verts = torch.from_numpy(cow["verts"]).to(dev); faces = torch.from_numpy(cow["faces"]).to(dev)
uvs = torch.from_numpy(cow["verts_uvs"]).to(dev); fuv = torch.from_numpy(cow["faces_uvs"]).to(dev)
tex = torch.rand(S, S, 3, device=dev)
ndc = ops.project_verts(verts, torch.from_numpy(R).to(dev), torch.from_numpy(T).to(dev))
frag = ops.raster_fwd(ndc, faces, S)
grad = torch.randn(B, 3, S, S, device=dev)
print("coverage %.3f" % float((frag[0] >= 0).float().mean()))


def timed(fn, name, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us")


gt = torch.zeros(S, S, 3, device=dev)
timed(lambda: ops.raster_fwd(ndc, faces, S), "raster_fwd (face setup + tiles)")
timed(lambda: ops.shade_fwd(frag, uvs, fuv, tex), "shade_fwd")
timed(lambda: ops.shade_bwd(grad, frag, uvs, fuv, tex, grad_texture=gt), "shade_bwd texture only")
timed(lambda: ops.shade_bwd(grad, frag, uvs, fuv, tex, grad_texture=gt, want_bary=True), "shade_bwd texture + d/d bary")
timed(lambda: ops.shade_bwd(grad, frag, uvs, fuv, tex, want_texture=False, want_bary=True), "shade_bwd d/d bary only")
gb = ops.shade_bwd(grad, frag, uvs, fuv, tex, want_texture=False, want_bary=True)
gb = gb[-1] if isinstance(gb, tuple) else gb
timed(lambda: ops.raster_bwd(gb, frag[0], ndc, faces), "raster_bwd")
# pure sweep cost: the same mesh pushed off screen (no tile is hit)
ndc_off = ndc.clone(); ndc_off[..., 0] += 50.0
timed(lambda: ops.raster_fwd(ndc_off, faces, S), "raster_fwd, mesh off screen (sweep only)")
# tiny mesh: a 10th of the size (few tiles hit, short lists)
ndc_small = ndc.clone(); ndc_small[..., :2] *= 0.1
timed(lambda: ops.raster_fwd(ndc_small, faces, S), "raster_fwd, mesh scaled 0.1")
