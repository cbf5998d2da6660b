"""Pins the oracle's hand-derived vertex-path backward (oracle/raster_ref.c: ref_raster_bwd,
ref_project_verts_bwd, the UV gradient of ref_shade_bwd) against torch AUTOGRAD of an independent
fp64 restatement of the forward (fixed coverage: pix_to_face from the rasteriser).  CPU only.
PARITY UNPINNED w.r.t. PyTorch3D (absent): this checks the derivation, not upstream's code."""
import math

import numpy as np
import torch

from oracle import render_ref as rr


def _forward_f64(verts, R, T, faces, p2f, uvs, fuv, tex, zbuf, dists, S):
    """project -> perspective-correct barycentrics of each pixel's face -> UV -> bilinear sample
    (flipped map, align_corners, border) -> K=1 blend.  All differentiable in verts and tex."""
    s = 1.0 / math.tan(math.radians(30.0))
    view = verts @ R + T
    ndc = torch.stack([s * view[:, 0] / view[:, 2], s * view[:, 1] / view[:, 2], view[:, 2]], dim=1)
    ys, xs = torch.nonzero(p2f >= 0, as_tuple=True)
    f = p2f[ys, xs].long()
    px = 1.0 - (2.0 * xs.double() + 1.0) / S
    py = 1.0 - (2.0 * ys.double() + 1.0) / S
    v = [ndc[faces[f, i]] for i in range(3)]

    def E(ax, ay, bx, by, qx, qy):
        return (qx - ax) * (by - ay) - (qy - ay) * (bx - ax)
    A = E(v[0][:, 0], v[0][:, 1], v[1][:, 0], v[1][:, 1], v[2][:, 0], v[2][:, 1]) + 1e-8
    w0 = E(v[1][:, 0], v[1][:, 1], v[2][:, 0], v[2][:, 1], px, py) / A
    w1 = E(v[2][:, 0], v[2][:, 1], v[0][:, 0], v[0][:, 1], px, py) / A
    w2 = E(v[0][:, 0], v[0][:, 1], v[1][:, 0], v[1][:, 1], px, py) / A
    z0, z1, z2 = v[0][:, 2], v[1][:, 2], v[2][:, 2]
    t0, t1, t2 = w0 * z1 * z2, z0 * w1 * z2, z0 * z1 * w2
    den = t0 + t1 + t2
    b = torch.stack([t0 / den, t1 / den, t2 / den], dim=1)
    uv = sum(b[:, i:i + 1] * uvs[fuv[f, i]] for i in range(3))
    Tn = tex.shape[0]
    ix = (uv[:, 0] * (Tn - 1)).clamp(0, Tn - 1)
    iy = (uv[:, 1] * (Tn - 1)).clamp(0, Tn - 1)
    x0 = ix.detach().floor().long().clamp(max=Tn - 1)
    y0 = iy.detach().floor().long().clamp(max=Tn - 1)
    x1, y1 = (x0 + 1).clamp(max=Tn - 1), (y0 + 1).clamp(max=Tn - 1)
    wx1, wy1 = (ix - x0)[:, None], (iy - y0)[:, None]
    r0, r1 = (Tn - 1) - y0, (Tn - 1) - y1                     # flipped map -> original rows
    texel = (tex[r0, x0] * (1 - wx1) * (1 - wy1) + tex[r0, x1] * wx1 * (1 - wy1)
             + tex[r1, x0] * (1 - wx1) * wy1 + tex[r1, x1] * wx1 * wy1)
    prob = 1.0 / (1.0 + torch.exp(dists[ys, xs].double() / 1e-4))
    delta = 1e-10
    rgb = (prob[:, None] * texel + delta) / (prob[:, None] + delta)
    return rgb, ys, xs, b


def test_vertex_and_texture_gradients_match_fp64_autograd(cow):
    S, Tn = 64, 24
    rng = np.random.default_rng(5)
    tex = rng.random((Tn, Tn, 3)).astype(np.float32)
    R, T = rr.look_at_view_transform(2.10, [15.0, -30.0], [35.0, 200.0], at=(0, 0.10, 0.25))
    verts, faces, uvs, fuv = cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"]
    imgs, _, frags = rr.render_views(verts, faces, uvs, fuv, tex, R, T, S, 4)
    g = rng.standard_normal(imgs.shape).astype(np.float32)
    gtex, gverts = rr.render_bwd_views(g, frags, verts, faces, uvs, fuv, tex, R, T)

    vt = torch.from_numpy(verts).double().requires_grad_(True)
    tt = torch.from_numpy(tex).double().requires_grad_(True)
    total = 0
    for b in range(2):
        p2f, zbuf, bary, dists = (torch.from_numpy(a) for a in frags[b])
        rgb, ys, xs, bb = _forward_f64(vt, torch.from_numpy(R[b]).double(), torch.from_numpy(T[b]).double(),
                                       torch.from_numpy(faces).long(), p2f, torch.from_numpy(uvs).double(),
                                       torch.from_numpy(fuv).long(), tt, zbuf, dists, S)
        # the restated forward reproduces the oracle's pixels and barycentrics
        np.testing.assert_allclose(rgb.detach().numpy().T, imgs[b][:, ys, xs], atol=5e-5)
        np.testing.assert_allclose(bb.detach().numpy(), bary.numpy()[ys, xs], atol=2e-4)   # fp32 barycentrics of thin faces
        total = total + (rgb.t() * torch.from_numpy(g[b]).double()[:, ys, xs]).sum()
    total.backward()
    rel_t = np.linalg.norm(gtex - tt.grad.numpy()) / np.linalg.norm(tt.grad.numpy())
    rel_v = np.linalg.norm(gverts - vt.grad.numpy()) / np.linalg.norm(vt.grad.numpy())
    assert rel_t <= 1e-4, rel_t
    assert rel_v <= 2e-3, rel_v           # float32 footprint/UV arithmetic in the C oracle vs fp64 here
    assert np.abs(gverts).max() > 0


def test_projection_backward_matches_autograd():
    rng = np.random.default_rng(0)
    verts = rng.standard_normal((50, 3)).astype(np.float32) * 0.3
    R, T = rr.look_at_view_transform(2.5, [20.0], [50.0])
    gn = rng.standard_normal((50, 3))
    gv = rr.project_verts_bwd(verts, R[0], T[0], gn)
    vt = torch.from_numpy(verts).double().requires_grad_(True)
    view = vt @ torch.from_numpy(R[0]).double() + torch.from_numpy(T[0]).double()
    s = float(rr.INV_TAN_HALF_FOV)
    ndc = torch.stack([s * view[:, 0] / view[:, 2], s * view[:, 1] / view[:, 2], view[:, 2]], dim=1)
    (ndc * torch.from_numpy(gn)).sum().backward()
    np.testing.assert_allclose(gv, vt.grad.numpy(), rtol=1e-5, atol=1e-7)
