"""oracle/soft_ref.py -- TEST INFRASTRUCTURE ONLY.

torch restatement (any dtype; fp64 + autograd in the gradient tests) of the GENERAL soft
rendering path the reference does not configure (it fixes blur_radius=0, faces_per_pixel=1,
first_approach.py:107) but BASELINE.json's north star names: K faces per pixel, blur radius,
barycentric clipping, ``softmax_rgb_blend`` over K with sigma / gamma / background.

PARITY UNPINNED: PyTorch3D is absent; formulas per SURVEY.md A.2-A.4.  ``soft_geometry`` recomputes
the differentiable fragment quantities (clipped perspective-correct barycentrics, depth, signed
squared edge distance) for a FIXED pixel->faces assignment (pix_to_face from the C oracle), so
autograd gives the exact gradient of the rendered image w.r.t. vertices and texture that the HIP
backward kernels are checked against.
"""
import math

import torch

K_EPS = 1e-8


def _edge(px, py, ax, ay, bx, by):
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def _pld2(px, py, ax, ay, bx, by):
    bax, bay = bx - ax, by - ay
    l2 = bax * bax + bay * bay
    t = ((bax * (px - ax) + bay * (py - ay)) / l2.clamp_min(1e-30)).clamp(0, 1)
    # PyTorch3D treats the projection parameter as a constant in the backward (envelope theorem:
    # exact for the closest point); detach reproduces that and keeps clamped cases exact
    t = t.detach()
    qx, qy = ax + t * bax, ay + t * bay
    d = (qx - px) ** 2 + (qy - py) ** 2
    deg = l2 <= K_EPS
    return torch.where(deg, (px - bx) ** 2 + (py - by) ** 2, d)


def project(verts, R, T):
    s = 1.0 / math.tan(math.radians(30.0))
    view = verts @ R + T
    return torch.stack([s * view[:, 0] / view[:, 2], s * view[:, 1] / view[:, 2], view[:, 2]], dim=1)


def soft_geometry(ndc, faces, p2f, S, clip_bary, perspective_correct=True):
    """ndc (V,3), faces (F,3) long, p2f (S,S,K) long -> bary (S,S,K,3), zbuf, dists, mask."""
    K = p2f.shape[-1]
    mask = p2f >= 0
    f = p2f.clamp_min(0)
    ys = torch.arange(S, dtype=ndc.dtype).view(S, 1, 1).expand(S, S, K)
    xs = torch.arange(S, dtype=ndc.dtype).view(1, S, 1).expand(S, S, K)
    px = 1.0 - (2.0 * xs + 1.0) / S
    py = 1.0 - (2.0 * ys + 1.0) / S
    v = [ndc[faces[f, i]] for i in range(3)]                       # each (S,S,K,3)
    x0, y0, z0 = v[0][..., 0], v[0][..., 1], v[0][..., 2]
    x1, y1, z1 = v[1][..., 0], v[1][..., 1], v[1][..., 2]
    x2, y2, z2 = v[2][..., 0], v[2][..., 1], v[2][..., 2]
    area = _edge(x2, y2, x0, y0, x1, y1) + K_EPS
    w0 = _edge(px, py, x1, y1, x2, y2) / area
    w1 = _edge(px, py, x2, y2, x0, y0) / area
    w2 = _edge(px, py, x0, y0, x1, y1) / area
    t0, t1, t2 = w0 * z1 * z2, z0 * w1 * z2, z0 * z1 * w2
    den = (t0 + t1 + t2).clamp_min(K_EPS)
    b = torch.stack([t0 / den, t1 / den, t2 / den], dim=-1) if perspective_correct else torch.stack([w0, w1, w2], dim=-1)
    inside = (b > 0).all(dim=-1)
    if clip_bary:
        c = b.clamp(0, 1)
        b_out = c / c.sum(dim=-1, keepdim=True).clamp_min(K_EPS)
    else:
        b_out = b
    pz = b_out[..., 0] * z0 + b_out[..., 1] * z1 + b_out[..., 2] * z2
    d = torch.minimum(torch.minimum(_pld2(px, py, x0, y0, x1, y1), _pld2(px, py, x1, y1, x2, y2)), _pld2(px, py, x2, y2, x0, y0))
    sd = torch.where(inside, -d, d)
    return b_out, pz, sd, mask


def sample_texture(bary, p2f, verts_uvs, faces_uvs, tex):
    """(S,S,K,3) texels: TexturesUV.sample_textures (flipped map, align_corners, border)."""
    f = p2f.clamp_min(0)
    uv = sum(bary[..., i:i + 1] * verts_uvs[faces_uvs[f, i]] for i in range(3))
    T = tex.shape[0]
    ix = (uv[..., 0] * (T - 1)).clamp(0, T - 1)
    iy = (uv[..., 1] * (T - 1)).clamp(0, T - 1)
    x0 = ix.detach().floor().long().clamp(max=T - 1)
    y0 = iy.detach().floor().long().clamp(max=T - 1)
    x1, y1 = (x0 + 1).clamp(max=T - 1), (y0 + 1).clamp(max=T - 1)
    wx1, wy1 = (ix - x0).unsqueeze(-1), (iy - y0).unsqueeze(-1)
    r0, r1 = (T - 1) - y0, (T - 1) - y1
    return (tex[r0, x0] * (1 - wx1) * (1 - wy1) + tex[r0, x1] * wx1 * (1 - wy1)
            + tex[r1, x0] * (1 - wx1) * wy1 + tex[r1, x1] * wx1 * wy1)


def softmax_rgb_blend(colors, zbuf, dists, mask, sigma=1e-4, gamma=1e-4, background=(1.0, 1.0, 1.0), znear=1.0, zfar=100.0):
    """PyTorch3D blending.softmax_rgb_blend: colors (S,S,K,3) -> rgb (S,S,3), alpha (S,S)."""
    eps = 1e-10
    m = mask.to(colors.dtype)
    prob = torch.sigmoid(-dists / sigma) * m
    alpha = torch.prod(1.0 - prob, dim=-1)
    z_inv = (zfar - zbuf) / (zfar - znear) * m
    z_max = torch.max(z_inv, dim=-1).values.unsqueeze(-1).clamp(min=eps)
    wnum = prob * torch.exp((z_inv - z_max) / gamma)
    delta = torch.exp((eps - z_max) / gamma).clamp(min=eps)
    denom = wnum.sum(dim=-1, keepdim=True) + delta
    bg = torch.tensor(background, dtype=colors.dtype)
    rgb = ((wnum.unsqueeze(-1) * colors).sum(dim=-2) + delta * bg) / denom
    return rgb, 1.0 - alpha


def soft_render(verts, R, T, faces, p2f, verts_uvs, faces_uvs, tex, S, clip_bary, sigma, gamma, background=(1.0, 1.0, 1.0),
                perspective_correct=True):
    """One view, fixed coverage: -> rgb (3,S,S), alpha (S,S); differentiable in verts and tex."""
    ndc = project(verts, R, T)
    bary, pz, sd, mask = soft_geometry(ndc, faces, p2f, S, clip_bary, perspective_correct)
    colors = sample_texture(bary, p2f, verts_uvs, faces_uvs, tex)
    rgb, alpha = softmax_rgb_blend(colors, pz, sd, mask, sigma, gamma, background)
    return rgb.permute(2, 0, 1), alpha


# ---------------------------------------------------------------------------------------------- near-plane clipping
def clip_face(v, z_clip, perspective_correct=True):
    """torch restatement of oracle/raster_ref.c:ref_clip_face for ONE face: v (3,3) rows (x_ndc, y_ndc, z_view) ->
    list of (sub-triangle (3,3), M (3,3)) with M the rows of barycentric coordinates of its vertices in the original face.
    Differentiable in v (the HIP backward through clipped faces is checked against autograd of this)."""
    behind = [bool(v[i, 2] < z_clip) for i in range(3)]
    nb = sum(behind)
    eye = torch.eye(3, dtype=v.dtype)
    if nb == 0:
        return [(v, eye)]
    if nb == 3:
        return []
    i1 = [i for i in range(3) if behind[i] == (nb == 1)][0]
    i2, i3 = (i1 + 1) % 3, (i1 + 2) % 3
    p1, p2, p3 = v[i1], v[i2], v[i3]
    w2 = (p1[2] - z_clip) / (p1[2] - p2[2])
    w3 = (p1[2] - z_clip) / (p1[2] - p3[2])
    zc = torch.as_tensor(z_clip, dtype=v.dtype)

    def cut(pa, pb, w):
        if perspective_correct:
            xy = ((1 - w) * pa[:2] * pa[2] + w * pb[:2] * pb[2]) / zc
        else:
            xy = (1 - w) * pa[:2] + w * pb[:2]
        return torch.cat([xy, zc.reshape(1)])
    p4, p5 = cut(p1, p2, w2), cut(p1, p3, w3)
    b1, b2, b3 = eye[i1], eye[i2], eye[i3]
    b4 = (1 - w2) * b1 + w2 * b2
    b5 = (1 - w3) * b1 + w3 * b3
    if nb == 1:
        return [(torch.stack([p4, p2, p5]), torch.stack([b4, b2, b5])), (torch.stack([p5, p2, p3]), torch.stack([b5, b2, b3]))]
    return [(torch.stack([p1, p4, p5]), torch.stack([b1, b4, b5]))]


def clipped_geometry(ndc, faces, slots, S, clip_bary, perspective_correct, z_clip):
    """Differentiable fragment quantities for a FIXED fragment -> record-slot assignment (slots (S,S,K) from the C oracle or
    the HIP kernel, slot = 2 * face + sub): bary in the ORIGINAL face, zbuf and signed distance of the clipped triangle."""
    K = slots.shape[-1]
    bary = torch.zeros(S, S, K, 3, dtype=ndc.dtype)
    pz = torch.zeros(S, S, K, dtype=ndc.dtype)
    sd = torch.zeros(S, S, K, dtype=ndc.dtype)
    mask = slots >= 0
    cache = {}
    for yi, xi, k in torch.nonzero(mask).tolist():
        sl = int(slots[yi, xi, k])
        if sl not in cache:
            subs = clip_face(ndc[faces[sl >> 1]], z_clip, perspective_correct) if z_clip is not None else [(ndc[faces[sl >> 1]], torch.eye(3, dtype=ndc.dtype))]
            cache[sl] = subs[sl & 1]
        tri, M = cache[sl]
        px = torch.tensor(1.0 - (2.0 * xi + 1.0) / S, dtype=ndc.dtype)
        py = torch.tensor(1.0 - (2.0 * yi + 1.0) / S, dtype=ndc.dtype)
        (x0, y0, z0), (x1, y1, z1), (x2, y2, z2) = tri[0], tri[1], tri[2]
        area = _edge(x2, y2, x0, y0, x1, y1) + K_EPS
        w = torch.stack([_edge(px, py, x1, y1, x2, y2), _edge(px, py, x2, y2, x0, y0), _edge(px, py, x0, y0, x1, y1)]) / area
        if perspective_correct:
            t = torch.stack([w[0] * z1 * z2, z0 * w[1] * z2, z0 * z1 * w[2]])
            b = t / t.sum().clamp_min(K_EPS)
        else:
            b = w
        inside = bool((b > 0).all())
        c = b
        if clip_bary:
            c = b.clamp(0, 1)
            c = c / c.sum().clamp_min(K_EPS)
        pz[yi, xi, k] = c[0] * z0 + c[1] * z1 + c[2] * z2
        bary[yi, xi, k] = c @ M
        d = torch.minimum(torch.minimum(_pld2(px, py, x0, y0, x1, y1), _pld2(px, py, x1, y1, x2, y2)), _pld2(px, py, x2, y2, x0, y0))
        sd[yi, xi, k] = -d if inside else d
    return bary, pz, sd, mask
