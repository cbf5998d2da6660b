"""Shared scene builders for the GPU parity tests: the reference's DATA files (committed as tests/golden/assets_*.npz,
/root/reference does not exist on the GPU box) turned into the inputs of BASELINE.json's configs, once as numpy for the
CPU oracle and once as device tensors behind the drop-in API."""
import os

import numpy as np
import torch
import torch.nn.functional as F

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_asset(name):
    d = np.load(os.path.join(GOLDEN, f"assets_{name}_mesh.npz"))
    return {k: d[k] for k in d.files}


def texture_at(asset, T):
    """The mesh's own map resized to T x T as the reference does (second_approach.py:84-94: bilinear,
    align_corners=False) -> (T,T,3) float32 numpy."""
    tex = torch.from_numpy(asset["texture_u8"]).float().div(255.0)
    tex = F.interpolate(tex.permute(2, 0, 1)[None], size=T, mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    return np.ascontiguousarray(tex.numpy())


def style_at(k, S):
    """imgs/Style_k through load_as_tensor's pipeline at 512^2 (fixture), then to S x S -> (1,3,S,S) float32."""
    u8 = np.load(os.path.join(GOLDEN, f"assets_style{k}_512.npz"))["rgb_u8"]
    t = torch.from_numpy(u8).permute(2, 0, 1).float().div(255.0)[None]
    if S != t.shape[2]:
        t = F.interpolate(t, size=S, mode="bilinear", align_corners=False, antialias=S < t.shape[2])
    return t.contiguous()


def random_cameras(B, seed):
    """build_random_cameras (utils.py:154-170) with a seeded generator -> R (B,3,3), T (B,3) float32 numpy."""
    from oracle import render_ref as rr
    g = torch.Generator().manual_seed(seed)
    elev, azim = rr.random_camera_angles(B, lambda k: torch.rand(k, generator=g).numpy())
    return rr.look_at_view_transform(2.10, elev, azim, at=(0, 0.10, 0.25))


def subdivide(verts, faces, verts_uvs, faces_uvs):
    """One 1 -> 4 midpoint subdivision of positions and UVs (stand-in for the high-polygon bunny whose OBJ the reference
    does not hold).  numpy in, numpy out."""
    def split(pts, idx):
        idx = idx.astype(np.int64)
        e = np.concatenate([idx[:, [0, 1]], idx[:, [1, 2]], idx[:, [2, 0]]], 0)
        key = np.sort(e, 1)
        uniq, inv = np.unique(key[:, 0] * (pts.shape[0] + 1) + key[:, 1], return_inverse=True)
        lo, hi = uniq // (pts.shape[0] + 1), uniq % (pts.shape[0] + 1)
        mids = 0.5 * (pts[lo] + pts[hi])
        n = idx.shape[0]
        m01, m12, m20 = (inv[0:n] + pts.shape[0], inv[n:2 * n] + pts.shape[0], inv[2 * n:3 * n] + pts.shape[0])
        new_idx = np.concatenate([np.stack([idx[:, 0], m01, m20], 1), np.stack([m01, idx[:, 1], m12], 1),
                                  np.stack([m20, m12, idx[:, 2]], 1), np.stack([m01, m12, m20], 1)], 0)
        return np.concatenate([pts, mids.astype(pts.dtype)], 0), new_idx.astype(np.int32)
    v, f = split(np.asarray(verts, np.float32), np.asarray(faces))
    uv, fuv = split(np.asarray(verts_uvs, np.float32), np.asarray(faces_uvs))
    return v, f, uv, fuv


def device_scene(U, dev, verts, faces, verts_uvs, faces_uvs, tex_np, R, T, S):
    """-> (mesh, renderer, cameras) behind the drop-in API for the numpy scene."""
    from st3d.render import FoVPerspectiveCameras, MeshRasterizer, MeshRenderer, RasterizationSettings, SoftPhongShader
    mesh = U.build_mesh(torch.from_numpy(np.asarray(verts_uvs, np.float32))[None].to(dev),
                        torch.from_numpy(np.asarray(faces_uvs).astype(np.int64))[None].to(dev),
                        torch.from_numpy(np.asarray(tex_np, np.float32))[None].to(dev),
                        torch.from_numpy(np.asarray(verts, np.float32)).to(dev),
                        torch.from_numpy(np.asarray(faces).astype(np.int64)).to(dev))
    renderer = MeshRenderer(MeshRasterizer(None, RasterizationSettings(image_size=S)), SoftPhongShader())
    cams = FoVPerspectiveCameras(R=torch.from_numpy(R), T=torch.from_numpy(T), device=dev)
    return mesh, renderer, cams
