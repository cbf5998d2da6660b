"""Tensor-level wrappers over the C ABI (one function per include/st3d.h entry point).
PyTorch owns the device memory and the stream; every call goes to libst3d.so."""
import ctypes
import math
import os

import torch

from . import _lib
from ._lib import call, dptr, stream_ptr

F32, I32, U8 = torch.float32, torch.int32, torch.uint8
INV_TAN_HALF_FOV = float(1.0 / math.tan(math.radians(60.0) / 2.0))   # FoVPerspectiveCameras default fov=60


def _f32c(t):
    return t.detach().to(F32).contiguous()


# The texture / vertex gradient scatters of the render backward -- the specialised K = 1 path and, since round 3, the
# general soft path -- accumulate in 64-bit fixed point (csrc/det.h): bitwise reproducible from run to run, and measured
# no slower than the float-atomic kernels (0.128 vs 0.137 ms for the texture scatter of config 2), so it is the default.  set_deterministic(False) / ST3D_DETERMINISTIC=0 selects the float atomics.
_DETERMINISTIC = os.environ.get("ST3D_DETERMINISTIC", "1") not in ("", "0")


def set_deterministic(on=True):
    global _DETERMINISTIC
    _DETERMINISTIC = bool(on)


def is_deterministic():
    return _DETERMINISTIC


# ------------------------------------------------------------------ render
def project_verts(verts, R, T):
    """verts (V,3), R (B,3,3), T (B,3) -> (B,V,3) (x_ndc, y_ndc, z_view)."""
    verts, R, T = _f32c(verts), _f32c(R), _f32c(T)
    B, V = R.shape[0], verts.shape[0]
    out = torch.empty((B, V, 3), dtype=F32, device=verts.device)
    call("st3d_project_verts", dptr(verts), V, dptr(R), dptr(T), B, INV_TAN_HALF_FOV, dptr(out), stream_ptr())
    return out


# ---- near-plane watch of the specialised K = 1 rasteriser.  PyTorch3D clips every mesh at z_clip_value = znear / 2 before
# rasterising; the K = 1 / blur 0 kernels do not clip (for the reference's cameras nothing comes nearer than 0.78), so they
# raise a device flag when a rasterised face has a vertex nearer than z_clip.  The flag travels to pinned host memory
# without blocking and is looked at when a later call finds its copy complete (or by check_near_plane(block=True)).
# Policy (ST3D_NEAR_PLANE, default "clip"): from then on every render of the process goes through the general kernels,
# which clip exactly like PyTorch3D (st3d.render.render_views asks near_plane_triggered()), with one warning.  "raise":
# fail loudly.  This asynchronous watch is the safety net of DIRECT raster_fwd callers: st3d.render.render_views asks
# reaches_near_plane() BEFORE it renders (round 3), so through the renderer no frame is ever rendered unclipped -- a vertex
# optimisation that drives the mesh into the near plane (bob, 'both', lr 0.01: after ~80 steps) moves to the clipping
# kernels with the first such frame and keeps running like the reference does.
_NEAR_PENDING = []
_NEAR_TRIGGERED = False
_NEAR_WARNED = False
NEAR_PLANE_POLICY = os.environ.get("ST3D_NEAR_PLANE", "clip")
NEAR_PLANE_MESSAGE = ("a rasterised face has a vertex nearer than z_clip_value (PyTorch3D clips meshes at znear / 2 = 0.5); the "
                      "specialised K = 1 kernels do not clip -- construct RasterizationSettings(z_clip_value=0.5) to render "
                      "through the general kernels, which do")


def near_plane_triggered():
    return _NEAR_TRIGGERED


def reset_near_plane():
    global _NEAR_TRIGGERED, _NEAR_WARNED
    _NEAR_TRIGGERED = _NEAR_WARNED = False
    _NEAR_PENDING.clear()


def note_near_plane():
    """One warning per process the first time a render is sent to the clipping kernels."""
    global _NEAR_WARNED
    if not _NEAR_WARNED:
        import warnings
        warnings.warn("st3d: the mesh reached the near clipping plane (z < znear / 2); rendering continues on the general "
                      "kernels, which clip like PyTorch3D (ST3D_NEAR_PLANE=raise turns this into an error)")
        _NEAR_WARNED = True


def check_near_plane(block=False):
    global _NEAR_TRIGGERED
    while _NEAR_PENDING:
        host, ev = _NEAR_PENDING[0]
        if block:
            ev.synchronize()
        elif not ev.query():
            break
        _NEAR_PENDING.pop(0)
        if int(host[0]) != 0:
            _NEAR_PENDING.clear()
            if NEAR_PLANE_POLICY == "raise":
                raise RuntimeError(NEAR_PLANE_MESSAGE)
            note_near_plane()
            _NEAR_TRIGGERED = True


def raster_fwd(verts_ndc, faces_i32, S, z_clip=None):
    """-> pix_to_face (B,S,S) int32, zbuf (B,S,S), bary (B,S,S,3), dists (B,S,S).  z_clip: depth the near-plane watch
    compares with (None = no watch)."""
    B, V, _ = verts_ndc.shape
    F = faces_i32.shape[0]
    dev = verts_ndc.device
    ws_bytes = _lib.load().st3d_raster_workspace_bytes_binned(B, F, S)       # room for the coarse face bins too
    ws = torch.empty(((ws_bytes + 3) // 4,), dtype=F32, device=dev)
    p2f = torch.empty((B, S, S), dtype=I32, device=dev)
    zbuf = torch.empty((B, S, S), dtype=F32, device=dev)
    bary = torch.empty((B, S, S, 3), dtype=F32, device=dev)
    dists = torch.empty((B, S, S), dtype=F32, device=dev)
    flag = torch.zeros((1,), dtype=I32, device=dev) if z_clip is not None else None
    call("st3d_raster_fwd", dptr(verts_ndc, F32), dptr(faces_i32, I32), B, V, F, S, dptr(ws), ws_bytes, dptr(p2f),
         dptr(zbuf), dptr(bary), dptr(dists), float(z_clip or 0.0), dptr(flag), stream_ptr())
    if flag is not None:
        check_near_plane()
        host = torch.empty((1,), dtype=I32, pin_memory=True)
        host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _NEAR_PENDING.append((host, ev))
    return p2f, zbuf, bary, dists


def shade_fwd(frag, verts_uvs, faces_uvs_i32, texture):
    p2f, zbuf, bary, dists = frag
    B, S, _ = p2f.shape
    T = texture.shape[0]
    rgb = torch.empty((B, 3, S, S), dtype=F32, device=p2f.device)
    mask = torch.empty((B, 1, S, S), dtype=F32, device=p2f.device)
    call("st3d_shade_fwd", dptr(p2f, I32), dptr(bary, F32), dptr(zbuf, F32), dptr(dists, F32), dptr(verts_uvs, F32),
         dptr(faces_uvs_i32, I32), dptr(texture, F32), B, S, T, faces_uvs_i32.shape[0], verts_uvs.shape[0], dptr(rgb),
         dptr(mask), stream_ptr())
    return rgb, mask


def shade_bwd(grad_rgb, frag, verts_uvs, faces_uvs_i32, texture, grad_texture=None, want_uv=False, want_bary=False,
              want_texture=True):
    """-> grad_texture (T,T,3) [, grad_uv (B,S,S,2)] [, grad_bary (B,S,S,3)]"""
    p2f, zbuf, bary, dists = frag
    B, S, _ = p2f.shape
    T = texture.shape[0]
    if grad_texture is None and want_texture:
        grad_texture = torch.zeros((T, T, 3), dtype=F32, device=p2f.device)
    guv = torch.empty((B, S, S, 2), dtype=F32, device=p2f.device) if want_uv else None
    gbary = torch.empty((B, S, S, 3), dtype=F32, device=p2f.device) if want_bary else None
    grad_rgb = grad_rgb.contiguous()
    if _DETERMINISTIC and grad_texture is not None:
        nb = _lib.load().st3d_shade_bwd_det_workspace_bytes(T)
        ws = torch.empty(((nb + 15) // 16 * 4,), dtype=F32, device=p2f.device)
        call("st3d_shade_bwd_det", dptr(grad_rgb, F32), dptr(p2f, I32), dptr(bary, F32), dptr(zbuf, F32), dptr(dists, F32),
             dptr(verts_uvs, F32), dptr(faces_uvs_i32, I32), dptr(texture, F32), B, S, T, faces_uvs_i32.shape[0],
             verts_uvs.shape[0], dptr(grad_texture, F32), dptr(guv), dptr(gbary), dptr(ws), nb, stream_ptr())
        out = (grad_texture,)
        if want_uv:
            out += (guv,)
        if want_bary:
            out += (gbary,)
        return out if len(out) > 1 else out[0]
    call("st3d_shade_bwd", dptr(grad_rgb, F32), dptr(p2f, I32), dptr(bary, F32), dptr(zbuf, F32), dptr(dists, F32),
         dptr(verts_uvs, F32), dptr(faces_uvs_i32, I32), dptr(texture, F32), B, S, T, faces_uvs_i32.shape[0],
         verts_uvs.shape[0], dptr(grad_texture, F32) if grad_texture is not None else None, dptr(guv), dptr(gbary),
         stream_ptr())
    out = (grad_texture,)
    if want_uv:
        out += (guv,)
    if want_bary:
        out += (gbary,)
    return out if len(out) > 1 else out[0]


def raster_bwd(grad_bary, p2f, verts_ndc, faces_i32):
    """grad_bary (B,S,S,3) -> grad_verts_ndc (B,V,3)"""
    B, V, _ = verts_ndc.shape
    S = p2f.shape[1]
    g = torch.empty((B, V, 3), dtype=F32, device=verts_ndc.device)
    if _DETERMINISTIC:
        nb = _lib.load().st3d_raster_bwd_det_workspace_bytes(B, V, S)
        ws = torch.empty(((nb + 15) // 16 * 4,), dtype=F32, device=verts_ndc.device)
        call("st3d_raster_bwd_det", dptr(grad_bary, F32), dptr(p2f, I32), dptr(verts_ndc, F32), dptr(faces_i32, I32), B, V,
             faces_i32.shape[0], S, dptr(g), dptr(ws), nb, stream_ptr())
        return g
    call("st3d_raster_bwd", dptr(grad_bary, F32), dptr(p2f, I32), dptr(verts_ndc, F32), dptr(faces_i32, I32), B, V,
         faces_i32.shape[0], S, dptr(g), stream_ptr())
    return g


def project_verts_bwd(verts, R, T, grad_ndc, out=None):
    verts, R, T = _f32c(verts), _f32c(R), _f32c(T)
    acc = 1
    if out is None:
        out = torch.empty_like(verts)
        acc = 0
    call("st3d_project_verts_bwd", dptr(verts), verts.shape[0], dptr(R), dptr(T), R.shape[0], INV_TAN_HALF_FOV,
         dptr(grad_ndc, F32), acc, dptr(out), stream_ptr())
    return out


def mesh_reg(verts, target, topo, weights, want_grad=True):
    """topo: dict(edges (E,2) i32, nbr_off (V+1) i32, nbr_idx i32, pairs (P,4) i32, pair_off (V+1) i32, pair_ref (4P) i32).
    -> (loss_out [weighted total, mse, edge, laplacian, normal], grad_verts (V,3) or None)"""
    verts, target = _f32c(verts), _f32c(target)
    V = verts.shape[0]
    dev = verts.device
    P = topo["pairs"].shape[0]
    scratch = torch.empty((_lib.load().st3d_mesh_reg_scratch_floats(V, P),), dtype=F32, device=dev)
    parts = torch.empty((4 * _lib.load().st3d_reduce_partials(),), dtype=F32, device=dev)
    out = torch.zeros((5,), dtype=F32, device=dev)
    g = torch.zeros_like(verts) if want_grad else None
    w = (ctypes.c_float * 4)(*[float(x) for x in weights])
    call("st3d_mesh_reg", dptr(verts), dptr(target), V, dptr(topo["edges"], I32), topo["edges"].shape[0],
         dptr(topo["nbr_off"], I32), dptr(topo["nbr_idx"], I32), dptr(topo["pairs"], I32) if P else None, P,
         dptr(topo["pair_off"], I32) if P else None, dptr(topo["pair_ref"], I32) if P else None, w,
         dptr(scratch), dptr(parts), dptr(out), dptr(g), stream_ptr())
    return out, g


# ------------------------------------------------------------------ general soft renderer (K faces per pixel, blur)
def raster_soft_fwd(verts_ndc, faces_i32, S, K, blur_radius=0.0, clip_bary=None, cull_backfaces=False,
                    perspective_correct=True, z_clip=None):
    """-> pix_to_face (B,S,S,K) int32, zbuf, bary (B,S,S,K,3), dists; clip_bary None = PyTorch3D default (blur > 0).
    z_clip: near-plane clipping depth (PyTorch3D: znear / 2); then a fifth tensor, the record slot of every fragment,
    is returned for raster_soft_bwd."""
    B, V, _ = verts_ndc.shape
    F = faces_i32.shape[0]
    dev = verts_ndc.device
    if clip_bary is None:
        clip_bary = blur_radius > 0.0
    slots = None
    if z_clip is None:
        ws_bytes = _lib.load().st3d_raster_workspace_bytes(B, F)
        rec = torch.empty((ws_bytes // 4,), dtype=F32, device=dev)
        call("st3d_face_setup", dptr(verts_ndc, F32), dptr(faces_i32, I32), B, V, F, dptr(rec), ws_bytes, stream_ptr())
    else:
        ws_bytes = _lib.load().st3d_clip_records_bytes(B, F)
        rec = torch.empty((ws_bytes // 4,), dtype=F32, device=dev)
        call("st3d_face_setup_clip", dptr(verts_ndc, F32), dptr(faces_i32, I32), B, V, F, float(z_clip),
             1 if perspective_correct else 0, dptr(rec), ws_bytes, stream_ptr())
        slots = torch.empty((B, S, S, K), dtype=I32, device=dev)
    p2f = torch.empty((B, S, S, K), dtype=I32, device=dev)
    zbuf = torch.empty((B, S, S, K), dtype=F32, device=dev)
    bary = torch.empty((B, S, S, K, 3), dtype=F32, device=dev)
    dists = torch.empty((B, S, S, K), dtype=F32, device=dev)
    call("st3d_raster_soft_fwd", dptr(rec), B, F, S, int(K), float(blur_radius), 1 if clip_bary else 0,
         1 if cull_backfaces else 0, 1 if perspective_correct else 0, 1 if slots is None else 2, dptr(slots), dptr(p2f),
         dptr(zbuf), dptr(bary), dptr(dists), stream_ptr())
    return (p2f, zbuf, bary, dists) if slots is None else (p2f, zbuf, bary, dists, slots)


def _bg3(background):
    return (ctypes.c_float * 3)(*[float(x) for x in background])


def shade_soft_fwd(frag, verts_uvs, faces_uvs_i32, texture, sigma=1e-4, gamma=1e-4, background=(1.0, 1.0, 1.0)):
    p2f, zbuf, bary, dists = frag
    B, S, _, K = p2f.shape
    rgb = torch.empty((B, 3, S, S), dtype=F32, device=p2f.device)
    alpha = torch.empty((B, 1, S, S), dtype=F32, device=p2f.device)
    call("st3d_shade_soft_fwd", dptr(p2f, I32), dptr(bary, F32), dptr(zbuf, F32), dptr(dists, F32), dptr(verts_uvs, F32),
         dptr(faces_uvs_i32, I32), dptr(texture, F32), B, S, texture.shape[0], K, float(sigma), float(gamma), _bg3(background),
         dptr(rgb), dptr(alpha), stream_ptr())
    return rgb, alpha


def shade_soft_bwd(grad_rgb, frag, verts_uvs, faces_uvs_i32, texture, sigma=1e-4, gamma=1e-4, background=(1.0, 1.0, 1.0),
                   want_texture=True, want_geometry=True):
    """-> (grad_texture (T,T,3) | None, (grad_bary, grad_zbuf, grad_dists) | None)"""
    p2f, zbuf, bary, dists = frag
    B, S, _, K = p2f.shape
    T = texture.shape[0]
    dev = p2f.device
    gt = torch.zeros((T, T, 3), dtype=F32, device=dev) if want_texture else None
    gb = torch.empty((B, S, S, K, 3), dtype=F32, device=dev) if want_geometry else None
    gz = torch.empty((B, S, S, K), dtype=F32, device=dev) if want_geometry else None
    gd = torch.empty((B, S, S, K), dtype=F32, device=dev) if want_geometry else None
    if _DETERMINISTIC and gt is not None:       # fixed-point texture scatter: bitwise reproducible (st3d_shade_soft_bwd_det)
        nb = _lib.load().st3d_shade_soft_bwd_det_workspace_bytes(T)
        ws = torch.empty(((nb + 15) // 16 * 4,), dtype=F32, device=dev)
        call("st3d_shade_soft_bwd_det", dptr(grad_rgb.contiguous(), F32), dptr(p2f, I32), dptr(bary, F32), dptr(zbuf, F32),
             dptr(dists, F32), dptr(verts_uvs, F32), dptr(faces_uvs_i32, I32), dptr(texture, F32), B, S, T, K, float(sigma),
             float(gamma), _bg3(background), dptr(gt), dptr(gb), dptr(gz), dptr(gd), dptr(ws), nb, stream_ptr())
        return gt, ((gb, gz, gd) if want_geometry else None)
    call("st3d_shade_soft_bwd", dptr(grad_rgb.contiguous(), F32), dptr(p2f, I32), dptr(bary, F32), dptr(zbuf, F32),
         dptr(dists, F32), dptr(verts_uvs, F32), dptr(faces_uvs_i32, I32), dptr(texture, F32), B, S, T, K, float(sigma),
         float(gamma), _bg3(background), dptr(gt), dptr(gb), dptr(gz), dptr(gd), stream_ptr())
    return gt, ((gb, gz, gd) if want_geometry else None)


def raster_soft_bwd(grads, p2f, verts_ndc, faces_i32, clip_bary, perspective_correct=True, slots=None, z_clip=None):
    gb, gz, gd = grads
    B, V, _ = verts_ndc.shape
    S, K = p2f.shape[1], p2f.shape[3]
    g = torch.empty((B, V, 3), dtype=F32, device=verts_ndc.device)
    if _DETERMINISTIC:                          # fixed-point vertex scatter (st3d_raster_soft_bwd_det)
        nb = _lib.load().st3d_raster_soft_bwd_det_workspace_bytes(B, V, S)
        ws = torch.empty(((nb + 15) // 16 * 4,), dtype=F32, device=verts_ndc.device)
        call("st3d_raster_soft_bwd_det", dptr(gb, F32), dptr(gz, F32), dptr(gd, F32), dptr(p2f, I32), dptr(verts_ndc, F32),
             dptr(faces_i32, I32), B, V, faces_i32.shape[0], S, K, 1 if clip_bary else 0, 1 if perspective_correct else 0,
             dptr(slots, I32) if slots is not None else None, float(z_clip) if z_clip is not None else 0.0, dptr(g), dptr(ws), nb,
             stream_ptr())
        return g
    call("st3d_raster_soft_bwd", dptr(gb, F32), dptr(gz, F32), dptr(gd, F32), dptr(p2f, I32), dptr(verts_ndc, F32),
         dptr(faces_i32, I32), B, V, faces_i32.shape[0], S, K, 1 if clip_bary else 0, 1 if perspective_correct else 0,
         dptr(slots, I32) if slots is not None else None, float(z_clip) if z_clip is not None else 0.0, dptr(g), stream_ptr())
    return g


def apply_background(img, mask, bg=None):
    B, _, S, _ = img.shape
    out = torch.empty_like(img)
    bgb = 1 if (bg is None or bg.dim() == 3 or bg.shape[0] == 1) else B
    call("st3d_apply_background", dptr(img.contiguous(), F32), dptr(mask.contiguous(), F32),
         dptr(bg.contiguous(), F32) if bg is not None else None, bgb, B, S, dptr(out), stream_ptr())
    return out


# ------------------------------------------------------------------ conv / pool
def conv3x3_pack(w):
    """w (Cout,Cin,3,3) -> (w_fwd_packed, w_dgrad_packed) flat float tensors."""
    w = _f32c(w)
    Cout, Cin = w.shape[:2]
    n = _lib.load().st3d_conv3x3_packed_floats(Cout, Cin)
    wf = torch.empty((n,), dtype=F32, device=w.device)
    wd = torch.empty((n,), dtype=F32, device=w.device)
    call("st3d_conv3x3_pack", dptr(w), Cout, Cin, dptr(wf), dptr(wd), stream_ptr())
    return wf, wd


def conv3x3_fwd(x, wf, bias, Cout, relu=True):
    N, Cin, H, W = x.shape
    y = torch.empty((N, Cout, H, W), dtype=F32, device=x.device)
    call("st3d_conv3x3_fwd", dptr(x.contiguous(), F32), dptr(wf, F32), dptr(bias, F32) if bias is not None else None,
         dptr(y), N, Cin, Cout, H, W, 1 if relu else 0, stream_ptr())
    return y


def conv3x3_dgrad(gy, act, wd, Cin):
    N, Cout, H, W = gy.shape
    gx = torch.empty((N, Cin, H, W), dtype=F32, device=gy.device)
    call("st3d_conv3x3_dgrad", dptr(gy.contiguous(), F32), dptr(act, F32) if act is not None else None, dptr(wd, F32),
         dptr(gx), N, Cin, Cout, H, W, stream_ptr())
    return gx


def conv1_bwd(gy, act, D, coef, wd):
    """gx (N,3,H,W) = conv1_1^T(gate(gy + coef * D act)) in one pass (st3d_conv1_bwd); gy or D may be None."""
    N, C, H, W = act.shape
    nb = _lib.load().st3d_conv1_bwd_workspace_bytes(N, H, W)
    ws = torch.empty((nb // 4,), dtype=F32, device=act.device)
    gx = torch.empty((N, 3, H, W), dtype=F32, device=act.device)
    call("st3d_conv1_bwd", dptr(gy, F32), dptr(act, F32), dptr(D, F32), float(coef), dptr(wd, F32), dptr(ws), nb, dptr(gx),
         N, H, W, stream_ptr())
    return gx


def conv3x3_dgrad_unpool(gy_pooled, pool_idx, pooled, wd, Cin):
    N, Cout, Hp, Wp = gy_pooled.shape
    H, W = 2 * Hp, 2 * Wp
    gx = torch.empty((N, Cin, H, W), dtype=F32, device=gy_pooled.device)
    call("st3d_conv3x3_dgrad_unpool", dptr(gy_pooled.contiguous(), F32), dptr(pool_idx, U8), dptr(pooled, F32),
         dptr(wd, F32), dptr(gx), N, Cin, Cout, H, W, stream_ptr())
    return gx


def wino_pack(w):
    """w (Cout,Cin,3,3) -> (u_fwd [16][Cin][Cout], u_dgrad [16][Cout][Cin]) flat tensors."""
    w = _f32c(w)
    Cout, Cin = w.shape[:2]
    n = _lib.load().st3d_wino_packed_floats(Cout, Cin)
    uf = torch.empty((n,), dtype=F32, device=w.device)
    ud = torch.empty((n,), dtype=F32, device=w.device)
    call("st3d_wino_pack", dptr(w), Cout, Cin, dptr(uf), dptr(ud), stream_ptr())
    return uf, ud


def wino_fwd(x, uf, bias, Cout, relu=True, pool=False, keep_full=True):
    N, Cin, H, W = x.shape
    y = torch.empty((N, Cout, H, W), dtype=F32, device=x.device) if (keep_full or not pool) else None
    yp = torch.empty((N, Cout, H // 2, W // 2), dtype=F32, device=x.device) if pool else None
    idx = torch.empty((N, Cout, H // 2, W // 2), dtype=U8, device=x.device) if pool else None
    call("st3d_wino_fwd", dptr(x.contiguous(), F32), dptr(uf, F32), dptr(bias, F32) if bias is not None else None, dptr(y),
         dptr(yp), dptr(idx), N, Cin, Cout, H, W, 1 if relu else 0, stream_ptr())
    return (y, yp, idx) if pool else y


def wino_dgrad(gy, act, ud, Cin):
    N, Cout, H, W = gy.shape
    gx = torch.empty((N, Cin, H, W), dtype=F32, device=gy.device)
    call("st3d_wino_dgrad", dptr(gy.contiguous(), F32), dptr(act, F32) if act is not None else None, dptr(ud, F32),
         dptr(gx), N, Cin, Cout, H, W, stream_ptr())
    return gx


def wino_dgrad_unpool(gy_pooled, pool_idx, pooled, ud, Cin):
    N, Cout, Hp, Wp = gy_pooled.shape
    H, W = 2 * Hp, 2 * Wp
    gx = torch.empty((N, Cin, H, W), dtype=F32, device=gy_pooled.device)
    call("st3d_wino_dgrad_unpool", dptr(gy_pooled.contiguous(), F32), dptr(pool_idx, U8), dptr(pooled, F32), dptr(ud, F32),
         dptr(gx), N, Cin, Cout, H, W, stream_ptr())
    return gx


def wino_dgrad_chain(gy, ud, Cin, act=None, pool_idx=None, pooled=None, out_gate=None, add_target=None, add_coef=0.0):
    """One link of the producer-gated backward chain (st3d_wino_dgrad_chain)."""
    N, Cout = gy.shape[:2]
    H, W = (2 * gy.shape[2], 2 * gy.shape[3]) if pool_idx is not None else gy.shape[2:]
    gx = torch.empty((N, Cin, H, W), dtype=F32, device=gy.device)
    call("st3d_wino_dgrad_chain", dptr(gy.contiguous(), F32), dptr(act, F32), dptr(pool_idx, U8), dptr(pooled, F32),
         dptr(ud, F32), dptr(out_gate, F32), dptr(add_target, F32), float(add_coef), dptr(gx), N, Cin, Cout, H, W, stream_ptr())
    return gx


def wino43_pack(w):
    """w (Cout,Cin,3,3) -> (u_fwd, u_dgrad): F(4x4,3x3) filter packs (36 floats per weight, st3d_wino43_pack)."""
    w = _f32c(w)
    Cout, Cin = w.shape[:2]
    n = _lib.load().st3d_wino43_packed_floats(Cout, Cin)
    uf = torch.empty((n,), dtype=F32, device=w.device)
    ud = torch.empty((n,), dtype=F32, device=w.device)
    call("st3d_wino43_pack", dptr(w), Cout, Cin, dptr(uf), dptr(ud), stream_ptr())
    return uf, ud


def wino43_fwd(x, uf, bias, Cout, relu=True, pool=False, keep_full=True):
    N, Cin, H, W = x.shape
    y = torch.empty((N, Cout, H, W), dtype=F32, device=x.device) if (keep_full or not pool) else None
    yp = torch.empty((N, Cout, H // 2, W // 2), dtype=F32, device=x.device) if pool else None
    idx = torch.empty((N, Cout, H // 2, W // 2), dtype=U8, device=x.device) if pool else None
    call("st3d_wino43_fwd", dptr(x.contiguous(), F32), dptr(uf, F32), dptr(bias, F32) if bias is not None else None, dptr(y),
         dptr(yp), dptr(idx), N, Cin, Cout, H, W, 1 if relu else 0, stream_ptr())
    return (y, yp, idx) if pool else y


def wino43_dgrad_chain(gy, ud, Cin, pool_idx=None, out_gate=None, add_target=None, add_coef=0.0):
    """One link of the producer-gated backward chain on the F(4x4,3x3) kernel (st3d_wino43_dgrad_chain)."""
    N, Cout = gy.shape[:2]
    H, W = (2 * gy.shape[2], 2 * gy.shape[3]) if pool_idx is not None else gy.shape[2:]
    gx = torch.empty((N, Cin, H, W), dtype=F32, device=gy.device)
    call("st3d_wino43_dgrad_chain", dptr(gy.contiguous(), F32), dptr(pool_idx, U8), dptr(ud, F32), dptr(out_gate, F32),
         dptr(add_target, F32), float(add_coef), dptr(gx), N, Cin, Cout, H, W, stream_ptr())
    return gx


def maxpool2x2(y, want_idx=True):
    N, C, H, W = y.shape
    p = torch.empty((N, C, H // 2, W // 2), dtype=F32, device=y.device)
    idx = torch.empty((N, C, H // 2, W // 2), dtype=U8, device=y.device) if want_idx else None
    call("st3d_maxpool2x2_fwd", dptr(y.contiguous(), F32), dptr(p), dptr(idx), N, C, H, W, stream_ptr())
    return (p, idx) if want_idx else p


# ------------------------------------------------------------------ gram / losses
def gram_fwd(feat):
    """(B,C,H,W) or (B,C,HW) -> (B,C,C) unnormalised Gram (style_transfer.py:31-35)."""
    B, C = feat.shape[:2]
    HW = feat[0, 0].numel()
    ws_bytes = _lib.load().st3d_gram_workspace_bytes(B, C, HW)
    ws = torch.empty((max(ws_bytes // 4, 1),), dtype=F32, device=feat.device)
    g = torch.empty((B, C, C), dtype=F32, device=feat.device)
    call("st3d_gram_fwd", dptr(feat.contiguous(), F32), B, C, HW, dptr(ws), ws_bytes, dptr(g), stream_ptr())
    return g


class _GramItem(ctypes.Structure):
    _fields_ = [("feat", ctypes.c_void_p), ("gram", ctypes.c_void_p), ("B", ctypes.c_int), ("C", ctypes.c_int), ("HW", ctypes.c_int)]


def gram_fwd_multi(feats):
    """[(B,C,H,W) ...] -> [(B,C,C) ...]: the Grams of several layers in one launch pair (st3d_gram_fwd_multi)."""
    feats = [f.contiguous() for f in feats]
    grams = [torch.empty((f.shape[0], f.shape[1], f.shape[1]), dtype=F32, device=f.device) for f in feats]
    items = (_GramItem * len(feats))()
    for it, f, g in zip(items, feats, grams):
        it.feat, it.gram, it.B, it.C, it.HW = dptr(f, F32), dptr(g, F32), f.shape[0], f.shape[1], f[0, 0].numel()
    nb = _lib.load().st3d_gram_multi_workspace_bytes(items, len(feats))
    ws = torch.empty((max(nb // 4, 64),), dtype=F32, device=feats[0].device)
    assert ws.data_ptr() % 256 == 0
    call("st3d_gram_fwd_multi", items, len(feats), dptr(ws), nb, stream_ptr())
    return grams


def gram_bwd(D, feat, coef, out=None, gated=False):
    """out (+)= coef * D feat; gated: then zeroed where feat <= 0 (st3d_gram_bwd_gated)."""
    B, C = feat.shape[:2]
    HW = feat[0, 0].numel()
    acc = 1
    if out is None:
        out = torch.empty_like(feat)
        acc = 0
    call("st3d_gram_bwd_gated" if gated else "st3d_gram_bwd", dptr(D.contiguous(), F32), dptr(feat.contiguous(), F32), B, C, HW, float(coef), acc, dptr(out),
         stream_ptr())
    return out


def sqdiff_sum(a, b, scale=1.0, want_diff=False):
    n, nb = a.numel(), b.numel()
    parts = torch.empty((_lib.load().st3d_reduce_partials(),), dtype=F32, device=a.device)
    out = torch.zeros((1,), dtype=F32, device=a.device)
    D = torch.empty_like(a) if want_diff else None
    call("st3d_sqdiff_sum", dptr(a.contiguous(), F32), dptr(b.contiguous(), F32), n, nb, float(scale), dptr(D),
         dptr(parts), dptr(out), stream_ptr())
    return (out, D) if want_diff else out


def masked_mse(rendered, target, mask, want_grad=True):
    B, _, S, _ = rendered.shape
    parts = torch.empty((_lib.load().st3d_reduce_partials(),), dtype=F32, device=rendered.device)
    out = torch.zeros((1,), dtype=F32, device=rendered.device)
    g = torch.empty_like(rendered) if want_grad else None
    call("st3d_masked_mse", dptr(rendered.contiguous(), F32), dptr(target.contiguous(), F32), dptr(mask.contiguous(), F32),
         B, S, dptr(g), dptr(parts), dptr(out), stream_ptr())
    return out, g


def tv_loss(images, masks, want_grad=True):
    """masked anisotropic L1 TV / sum(masks) -> (loss (1,), grad_images | None)"""
    B, C, H, W = images.shape
    parts = torch.empty((2 * _lib.load().st3d_reduce_partials(),), dtype=F32, device=images.device)
    out = torch.zeros((2,), dtype=F32, device=images.device)
    g = torch.empty_like(images, memory_format=torch.contiguous_format) if want_grad else None
    call("st3d_tv_loss", dptr(images.contiguous(), F32), dptr(masks.contiguous(), F32), B, C, H, W, dptr(parts), dptr(out),
         dptr(g), stream_ptr())
    return out[:1], g


def range_loss(values, want_grad=True):
    """sum relu(v - 1) + relu(-v) -> (loss (1,), grad | None)"""
    v = values.contiguous()
    parts = torch.empty((_lib.load().st3d_reduce_partials(),), dtype=F32, device=v.device)
    out = torch.zeros((1,), dtype=F32, device=v.device)
    g = torch.empty_like(v) if want_grad else None
    call("st3d_range_loss", dptr(v, F32), v.numel(), dptr(parts), dptr(out), dptr(g), stream_ptr())
    return out, g


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    call("st3d_adam_step", dptr(p, F32), dptr(g.contiguous(), F32), dptr(m, F32), dptr(v, F32), p.numel(), int(step),
         float(lr), float(b1), float(b2), float(eps), stream_ptr())


def device_info(device=0):
    cu = ctypes.c_int(0)
    hbm = ctypes.c_size_t(0)
    name = ctypes.create_string_buffer(128)
    call("st3d_device_info", device, ctypes.byref(cu), ctypes.byref(hbm), name, 128)
    return {"cu_count": cu.value, "hbm_bytes": hbm.value, "name": name.value.decode()}



# ---------------------------------------------------------------------------- roctx ranges (ST3D_ROCTX=1; include/st3d.h)
class trace:
    """with ops.trace("render"): ...  -- a named range for rocprofv3 --marker-trace around a host phase of the step (the
    library marks the VGG phases itself).  Costs nothing unless ST3D_ROCTX=1 was set before the first call."""
    _on = None

    def __init__(self, name):
        self.name = name.encode()

    def __enter__(self):
        if trace._on is None:
            trace._on = os.environ.get("ST3D_ROCTX") == "1" and _lib.load().st3d_trace_enabled() == 1
        if trace._on:
            _lib.load().st3d_trace_push(self.name)
        return self

    def __exit__(self, *exc):
        if trace._on:
            _lib.load().st3d_trace_pop()
        return False
