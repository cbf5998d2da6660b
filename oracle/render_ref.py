"""oracle/render_ref.py -- TEST INFRASTRUCTURE ONLY.

numpy/ctypes front-end of the C restatement in ``oracle/raster_ref.c`` plus the camera
builders of the reference (``utils.py:121-170``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module;
the product package never does.

PARITY UNPINNED for the render half: PyTorch3D (unpinned version, not vendored, not
installed) cannot be run here and the reference holds no golden vectors for it, so these
functions restate its published algorithm and are pinned by the
analytic tests in ``tests/test_oracle_raster.py`` only.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_raster.so")
_lib = None

F32P = ctypes.POINTER(ctypes.c_float)
F64P = ctypes.POINTER(ctypes.c_double)
I32P = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """Compile the C oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "raster_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.ref_project_verts.argtypes = [F32P, ctypes.c_int, F32P, F32P, ctypes.c_float, F32P]
        _lib.ref_rasterize.argtypes = [F32P, I32P, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                       I32P, F32P, F32P, F32P]
        _lib.ref_rasterize_naive.argtypes = _lib.ref_rasterize.argtypes
        _lib.ref_rasterize_naive.restype = None
        _lib.ref_shade_fwd.argtypes = [I32P, F32P, F32P, F32P, F32P, I32P, F32P, ctypes.c_int, ctypes.c_int, F32P, F32P]
        _lib.ref_shade_bwd.argtypes = [F32P, I32P, F32P, F32P, F32P, F32P, I32P, F32P, ctypes.c_int, ctypes.c_int,
                                       F64P, F32P]
        _lib.ref_adam_step.argtypes = [F32P, F32P, F32P, F32P, ctypes.c_size_t, ctypes.c_int,
                                       ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float]
        _lib.ref_uv_to_bary_grad.argtypes = [F32P, I32P, F32P, I32P, ctypes.c_int, F32P]
        _lib.ref_raster_bwd.argtypes = [F32P, I32P, F32P, I32P, ctypes.c_int, F64P]
        _lib.ref_project_verts_bwd.argtypes = [F32P, ctypes.c_int, F32P, F32P, ctypes.c_float, F64P, F64P]
        _lib.ref_rasterize_k.argtypes = [F32P, I32P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                         ctypes.c_int, I32P, F32P, F32P, F32P]
        _lib.ref_rasterize_k2.argtypes = [F32P, I32P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, I32P, F32P, F32P, F32P]
        _lib.ref_rasterize_k2.restype = None
        _lib.ref_rasterize_k3.argtypes = [F32P, I32P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, I32P, F32P, F32P, F32P, I32P]
        _lib.ref_rasterize_k3.restype = None
        for f in ("ref_project_verts", "ref_rasterize", "ref_shade_fwd", "ref_shade_bwd", "ref_adam_step",
                  "ref_uv_to_bary_grad", "ref_raster_bwd", "ref_project_verts_bwd", "ref_rasterize_k"):
            getattr(_lib, f).restype = None
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


INV_TAN_HALF_FOV = float(np.float32(1.0 / math.tan(math.radians(60.0) / 2.0)))   # FoVPerspectiveCameras fov=60

# ----------------------------------------------------------------------------- cameras


def look_at_view_transform(dist, elev, azim, at=(0.0, 0.0, 0.0)):
    """PyTorch3D look_at_view_transform (degrees), as called at utils.py:161-166.

    Returns R (n,3,3), T (n,3) float32 in the row-vector convention X_view = X_world @ R + T.
    """
    elev = np.atleast_1d(np.asarray(elev, dtype=np.float64)) * math.pi / 180.0
    azim = np.atleast_1d(np.asarray(azim, dtype=np.float64)) * math.pi / 180.0
    at = np.asarray(at, dtype=np.float64).reshape(1, 3)
    x = dist * np.cos(elev) * np.sin(azim)
    y = dist * np.sin(elev)
    z = dist * np.cos(elev) * np.cos(azim)
    C = np.stack([x, y, z], axis=1) + at
    up = np.array([[0.0, 1.0, 0.0]])

    def _n(v):
        return v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-5)

    z_axis = _n(at - C)
    x_axis = _n(np.cross(up, z_axis))
    y_axis = _n(np.cross(z_axis, x_axis))
    degenerate = np.all(np.isclose(x_axis, 0.0, atol=5e-3), axis=1)
    if degenerate.any():
        x_axis[degenerate] = _n(np.cross(y_axis, z_axis))[degenerate]
    R = np.stack([x_axis, y_axis, z_axis], axis=2)          # columns are the axes
    T = -np.einsum("nji,nj->ni", R, C)                      # -R^T C
    return R.astype(np.float32), T.astype(np.float32)


def random_camera_angles(n_views, rng_uniform):
    """utils.py:154-159: cos_elev ~ U(-1,1), elev = acos()*180/pi - 90, azim ~ U(-180,180).

    rng_uniform(n) must return n float32 uniforms in [0,1) (the reference uses unseeded
    torch.rand; tests inject a seeded generator)."""
    cos_e = rng_uniform(n_views).astype(np.float32) * 2 - 1
    elev = np.arccos(cos_e) * np.float32(180.0) / np.float32(math.pi) - 90
    azim = rng_uniform(n_views).astype(np.float32) * 360 - 180
    return elev, azim


def rotate_axis_angle(angle_deg, axis):
    """3x3 block of PyTorch3D RotateAxisAngle(angle, axis).get_matrix() (utils.py:142)."""
    a = math.radians(angle_deg)
    c, s = math.cos(a), math.sin(a)
    if axis == "X":
        m = np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)
    elif axis == "Y":
        m = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)
    else:
        m = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)
    return m.T.astype(np.float32)


def fixed_cameras(n_views, dist=3.0):
    """utils.py:121-151 without the shuffle."""
    xv = n_views // 2
    yv = n_views - xv
    angles = [(float(a), "X") for a in np.linspace(0, 315, xv, dtype=np.float32)] + \
             [(float(a), "Y") for a in np.linspace(45, 315, yv, dtype=np.float32)]
    R = np.stack([rotate_axis_angle(a, ax) for a, ax in angles])
    T = np.tile(np.array([[0.0, 0.0, dist]], dtype=np.float32), (n_views, 1))
    return R, T


# ----------------------------------------------------------------------------- render


def project_verts(verts, R, T):
    verts = _f32(verts)
    out = np.empty_like(verts)
    R = _f32(R)
    T = _f32(T)
    lib().ref_project_verts(_p(verts, F32P), verts.shape[0], _p(R, F32P), _p(T, F32P), INV_TAN_HALF_FOV, _p(out, F32P))
    return out


def rasterize(verts_ndc, faces, S, blur_radius=0.0, nthreads=1, naive=False):
    """naive=True: every face for every pixel; default: per-row candidate lists (same tests, same results)."""
    verts_ndc = _f32(verts_ndc)
    faces = _i32(faces)
    p2f = np.empty((S, S), np.int32)
    zbuf = np.empty((S, S), np.float32)
    bary = np.empty((S, S, 3), np.float32)
    dists = np.empty((S, S), np.float32)
    fn = lib().ref_rasterize_naive if naive else lib().ref_rasterize
    fn(_p(verts_ndc, F32P), _p(faces, I32P), faces.shape[0], S, blur_radius, nthreads,
       _p(p2f, I32P), _p(zbuf, F32P), _p(bary, F32P), _p(dists, F32P))
    return p2f, zbuf, bary, dists


def rasterize_k(verts_ndc, faces, S, K, blur_radius=0.0, clip_bary=None, nthreads=1, cull_backfaces=False,
                perspective_correct=True, z_clip=None, return_slots=False):
    """General soft rasteriser: (S,S,K) fragments sorted by depth.  clip_bary=None follows
    PyTorch3D's default (clip when blur_radius > 0)."""
    assert 1 <= K <= 16
    if clip_bary is None:
        clip_bary = blur_radius > 0.0
    verts_ndc = _f32(verts_ndc)
    faces = _i32(faces)
    p2f = np.empty((S, S, K), np.int32)
    zbuf = np.empty((S, S, K), np.float32)
    bary = np.empty((S, S, K, 3), np.float32)
    dists = np.empty((S, S, K), np.float32)
    if z_clip is None and not return_slots:
        lib().ref_rasterize_k2(_p(verts_ndc, F32P), _p(faces, I32P), faces.shape[0], S, K, blur_radius, int(bool(clip_bary)),
                               int(bool(cull_backfaces)), int(bool(perspective_correct)), nthreads, _p(p2f, I32P), _p(zbuf, F32P),
                               _p(bary, F32P), _p(dists, F32P))
        return p2f, zbuf, bary, dists
    # near-plane clipping (PyTorch3D: z_clip_value = znear / 2 for perspective cameras)
    slots = np.empty((S, S, K), np.int32)
    lib().ref_rasterize_k3(_p(verts_ndc, F32P), _p(faces, I32P), faces.shape[0], S, K, blur_radius, int(bool(clip_bary)),
                           int(bool(cull_backfaces)), int(bool(perspective_correct)), -1.0 if z_clip is None else float(z_clip),
                           nthreads, _p(p2f, I32P), _p(zbuf, F32P), _p(bary, F32P), _p(dists, F32P), _p(slots, I32P))
    return (p2f, zbuf, bary, dists, slots) if return_slots else (p2f, zbuf, bary, dists)


def shade_fwd(frag, verts_uvs, faces_uvs, texture):
    p2f, zbuf, bary, dists = frag
    S = p2f.shape[0]
    verts_uvs = _f32(verts_uvs)
    faces_uvs = _i32(faces_uvs)
    texture = _f32(texture)
    T = texture.shape[0]
    assert texture.shape == (T, T, 3)
    rgb = np.empty((3, S, S), np.float32)
    mask = np.empty((1, S, S), np.float32)
    lib().ref_shade_fwd(_p(p2f, I32P), _p(bary, F32P), _p(zbuf, F32P), _p(dists, F32P), _p(verts_uvs, F32P),
                        _p(faces_uvs, I32P), _p(texture, F32P), S, T, _p(rgb, F32P), _p(mask, F32P))
    return rgb, mask


def shade_bwd(grad_rgb, frag, verts_uvs, faces_uvs, texture, grad_texture=None, want_uv=False):
    p2f, zbuf, bary, dists = frag
    S = p2f.shape[0]
    grad_rgb = _f32(grad_rgb)
    verts_uvs = _f32(verts_uvs)
    faces_uvs = _i32(faces_uvs)
    texture = _f32(texture)
    T = texture.shape[0]
    if grad_texture is None:
        grad_texture = np.zeros((T, T, 3), np.float64)
    guv = np.empty((S, S, 2), np.float32) if want_uv else None
    lib().ref_shade_bwd(_p(grad_rgb, F32P), _p(p2f, I32P), _p(bary, F32P), _p(zbuf, F32P), _p(dists, F32P),
                        _p(verts_uvs, F32P), _p(faces_uvs, I32P), _p(texture, F32P), S, T,
                        _p(grad_texture, F64P), _p(guv, F32P) if want_uv else None)
    return (grad_texture, guv) if want_uv else grad_texture


def render_views(verts, faces, verts_uvs, faces_uvs, texture, R, T, S, nthreads=1):
    """CPU restatement of utils.py:65-77 for a batch of cameras: -> (B,3,S,S), (B,1,S,S), frags."""
    imgs, masks, frags = [], [], []
    for b in range(R.shape[0]):
        ndc = project_verts(verts, R[b], T[b])
        frag = rasterize(ndc, faces, S, 0.0, nthreads)
        rgb, m = shade_fwd(frag, verts_uvs, faces_uvs, texture)
        imgs.append(rgb)
        masks.append(m)
        frags.append(frag)
    return np.stack(imgs), np.stack(masks), frags


def adam_step(p, g, m, v, step, lr=0.01, b1=0.9, b2=0.999, eps=1e-8):
    for a in (p, g, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    lib().ref_adam_step(_p(p, F32P), _p(g, F32P), _p(m, F32P), _p(v, F32P), p.size, step, lr, b1, b2, eps)


# ----------------------------------------------------------------------------- vertex path


def uv_to_bary_grad(grad_uv, p2f, verts_uvs, faces_uvs):
    S = p2f.shape[0]
    out = np.empty((S, S, 3), np.float32)
    grad_uv, verts_uvs, faces_uvs = _f32(grad_uv), _f32(verts_uvs), _i32(faces_uvs)
    lib().ref_uv_to_bary_grad(_p(grad_uv, F32P), _p(p2f, I32P), _p(verts_uvs, F32P), _p(faces_uvs, I32P), S, _p(out, F32P))
    return out


def raster_bwd(grad_bary, p2f, verts_ndc, faces, grad_verts_ndc=None):
    S = p2f.shape[0]
    verts_ndc, faces, grad_bary = _f32(verts_ndc), _i32(faces), _f32(grad_bary)
    if grad_verts_ndc is None:
        grad_verts_ndc = np.zeros(verts_ndc.shape, np.float64)
    lib().ref_raster_bwd(_p(grad_bary, F32P), _p(p2f, I32P), _p(verts_ndc, F32P), _p(faces, I32P), S, _p(grad_verts_ndc, F64P))
    return grad_verts_ndc


def project_verts_bwd(verts, R, T, grad_ndc, grad_verts=None):
    verts, R, T = _f32(verts), _f32(R), _f32(T)
    grad_ndc = np.ascontiguousarray(grad_ndc, dtype=np.float64)
    if grad_verts is None:
        grad_verts = np.zeros(verts.shape, np.float64)
    lib().ref_project_verts_bwd(_p(verts, F32P), verts.shape[0], _p(R, F32P), _p(T, F32P), INV_TAN_HALF_FOV,
                                _p(grad_ndc, F64P), _p(grad_verts, F64P))
    return grad_verts


def render_bwd_views(grad_imgs, frags, verts, faces, verts_uvs, faces_uvs, texture, R, T):
    """d loss / d (texture, verts) for a batch of views given d loss / d images (B,3,S,S)."""
    gtex = np.zeros(texture.shape, np.float64)
    gverts = np.zeros(np.asarray(verts).shape, np.float64)
    for b in range(R.shape[0]):
        _, guv = shade_bwd(grad_imgs[b], frags[b], verts_uvs, faces_uvs, texture, gtex, want_uv=True)
        gb = uv_to_bary_grad(guv, frags[b][0], verts_uvs, faces_uvs)
        ndc = project_verts(verts, R[b], T[b])
        gndc = raster_bwd(gb, frags[b][0], ndc, faces)
        project_verts_bwd(verts, R[b], T[b], gndc, gverts)
    return gtex, gverts
