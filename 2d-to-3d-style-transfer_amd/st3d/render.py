"""Scene containers, cameras and the differentiable renderer on libst3d.

Host-side mirror of the PyTorch3D objects the reference builds (first_approach.py:104-113,
second_approach.py:98-108, utils.py:6-9,121-170,207-210): thin structs with the same
constructor arguments and accessor names, so the drop-in ``utils.py`` / approach scripts read
like the reference.  Rendering itself is NOT a per-camera Python loop of ~20 small launches as
upstream: all views of a batch go through four HIP launches (project, face setup + tile
raster, fused shade) and the backward is one launch (texture scatter).

The reference's own configuration (blur_radius=0, faces_per_pixel=1, default BlendParams;
first_approach.py:107) runs on the specialised hard kernels.  Any other RasterizationSettings /
BlendParams (K <= 8 faces per pixel, blur_radius > 0, clipped barycentrics, sigma/gamma/background)
runs on the general soft rasteriser + softmax blend (csrc/soft.hip).  Lights:
AmbientLights only; cameras: FoV perspective with default fov/znear/zfar.
"""
import math

import torch

from . import ops

# ------------------------------------------------------------------------ containers

_I32_CACHE = {}


def _checked_i32(idx, limit, what):
    """int64 index tensor -> validated contiguous int32 copy, cached on the tensor's identity:
    the reference rebuilds its Meshes every step (second_approach.py:164) from the SAME index
    tensors, and the range check costs a host sync, so it must not be repeated per step.
    (An out-of-range index would fault the GPU inside the raster/shade kernels.)"""
    key = (idx.data_ptr(), tuple(idx.shape), idx._version, int(limit), str(idx.device))
    hit = _I32_CACHE.get(key)
    if hit is None:
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= limit):
            raise ValueError(f"{what} holds indices outside [0, {limit})")
        if len(_I32_CACHE) > 64:
            _I32_CACHE.clear()
        hit = (idx.to(torch.int32).contiguous(), idx)        # keep the source alive: data_ptr stays unique
        _I32_CACHE[key] = hit
    return hit[0]



class TexturesUV:
    """utils.py:208 ``TexturesUV(verts_uvs=..., faces_uvs=..., maps=...)``; maps (1,T,T,3)."""

    def __init__(self, maps, faces_uvs, verts_uvs):
        if isinstance(maps, (list, tuple)):
            maps = torch.stack(list(maps))
        if isinstance(verts_uvs, (list, tuple)):
            verts_uvs = torch.stack(list(verts_uvs))
        if isinstance(faces_uvs, (list, tuple)):
            faces_uvs = torch.stack(list(faces_uvs))
        self._maps = maps if maps.dim() == 4 else maps[None]
        self._verts_uvs = verts_uvs if verts_uvs.dim() == 3 else verts_uvs[None]
        self._faces_uvs = faces_uvs if faces_uvs.dim() == 3 else faces_uvs[None]
        self._faces_uvs_i32 = None

    def maps_padded(self):
        return self._maps

    def verts_uvs_padded(self):
        return self._verts_uvs

    def faces_uvs_padded(self):
        return self._faces_uvs

    def faces_uvs_i32(self):
        if self._faces_uvs_i32 is None:
            self._faces_uvs_i32 = _checked_i32(self._faces_uvs[0], self._verts_uvs.shape[1], "faces_uvs")
        return self._faces_uvs_i32

    def clone(self):
        t = TexturesUV(self._maps.clone(), self._faces_uvs.clone(), self._verts_uvs.clone())
        return t

    def detach(self):
        return TexturesUV(self._maps.detach(), self._faces_uvs, self._verts_uvs.detach())


class Meshes:
    """utils.py:209 ``Meshes(verts=[verts], faces=[faces], textures=textures)`` (one mesh)."""

    def __init__(self, verts, faces, textures=None):
        if isinstance(verts, (list, tuple)):
            assert len(verts) == 1, "one mesh per batch (the reference never batches meshes)"
            verts = verts[0]
        if isinstance(faces, (list, tuple)):
            faces = faces[0]
        if verts.dim() == 3:
            assert verts.shape[0] == 1
            self._verts_padded = verts
            self._verts = None
        else:
            self._verts = verts
            self._verts_padded = None
        self._faces = faces[0] if faces.dim() == 3 else faces
        self.textures = textures
        self._faces_i32 = None

    def verts_packed(self):
        if self._verts is None:
            self._verts = self._verts_padded[0]
        return self._verts

    def verts_padded(self):
        if self._verts_padded is None:
            self._verts_padded = self._verts[None]
        return self._verts_padded

    def faces_packed(self):
        return self._faces

    def faces_padded(self):
        return self._faces[None]

    def faces_i32(self):
        if self._faces_i32 is None:
            self._faces_i32 = _checked_i32(self._faces, self.verts_packed().shape[0], "faces")
        return self._faces_i32

    def clone(self):
        return Meshes(self.verts_packed().clone(), self._faces.clone(),
                      self.textures.clone() if self.textures is not None else None)

    def detach(self):
        return Meshes(self.verts_packed().detach(), self._faces,
                      self.textures.detach() if self.textures is not None else None)

    @property
    def device(self):
        return self.verts_packed().device


# ------------------------------------------------------------------------ cameras


class FoVPerspectiveCameras:
    """R (n,3,3), T (n,3) in the row-vector convention X_view = X_world R + T; defaults
    fov=60 deg, znear=1, zfar=100, aspect=1 (the reference never overrides them)."""

    def __init__(self, R=None, T=None, device="cpu", fov=60.0, znear=1.0, zfar=100.0):
        dev = torch.device(device)
        self.R = (torch.eye(3)[None] if R is None else R).to(dev, torch.float32).reshape(-1, 3, 3)
        self.T = (torch.zeros(1, 3) if T is None else T).to(dev, torch.float32).reshape(-1, 3)
        if (fov, znear, zfar) != (60.0, 1.0, 100.0):
            raise NotImplementedError("only the FoVPerspectiveCameras defaults the reference uses are supported")
        self.device = dev

    def __len__(self):
        return self.R.shape[0]

    def __getitem__(self, i):
        if isinstance(i, int):
            i = slice(i, i + 1)
        return FoVPerspectiveCameras(self.R[i], self.T[i], device=self.device)


def join_cameras(cameras):
    """list of cameras (as utils.py:68 iterates) or one batched camera -> (R (B,3,3), T (B,3))."""
    if isinstance(cameras, FoVPerspectiveCameras):
        return cameras.R, cameras.T
    return torch.cat([c.R for c in cameras], 0), torch.cat([c.T for c in cameras], 0)


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, at=((0, 0, 0),), up=((0, 1, 0),), device="cpu"):
    """PyTorch3D look_at_view_transform with degrees=True (utils.py:161-166)."""
    def _t(x):
        x = torch.as_tensor(x, dtype=torch.float32)
        return x.reshape(-1) if x.dim() <= 1 else x
    dist, elev, azim = _t(dist), _t(elev) * (math.pi / 180.0), _t(azim) * (math.pi / 180.0)
    at = torch.as_tensor(at, dtype=torch.float32).reshape(-1, 3)
    up = torch.as_tensor(up, dtype=torch.float32).reshape(-1, 3)
    n = max(dist.numel(), elev.numel(), azim.numel(), at.shape[0])
    dist, elev, azim = (t.expand(n) for t in (dist, elev, azim))
    at, up = at.expand(n, 3), up.expand(n, 3)
    x = dist * torch.cos(elev) * torch.sin(azim)
    y = dist * torch.sin(elev)
    z = dist * torch.cos(elev) * torch.cos(azim)
    C = torch.stack([x, y, z], dim=1) + at
    z_axis = torch.nn.functional.normalize(at - C, eps=1e-5)
    x_axis = torch.nn.functional.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = torch.nn.functional.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    is_close = torch.isclose(x_axis, torch.tensor(0.0), atol=5e-3).all(dim=1, keepdim=True)
    if is_close.any():
        repl = torch.nn.functional.normalize(torch.cross(y_axis, z_axis, dim=1), eps=1e-5)
        x_axis = torch.where(is_close, repl, x_axis)
    R = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1).transpose(1, 2)
    T = -torch.bmm(R.transpose(1, 2), C[:, :, None])[:, :, 0]
    return R.to(device), T.to(device)


class RotateAxisAngle:
    """utils.py:142 ``RotateAxisAngle(angle, axis=axis).get_matrix()[..., :3, :3]``."""

    def __init__(self, angle, axis="X", degrees=True, device="cpu"):
        a = float(angle) * (math.pi / 180.0 if degrees else 1.0)
        c, s = math.cos(a), math.sin(a)
        if axis == "X":
            m = [[1, 0, 0], [0, c, -s], [0, s, c]]
        elif axis == "Y":
            m = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
        elif axis == "Z":
            m = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
        else:
            raise ValueError("axis must be X, Y or Z")
        M = torch.eye(4, dtype=torch.float32)
        M[:3, :3] = torch.tensor(m, dtype=torch.float32).t()     # row-vector convention
        self._m = M[None].to(device)

    def get_matrix(self):
        return self._m


# ------------------------------------------------------------------------ renderer


class RasterizationSettings:
    """PyTorch3D RasterizationSettings: image_size, blur_radius, faces_per_pixel, clip_barycentric_coords
    (None = clip iff blur_radius > 0, the PyTorch3D default), perspective_correct (None = True: every camera here is a
    perspective camera), cull_backfaces.  bin_size / max_faces_per_bin only pick PyTorch3D's binning strategy and are
    accepted and ignored.  Anything but the reference's own values (first_approach.py:107) runs on the general kernels."""
    MAX_FACES_PER_PIXEL = 8

    Z_CLIP_DEFAULT = 0.5        # PyTorch3D MeshRasterizer: z_clip_value None -> znear / 2 for perspective cameras (znear = 1)

    def __init__(self, image_size=256, blur_radius=0.0, faces_per_pixel=1, bin_size=None, max_faces_per_bin=None,
                 perspective_correct=None, clip_barycentric_coords=None, cull_backfaces=False, z_clip_value=None,
                 cull_to_frustum=False, **kw):
        if isinstance(image_size, (tuple, list)):
            if len(image_size) != 2 or image_size[0] != image_size[1]:
                raise NotImplementedError("square images only")
            image_size = image_size[0]
        if not 1 <= int(faces_per_pixel) <= self.MAX_FACES_PER_PIXEL:
            raise NotImplementedError(f"faces_per_pixel must be in 1..{self.MAX_FACES_PER_PIXEL}")
        if blur_radius < 0.0:
            raise ValueError("blur_radius must be >= 0")
        self.image_size, self.blur_radius, self.faces_per_pixel = int(image_size), float(blur_radius), int(faces_per_pixel)
        self.clip_barycentric_coords = (self.blur_radius > 0.0) if clip_barycentric_coords is None \
            else bool(clip_barycentric_coords)
        self.perspective_correct = True if perspective_correct is None else bool(perspective_correct)
        self.cull_backfaces = bool(cull_backfaces)
        if cull_to_frustum:
            raise NotImplementedError("cull_to_frustum=True is not implemented (PyTorch3D's default is False)")
        if z_clip_value is not None and not z_clip_value > 0.0:
            raise ValueError("z_clip_value must be positive")
        # None: PyTorch3D's default plane.  The general kernels clip at it; the specialised K = 1 kernels only WATCH it
        # (st3d.ops.check_near_plane) -- an explicit value sends the render to the general kernels
        self.z_clip_value = None if z_clip_value is None else float(z_clip_value)

    @property
    def z_clip(self):
        return self.Z_CLIP_DEFAULT if self.z_clip_value is None else self.z_clip_value

    @property
    def is_hard(self):
        return (self.faces_per_pixel == 1 and self.blur_radius == 0.0 and not self.clip_barycentric_coords
                and self.perspective_correct and not self.cull_backfaces and self.z_clip_value is None)


class BlendParams:
    """PyTorch3D BlendParams (sigma, gamma, background_color) for softmax_rgb_blend."""

    def __init__(self, sigma=1e-4, gamma=1e-4, background_color=(1.0, 1.0, 1.0)):
        self.sigma, self.gamma = float(sigma), float(gamma)
        bg = torch.as_tensor(background_color, dtype=torch.float32).reshape(-1).tolist()
        if len(bg) != 3:
            raise ValueError("background_color must have 3 components")
        self.background_color = tuple(bg)
        if self.sigma <= 0.0 or self.gamma <= 0.0:
            raise ValueError("sigma and gamma must be positive")


class AmbientLights:
    def __init__(self, ambient_color=((1.0, 1.0, 1.0),), device="cpu"):
        c = torch.as_tensor(ambient_color, dtype=torch.float32).reshape(-1)
        if not torch.allclose(c, torch.ones(3)):
            raise NotImplementedError("only the default white ambient light is supported")


class MeshRasterizer:
    def __init__(self, cameras=None, raster_settings=None):
        self.cameras, self.raster_settings = cameras, raster_settings or RasterizationSettings()


class SoftPhongShader:
    def __init__(self, device="cpu", cameras=None, lights=None, materials=None, blend_params=None, **kw):
        self.cameras, self.lights = cameras, lights
        self.blend_params = blend_params if blend_params is not None else BlendParams()


class _RenderFn(torch.autograd.Function):
    """(verts, texture_map) -> (rgb (B,3,S,S), mask (B,1,S,S)); backward = texture scatter and,
    when the vertices need a gradient, shade d/d(bary) -> raster backward -> projection backward."""

    @staticmethod
    def forward(ctx, verts, tex_map, faces_i32, verts_uvs, faces_uvs_i32, R, T, S):
        v = verts.detach().to(torch.float32).contiguous()
        tex = tex_map.detach().to(torch.float32).reshape(tex_map.shape[-3], tex_map.shape[-2], 3).contiguous()
        if tex.shape[0] != tex.shape[1]:
            raise NotImplementedError("square texture maps only (the reference resizes to size x size)")
        uvs = verts_uvs.detach().to(torch.float32).reshape(-1, 2).contiguous()
        ndc = ops.project_verts(v, R, T)
        # (no near-plane watch here: render_views has looked at the vertices' depths before choosing these kernels)
        frag = ops.raster_fwd(ndc, faces_i32, S)
        rgb, mask = ops.shade_fwd(frag, uvs, faces_uvs_i32, tex)
        ctx.frag, ctx.uvs, ctx.fuv, ctx.tex = frag, uvs, faces_uvs_i32, tex
        ctx.tex_shape = tex_map.shape
        ctx.geom = (v, ndc, faces_i32, R, T)
        ctx.verts_shape = verts.shape
        ctx.mark_non_differentiable(mask)
        return rgb, mask

    @staticmethod
    def backward(ctx, grad_rgb, _grad_mask):
        with ops.trace("render_backward"):
            return _RenderFn._backward(ctx, grad_rgb, _grad_mask)

    @staticmethod
    def _backward(ctx, grad_rgb, _grad_mask):
        need_v, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gtex = gverts = None
        if need_v or need_t:
            res = ops.shade_bwd(grad_rgb.to(torch.float32), ctx.frag, ctx.uvs, ctx.fuv, ctx.tex, want_bary=need_v,
                                want_texture=need_t)
            gt, gbary = (res if need_v else (res, None))
            if need_t:
                gtex = gt.reshape(ctx.tex_shape)
            if need_v:      # uv -> barycentrics -> projected vertices -> world vertices
                v, ndc, faces_i32, R, T = ctx.geom
                gndc = ops.raster_bwd(gbary, ctx.frag[0], ndc, faces_i32)
                gverts = ops.project_verts_bwd(v, R, T, gndc).reshape(ctx.verts_shape)
        return gverts, gtex, None, None, None, None, None, None


class _SoftRenderFn(torch.autograd.Function):
    """General path: K faces per pixel, blur_radius, softmax_rgb_blend -> (rgb (B,3,S,S), alpha (B,1,S,S)).
    Gradients flow from rgb to the texture and, through barycentrics, depth and the signed edge distance, to
    the vertices; alpha is returned without a gradient (the reference only ever thresholds it, utils.py:72)."""

    @staticmethod
    def forward(ctx, verts, tex_map, faces_i32, verts_uvs, faces_uvs_i32, R, T, S, K, blur, clip, sigma, gamma, bg,
                cull=False, persp=True, z_clip=None):
        v = verts.detach().to(torch.float32).contiguous()
        tex = tex_map.detach().to(torch.float32).reshape(tex_map.shape[-3], tex_map.shape[-2], 3).contiguous()
        if tex.shape[0] != tex.shape[1]:
            raise NotImplementedError("square texture maps only (the reference resizes to size x size)")
        uvs = verts_uvs.detach().to(torch.float32).reshape(-1, 2).contiguous()
        ndc = ops.project_verts(v, R, T)
        frag = ops.raster_soft_fwd(ndc, faces_i32, S, K, blur, clip, cull, persp, z_clip)
        ctx.slots = frag[4] if z_clip is not None else None
        ctx.z_clip = z_clip
        frag = frag[:4]
        rgb, alpha = ops.shade_soft_fwd(frag, uvs, faces_uvs_i32, tex, sigma, gamma, bg)
        ctx.persp = persp
        ctx.frag, ctx.uvs, ctx.fuv, ctx.tex = frag, uvs, faces_uvs_i32, tex
        ctx.blend = (sigma, gamma, bg)
        ctx.clip = clip
        ctx.tex_shape, ctx.verts_shape = tex_map.shape, verts.shape
        ctx.geom = (v, ndc, faces_i32, R, T)
        ctx.mark_non_differentiable(alpha)
        return rgb, alpha

    @staticmethod
    def backward(ctx, grad_rgb, _grad_alpha):
        with ops.trace("render_backward"):
            return _SoftRenderFn._backward(ctx, grad_rgb, _grad_alpha)

    @staticmethod
    def _backward(ctx, grad_rgb, _grad_alpha):
        need_v, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gtex = gverts = None
        if need_v or need_t:
            sigma, gamma, bg = ctx.blend
            gt, geo = ops.shade_soft_bwd(grad_rgb.to(torch.float32), ctx.frag, ctx.uvs, ctx.fuv, ctx.tex, sigma, gamma, bg,
                                         want_texture=need_t, want_geometry=need_v)
            if need_t:
                gtex = gt.reshape(ctx.tex_shape)
            if need_v:
                v, ndc, faces_i32, R, T = ctx.geom
                gndc = ops.raster_soft_bwd(geo, ctx.frag[0], ndc, faces_i32, ctx.clip, ctx.persp, ctx.slots, ctx.z_clip)
                gverts = ops.project_verts_bwd(v, R, T, gndc).reshape(ctx.verts_shape)
        return (gverts, gtex) + (None,) * 15


def uses_hard_path(raster_settings, blend_params):
    """True for the reference's own configuration (K=1, blur 0, unclipped, default BlendParams): there
    softmax_rgb_blend reduces to texel-or-white and the specialised kernels apply."""
    rs, bp = raster_settings, blend_params
    if rs is not None and not rs.is_hard:
        return False
    return bp is None or (bp.sigma, bp.gamma, bp.background_color) == (1e-4, 1e-4, (1.0, 1.0, 1.0))


_DEPTH_CACHE = {}


def reaches_near_plane(verts, R, T, z_clip):
    """True if any vertex lies nearer than `z_clip` to any of the cameras (view depth = X_world R[:, :, 2] + T[2]).
    PyTorch3D clips every mesh at z_clip before rasterising; the specialised K = 1 kernels do not, so render_views asks
    BEFORE it renders and sends such a batch to the clipping kernels -- no frame is ever rendered unclipped.  The answer
    needs one host read of a device scalar: for tensors that are not being optimised it is cached on their identity and
    version (texture-only runs: one read per camera batch, ever); vertices under optimisation are asked every step (the
    read waits for the previous step, ~1 % of a config-5 step)."""
    v = verts.detach()
    key = None
    if not verts.requires_grad:
        key = (v.data_ptr(), v._version, tuple(v.shape), R.data_ptr(), R._version, T.data_ptr(), T._version, R.shape[0],
               float(z_clip), str(v.device))
        hit = _DEPTH_CACHE.get(key)
        if hit is not None:
            return hit[0]
    zmin = (v.to(torch.float32) @ R[:, :, 2].to(torch.float32).t() + T[:, 2].to(torch.float32)).min()
    near = bool(zmin.item() < z_clip)
    if key is not None:
        if len(_DEPTH_CACHE) > 256:
            _DEPTH_CACHE.clear()
        _DEPTH_CACHE[key] = (near, verts, R, T)         # keep the keyed tensors alive: their addresses stay unique
    return near


def render_views(meshes, R, T, image_size, raster_settings=None, blend_params=None):
    """``_render_views`` inside a named range (``ST3D_ROCTX=1``: rocprofv3 --marker-trace shows the step's phases)."""
    with ops.trace("render"):
        return _render_views(meshes, R, T, image_size, raster_settings, blend_params)


def _render_views(meshes, R, T, image_size, raster_settings=None, blend_params=None):
    """All B views in one batch of launches -> (rgb (B,3,S,S), coverage (B,1,S,S)).  Coverage is the 0/1 mask under
    the reference's hard settings (whichever kernels render them) and softmax_rgb_blend's alpha under soft settings;
    both satisfy ``coverage > 0`` == covered."""
    tex = meshes.textures
    dev = meshes.device
    rs, bp = raster_settings, blend_params
    R, T = R.to(dev), T.to(dev)
    hard_settings = uses_hard_path(rs, bp)
    if hard_settings and (ops.near_plane_triggered() or
                          reaches_near_plane(meshes.verts_packed(), R, T, RasterizationSettings.Z_CLIP_DEFAULT)):
        # the specialised kernels do not clip: a mesh at the near plane renders with the same settings on the general
        # kernels at PyTorch3D's default clipping depth.  Every rank decides for its own views; both kernel families give
        # the same pixels where nothing is clipped, so ranks need not agree.
        if ops.NEAR_PLANE_POLICY == "raise":
            raise RuntimeError(ops.NEAR_PLANE_MESSAGE)
        ops.note_near_plane()
        rs = RasterizationSettings(image_size=image_size, z_clip_value=RasterizationSettings.Z_CLIP_DEFAULT)
    if uses_hard_path(rs, bp):
        # K=1, blur 0: the blend weight cancels and the pixel is the sampled texel itself
        return _RenderFn.apply(meshes.verts_packed(), tex.maps_padded(), meshes.faces_i32(), tex.verts_uvs_padded(),
                               tex.faces_uvs_i32(), R, T, int(image_size))
    bp = bp if bp is not None else BlendParams()
    rs = rs if rs is not None else RasterizationSettings(image_size=image_size)
    rgb, alpha = _SoftRenderFn.apply(meshes.verts_packed(), tex.maps_padded(), meshes.faces_i32(), tex.verts_uvs_padded(),
                                     tex.faces_uvs_i32(), R, T, int(image_size), rs.faces_per_pixel,
                                     rs.blur_radius, rs.clip_barycentric_coords, bp.sigma, bp.gamma, bp.background_color,
                                     rs.cull_backfaces, rs.perspective_correct, rs.z_clip)
    if hard_settings:
        # the caller asked for the hard configuration and is handed what the hard path hands out: the 0/1 coverage mask
        # (alpha of a K = 1 / blur 0 blend is in [0.5, 1) on covered pixels; the reference thresholds it, utils.py:72)
        alpha = (alpha.detach() > 0).to(torch.float32)
    return rgb, alpha


class MeshRenderer:
    """``renderer(meshes_world=mesh, cameras=camera)`` -> (n,S,S,4) RGBA like PyTorch3D's
    MeshRenderer (utils.py:69); ``render_meshes`` in the drop-in utils.py calls
    ``render`` and skips the RGBA repack."""

    def __init__(self, rasterizer, shader):
        self.rasterizer, self.shader = rasterizer, shader

    @property
    def image_size(self):
        return self.rasterizer.raster_settings.image_size

    @property
    def is_hard(self):
        return uses_hard_path(self.rasterizer.raster_settings, getattr(self.shader, "blend_params", None))

    def render(self, meshes_world, cameras=None):
        """-> (rgb (n,3,S,S), coverage (n,1,S,S)) under this renderer's raster settings and blend params."""
        cameras = cameras if cameras is not None else self.rasterizer.cameras
        R, T = join_cameras(cameras)
        return render_views(meshes_world, R, T, self.image_size, self.rasterizer.raster_settings,
                            getattr(self.shader, "blend_params", None))

    def __call__(self, meshes_world, cameras=None, **kw):
        rgb, cov = self.render(meshes_world, cameras)
        # hard path: alpha of softmax_rgb_blend with K=1 is in [0.5,1) on covered pixels, 0 elsewhere; only
        # (alpha > 0) is ever consumed (utils.py:72), so the 0/1 mask stands in for it.  Soft path: the real alpha.
        return torch.cat([rgb, cov], dim=1).permute(0, 2, 3, 1)

    def to(self, *a, **k):
        return self
