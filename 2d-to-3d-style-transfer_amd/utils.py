"""Host helpers with the reference's ``utils.py`` surface (names, arguments, return layouts of
reference utils.py:19-210) implemented on libst3d -- no torchvision, no PyTorch3D.

What differs from the reference, none of it visible in results:
  * ``render_meshes`` submits every camera of the batch in one set of launches (the reference calls the
    renderer once per camera, :68-69) and takes NCHW colour + coverage straight from the shade kernel
    instead of slicing and permuting an RGBA image (:70-76);
  * ``get_vgg`` never downloads: a local state_dict (argument or ST3D_VGG19_WEIGHTS) or seeded weights;
  * ``setup_optimizations`` returns the fused HIP Adam (st3d.optim.Adam), which also sums the gradient
    over ranks when the view batch is sharded across GPUs.
"""
import os
import random

import numpy as np
import torch
from PIL import Image

from st3d import ops as _ops
from st3d import optim as _st3d_optim
from st3d import render as _render
from st3d import vgg as _vgg
from st3d.render import FoVPerspectiveCameras, Meshes, RotateAxisAngle, TexturesUV, look_at_view_transform  # noqa: F401
from style_transfer import *  # noqa: F401,F403  (star re-export relied on by the CLIs, reference utils.py:12)

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


# ------------------------------------------------------------------------------------------ compositing
class _Composite(torch.autograd.Function):
    """out = img * mask + bg * (1 - mask) in one launch; d out / d img = mask."""

    @staticmethod
    def forward(ctx, img, mask, bg):
        ctx.mask = mask
        return _ops.apply_background(img.detach(), mask, bg)

    @staticmethod
    def backward(ctx, grad_out):
        return _ops.apply_background(grad_out.contiguous(), ctx.mask, None), None, None


def apply_background(tensors, masks, background_type='noise', background=None):
    """reference :19-30 -- 'white' is the identity because the renderer's own background is white;
    'noise' draws a fresh U[0,1) image per call; any other type returns None like the reference."""
    if background_type == 'white':
        return tensors
    if background_type == 'style':
        return _Composite.apply(tensors, masks, background)
    if background_type == 'noise':
        return _Composite.apply(tensors, masks, torch.rand(tensors.shape, device=tensors.device))
    return None


# ------------------------------------------------------------------------------------------ images
def load_as_tensor(image_path, size=512):
    """File -> (3,size,size) float32 in [0,1] on `device`: RGB decode, PIL's antialiased bilinear
    resize, /255 -- what transforms.Resize((size,size)) + ToTensor() give for a PIL image (:34-44)."""
    with Image.open(image_path) as im:
        rgb = im.convert('RGB').resize((size, size), Image.BILINEAR)
    hwc = torch.from_numpy(np.array(rgb, dtype=np.uint8))
    return (hwc.permute(2, 0, 1).float() / 255.0).contiguous().to(device)


def tensor_to_image(tensor):
    """(1,3,H,W) or (3,H,W) -> PIL image; values clamped to [0,1], scaled by 255 and truncated
    (ToPILImage semantics, :56-61)."""
    chw = tensor.detach().reshape(tensor.shape[-3:]).clamp(0, 1).cpu()
    return Image.fromarray((chw * 255).to(torch.uint8).permute(1, 2, 0).numpy())


def get_vgg(weights=None, seed=0):
    """Frozen VGG-19 feature stack (:48-52) as an st3d.vgg.Vgg19Features."""
    return _vgg.get_vgg(weights=weights, device=device, seed=seed)


# ------------------------------------------------------------------------------------------ rendering
def render_meshes(renderer, meshes, cameras):
    """-> colour (n,3,H,W), mask (n,1,H,W) with mask = (alpha > 0) as float (:65-77)."""
    colour, coverage = renderer.render(meshes, cameras)
    if not renderer.is_hard:                 # soft settings hand back alpha itself
        coverage = (coverage.detach() > 0).float()
    return colour, coverage


def save_render(renderer, meshes, cameras, path):
    """view_<k>.png for every camera (:81-91)."""
    os.makedirs(path, exist_ok=True)
    colour, _ = render_meshes(renderer, meshes, cameras)
    for k, img in enumerate(colour):
        tensor_to_image(img).save(os.path.join(path, f"view_{k}.png"))


def finalize_tensor(tensor):
    return tensor.detach().clamp(0.0, 1.0)


def finalize_mesh(mesh):
    """Same geometry and UVs, texture clamped to displayable range and detached (:95-104)."""
    tex = mesh.textures
    clamped = TexturesUV(maps=finalize_tensor(tex.maps_padded()), faces_uvs=tex.faces_uvs_padded(),
                         verts_uvs=tex.verts_uvs_padded())
    return Meshes(verts=mesh.verts_padded(), faces=mesh.faces_padded(), textures=clamped)


def build_mesh(verts_uvs, faces_uvs, texture_map, verts, faces):
    """Fresh containers around the (possibly leaf) tensors, rebuilt every step like the reference (:207-210)."""
    return Meshes(verts=[verts], faces=[faces],
                  textures=TexturesUV(maps=texture_map, faces_uvs=faces_uvs, verts_uvs=verts_uvs))


# ------------------------------------------------------------------------------------------ cameras
def build_fixed_cameras(n_views, dist=3.0, shuffle=True):
    """Turntable: the first n//2 views rotate the object about X by linspace(0,315), the rest about Y by
    linspace(45,315); the camera sits at T=(0,0,dist); order shuffled with Python's RNG (:121-151)."""
    n_x = n_views // 2
    plan = [("X", float(a)) for a in torch.linspace(0, 315, n_x)]
    plan += [("Y", float(a)) for a in torch.linspace(45, 315, n_views - n_x)]
    if shuffle:
        random.shuffle(plan)
    rot = [RotateAxisAngle(angle, axis=axis).get_matrix()[0, :3, :3] for axis, angle in plan]
    R = torch.stack(rot) if rot else torch.zeros(0, 3, 3)
    T = torch.tensor([0.0, 0.0, dist]).expand(len(plan), 3).clone()
    return FoVPerspectiveCameras(R=R, T=T, device=device)


def build_random_cameras(n_views, dist=2.10, generator=None):
    """Uniform directions on the sphere around the look-at point (0, 0.10, 0.25): cos(polar) ~ U(-1,1),
    azimuth ~ U(-180,180) degrees (:154-170).  `generator` seeds the draws (the reference is unseeded)."""
    u = torch.rand(n_views, generator=generator)
    elev = torch.acos(2.0 * u - 1.0) * 180 / torch.pi - 90        # same op order as the reference: bit-equal angles
    azim = 360.0 * torch.rand(n_views, generator=generator) - 180.0
    R, T = look_at_view_transform(dist=dist, elev=elev, azim=azim, at=((0, 0.10, 0.25),))
    return FoVPerspectiveCameras(R=R, T=T, device=device)


# ------------------------------------------------------------------------------------------ optimiser
_LEAVES = {'texture': ('texture_map',), 'mesh': ('verts',), 'both': ('verts', 'texture_map')}


def setup_optimizations(optimization_target, mesh, lr):
    """Clone the mesh, mark the tensors `optimization_target` names as leaves and put ONE Adam (single lr,
    default betas/eps) over them (:173-204).  Keys of the returned dict are the reference's."""
    work = mesh.clone()
    parts = {
        'texture_map': work.textures.maps_padded(),
        'verts': work.verts_packed(),
        'faces': work.faces_packed(),
        'verts_uvs': work.textures.verts_uvs_padded(),
        'faces_uvs': work.textures.faces_uvs_padded(),
    }
    if optimization_target not in _LEAVES:      # the reference falls through to an unbound `optimizer`
        raise UnboundLocalError("local variable 'optimizer' referenced before assignment "
                                f"(optimization_target={optimization_target!r})")
    leaves = [parts[name].requires_grad_(True) for name in _LEAVES[optimization_target]]
    return dict(parts, optimizable_mesh=work, optimizer=_st3d_optim.Adam(leaves, lr=lr))
