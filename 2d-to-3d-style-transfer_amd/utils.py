"""Drop-in for the reference's ``utils.py`` (same function names, arguments and return layouts:
reference utils.py:19-210) on libst3d -- no torchvision, no PyTorch3D.

Differences that are NOT visible in results:
  * ``render_meshes`` renders all cameras of the batch in one set of launches instead of one
    renderer call per camera (reference :68-69) and gets NCHW RGB + mask straight from the
    shade kernel instead of permuting an RGBA image (reference :70-76);
  * ``get_vgg`` never downloads: local state_dict via ST3D_VGG19_WEIGHTS, else seeded weights;
  * ``setup_optimizations`` hands back the fused HIP Adam (st3d.optim.Adam), which also sums the
    gradient over ranks when the view batch is sharded across GPUs.
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
from style_transfer import *  # noqa: F401,F403  (the reference does the same, utils.py:12)

# Check if CUDA is available
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class _BackgroundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tensors, masks, backgrounds):
        ctx.masks = masks
        return _ops.apply_background(tensors.detach(), masks, backgrounds)

    @staticmethod
    def backward(ctx, g):
        return _ops.apply_background(g.contiguous(), ctx.masks, None), None, None


# Helper function to blend image with background
def apply_background(tensors, masks, background_type = 'noise', background = None):

    if background_type == 'noise':
        backgrounds = torch.rand(tensors.shape, device = tensors.device)
        return _BackgroundFn.apply(tensors, masks, backgrounds)

    elif background_type == 'style':
        return _BackgroundFn.apply(tensors, masks, background)

    elif background_type == 'white':
        return tensors


# Load and preprocess the images
def load_as_tensor(image_path, size=512):
    """PIL RGB -> bilinear (antialiased) resize to (size,size) -> float32 CHW /255, what
    transforms.Resize + ToTensor do to a PIL image (reference :34-44)."""
    image = Image.open(image_path).convert('RGB').resize((size, size), Image.BILINEAR)
    arr = np.asarray(image, dtype=np.uint8)
    tensor = torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div(255.0)
    return tensor[:3, :, :].contiguous().to(device)


# Load the VGG19 feature extractor (frozen)
def get_vgg(weights=None, seed=0):
    return _vgg.get_vgg(weights=weights, device=device, seed=seed)


# Convert tensor to image for display
def tensor_to_image(tensor):
    image = tensor.clone().detach()
    image = image.squeeze(0)  # Remove batch dimension
    image = image.clamp(0, 1).cpu()
    # ToPILImage on a float tensor: mul(255) then truncating byte cast
    arr = image.mul(255).to(torch.uint8).permute(1, 2, 0).numpy()
    return Image.fromarray(arr)


# Render the content tensor
def render_meshes(renderer, meshes, cameras):
    tensors, coverage = renderer.render(meshes, cameras)
    if not renderer.is_hard:
        coverage = (coverage.detach() > 0).float()      # soft settings: alpha -> mask (reference :72)
    return tensors, coverage      # (BATCH, 3, H, W), (BATCH, 1, H, W) with mask = alpha > 0


# Save final optimized images
def save_render(renderer, meshes, cameras, path):

    os.makedirs(path, exist_ok=True)

    # Render optimized mesh
    tensors, _ = render_meshes(renderer, meshes, cameras)

    for i in range(tensors.shape[0]):
        tensor_to_image(tensors[i, ...]).save(f"{path}/view_{i}.png")


def finalize_mesh(mesh):
    textures = mesh.textures
    # colours clamped to (0,1); geometry and UVs as they are
    final_textures = TexturesUV(verts_uvs=textures.verts_uvs_padded(), faces_uvs=textures.faces_uvs_padded(),
                                maps=finalize_tensor(textures.maps_padded()))
    return Meshes(verts=mesh.verts_padded(), faces=mesh.faces_padded(), textures=final_textures)


def finalize_tensor(tensor):
    final_tensor = torch.clamp(tensor, 0.0, 1.0).detach()
    return final_tensor


def build_fixed_cameras(n_views, dist=3.0, shuffle = True):

    # viewpoints: half rotate about X, half about Y (reference :124-128)
    x_views = (n_views // 2)
    y_views = n_views - x_views
    angles = [(a.item(), "X") for a in torch.linspace(0, 315, x_views)] + \
             [(a.item(), "Y") for a in torch.linspace(45, 315, y_views)]

    if shuffle:
        random.shuffle(angles)

    R_list = torch.stack([RotateAxisAngle(angle, axis=axis).get_matrix()[..., :3, :3].squeeze(0) for angle, axis in angles], dim=0)
    T_list = torch.tensor([[0.0, 0.0, dist]]).repeat(len(angles), 1)

    return FoVPerspectiveCameras(R=R_list, T=T_list, device=device)


def build_random_cameras(n_views, dist=2.10, generator=None):

    cos_elevs = torch.rand(n_views, generator=generator) * 2 - 1
    elevs = torch.acos(cos_elevs) * 180 / torch.pi - 90

    azims = torch.rand(n_views, generator=generator) * 360 - 180

    R_list, T_list = look_at_view_transform(dist = dist, elev = elevs, azim = azims, at=((0, 0.10, 0.25),))

    return FoVPerspectiveCameras(R=R_list, T=T_list, device=device)


def setup_optimizations(optimization_target, mesh, lr):

    optimizable_mesh = mesh.clone()

    texture_map = optimizable_mesh.textures.maps_padded()
    verts = optimizable_mesh.verts_packed()
    faces = optimizable_mesh.faces_packed()
    verts_uvs = optimizable_mesh.textures.verts_uvs_padded()
    faces_uvs = optimizable_mesh.textures.faces_uvs_padded()

    if optimization_target == 'texture':
        texture_map.requires_grad_(True)
        optimizer = _st3d_optim.Adam([texture_map], lr=lr)

    elif optimization_target == 'mesh':
        verts.requires_grad_(True)
        optimizer = _st3d_optim.Adam([verts], lr=lr)

    elif optimization_target == 'both':
        texture_map.requires_grad_(True)
        verts.requires_grad_(True)
        optimizer = _st3d_optim.Adam([verts, texture_map], lr = lr)

    return {'optimizable_mesh': optimizable_mesh,
            'optimizer': optimizer,
            'texture_map': texture_map,
            'verts': verts,
            'faces': faces,
            'verts_uvs': verts_uvs,
            'faces_uvs': faces_uvs
            }


def build_mesh(verts_uvs, faces_uvs, texture_map, verts, faces):
    textures = TexturesUV(verts_uvs=verts_uvs, faces_uvs=faces_uvs, maps=texture_map)
    mesh = Meshes(verts=[verts], faces=[faces], textures=textures)
    return mesh
