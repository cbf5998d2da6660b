"""Drop-in for the reference's ``first_approach.py`` (2-D stylise each view batch, then fit the
texture / mesh to the stylised images through the renderer): same flags and defaults
(reference first_approach.py:22-45), same artefacts (``log.txt`` with
``Batch i, Step s, Loss x`` lines, ``2d_style_transfer/view_k.png``, ``final_render/``,
``final.obj``).

Phase A (reference :178-179) is ``style_transfer()``: fused VGG forward/loss/backward + fused
Adam on the pixels.  Phase B (reference :191-217) is raster + fused shade + masked MSE + texture
scatter + fused Adam: five HIP launches per step, no VGG.  With several ranks
(``python -m torch.distributed.run ... first_approach.py``) each rank stylises and renders its
slice of the batch's views and the phase-B gradient is summed inside ``optimizer.step()``.
"""
import argparse
import math
import os

import torch
from tqdm import tqdm

from style_transfer import *  # noqa: F401,F403
from utils import *  # noqa: F401,F403
from losses import *  # noqa: F401,F403

from second_approach import load_scene
from st3d import io as st3d_io
from st3d import optim as st3d_optim
from st3d.render import (AmbientLights, FoVPerspectiveCameras, MeshRasterizer, MeshRenderer, RasterizationSettings,
                         SoftPhongShader)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--n_views", default=6, type=int, help="Number of views considered by the renderer")
    parser.add_argument("--n_mse_steps", default=100, type=int, help="Number of steps for MSE optimization")
    parser.add_argument("--n_style_transfer_steps", default=3000, type=int, help="Number of steps for style transfer")
    parser.add_argument("--obj_path", default="./objects/cow_mesh/cow.obj", type=str, help="Path to the object")
    parser.add_argument("--style_path", default="./imgs/Style_1.jpg", type=str, help="Path to the style image")
    parser.add_argument("--style_weight", default=1e6, type=float, help="Weight of the style loss")
    parser.add_argument("--content_weight", default=1.0, type=float, help="Weight of the content loss")
    parser.add_argument("--resize_texture", default=True, type=bool, help="Whether to resize the texture to the same size of the images")
    parser.add_argument("--size", default=768, type=int, help="Dimension of the images")
    parser.add_argument("--output_path", default="/content/output_first", type=str, help="Output folder path")
    parser.add_argument("--batch_size", default=4, type=int, help="Batch size")
    parser.add_argument("--style_transfer_init", default='content', type=str, choices=['noise', 'current', 'content'], help="Initialization for the 2D Style Transfer")
    parser.add_argument("--content_background", default='white', type=str, choices=['noise', 'style', 'white'], help="Type of background for the content image")
    parser.add_argument("--current_background", default='white', type=str, choices=['noise', 'style', 'white'], help="Type of background for the current image")
    parser.add_argument("--style_transfer_lr", default=0.01, type=float, help="Style Transfer Learning Rate")
    parser.add_argument("--mse_lr", default=0.01, type=float, help="2D to 3D Learning Rate")
    parser.add_argument("--randomize_views", type=bool, default=True, help="Whether or not to randomize views")
    parser.add_argument("--optimization_target", type=str, choices=['texture', 'mesh', 'both'], default="texture", help="Decide what to optimize")
    parser.add_argument("--main_loss_weight", type=float, default=3.0, help="Weight of the main computed loss (i.e., mse)")
    parser.add_argument("--mesh_edge_loss_weight", type=float, default=1.0, help="Weight of edge loss (enforces admissible weights for the edges)")
    parser.add_argument("--mesh_laplacian_smoothing_weight", type=float, default=1.0, help="Weight of smoothing (smooth surface)")
    parser.add_argument("--mesh_normal_consistency_weight", type=float, default=1.0, help="Weight of normal consistency")
    parser.add_argument("--mesh_verts_weight", type=float, default=1.0, help="Mesh verts (uvs and not uvs) regularization weight")
    # additions (defaults keep the reference behaviour)
    parser.add_argument("--vgg_weights", default=None, type=str, help="Local VGG-19 state_dict (no download is attempted)")
    parser.add_argument("--seed", default=None, type=int, help="Seed the camera sampling / noise (the reference is unseeded)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, world, local = st3d_optim.init_distributed()
    device = torch.device(f"cuda:{local}" if torch.cuda.is_available() else "cpu")
    if device.type != "cuda":
        raise RuntimeError("st3d needs an MI355X (libst3d has no CPU fallback)")
    torch.cuda.set_device(device)
    import utils as _u, style_transfer as _s, losses as _l
    _u.device = _s.device = _l.device = device
    if args.seed is not None:
        torch.manual_seed(args.seed)

    loss_weights = {
        'mesh_edge_loss_weight': args.mesh_edge_loss_weight,
        'mesh_laplacian_smoothing_weight': args.mesh_laplacian_smoothing_weight,
        'mesh_normal_consistency_weight': args.mesh_normal_consistency_weight,
        'mesh_verts_weight': args.mesh_verts_weight,
        'main_loss_weight': args.main_loss_weight,
    }
    output_path, size, n_views, batch_size = args.output_path, args.size, args.n_views, args.batch_size
    if rank == 0:
        os.makedirs(output_path, exist_ok=True)
        os.makedirs(output_path + "/2d_style_transfer", exist_ok=True)

    print("Loading mesh...")
    original_verts, original_faces, original_verts_uvs, original_faces_uvs, texture_image = load_scene(
        args.obj_path, size, args.resize_texture, device)
    content_mesh = build_mesh(original_verts_uvs, original_faces_uvs, texture_image, original_verts, original_faces)

    cameras = FoVPerspectiveCameras(device=device)
    raster_settings = RasterizationSettings(image_size=size, blur_radius=0.0, faces_per_pixel=1)
    renderer = MeshRenderer(rasterizer=MeshRasterizer(cameras=cameras, raster_settings=raster_settings),
                            shader=SoftPhongShader(device=device, cameras=cameras, lights=AmbientLights(device=device)))

    print("Loading model...")
    vgg = get_vgg(weights=args.vgg_weights)

    print("Building cameras...")
    gen = torch.Generator().manual_seed(args.seed) if args.seed is not None else None
    cameras_list = build_random_cameras(n_views, generator=gen) if args.randomize_views else build_fixed_cameras(n_views)
    if world > 1:
        torch.distributed.broadcast(cameras_list.R, 0)
        torch.distributed.broadcast(cameras_list.T, 0)

    # one optimiser (and one Adam state) across all view batches, as the reference (:129)
    out = setup_optimizations(args.optimization_target, content_mesh, args.mse_lr)
    current_mesh, optimizer = out['optimizable_mesh'], out['optimizer']
    texture_map, verts, faces = out['texture_map'], out['verts'], out['faces']
    verts_uvs, faces_uvs = out['verts_uvs'], out['faces_uvs']

    if rank == 0:
        with open(output_path + '/log.txt', 'w') as file:
            file.write('Logger:\n')

    style_image = load_as_tensor(args.style_path, size=size)

    print("Starting optimization...")
    for i in range(math.ceil(n_views / batch_size)):
        if rank == 0:
            print(f"\nBatch {i}")
        batch_start, batch_end = i * batch_size, min((i + 1) * batch_size, n_views)
        current_batch_size = batch_end - batch_start
        lo, hi = st3d_optim.shard_views(current_batch_size, rank, world)
        lo, hi = batch_start + lo, batch_start + hi
        n_local = hi - lo
        have = n_local > 0
        applied_style_tensors = object_masks = None
        if have:
            batch_cameras = cameras_list[lo:hi]
            style_tensors = style_image.expand(n_local, -1, -1, -1)
            with torch.no_grad():
                content_tensors, content_masks = render_meshes(renderer, content_mesh, batch_cameras)
                content_tensors = apply_background(content_tensors, content_masks, background_type=args.content_background, background=style_tensors)

                # initial images of the 2-D style transfer (reference :167-175)
                if args.style_transfer_init == 'noise':
                    applied_style_tensors = torch.rand(content_tensors.shape, device=device)
                elif args.style_transfer_init == 'content':
                    applied_style_tensors = content_tensors
                elif args.style_transfer_init == 'current':
                    current_mesh = build_mesh(verts_uvs, faces_uvs, texture_map, verts, faces)
                    current_tensors, current_masks = render_meshes(renderer, current_mesh, batch_cameras)
                    applied_style_tensors = apply_background(current_tensors, current_masks, background_type=args.current_background, background=style_tensors)

            # phase A: batch style transfer on the pixels
            applied_style_tensors = style_transfer(applied_style_tensors, content_tensors, style_tensors, vgg,
                                                   steps=args.n_style_transfer_steps, style_weight=args.style_weight,
                                                   content_weight=args.content_weight, lr=args.style_transfer_lr)
            applied_style_tensors = finalize_tensor(applied_style_tensors)      # values may leave (0,1)
            for j, applied_style_tensor in enumerate(applied_style_tensors):
                tensor_to_image(applied_style_tensor).save(output_path + f"/2d_style_transfer/view_{lo + j}.png")

        # phase B: fit the texture / mesh to the stylised views through the renderer
        loss_value = 0
        for step in tqdm(range(args.n_mse_steps), desc="Optimizing", postfix=loss_value, disable=rank != 0):
            optimizer.zero_grad()
            if have:
                current_mesh = build_mesh(verts_uvs, faces_uvs, texture_map, verts, faces)
                rendered_tensors, object_masks = render_meshes(renderer, current_mesh, batch_cameras)
                loss = compute_first_approach_loss(rendered=rendered_tensors, masks=object_masks,
                                                   target_rendered=applied_style_tensors, verts=verts,
                                                   target_verts=original_verts, mesh=current_mesh, weights=loss_weights,
                                                   opt_type=args.optimization_target)
                if world > 1:
                    loss = loss * (n_local / current_batch_size)     # local mean -> share of the global mean
                loss.backward()
            else:
                loss = torch.zeros((), device=device)
                for p in optimizer.params:
                    p.grad = torch.zeros_like(p)
            optimizer.step()
            loss_t = loss.detach().clone()
            if world > 1:
                torch.distributed.all_reduce(loss_t)
            loss_value = loss_t.item()
            if rank == 0:
                with open(output_path + '/log.txt', 'a') as file:
                    file.write(f'Batch {i}, Step {step}, Loss {loss_value}\n')

    if rank == 0:
        final_mesh = finalize_mesh(current_mesh)
        fixed = build_fixed_cameras(12)
        save_render(renderer, final_mesh, fixed, output_path + "/final_render")
        tex = final_mesh.textures
        st3d_io.save_obj(output_path + "/final.obj", final_mesh.verts_packed(), final_mesh.faces_packed(),
                         tex.verts_uvs_padded()[0], tex.faces_uvs_padded()[0], tex.maps_padded()[0])
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
