"""Shared scaffolding of the two command-line drivers (first_approach.py / second_approach.py).

The reference repeats set-up, batching, logging and export in both scripts (first_approach.py:47-147,
219-225; second_approach.py:44-140,196-202).  Here that is one ``Run`` object: flag tables, device /
rank set-up, scene + renderer + VGG + cameras + optimiser, the view-batch schedule with its per-rank
slice, the log file and the final export.  The scripts keep only their loop bodies.
"""
import argparse
import math
import os
from collections import namedtuple

import torch
import torch.nn.functional as F

from . import io as st3d_io
from . import optim as st3d_optim
from .render import (AmbientLights, FoVPerspectiveCameras, MeshRasterizer, MeshRenderer, RasterizationSettings,
                     SoftPhongShader)

Flag = namedtuple("Flag", "name type default help choices", defaults=(None,))

_BACKGROUNDS = ['noise', 'style', 'white']
_TARGETS = ['texture', 'mesh', 'both']

# name, type, default and choices are the reference's (first_approach.py:23-45, second_approach.py:23-42); note
# `type=bool` flags keep argparse's "any non-empty string is True" behaviour of the reference.
SHARED_FLAGS = [
    Flag("n_views", int, 6, "how many camera views surround the object"),
    Flag("obj_path", str, "./objects/cow_mesh/cow.obj", "Wavefront OBJ to stylise"),
    Flag("style_path", str, "./imgs/Style_1.jpg", "style image"),
    Flag("style_weight", float, 1e6, "multiplier of the Gram (style) term"),
    Flag("content_weight", float, 1.0, "multiplier of the conv4_2 (content) term"),
    Flag("resize_texture", bool, True, "resample the texture map to size x size"),
    Flag("size", int, 768, "side of the rendered images in pixels"),
    Flag("batch_size", int, 4, "views per optimisation step"),
    Flag("content_background", str, 'white', "what fills the uncovered pixels of the content renders", _BACKGROUNDS),
    Flag("current_background", str, 'white', "what fills the uncovered pixels of the current renders", _BACKGROUNDS),
    Flag("randomize_views", bool, True, "random cameras on a sphere instead of the fixed turntable"),
    Flag("optimization_target", str, "texture", "which tensors Adam updates", _TARGETS),
    Flag("main_loss_weight", float, 3.0, "weight of the image term when the mesh is optimised"),
    Flag("mesh_edge_loss_weight", float, 1.0, "weight of the edge-length regulariser"),
    Flag("mesh_laplacian_smoothing_weight", float, 1.0, "weight of the uniform-Laplacian regulariser"),
    Flag("mesh_normal_consistency_weight", float, 1.0, "weight of the normal-consistency regulariser"),
    Flag("mesh_verts_weight", float, 1.0, "weight of the distance to the original vertices"),
    # additions; the defaults keep the reference behaviour
    Flag("vgg_weights", str, None, "local VGG-19 state_dict (never downloaded); default ST3D_VGG19_WEIGHTS or seeded weights"),
    Flag("seed", int, None, "seed for camera sampling / noise (the reference is unseeded)"),
    Flag("verts_lr", float, None, "separate Adam step size for the vertices when both are optimised (notes.txt:29 of the reference)"),
    Flag("checkpoint_every", int, 0, "write <output_path>/checkpoint.pt (parameters + Adam state) every N epochs/batches; 0 = never"),
    Flag("resume", str, None, "checkpoint.pt to continue from"),
]

# regularisers the reference defines but never switches on (losses.py:48-65, notes.txt:36,39); weight 0 = off
REGULARISER_FLAGS = [
    Flag("tv_weight", float, 0.0, "weight of the masked total-variation term on the current renders"),
    Flag("rgb_range_weight", float, 0.0, "weight of the out-of-[0,1] penalty on the texture map"),
    Flag("texture_l2_weight", float, 0.0, "weight of the squared distance to the original texture map"),
]


def make_parser(extra_flags):
    parser = argparse.ArgumentParser()
    for fl in list(extra_flags) + SHARED_FLAGS:
        kw = {"type": fl.type, "default": fl.default, "help": fl.help}
        if fl.choices:
            kw["choices"] = fl.choices
        parser.add_argument("--" + fl.name, **kw)
    return parser


def load_scene(obj_path, size, resize_texture, device):
    """OBJ + its texture -> (verts (V,3), faces (F,3), verts_uvs (1,VT,2), faces_uvs (1,F,3), map (1,T,T,3)) on
    `device`, the map resampled to size x size when asked (reference second_approach.py:77-97)."""
    verts, faces, aux = st3d_io.load_obj(obj_path)
    if aux.verts_uvs is None or faces.textures_idx is None or not aux.texture_images:
        # e.g. objects/teapot_mesh/teapot.obj (faces `v//vn`, no mtllib): the reference crashes at
        # first_approach.py:85-88, so there is no behaviour to match.  Per-vertex spherical UVs
        # and a mid-grey texture with seeded noise are synthesised so BASELINE config 4 can run.
        print(f"WARNING: {obj_path} has no UVs / texture; synthesising spherical UVs and a grey noise texture")
        uvs, uv_faces = st3d_io.synthesize_uvs(verts), faces.verts_idx.clone()
        noise = torch.randn((size, size, 3), generator=torch.Generator().manual_seed(0))
        tex = (0.5 + 0.1 * noise).clamp(0, 1)
    else:
        uvs, uv_faces = aux.verts_uvs, faces.textures_idx
        tex = next(iter(aux.texture_images.values()))
    tex = tex[None].to(device)
    if resize_texture:
        nchw = F.interpolate(tex.permute(0, 3, 1, 2), size=size, mode='bilinear', align_corners=False)
        tex = nchw.permute(0, 2, 3, 1).contiguous()
    return verts.to(device), faces.verts_idx.to(device), uvs[None].to(device), uv_faces[None].to(device), tex


ViewBatch = namedtuple("ViewBatch", "index size lo hi")      # batch number, its global size, this rank's [lo, hi)


class AsyncImageWriter:
    """PNG dumps off the critical path.  The reference encodes every view of every step on the main thread
    (second_approach.py:183-185: ~30 ms per 512^2 image, 15x the GPU step at config 2).  Here the batch is quantised on
    the GPU exactly like ``tensor_to_image`` (clamp to [0,1], x255, truncate), copied to pinned host memory without
    blocking, and encoded by worker threads once the copy's event has fired; at most `depth` batches are in flight
    (back-pressure instead of unbounded memory).  Pixels are identical; only zlib's effort level is lower."""

    def __init__(self, workers=8, depth=4):
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="st3d-png")
        self._pending = []
        self._depth = depth

    def submit(self, images, paths):
        """images (n,3,H,W) float on the GPU; paths: n file names."""
        u8 = (images.detach().clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
        host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(u8, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self._pending.append([self._pool.submit(self._write, done, host, k, path) for k, path in enumerate(paths)])
        while len(self._pending) > self._depth:
            self._wait(self._pending.pop(0))

    @staticmethod
    def _write(done, host, k, path):
        from PIL import Image
        done.synchronize()
        Image.fromarray(host[k].numpy()).save(path, compress_level=1)

    @staticmethod
    def _wait(futures):
        for f in futures:
            f.result()          # re-raises a failed write

    def flush(self):
        while self._pending:
            self._wait(self._pending.pop(0))


class Run:
    """Everything both drivers need before their loop starts."""

    def __init__(self, args, lr, image_dir):
        import losses as _l
        import style_transfer as _s
        import utils as _u
        self.args = args
        self.rank, self.world, local = st3d_optim.init_distributed()
        if not torch.cuda.is_available():
            raise RuntimeError("st3d needs an MI355X (libst3d has no CPU fallback)")
        # one GPU per rank; only a gloo rehearsal (ST3D_DIST_BACKEND=gloo) may put several ranks on one card
        self.device = torch.device(f"cuda:{local % max(torch.cuda.device_count(), 1)}")
        torch.cuda.set_device(self.device)
        _u.device = _s.device = _l.device = self.device
        if args.seed is not None:
            torch.manual_seed(args.seed)
        self.out_dir = args.output_path
        self.image_dir = os.path.join(self.out_dir, image_dir)
        if self.main:
            os.makedirs(self.image_dir, exist_ok=True)
        self.loss_weights = {k: getattr(args, k) for k in (
            'mesh_edge_loss_weight', 'mesh_laplacian_smoothing_weight', 'mesh_normal_consistency_weight',
            'mesh_verts_weight', 'main_loss_weight')}

        self.say("Loading mesh...")
        verts, faces, verts_uvs, faces_uvs, tex = load_scene(args.obj_path, args.size, args.resize_texture, self.device)
        self.original_verts = verts
        self.content_mesh = _u.build_mesh(verts_uvs, faces_uvs, tex, verts, faces)

        cams = FoVPerspectiveCameras(device=self.device)
        settings = RasterizationSettings(image_size=args.size, blur_radius=0.0, faces_per_pixel=1)
        self.renderer = MeshRenderer(rasterizer=MeshRasterizer(cameras=cams, raster_settings=settings),
                                     shader=SoftPhongShader(device=self.device, cameras=cams,
                                                            lights=AmbientLights(device=self.device)))
        self.say("Loading model...")
        self.vgg = _u.get_vgg(weights=args.vgg_weights)

        self.say("Building cameras...")
        gen = torch.Generator().manual_seed(args.seed) if args.seed is not None else None
        self.cameras = (_u.build_random_cameras(args.n_views, generator=gen) if args.randomize_views
                        else _u.build_fixed_cameras(args.n_views))
        if self.world > 1:                      # every rank must look through the same cameras
            torch.distributed.broadcast(self.cameras.R, 0)
            torch.distributed.broadcast(self.cameras.T, 0)

        # one optimiser (one Adam state) for the whole run, over all view batches
        self.opt = _u.setup_optimizations(args.optimization_target, self.content_mesh, lr)
        if args.verts_lr is not None and args.optimization_target == 'both':
            self.opt['optimizer'] = st3d_optim.Adam([{"params": [self.opt['verts']], "lr": args.verts_lr},
                                                     {"params": [self.opt['texture_map']], "lr": lr}])
        self.optimizer = self.opt['optimizer']
        self.original_map = tex
        self.progress = 0               # epochs (second approach) / view batches (first approach) already done
        if args.resume:
            self.load_checkpoint(args.resume)
        self.style_image = _u.load_as_tensor(args.style_path, size=args.size)   # loop-invariant; the reference reloads it
        self._log = os.path.join(self.out_dir, 'log.txt')
        if self.main:
            with open(self._log, 'w') as fh:
                fh.write('Logger:\n')
        self._utils = _u
        self.writer = AsyncImageWriter()

    # ---- small helpers
    @property
    def main(self):
        return self.rank == 0

    def say(self, msg):
        if self.main:
            print(msg)

    def log(self, line):
        if self.main:
            with open(self._log, 'a') as fh:
                fh.write(line + '\n')

    def save_views(self, images, first_index):
        """view_<k>.png for this rank's views of the batch, k counted over the whole view set (asynchronous)."""
        self.writer.submit(images, [os.path.join(self.image_dir, f"view_{first_index + j}.png") for j in range(images.shape[0])])

    def current_mesh(self):
        o = self.opt
        return self._utils.build_mesh(o['verts_uvs'], o['faces_uvs'], o['texture_map'], o['verts'], o['faces'])

    def batches(self):
        """The reference's schedule (ceil(n_views / batch_size) consecutive slices) with this rank's share of each."""
        n, bs = self.args.n_views, self.args.batch_size
        for i in range(math.ceil(n / bs)):
            first, size = i * bs, min((i + 1) * bs, n) - i * bs
            lo, hi = st3d_optim.shard_views(size, self.rank, self.world)
            yield ViewBatch(i, size, first + lo, first + hi)

    def zero_contribution(self):
        """A rank without views in this batch still joins the gradient all-reduce (with zeros)."""
        for p in self.optimizer.params:
            p.grad = torch.zeros_like(p)

    def idle_contribution(self):
        """This rank has no views in the batch: it still joins the gradient all-reduce, and it still owes its 1/world
        share of the view-INDEPENDENT terms (the mesh regularisers of 'mesh'/'both' and the optional texture
        regularisers), which every rank adds with weight 1/world so that the SUM over ranks counts them once.
        Leaves the gradients in place for ``optimizer.step()`` and returns this rank's (detached) loss share."""
        import losses as _l
        self.zero_contribution()
        mesh = self.current_mesh()
        total = self.regularisers(None, None, mesh, 0, 1)
        if self.args.optimization_target in ('mesh', 'both'):
            total = total + _l._mesh_terms(self.opt['verts'], self.original_verts, mesh, self.loss_weights)
        if torch.is_tensor(total) and total.requires_grad:
            total.backward()                      # accumulates into the zero gradients
        return total.detach() if torch.is_tensor(total) else torch.zeros((), device=self.device)

    def global_sum(self, value):
        t = value.detach().clone()
        if self.world > 1:
            torch.distributed.all_reduce(t)
        return t

    # ---- checkpoint / resume: parameters + Adam moments + how far the run got
    def save_checkpoint(self, progress):
        if not self.main:
            return
        blob = {"progress": int(progress), "optimization_target": self.args.optimization_target,
                "texture_map": self.opt['texture_map'].detach().cpu(), "verts": self.opt['verts'].detach().cpu(),
                "optimizer": self.optimizer.state_dict()}
        tmp = os.path.join(self.out_dir, "checkpoint.pt.tmp")
        torch.save(blob, tmp)
        os.replace(tmp, os.path.join(self.out_dir, "checkpoint.pt"))

    def load_checkpoint(self, path):
        blob = torch.load(path, map_location="cpu", weights_only=True)
        if blob["optimization_target"] != self.args.optimization_target:
            raise ValueError("checkpoint was written for optimization_target=%r" % blob["optimization_target"])
        with torch.no_grad():
            self.opt['texture_map'].copy_(blob["texture_map"])
            self.opt['verts'].copy_(blob["verts"])
        self.optimizer.load_state_dict(blob["optimizer"])
        self.progress = int(blob["progress"])
        self.say(f"Resumed from {path} at {self.progress}")

    def maybe_checkpoint(self, done):
        every = self.args.checkpoint_every
        if every and done % every == 0:
            self.save_checkpoint(done)

    def regularisers(self, current, coverage, mesh, n_local, batch_size):
        """Optional extra terms (all weights default to 0 = reference behaviour).  The image term is a per-view mean and
        is weighted by this rank's share of the batch; the texture terms are view-independent and enter once over all
        ranks (the gradient all-reduce SUMs)."""
        import losses as _l
        a, total = self.args, 0
        if getattr(a, "tv_weight", 0.0) and n_local:
            total = total + a.tv_weight * (n_local / batch_size) * _l.compute_tv_loss(current, coverage)
        if getattr(a, "rgb_range_weight", 0.0):
            total = total + (a.rgb_range_weight / self.world) * _l.rgb_range_loss(mesh)
        if getattr(a, "texture_l2_weight", 0.0):
            total = total + (a.texture_l2_weight / self.world) * _l.texture_l2_loss(mesh, self.original_map)
        return total

    def export(self, mesh):
        """final_render/view_k.png from 12 turntable cameras + final.obj/.mtl/.png (first_approach.py:219-225)."""
        from . import ops as _ops
        self.writer.flush()
        _ops.check_near_plane(block=True)       # a mesh that reached the near clipping plane fails loudly (no clipping at K = 1)
        if self.main:
            u = self._utils
            final = u.finalize_mesh(mesh)
            u.save_render(self.renderer, final, u.build_fixed_cameras(12), os.path.join(self.out_dir, "final_render"))
            tex = final.textures
            st3d_io.save_obj(os.path.join(self.out_dir, "final.obj"), final.verts_packed(), final.faces_packed(),
                             tex.verts_uvs_padded()[0], tex.faces_uvs_padded()[0], tex.maps_padded()[0])
        if self.world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
