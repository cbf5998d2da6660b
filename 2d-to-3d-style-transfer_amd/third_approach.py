"""The "3rd approach" sketched in the reference's notes.txt:38 ("do little steps on the texture using 1st approach with
many camera views"), which the reference never implemented: instead of stylising a view batch to convergence and then
fitting the texture to it once (first_approach.py), many short rounds alternate

    A. a FEW 2-D style-transfer steps on the batch, started from the CURRENT renders, and
    B. a FEW masked-MSE steps pulling the texture / mesh towards those lightly stylised images,

cycling over all view batches every round, so every part of the surface keeps being revisited while the style builds
up gradually and consistently across views.  Same building blocks, flags and artefacts as first_approach.py
(``log.txt`` lines ``Round r, Batch i, Loss x``); runs sharded over ranks the same way.
"""
import torch
from tqdm import tqdm

from style_transfer import *  # noqa: F401,F403
from utils import *  # noqa: F401,F403
from losses import *  # noqa: F401,F403

from st3d.cli import Flag, Run, make_parser

FLAGS = [
    Flag("n_rounds", int, 30, "passes over the whole view set"),
    Flag("n_style_transfer_steps", int, 20, "2-D style-transfer steps per batch and round"),
    Flag("n_mse_steps", int, 5, "texture / mesh fitting steps per batch and round"),
    Flag("output_path", str, "/content/output_third", "folder for log.txt, renders and the final mesh"),
    Flag("style_transfer_lr", float, 0.01, "Adam step size on the pixels"),
    Flag("mse_lr", float, 0.01, "Adam step size on the texture / vertices"),
]


def build_parser():
    return make_parser(FLAGS)


def main(argv=None):
    args = build_parser().parse_args(argv)
    run = Run(args, lr=args.mse_lr, image_dir="2d_style_transfer")
    mesh = run.opt['optimizable_mesh']
    content_of_batch = {}

    run.say("Starting optimization...")
    for rnd in tqdm(range(run.progress, args.n_rounds), desc="Round", disable=not run.main):
        for vb in run.batches():
            n_local = vb.hi - vb.lo
            targets = cams = None
            if n_local:
                cams = run.cameras[vb.lo:vb.hi]
                style = run.style_image.expand(n_local, -1, -1, -1)
                with torch.no_grad():
                    if vb.index not in content_of_batch or args.content_background == 'noise':
                        img, cov = render_meshes(run.renderer, run.content_mesh, cams)
                        content_of_batch[vb.index] = apply_background(img, cov, background_type=args.content_background,
                                                                      background=style)
                    img, cov = render_meshes(run.renderer, run.current_mesh(), cams)
                    start = apply_background(img, cov, background_type=args.current_background, background=style)
                targets = finalize_tensor(style_transfer(start, content_of_batch[vb.index], style, run.vgg,
                                                         steps=args.n_style_transfer_steps, style_weight=args.style_weight,
                                                         content_weight=args.content_weight, lr=args.style_transfer_lr))
                if rnd == args.n_rounds - 1:
                    run.save_views(targets, vb.lo)
            loss = torch.zeros((), device=run.device)
            for _ in range(args.n_mse_steps):
                run.optimizer.zero_grad()
                if n_local:
                    mesh = run.current_mesh()
                    rendered, cov = render_meshes(run.renderer, mesh, cams)
                    loss = compute_first_approach_loss(rendered=rendered, masks=cov, target_rendered=targets,
                                                       verts=run.opt['verts'], target_verts=run.original_verts, mesh=mesh,
                                                       weights=run.loss_weights, opt_type=args.optimization_target,
                                                       batch_denom=vb.size)
                    loss.backward()
                else:
                    loss = run.idle_contribution()
                run.optimizer.step()
            run.log(f'Round {rnd}, Batch {vb.index}, Loss {run.global_sum(loss).item()}')
        run.maybe_checkpoint(rnd + 1)

    run.export(mesh)


if __name__ == "__main__":
    main()
