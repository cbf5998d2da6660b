"""``first_approach.py`` of the reference (stylise each batch of views in 2-D, then fit the texture /
mesh to the stylised images through the renderer) on MI355X: same flags, defaults and artefacts
(``log.txt`` with ``Batch i, Step s, Loss x`` lines, ``2d_style_transfer/view_k.png``,
``final_render/``, ``final.obj``; reference first_approach.py:22-45, 147-225).

Phase A (reference :178-179) is ``style_transfer()``: fused VGG forward / loss / backward and a fused
Adam on the pixels.  Phase B (reference :191-217) is raster + fused shade + masked MSE + texture scatter
+ fused Adam: five HIP launches per step, no VGG.  Under ``python -m torch.distributed.run ...`` each
rank stylises and renders its slice of the batch's views and the phase-B gradient is summed inside
``optimizer.step()``.
"""
import torch
from tqdm import tqdm

from style_transfer import *  # noqa: F401,F403
from utils import *  # noqa: F401,F403
from losses import *  # noqa: F401,F403

from st3d.cli import Flag, Run, make_parser

FLAGS = [
    Flag("n_mse_steps", int, 100, "phase-B steps (fit the 3-D parameters to the stylised views) per batch"),
    Flag("n_style_transfer_steps", int, 3000, "phase-A steps (2-D style transfer on the pixels) per batch"),
    Flag("output_path", str, "/content/output_first", "folder for log.txt, renders and the final mesh"),
    Flag("style_transfer_init", str, 'content', "what the 2-D style transfer starts from", ['noise', 'current', 'content']),
    Flag("style_transfer_lr", float, 0.01, "Adam step size of phase A"),
    Flag("mse_lr", float, 0.01, "Adam step size of phase B"),
]


def build_parser():
    return make_parser(FLAGS)


def _phase_a_start(run, args, content, cams, style):
    """Initial pixels of the 2-D style transfer (reference :167-175)."""
    if args.style_transfer_init == 'content':
        return content
    if args.style_transfer_init == 'noise':
        return torch.rand(content.shape, device=run.device)
    img, cov = render_meshes(run.renderer, run.current_mesh(), cams)          # 'current'
    return apply_background(img, cov, background_type=args.current_background, background=style)


def main(argv=None):
    args = build_parser().parse_args(argv)
    run = Run(args, lr=args.mse_lr, image_dir="2d_style_transfer")
    mesh = run.opt['optimizable_mesh']

    run.say("Starting optimization...")
    for vb in run.batches():
        if vb.index < run.progress:         # --resume: these view batches were fitted before the checkpoint
            continue
        run.say(f"\nBatch {vb.index}")
        n_local = vb.hi - vb.lo
        targets = cams = None
        if n_local:
            cams = run.cameras[vb.lo:vb.hi]
            style = run.style_image.expand(n_local, -1, -1, -1)
            with torch.no_grad():
                img, cov = render_meshes(run.renderer, run.content_mesh, cams)
                content = apply_background(img, cov, background_type=args.content_background, background=style)
                start = _phase_a_start(run, args, content, cams, style)
            # phase A: the whole batch is stylised at once; the result may leave [0,1], hence the clamp
            targets = finalize_tensor(style_transfer(start, content, style, run.vgg, steps=args.n_style_transfer_steps,
                                                     style_weight=args.style_weight, content_weight=args.content_weight,
                                                     lr=args.style_transfer_lr))
            run.save_views(targets, vb.lo)

        # phase B: masked MSE between the renders and the stylised views
        shown = 0
        for step in tqdm(range(args.n_mse_steps), desc="Optimizing", postfix=shown, disable=not run.main):
            run.optimizer.zero_grad()
            if n_local:
                mesh = run.current_mesh()
                rendered, cov = render_meshes(run.renderer, mesh, cams)
                loss = compute_first_approach_loss(rendered=rendered, masks=cov, target_rendered=targets,
                                                   verts=run.opt['verts'], target_verts=run.original_verts, mesh=mesh,
                                                   weights=run.loss_weights, opt_type=args.optimization_target,
                                                   batch_denom=vb.size)     # image term -> share of the batch mean
                loss.backward()
            else:
                loss = run.idle_contribution()
            run.optimizer.step()
            shown = run.global_sum(loss).item()
            run.log(f'Batch {vb.index}, Step {step}, Loss {shown}')
        run.maybe_checkpoint(vb.index + 1)

    run.export(mesh)


if __name__ == "__main__":
    main()
