"""``second_approach.py`` of the reference (direct 3-D optimisation of texture / vertices under the
perceptual loss) on MI355X: same flags, defaults and artefacts (``log.txt`` with ``Epoch e, Loss x``
lines, ``current_images/view_k.png``, ``final_render/view_k.png``, ``final.obj``; reference
second_approach.py:22-42, 140-202).

    python second_approach.py --size 512 --batch_size 8 --n_views 8 ...
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        second_approach.py --size 512 --batch_size 64 --n_views 64 ...

Hoisted out of the step because it does not depend on the optimised parameters: the style image load
(reference :157), the content renders (:160) and with them the content/style VGG targets -- unless
``--content_background noise`` draws fresh noise every step (then they are recomputed, as the reference
does).  With several ranks each one renders its slice of the batch's views; the gradient is summed over
ranks inside ``optimizer.step()``.
"""
import torch
from tqdm import tqdm

# the reference's star imports: scripts and notebooks reach everything through these three modules
from style_transfer import *  # noqa: F401,F403
from utils import *  # noqa: F401,F403
from losses import *  # noqa: F401,F403

from st3d.cli import REGULARISER_FLAGS, Flag, Run, load_scene, make_parser  # noqa: F401  (load_scene re-exported)

FLAGS = [
    Flag("epochs", int, 3000, "passes over the view set"),
    Flag("output_path", str, "/content/output_second", "folder for log.txt, renders and the final mesh"),
    Flag("lr", float, 0.01, "Adam step size"),
    Flag("save_every", int, 1, "write current_images/*.png every N steps (reference: every step); 0 = never"),
] + REGULARISER_FLAGS


def build_parser():
    return make_parser(FLAGS)


def main(argv=None):
    args = build_parser().parse_args(argv)
    run = Run(args, lr=args.lr, image_dir="current_images")
    reuse_content = args.content_background != 'noise'
    content_of_batch = {}
    steps_done = 0

    run.say("Starting optimization...")
    for epoch in range(run.progress, args.epochs):
        run.say(f"\nEpoch {epoch}")
        epoch_loss = torch.zeros((), device=run.device)
        for vb in tqdm(list(run.batches()), leave=True, desc="Batch", disable=not run.main):
            run.optimizer.zero_grad()
            if vb.hi == vb.lo:                      # more ranks than views in this batch
                epoch_loss += run.idle_contribution()
                run.optimizer.step()
                continue
            cams = run.cameras[vb.lo:vb.hi]
            style = run.style_image.expand(vb.hi - vb.lo, -1, -1, -1)

            content = content_of_batch.get(vb.index) if reuse_content else None
            if content is None:
                with torch.no_grad():
                    img, cov = render_meshes(run.renderer, run.content_mesh, cams)
                    content = apply_background(img, cov, background_type=args.content_background, background=style)
                if reuse_content:
                    content_of_batch[vb.index] = content

            mesh = run.current_mesh()
            img, cov = render_meshes(run.renderer, mesh, cams)
            current = apply_background(img, cov, background_type=args.current_background, background=style)

            loss = compute_second_approach_loss(
                current=current, content=content, style=style, model=run.vgg, style_weight=args.style_weight,
                content_weight=args.content_weight, verts=run.opt['verts'], target_verts=run.original_verts, mesh=mesh,
                weights=run.loss_weights, opt_type=args.optimization_target, batch_denom=vb.size)
            extra = run.regularisers(current, cov, mesh, vb.hi - vb.lo, vb.size)
            if torch.is_tensor(extra):              # every weight is 0 by default: nothing is added, as in the reference
                loss = loss + extra

            if args.save_every and steps_done % args.save_every == 0:
                run.save_views(current, vb.lo)          # encoded by worker threads, off the step's critical path

            loss.backward()
            run.optimizer.step()                    # gradient all-reduce over ranks + fused Adam
            epoch_loss += loss.detach()
            steps_done += 1

        run.log(f'Epoch {epoch}, Loss {run.global_sum(epoch_loss).item()}')
        run.maybe_checkpoint(epoch + 1)

    run.export(run.current_mesh())


if __name__ == "__main__":
    main()
