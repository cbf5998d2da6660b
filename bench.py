#!/usr/bin/env python3
"""bench.py -- iterations/sec of the 2D->3D style-transfer optimisation step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = the loop body of the reference's second_approach.py:145-190 for
BASELINE.json configs[1] (cow_mesh + Style_1, 512x512, 8 views per GPU, texture-only):
differentiable render of the current textured mesh from the rank's views -> frozen VGG-19
forward to conv5_1 -> Gram/content losses -> backward to the pixels -> texture-scatter backward ->
(N > 1: RCCL all-reduce of the 3 MiB texture gradient) -> fused Adam.  Content renders and the
content/style VGG targets do not depend on the texture and are computed once before the timed
region (`--no-hoist` recomputes them every step like the reference does).  Scaling is WEAK:
every rank renders 8 views of the SAME texture, so the job does N*8 views per step;
`value` counts 8-view iterations per second over the whole job (= steps/s * N).

Prints ONE JSON line (rank 0).
  roofline       the dominant kernel (the Winograd conv launches, ~80 % of the step) against the fp32-MFMA peak:
                 achieved = MFMA flops ISSUED (16 multiplies per 2x2 outputs, i.e. 16/36 of the direct convolution's
                 count) / HIP-event time on the launch stream, so frac <= 1 by construction; the direct-convolution
                 equivalent rate is reported separately as alg_equiv_tflops.
  step_roofline  the same accounting for the whole step (all MFMA flops issued / step time).
  kernels        per kernel family: ms/step, launches, and either mfma_frac (issued flops / peak) or hbm_frac
                 (algorithmic bytes of SURVEY.md 8d / time / 8 TB/s) -- whichever roof bounds it.
  cpu_baseline   the CPU restatement (oracle/) of ONE full reference step (8 views: 16 renders, 3 VGG forwards + 1
                 backward, texture backward, Adam) timed on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "2d-to-3d-style-transfer_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
PEAK_FP32_MFMA = 157.3e12         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM = 8.0e12                 # MI355X_MICROARCH.md: HBM3E ~8 TB/s

# torchvision vgg19().features conv modules up to conv5_1: (module index, Cin, Cout, resolution divisor)
CONVS = [(0, 3, 64, 1), (2, 64, 64, 1), (5, 64, 128, 2), (7, 128, 128, 2), (10, 128, 256, 4), (12, 256, 256, 4),
         (14, 256, 256, 4), (16, 256, 256, 4), (19, 256, 512, 8), (21, 512, 512, 8), (23, 512, 512, 8), (25, 512, 512, 8),
         (28, 512, 512, 16)]
STYLE_TAPS = {0: (64, 1), 5: (128, 2), 10: (256, 4), 19: (512, 8), 28: (512, 16)}      # module -> (C, divisor)


def conv_alg_flops(module, S, B):
    """2*9*Cin*Cout flop per output pixel -- one direction (forward or input-gradient)."""
    for m, cin, cout, d in CONVS:
        if m == module:
            return 2.0 * 9 * cin * cout * (S // d) ** 2 * B
    return 0.0


def gram_alg_flops(module, S, B):
    if module not in STYLE_TAPS:        # the one launch pair that holds all five style layers (st3d_gram_fwd_multi)
        return sum(gram_alg_flops(m, S, B) for m in STYLE_TAPS)
    C, d = STYLE_TAPS[module]
    return 2.0 * C * C * (S // d) ** 2 * B


def gram_fwd_issued_fraction(module):
    """MFMA work the Gram forward ISSUES over the full C x C product (gram.hip): only blocks on or above the diagonal
    exist -- 32x32 blocks in gram_diag_kernel (C = 64: 3 of 4, C = 128: 10 of 16), 64x64 wave tiles in the multi-tile
    launches (the mirror wave of a diagonal 128x128 tile sits the MFMAs out: C = 256 10 of 16, C = 512 36 of 64)."""
    if module not in STYLE_TAPS:        # all five layers in one launch: flop-weighted over the layers (needs S, B: see caller)
        raise KeyError(module)
    C = STYLE_TAPS[module][0]
    n = C // 32 if C <= 128 else C // 64
    return (n * (n + 1) / 2) / (n * n)


def gram_fwd_issued_flops(module, S, B):
    if module not in STYLE_TAPS:
        return sum(gram_fwd_issued_flops(m, S, B) for m in STYLE_TAPS)
    return gram_alg_flops(module, S, B) * gram_fwd_issued_fraction(module)


def check_fractions(obj, path="line"):
    """Every roofline fraction in the line is a fraction: fail loudly instead of printing a number above 1."""
    if isinstance(obj, dict):
        for k, v in obj.items():
            if k in ("frac", "mfma_frac", "hbm_frac") and v is not None:
                assert 0.0 <= v <= 1.0, "%s.%s = %r is not a fraction of a roofline" % (path, k, v)
            check_fractions(v, path + "." + str(k))
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            check_fractions(v, "%s[%d]" % (path, i))


def load_assets(size, device, mesh="cow", style_k=1):
    d = np.load(os.path.join(GOLDEN, f"assets_{mesh}_mesh.npz"))
    sty = np.load(os.path.join(GOLDEN, f"assets_style{style_k}_512.npz"))["rgb_u8"]
    verts = torch.from_numpy(d["verts"]).to(device)
    faces = torch.from_numpy(d["faces"].astype(np.int64)).to(device)
    if "verts_uvs" in d.files:
        verts_uvs = torch.from_numpy(d["verts_uvs"])[None].to(device)
        faces_uvs = torch.from_numpy(d["faces_uvs"].astype(np.int64))[None].to(device)
        tex = torch.from_numpy(d["texture_u8"]).to(torch.float32).div(255.0)[None].to(device)
        # second_approach.py:84-94: texture resized to size x size (bilinear, align_corners=False)
        tex = F.interpolate(tex.permute(0, 3, 1, 2), size=size, mode="bilinear", align_corners=False).permute(0, 2, 3, 1).contiguous()
    else:                            # teapot: no UVs / texture in the reference's data
        from st3d import io as stio
        verts_uvs = stio.synthesize_uvs(verts.cpu())[None].to(device)
        faces_uvs = faces[None].clone()
        noise = torch.randn((size, size, 3), generator=torch.Generator().manual_seed(0))
        tex = (0.5 + 0.1 * noise).clamp(0, 1)[None].to(device)
    style = torch.from_numpy(sty).permute(2, 0, 1).to(torch.float32).div(255.0)
    if size != style.shape[1]:
        style = F.interpolate(style[None], size=size, mode="bilinear", align_corners=False, antialias=size < style.shape[1])[0]
    return verts, faces, verts_uvs, faces_uvs, tex, style.contiguous().to(device)


def cpu_baseline(size, views, seed_cam):
    """CPU restatement (oracle/loop_ref.py) of ONE step of the reference loop as the reference executes it
    (second_approach.py:157-189): `views` content renders + `views` current renders (naive rasteriser, OpenMP over
    rows), compute_perceptual_loss = 3 VGG-19 forwards + 1 backward on torch-CPU fp32, texture backward, Adam."""
    from oracle import loop_ref as LR
    from oracle import render_ref as RR
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    cow = np.load(os.path.join(GOLDEN, "assets_cow_mesh.npz"))
    sty = np.load(os.path.join(GOLDEN, "assets_style1_512.npz"))["rgb_u8"]
    tex = torch.from_numpy(cow["texture_u8"]).to(torch.float32).div(255.0)[None]
    tex = F.interpolate(tex.permute(0, 3, 1, 2), size=size, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)[0].contiguous().numpy()
    style = torch.from_numpy(sty).permute(2, 0, 1).to(torch.float32).div(255.0)[None]
    if size != style.shape[2]:
        style = F.interpolate(style, size=size, mode="bilinear", align_corners=False, antialias=True)
    g = torch.Generator().manual_seed(seed_cam)
    elev, azim = RR.random_camera_angles(views, lambda k: torch.rand(k, generator=g).numpy())
    R, T = RR.look_at_view_transform(2.10, elev, azim, at=(0, 0.10, 0.25))

    def loop(n):
        return LR.SecondApproachRef(cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"], tex, R[:n], T[:n], size,
                                    style, target="texture", lr=0.01, nthreads=threads, hoist=False)
    loop(1).step()                        # warm-up on one view (page-in, MKL/OpenMP thread pools)
    ref = loop(views)
    t0 = time.time()
    loss = ref.step()
    t_step = time.time() - t0
    return {"value": round(1.0 / t_step * (views / 8.0), 5), "unit": "iter/s (8 views, %dx%d)" % (size, size), "cores": threads,
            "kind": "port", "seconds_per_step": round(t_step, 2), "loss": loss,
            "sample": "ONE full %d-view step of the CPU restatement (oracle/loop_ref.py: %d naive renders with %d OpenMP "
                      "threads, 3 VGG-19 forwards + 1 backward on torch-CPU fp32 with %d threads, texture backward, Adam), "
                      "%.1f s, after a 1-view warm-up step; the reference additionally runs the unused VGG tail (modules "
                      "29-36, +7 %% flops) in each forward" % (views, 2 * views, threads, threads, t_step)}


def time_ms(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--views", type=int, default=8, help="views per GPU per step")
    ap.add_argument("--mesh", choices=["cow", "bob", "teapot"], default="cow")
    ap.add_argument("--style", type=int, choices=[1, 3, 4, 5], default=1)
    ap.add_argument("--no-hoist", action="store_true", help="recompute content renders + VGG targets every step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-kernels", action="store_true", help="per-kernel HIP events also when --gpus > 1")
    ap.add_argument("--layers", action="store_true", help="add the per-layer table to the JSON line")
    ap.add_argument("--target", choices=["texture", "mesh", "both"], default="texture",
                    help="optimization_target (BASELINE configs[1] = texture; configs[4] = both)")
    ap.add_argument("--approach", choices=["second", "first_a", "first_b"], default="second",
                    help="second: the 3-D loop of second_approach.py:145-190 (the headline).  first_a / first_b: the two phases "
                         "of first_approach.py -- A = the style_transfer() loop on the pixels of the views (style_transfer.py:59-83, "
                         "VGG only), B = render -> masked MSE -> texture scatter -> Adam (first_approach.py:191-217, no VGG)")
    ap.add_argument("--background", choices=["white", "noise"], default="white",
                    help="content_background and current_background of the reference's CLIs.  noise (notes.txt:1 of the "
                         "reference: the recommended setting) draws fresh U[0,1) backgrounds every step (utils.py:22), so the "
                         "content features cannot be hoisted: one more VGG forward to conv4_2 per step")
    args = ap.parse_args()
    if args.approach != "second" and (args.gpus > 1 or args.background != "white" or args.no_hoist):
        raise SystemExit("bench.py: --approach first_a / first_b are single-GPU variants with the default backgrounds")

    from st3d import launch as st3d_launch
    if not st3d_launch.under_launcher():
        # the parent process: nothing here has touched a GPU yet (importing torch does not)
        if not os.path.exists(os.path.join(PKG, "lib", "libst3d.so")):
            import __graft_entry__ as _ge       # a checkout without the prebuilt library: compile it (there is no CPU fallback)
            _ge.build()
        if args.gpus > 1:
            # `python bench.py --gpus N` on its own: one fresh child per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
            # torch.distributed.run would set them), rank 0's JSON line relayed, non-zero exit if any rank fails
            sys.exit(st3d_launch.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    from st3d import optim as st3d_optim
    if st3d_optim.dist_info()[1] != args.gpus:      # (checked before the rendezvous, which would wait for the missing ranks)
        world = st3d_optim.dist_info()[1]
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher environment says WORLD_SIZE={world}; start it as "
                         f"`python bench.py --gpus {args.gpus}` or under torch.distributed.run with --nproc-per-node {args.gpus}")
    rank, world, local = st3d_optim.init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libst3d has no CPU fallback")
    device = torch.device(f"cuda:{local % max(torch.cuda.device_count(), 1)}")     # > device_count only in gloo rehearsals
    torch.cuda.set_device(device)

    import losses as L
    import style_transfer as ST
    import utils as U
    U.device = ST.device = L.device = device
    from st3d import ops
    from st3d.render import (AmbientLights, FoVPerspectiveCameras, MeshRasterizer, MeshRenderer, RasterizationSettings,
                             SoftPhongShader)

    S, Bv = args.size, args.views
    global_views = Bv * world
    verts, faces, verts_uvs, faces_uvs, tex, style_image = load_assets(S, device, args.mesh, args.style)
    content_mesh = U.build_mesh(verts_uvs, faces_uvs, tex, verts, faces)
    cams0 = FoVPerspectiveCameras(device=device)
    renderer = MeshRenderer(MeshRasterizer(cams0, RasterizationSettings(image_size=S, blur_radius=0.0, faces_per_pixel=1)),
                            SoftPhongShader(device=device, cameras=cams0, lights=AmbientLights(device=device)))
    vgg = U.get_vgg(seed=0)
    gen = torch.Generator().manual_seed(0)                  # fixed cameras: every run renders the same views
    cameras = U.build_random_cameras(global_views, generator=gen)
    my_cams = cameras[rank * Bv:(rank + 1) * Bv]
    out = U.setup_optimizations(args.target, content_mesh, 0.01)
    optimizer, texture_map = out["optimizer"], out["texture_map"]
    reg_weights = {"main_loss_weight": 3.0, "mesh_verts_weight": 1.0, "mesh_edge_loss_weight": 1.0,
                   "mesh_laplacian_smoothing_weight": 1.0, "mesh_normal_consistency_weight": 1.0}   # second_approach.py:33-37
    style_tensors = style_image.expand(Bv, -1, -1, -1)

    bg = args.background
    if bg == "noise":
        torch.manual_seed(1 + rank)             # the noise backgrounds (torch.rand inside utils.apply_background)

    def content_render():
        with torch.no_grad():
            return U.render_meshes(renderer, content_mesh, my_cams)

    content_rgb, content_cov = content_render()         # loop-invariant: cameras and content mesh never change

    def content_targets():
        c, cm = (content_rgb, content_cov) if not args.no_hoist else content_render()
        with torch.no_grad():
            return U.apply_background(c, cm, background_type=bg, background=style_tensors)   # white: identity; noise: fresh draw

    content_tensors = content_targets()
    content_each_step = args.no_hoist or bg == "noise"  # the content VGG forward to conv4_2 cannot be (or is not) hoisted

    def step_second():          # second_approach.py:145-190
        optimizer.zero_grad()
        content = content_tensors if not content_each_step else content_targets()
        mesh = U.build_mesh(out["verts_uvs"], out["faces_uvs"], texture_map, out["verts"], out["faces"])
        cur, masks = U.render_meshes(renderer, mesh, my_cams)
        cur = U.apply_background(cur, masks, background_type=bg, background=style_tensors)
        loss = L.compute_second_approach_loss(cur, content, style_tensors, vgg, 1e6, 1.0, out["verts"], verts, mesh,
                                              reg_weights, args.target, batch_denom=global_views)
        loss.backward()
        optimizer.step()
        return loss

    # first_approach.py phase A = style_transfer(init = content renders, content, style, vgg, steps, lr): the loop body of
    # the drop-in style_transfer() (2d-to-3d-style-transfer_amd/style_transfer.py: plan.loss + fused Adam on the pixels;
    # reference style_transfer.py:59-83), its one-time targets (reference :44-51) computed before the timed region
    if args.approach == "first_a":
        plan_a = vgg.plan(Bv, S)
        plan_a.set_content(content_tensors, force=True)
        plan_a.set_style(style_tensors, Bv, force=True)
        pixels = content_tensors.clone().detach().contiguous().requires_grad_(True)
        opt_a = st3d_optim.Adam([pixels], lr=0.01, reduce_grads=False)         # first_approach.py:33 style_transfer_lr
        optimizer = opt_a

    def step_first_a():
        lossbuf, grad = plan_a.loss(pixels, 1e6, 1.0)
        pixels.grad = grad
        opt_a.step()
        return lossbuf[0]

    # first_approach.py phase B (:191-217): fit the texture to fixed stylised views through the renderer; the targets here
    # are the content renders pushed towards the style image (any fixed images do: the loss is a masked MSE)
    if args.approach == "first_b":
        targets_b = (0.5 * content_tensors + 0.5 * style_tensors).clamp(0, 1).contiguous()

    def step_first_b():
        optimizer.zero_grad()
        mesh = U.build_mesh(out["verts_uvs"], out["faces_uvs"], texture_map, out["verts"], out["faces"])
        rendered, cov = U.render_meshes(renderer, mesh, my_cams)
        loss = L.compute_first_approach_loss(rendered=rendered, masks=cov, target_rendered=targets_b, verts=out["verts"],
                                             target_verts=verts, mesh=mesh, weights=reg_weights, opt_type=args.target,
                                             batch_denom=global_views)
        loss.backward()
        optimizer.step()
        return loss

    step = {"second": step_second, "first_a": step_first_a, "first_b": step_first_b}[args.approach]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    comm = {}
    if world > 1:       # RCCL builds its communicator lazily on the first collective: keep that out of the timed steps even at --warmup 0
        torch.distributed.all_reduce(torch.zeros(texture_map.numel(), device=device))
        comm = st3d_optim.comm_info(device)         # backend, ranks that answered a SUM of ones, RCCL version
        ids = [None] * world
        torch.distributed.all_gather_object(ids, "%s #%d" % (torch.cuda.get_device_name(device), device.index))
        comm["devices_seen"] = ids
    first_loss = None
    for _ in range(args.warmup):
        loss = step()
        if first_loss is None:
            first_loss = loss.detach().clone()
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
        if first_loss is None:
            first_loss = loss.detach().clone()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = marks[0].elapsed_time(marks[-1])
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = per_step[len(per_step) // 2]
    final_loss, first_step_loss = float(loss.detach()), float(first_loss)
    collectives_per_step = optimizer.collectives / float(max(args.warmup + args.steps, 1))
    allreduce_ms = allreduce_ms_min = None
    if world > 1:
        # the one collective of the step (the flat gradient buffer of all optimised tensors), timed on its own right after
        # the timed region (same message size, same stream)
        g = torch.zeros(sum(q.numel() for q in optimizer.params), device=device)
        allreduce_ms = time_ms(lambda: torch.distributed.all_reduce(g), reps=20, warm=3)
        t = torch.tensor([elapsed, dev_ms, median_ms, allreduce_ms, -allreduce_ms], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed, dev_ms, median_ms, allreduce_ms, allreduce_ms_min = (float(x) for x in t)
        allreduce_ms_min = -allreduce_ms_min
        lt = torch.tensor([final_loss, first_step_loss], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(lt)
        final_loss, first_step_loss = float(lt[0]), float(lt[1])

    kernels, layers = None, None
    nprof = 5
    wino = os.environ.get("ST3D_CONV") != "direct"
    if args.profile_kernels or world == 1:
        # per-launch HIP events (recorded by libst3d on the stream the kernels are launched on) over nprof further
        # steps of the same loop, right after the timed region
        launches = []
        if args.approach != "first_b":          # (phase B of the first approach has no VGG plan)
            plan = vgg.plan(Bv, S)
            plan.profile(True)
            for _ in range(nprof):
                step()
            torch.cuda.synchronize()
            launches = plan.profile_launches()
            plan.profile_read()
            plan.profile(False)
        agg = {}
        for fam, module, ms in launches:
            e = agg.setdefault((fam, module), [0.0, 0])
            e[0] += ms
            e[1] += 1
        kernels, layers = {}, []
        for (fam, module), (ms, n) in sorted(agg.items()):
            ms /= nprof
            k = kernels.setdefault(fam, {"ms_per_step": 0.0, "launches_per_step": 0, "_issued": 0.0, "_alg": 0.0, "_bytes": 0.0})
            k["ms_per_step"] += ms
            k["launches_per_step"] += n // nprof
            issued = alg = nbytes = 0.0
            hw4 = lambda c, d: 4.0 * c * (S // d) ** 2 * Bv          # bytes of a (Bv, c, S/d, S/d) fp32 tensor
            if fam in ("conv_fwd", "conv_dgrad"):
                alg = conv_alg_flops(module, S, Bv)
                issued = alg * (16.0 / 36.0)            # Winograd F(2x2,3x3): 16 multiplies per 2x2 outputs
            elif fam in ("conv43_fwd", "conv43_dgrad"):
                alg = conv_alg_flops(module, S, Bv)
                issued = alg * (36.0 / 144.0)           # Winograd F(4x4,3x3): 36 multiplies per 4x4 outputs
            elif fam == "convx_dgrad" and module == 0 and ("gram_bwd", 0) not in agg:
                # csrc/tap0.hip: relu1_1 style gradient + gate + conv1_1 input gradient in one pass, both products on the
                # matrix pipe (64x64 and 32(27)x64 per pixel); algorithmic bytes: read gradient + activation, write 3 channels
                alg = conv_alg_flops(0, S, Bv) + gram_alg_flops(0, S, Bv)
                issued = gram_alg_flops(0, S, Bv) + 2.0 * 32 * 64 * S * S * Bv
                nbytes = hw4(64 + 64 + 3, 1)
            elif fam in ("convx_fwd", "convx_dgrad") and module == 0:
                alg = conv_alg_flops(0, S, Bv)
                # conv1_1 on the vector ALU: HBM-bound.  fwd reads 3 + writes 64 channels; dgrad reads the gradient and the
                # ReLU gate (64 + 64) and writes 3
                nbytes = hw4(3 + 64, 1) if fam == "convx_fwd" else hw4(64 + 64 + 3, 1)
            elif fam in ("convx_fwd", "convx_dgrad"):
                alg = issued = conv_alg_flops(module, S, Bv)          # direct MFMA kernels (ST3D_CONV=direct, odd shapes)
            elif fam == "gram_fwd":
                alg = gram_alg_flops(module, S, Bv)
                issued = gram_fwd_issued_flops(module, S, Bv)
            elif fam == "gram_bwd":
                alg = issued = gram_alg_flops(module, S, Bv)
                C, d = STYLE_TAPS[module]
                nbytes = hw4(C, d) * (2 if module == 28 else 3)      # read F, (read +) write dF: conv5_1 stores, the rest accumulate
            reps = max(n // nprof, 1)                       # launches of this (family, module) per step (2 with --no-hoist)
            issued, alg, nbytes = issued * reps, alg * reps, nbytes * reps
            k["_issued"] += issued
            k["_alg"] += alg
            k["_bytes"] += nbytes
            row = {"family": fam, "module": module, "ms": round(ms, 4)}
            if issued:
                row["mfma_frac"] = round(issued / (ms * 1e-3) / PEAK_FP32_MFMA, 4)
            if nbytes:
                row["hbm_frac"] = round(nbytes / (ms * 1e-3) / PEAK_HBM, 4)
            layers.append(row)
        for fam, k in kernels.items():
            ms = k["ms_per_step"]
            k["ms_per_step"] = round(ms, 4)
            issued, alg, nbytes = k.pop("_issued"), k.pop("_alg"), k.pop("_bytes")
            k["_issued_flops"], k["_alg_flops"] = issued, alg
            if ms > 0 and issued:
                k["issued_tflops"] = round(issued / (ms * 1e-3) / 1e12, 2)
                k["mfma_frac"] = round(issued / (ms * 1e-3) / PEAK_FP32_MFMA, 4)
                k["alg_equiv_tflops"] = round(alg / (ms * 1e-3) / 1e12, 2)
            if ms > 0 and nbytes:
                k["alg_gbps"] = round(nbytes / (ms * 1e-3) / 1e9, 1)
                k["hbm_frac"] = round(nbytes / (ms * 1e-3) / PEAK_HBM, 4)
        # the HBM/latency-bound kernels outside the VGG plan, each timed alone on this rank's inputs (torch events on the
        # current stream, which is the stream libst3d launches them on); algorithmic bytes per SURVEY.md 8d
        hb = {}
        with torch.no_grad():
            mesh = U.build_mesh(out["verts_uvs"], out["faces_uvs"], texture_map, out["verts"], out["faces"])
            f32 = mesh.faces_i32()
            fuv = mesh.textures.faces_uvs_i32()
            vuv = out["verts_uvs"][0].contiguous()
            tex2 = texture_map.detach()[0].contiguous()
            Fn, px = f32.shape[0], Bv * S * S
            ndc = ops.project_verts(out["verts"].detach(), my_cams.R, my_cams.T)
            frag = ops.raster_fwd(ndc, f32, S)
            gimg = torch.randn((Bv, 3, S, S), device=device)
            st = optimizer._state_of(texture_map) if any(q is texture_map for q in optimizer.params) else None
            hb = {} if args.approach == "first_a" else {
                "raster_fwd": (time_ms(lambda: ops.raster_fwd(ndc, f32, S)), Bv * Fn * 52.0 + px * 24.0),
                "shade_fwd": (time_ms(lambda: ops.shade_fwd(frag, vuv, fuv, tex2)), px * (24.0 + 16.0)),
                "shade_bwd_texture_scatter": (time_ms(lambda: ops.shade_bwd(gimg, frag, vuv, fuv, tex2)), px * (24.0 + 12.0) + 2 * tex2.numel() * 4.0),
            }
            if hb:
                ops.set_deterministic(False)
                hb["shade_bwd_texture_scatter_float_atomics"] = (time_ms(lambda: ops.shade_bwd(gimg, frag, vuv, fuv, tex2)), px * (24.0 + 12.0) + 2 * tex2.numel() * 4.0)
                ops.set_deterministic(True)
            if args.approach == "first_b":      # losses.py:71-75: reads rendered, target (3 channels each) and the mask, writes the gradient
                rend = torch.rand((Bv, 3, S, S), device=device)
                mk = (frag[0] >= 0).to(torch.float32)[:, None].contiguous()
                hb["masked_mse"] = (time_ms(lambda: ops.masked_mse(rend, targets_b, mk)), px * (12.0 + 12.0 + 4.0 + 12.0))
            if hb and args.target != "texture":
                # the vertex path of 'mesh' / 'both' (SURVEY.md K14, K15): d/d(barycentrics) out of the shade backward ->
                # raster backward (fixed-point scatter over B x V x 3) -> projection backward; the four mesh regularisers
                from st3d import mesh_losses as ML
                _, gbary = ops.shade_bwd(gimg, frag, vuv, fuv, tex2, want_bary=True)
                gndc = ops.raster_bwd(gbary, frag[0], ndc, f32)
                topo = ML._topology(mesh)
                vd = out["verts"].detach()
                hb["shade_bwd_with_bary"] = (time_ms(lambda: ops.shade_bwd(gimg, frag, vuv, fuv, tex2, want_bary=True)),
                                             px * (24.0 + 12.0 + 12.0) + 2 * tex2.numel() * 4.0)
                hb["raster_bwd_vertex_scatter"] = (time_ms(lambda: ops.raster_bwd(gbary, frag[0], ndc, f32)), px * (12.0 + 4.0) + Bv * Fn * 48.0)
                hb["project_verts_bwd"] = (time_ms(lambda: ops.project_verts_bwd(vd, my_cams.R, my_cams.T, gndc)), Bv * vd.shape[0] * 12.0 * 2)
                hb["mesh_regularisers"] = (time_ms(lambda: ops.mesh_reg(vd, verts, topo, [1.0, 1.0, 1.0, 1.0])), vd.shape[0] * 12.0 * 8)
            if st is not None:
                gdummy = torch.zeros_like(tex2)
                pdummy, m1, m2 = tex2.clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone()
                hb["adam"] = (time_ms(lambda: ops.adam_step(pdummy, gdummy, m1, m2, 3, 0.01)), 7.0 * tex2.numel() * 4.0)
        for name, (ms, nbytes) in hb.items():
            kernels[name] = {"ms_per_call": round(ms, 4), "alg_gbps": round(nbytes / (ms * 1e-3) / 1e9, 1),
                             "hbm_frac": round(nbytes / (ms * 1e-3) / PEAK_HBM, 4), "timed": "standalone, %d views" % Bv}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        s2 = (S / 512.0) ** 2
        # algorithmic and issued MFMA flops of one step on this GPU
        f_alg_step = sum(conv_alg_flops(m, S, Bv) for m, *_ in CONVS) * 2 + sum(gram_alg_flops(m, S, Bv) for m in STYLE_TAPS) * 2
        if content_each_step:
            f_alg_step += sum(conv_alg_flops(m, S, Bv) for m, *_ in CONVS if m <= 21)
        f_wino_alg = sum(conv_alg_flops(m, S, Bv) for m, *_ in CONVS if m != 0) * 2
        if content_each_step:
            f_wino_alg += sum(conv_alg_flops(m, S, Bv) for m, *_ in CONVS if 0 < m <= 21)
        wfams = ("conv_fwd", "conv_dgrad", "conv43_fwd", "conv43_dgrad")
        if kernels and any(f in kernels for f in wfams):
            # what the Winograd launches of a step actually issued: per launch F(2x2,3x3) = 16/36, F(4x4,3x3) = 36/144 of the
            # direct count, as tagged by the plan's profile (conv_* / conv43_* families)
            f_conv_issued_measured = sum(kernels[f]["_issued_flops"] for f in wfams if f in kernels)
        else:
            f_conv_issued_measured = None
        f_issued_step = ((f_conv_issued_measured if f_conv_issued_measured is not None else f_wino_alg * (16.0 / 36.0)) if wino
                         else f_wino_alg + 2 * conv_alg_flops(0, S, Bv)) \
            + sum(gram_fwd_issued_flops(m, S, Bv) + gram_alg_flops(m, S, Bv) for m in STYLE_TAPS)
        if os.environ.get("ST3D_TAP0_FUSED") != "0":
            f_issued_step += 2.0 * 32 * 64 * S * S * Bv        # conv1_1's input gradient as an MFMA product (csrc/tap0.hip)
        step_s = dev_ms / args.steps * 1e-3
        std_cfg = (S == 512 and Bv == 8 and not content_each_step and wino and args.mesh == "cow" and args.target == "texture"
                   and args.approach == "second")
        traffic, tpath = {}, ""
        if std_cfg:
            import glob
            tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))     # newest round last
            if tfiles:
                tpath = tfiles[-1]
                with open(tpath) as fh:
                    traffic = json.load(fh)
        tsrc = ("committed rocprofv3 PMC passes %s (FETCH_SIZE raw + WRITE_SIZE per launch; gfx950 under-reports 16-B/lane "
                "read streams by up to 2x), not re-measured in this run" % os.path.relpath(tpath, ROOT)) if traffic else None
        step_roofline = {
            "bound": "mfma", "achieved": round(f_issued_step / step_s / 1e12, 3), "peak": PEAK_FP32_MFMA / 1e12, "unit": "TFLOP/s",
            "frac": round(f_issued_step / step_s / PEAK_FP32_MFMA, 4),
            "alg_equiv_tflops": round(f_alg_step / step_s / 1e12, 3),
            "traffic": (traffic["fetch_bytes_raw_per_step"] + traffic["write_bytes_per_step"]) if traffic else None,
            "traffic_source": tsrc,
            "note": "whole step per GPU (renders, losses, Adam and host gaps included): MFMA flops issued per step (%.1f GF: "
                    "Winograd convs 16/36 (F(2x2,3x3)) or 36/144 (F(4x4,3x3)) of the direct count, Gram forward upper tiles only) over the HIP-event time of the "
                    "timed region (%.3f ms/step); alg_equiv_tflops prices the same time with the direct-convolution count of "
                    "SURVEY.md 8d (%.1f GF/step)" % (f_issued_step / 1e9, dev_ms / args.steps, f_alg_step / 1e9)}
        roofline = step_roofline
        if kernels and sum(kernels.get(f, {}).get("ms_per_step", 0) for f in wfams) > 0 and args.approach != "first_b":
            # the dominant kernel: the Winograd conv launches (forward + input-gradient, both tile sizes), priced together
            kms = sum(kernels[f]["ms_per_step"] for f in wfams if f in kernels)
            kn = sum(kernels[f]["launches_per_step"] for f in wfams if f in kernels)
            f_conv_alg = sum(kernels[f]["_alg_flops"] for f in wfams if f in kernels)
            f_conv_issued = f_conv_issued_measured if wino else f_conv_alg
            n43 = sum(kernels[f]["launches_per_step"] for f in ("conv43_fwd", "conv43_dgrad") if f in kernels)
            ktraffic = None
            if traffic and kn == traffic.get("wino_kernel_launches_per_step"):
                ktraffic = (traffic["wino_fetch_bytes_raw_per_step"] + traffic["wino_write_bytes_per_step"]) / kn
            roofline = {
                "bound": "mfma", "achieved": round(f_conv_issued / (kms * 1e-3) / 1e12, 3), "peak": PEAK_FP32_MFMA / 1e12,
                "unit": "TFLOP/s", "frac": round(f_conv_issued / (kms * 1e-3) / PEAK_FP32_MFMA, 4),
                "traffic": ktraffic, "traffic_source": tsrc if ktraffic else None,
                "alg_equiv_tflops": round(f_conv_alg / (kms * 1e-3) / 1e12, 3),
                # the number earlier rounds reported as `frac`, when every launch was F(2x2,3x3): 16/36 of the direct count
                # over the same time / peak.  Above 1 once launches run F(4x4,3x3) -- a comparison across rounds, not a roofline
                "f23_equiv_of_peak": round(f_conv_alg * 16.0 / 36.0 / (kms * 1e-3) / PEAK_FP32_MFMA, 4),
                "kernel": (("wino43_kernel (csrc/wino43.hip, F(4x4,3x3): %d launches)" % n43 if n43 else "")
                           + (" + " if n43 and kn > n43 else "")
                           + ("wino4_kernel (csrc/wino.hip, F(2x2,3x3): %d launches)" % (kn - n43) if kn > n43 else ""))
                          if wino else "conv3x3_kernel (csrc/conv.hip)",
                "launches_per_step": kn, "avg_launch_ms": round(kms / kn, 4), "ms_per_step": round(kms, 4),
                "share_of_step": round(kms / (dev_ms / args.steps), 4),
                "flops_per_launch_issued": round(f_conv_issued / kn), "flops_per_launch_alg": round(f_conv_alg / kn),
                "note": "achieved = MFMA flops ISSUED by these launches (Winograd F(2x2,3x3): 16 multiplies per 2x2 outputs = "
                        "16/36, F(4x4,3x3): 36 per 4x4 outputs = 36/144 of the 2*9*Cin*Cout per output pixel of the direct "
                        "convolutions they replace, forward + "
                        "input-gradient of every VGG conv but conv1_1: %.1f GF issued, %.1f GF algorithmic per step) / their "
                        "HIP-event time on the launch stream over %d steps after the timed region; frac = matrix-pipe "
                        "utilisation.  alg_equiv_tflops = the direct-convolution count over the same time (may exceed the "
                        "peak; it is not a roofline fraction); f23_equiv_of_peak = the round-1/2 `frac` metric (every launch "
                        "priced as F(2x2,3x3)) for comparison across rounds." % (f_conv_issued / 1e9, f_conv_alg / 1e9, nprof)}
        if args.approach == "first_b" and kernels:
            # phase B of the first approach has no VGG: the step IS the HBM/latency-bound render + scatter kernels.  Its
            # roofline is HBM: algorithmic bytes of SURVEY.md 8d per kernel over the kernel's own (standalone, back-to-back)
            # time; the dominant one is the rasteriser.  step_roofline prices all of them over the measured step, host gaps included.
            names = [k for k in ("raster_fwd", "shade_fwd", "masked_mse",
                                  "shade_bwd_with_bary" if "shade_bwd_with_bary" in kernels else "shade_bwd_texture_scatter",
                                  "raster_bwd_vertex_scatter", "project_verts_bwd", "mesh_regularisers", "adam") if k in kernels]
            kb = {k: kernels[k]["alg_gbps"] * 1e9 * kernels[k]["ms_per_call"] * 1e-3 for k in names}
            ksum_ms = sum(kernels[k]["ms_per_call"] for k in names)
            dom = max(names, key=lambda k: kernels[k]["ms_per_call"])
            step_roofline = {"bound": "hbm", "achieved": round(sum(kb.values()) / step_s / 1e9, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                             "frac": round(sum(kb.values()) / step_s / PEAK_HBM, 4), "traffic": None,
                             "kernel_ms_sum_standalone": round(ksum_ms, 4), "device_ms_per_step": round(step_s * 1e3, 4),
                             "host_or_launch_bound_share": round(max(0.0, 1.0 - ksum_ms / (step_s * 1e3)), 3),
                             "note": "algorithmic bytes of the step's kernels (%s) over the HIP-event time of the timed region; "
                                     "the difference between that time and the kernels' standalone sum is launch / host time "
                                     "(Python autograd + ~12 launches per step)" % ", ".join(names)}
            roofline = {"bound": "hbm", "kernel": dom + " (csrc/raster.hip: face setup + coarse bins + 16x16-pixel tiles)" if dom == "raster_fwd" else dom,
                        "achieved": kernels[dom]["alg_gbps"], "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": kernels[dom]["hbm_frac"],
                        "traffic": None, "avg_launch_ms": kernels[dom]["ms_per_call"],
                        "note": "the longest kernel of a phase-B step, timed alone on this step's inputs; latency-bound "
                                "(tile lists, per-pixel face walks), not a stream"}
        tgt = {"texture": "texture-only optimisation", "mesh": "vertex optimisation", "both": "joint vertex + texture optimisation"}[args.target]
        cfg_name = "BASELINE.json configs[1]" if std_cfg else "variant of BASELINE.json configs[1]"
        loop = {"second": "second_approach loop body", "first_a": "first_approach phase A = style_transfer() loop on the views' pixels, no renderer",
                "first_b": "first_approach phase B = render + masked MSE + texture scatter + Adam, no VGG"}[args.approach]
        unit = {"second": "iter/s (one iter = 8 views of %dx%d: render + VGG-19 fwd/bwd + Gram/content loss + Adam)",
                "first_a": "iter/s (one iter = one style_transfer() step on 8 images of %dx%d: VGG-19 fwd/bwd + Gram/content loss + Adam on the pixels)",
                "first_b": "iter/s (one iter = one masked-MSE fitting step on 8 views of %dx%d: render + loss + texture scatter + Adam)"}[args.approach]
        metric = "style-transfer iters/sec (512x512, 8 views, cow_mesh)"
        if args.approach != "second":
            metric = "first_approach phase %s steps/sec (512x512, 8 views, cow_mesh)" % args.approach[-1].upper()
        hoisted = ("content renders + content features + style Grams" if not content_each_step else
                   "style Grams + content renders; content features recomputed every step (fresh noise background)" if not args.no_hoist
                   else "style Grams only (--no-hoist: content re-rendered and its VGG forward redone every step)")
        res = {
            "metric": metric,
            "value": round(args.steps * world * (Bv / 8.0) / elapsed, 4),
            "unit": unit % (S, S),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "median_ms_per_step": round(median_ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic views (seeded cameras) of %s_mesh + Style_%d fixtures; seeded He-normal VGG-19 weights "
                    "(pretrained weights need a download)" % (args.mesh, args.style),
            "config": {"workload": "%s: %s_mesh + Style_%d, %dx%d, %d views/GPU/iter, %s, %s backgrounds (%s)"
                                   % (cfg_name, args.mesh, args.style, S, S, Bv, tgt, bg, loop),
                       "global_views_per_step": global_views, "texture": "%dx%d" % (S, S),
                       "targets_hoisted": hoisted,
                       "parallelism": "views sharded dp%d, RCCL all-reduce of the texture gradient" % world},
            "first_step_loss": first_step_loss, "final_loss": final_loss,
            "roofline": roofline,
            "step_roofline": step_roofline,
        }
        if allreduce_ms is not None:
            res.update(comm)
            res["collectives_per_step"] = collectives_per_step
            res["allreduce_ms"] = round(allreduce_ms, 4)                # slowest rank
            res["allreduce_ms_min"] = round(allreduce_ms_min, 4)        # fastest rank
            res["allreduce_bytes"] = int(sum(q.numel() for q in optimizer.params) * 4)
        if kernels:
            for k in kernels.values():
                k.pop("_issued_flops", None)
                k.pop("_alg_flops", None)
            res["kernels"] = kernels
        if layers and args.layers:
            res["layers"] = layers
        if not args.no_cpu_baseline and world == 1 and args.approach == "second":
            try:
                # like for like with the CPU leg, which executes the step as the reference does (content re-rendered and its
                # VGG forward redone every step, second_approach.py:157-166): the same on the GPU, 20 steps after the timed
                # region (the headline `value` stays the hoisted step; this is only the denominator-matched comparison)
                args.no_hoist, keep = True, (args.no_hoist, content_each_step)
                content_each_step = True
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    step()
                torch.cuda.synchronize()
                unhoisted_ms = (time.perf_counter() - t1) / 20 * 1e3
                args.no_hoist, content_each_step = keep
                res["cpu_baseline"] = cb = cpu_baseline(S, Bv, 0)
                cb["gpu_ms_per_step_same_work"] = round(unhoisted_ms, 3)
                cb["gpu_over_cpu_same_work"] = round(cb["seconds_per_step"] * 1e3 / unhoisted_ms, 1)
                if std_cfg:
                    # same seed-0 cameras, same initial texture: the CPU restatement's (first and only) step and the GPU's
                    # first step compute the same loss
                    cb["loss_rel_diff_vs_gpu_first_step"] = abs(cb["loss"] - first_step_loss) / abs(cb["loss"])
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                res["cpu_baseline"] = {"value": None, "error": repr(e)}
        check_fractions(res)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
