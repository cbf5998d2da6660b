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

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (the Winograd conv launches, ~83 % of
the step) against the fp32-MFMA peak from HIP events on the launch stream; `step_roofline` prices the whole
step with the algorithmic flop count of SURVEY.md 8d (396.9 GFLOP per 512^2 view-step, targets hoisted);
`cpu_baseline` times the CPU restatement (oracle/) of the reference's step on the host cores on a
bounded sample (1 view) and scales it to the 8-view step.
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
F_ALG_VIEW_512 = 396.9e9          # VGG fwd 189.35 + dgrad 189.35 + Gram fwd/bwd 2*9.13 GFLOP
PEAK_FP32_MFMA = 157.3e12         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
# MFMA flops the kernels actually EXECUTE per 512^2 view-step: Winograd F(2x2,3x3) does
# 16/36 of the direct-convolution multiplies for every layer but conv1_1 (direct: K padded 3->4 forward,
# M padded 3->32 in the input-gradient), the Gram forward computes only tiles on/above the diagonal.
F_EXEC_VIEW_512 = (0.906 * 4 / 3 + (189.35 - 0.906) * 16 / 36) * 1e9 \
    + (0.906 * 32 / 3 + (189.35 - 0.906) * 16 / 36) * 1e9 \
    + (2.147 + 2.147 + 2.147 * 3 / 4 + 2.147 * 10 / 16 + 0.537 * 10 / 16) * 1e9 + 9.13e9


def load_assets(size, device):
    cow = np.load(os.path.join(GOLDEN, "assets_cow_mesh.npz"))
    sty = np.load(os.path.join(GOLDEN, "assets_style1_512.npz"))["rgb_u8"]
    verts = torch.from_numpy(cow["verts"]).to(device)
    faces = torch.from_numpy(cow["faces"].astype(np.int64)).to(device)
    verts_uvs = torch.from_numpy(cow["verts_uvs"])[None].to(device)
    faces_uvs = torch.from_numpy(cow["faces_uvs"].astype(np.int64))[None].to(device)
    tex = torch.from_numpy(cow["texture_u8"]).to(torch.float32).div(255.0)[None].to(device)
    # second_approach.py:84-94: texture resized to size x size (bilinear, align_corners=False)
    tex = F.interpolate(tex.permute(0, 3, 1, 2), size=size, mode="bilinear", align_corners=False).permute(0, 2, 3, 1).contiguous()
    style = torch.from_numpy(sty).permute(2, 0, 1).to(torch.float32).div(255.0)
    if size != style.shape[1]:
        style = F.interpolate(style[None], size=size, mode="bilinear", align_corners=False, antialias=True)[0]
    return verts, faces, verts_uvs, faces_uvs, tex, style.contiguous().to(device)


def cpu_baseline(size, seed_cam):
    """CPU restatement of ONE view of the reference step, as the reference executes it
    (second_approach.py:157-189: content render + current render, compute_perceptual_loss = 3 VGG
    forwards + 1 backward, texture backward, Adam), on the host cores; scaled to 8 views."""
    from oracle import perceptual_ref as P
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
    elev, azim = RR.random_camera_angles(1, lambda k: torch.rand(k, generator=g).numpy())
    R, T = RR.look_at_view_transform(2.10, elev, azim, at=(0, 0.10, 0.25))
    model = P.make_vgg19_features(seed=0)
    m = np.zeros_like(tex); v = np.zeros_like(tex)

    def one_view_step():
        content, _, _ = RR.render_views(cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"], tex, R, T, size, threads)
        cur, _, frags = RR.render_views(cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"], tex, R, T, size, threads)
        cur_t = torch.from_numpy(cur).requires_grad_(True)
        loss = P.perceptual_loss_ref(cur_t, torch.from_numpy(content), style, model)
        loss.backward()
        gt = RR.shade_bwd(cur_t.grad.numpy()[0], frags[0], cow["verts_uvs"], cow["faces_uvs"], tex).astype(np.float32)
        RR.adam_step(tex, gt, m, v, 1)
        return float(loss)

    one_view_step()                       # warm-up (page-in, MKL threads)
    t0 = time.time()
    reps = 0
    while reps < 2 or time.time() - t0 < 10.0:
        one_view_step()
        reps += 1
        if time.time() - t0 > 40.0:
            break
    t_view = (time.time() - t0) / reps
    return {"value": 1.0 / (8.0 * t_view), "unit": "iter/s (8 views, %dx%d)" % (size, size), "cores": threads,
            "kind": "port",
            "sample": "%d x one-view step (%.2f s each: 2 naive renders with %d OpenMP threads, 3 VGG-19 forwards + 1 "
                      "backward on torch-CPU fp32, texture backward, Adam), scaled x8 views" % (reps, t_view, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--views", type=int, default=8, help="views per GPU per step")
    ap.add_argument("--no-hoist", action="store_true", help="recompute content renders + VGG targets every step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-kernels", action="store_true", help="extra untimed steps with per-kernel-family HIP events")
    ap.add_argument("--target", choices=["texture", "mesh", "both"], default="texture",
                    help="optimization_target (BASELINE configs[1] = texture; configs[4] = both)")
    args = ap.parse_args()

    if not os.path.exists(os.path.join(PKG, "lib", "libst3d.so")) and int(os.environ.get("RANK", "0")) == 0 \
            and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        import __graft_entry__ as _ge           # a checkout without the prebuilt library: compile it (there is no CPU fallback)
        _ge.build()
    from st3d import optim as st3d_optim
    rank, world, local = st3d_optim.init_distributed()
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libst3d has no CPU fallback")
    device = torch.device(f"cuda:{local % max(torch.cuda.device_count(), 1)}")     # > device_count only in gloo rehearsals
    torch.cuda.set_device(device)

    import losses as L
    import style_transfer as ST
    import utils as U
    U.device = ST.device = L.device = device
    from st3d.render import (AmbientLights, FoVPerspectiveCameras, MeshRasterizer, MeshRenderer, RasterizationSettings,
                             SoftPhongShader)

    S, Bv = args.size, args.views
    global_views = Bv * world
    verts, faces, verts_uvs, faces_uvs, tex, style_image = load_assets(S, device)
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

    def content_targets():
        with torch.no_grad():
            c, cm = U.render_meshes(renderer, content_mesh, my_cams)
            return U.apply_background(c, cm, background_type="white", background=style_tensors)

    content_tensors = content_targets()

    def step():
        optimizer.zero_grad()
        content = content_tensors if not args.no_hoist else content_targets()
        mesh = U.build_mesh(out["verts_uvs"], out["faces_uvs"], texture_map, out["verts"], out["faces"])
        cur, masks = U.render_meshes(renderer, mesh, my_cams)
        cur = U.apply_background(cur, masks, background_type="white", background=style_tensors)
        loss = L.compute_second_approach_loss(cur, content, style_tensors, vgg, 1e6, 1.0, out["verts"], verts, mesh,
                                              reg_weights, args.target, batch_denom=global_views)
        loss.backward()
        optimizer.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if world > 1:       # RCCL builds its communicator lazily on the first collective: keep that out of the timed steps even at --warmup 0
        torch.distributed.all_reduce(torch.zeros(texture_map.numel(), device=device))
    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        loss = step()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    final_loss = float(loss.detach())
    if world > 1:
        t = torch.tensor([elapsed, dev_ms], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed, dev_ms = float(t[0]), float(t[1])
        lt = torch.tensor([final_loss], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(lt)
        final_loss = float(lt[0])

    kernels = None
    nprof = 5
    if args.profile_kernels or world == 1:
        # per-kernel-family HIP events (recorded by libst3d on the stream the kernels are launched on) over nprof
        # further steps of the same loop, right after the timed region
        plan = vgg.plan(Bv, S)
        plan.profile(True)
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        pr = plan.profile_read()
        plan.profile(False)
        s2 = (S / 512.0) ** 2
        wino = os.environ.get("ST3D_CONV") != "direct"
        f_conv = (189.35e9 - 0.906e9 if wino else 189.35e9) * s2 * Bv
        flops = {"conv_fwd": f_conv, "conv_dgrad": f_conv, "convx_fwd": 0.906e9 * s2 * Bv if wino else 0.0,
                 "convx_dgrad": 0.906e9 * s2 * Bv if wino else 0.0, "gram_fwd": 9.13e9 * s2 * Bv, "gram_bwd": 9.13e9 * s2 * Bv}
        kernels = {}
        for k, v in pr.items():
            ms = v["ms"] / nprof
            kernels[k] = {"ms_per_step": round(ms, 4), "launches_per_step": v["launches"] // nprof}
            if flops.get(k) and ms > 0:
                kernels[k]["alg_tflops"] = round(flops[k] / (ms * 1e-3) / 1e12, 2)
                kernels[k]["alg_frac_of_peak"] = round(flops[k] / (ms * 1e-3) / PEAK_FP32_MFMA, 4)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        s2 = (S / 512.0) ** 2
        wino = os.environ.get("ST3D_CONV") != "direct"
        f_alg_step = F_ALG_VIEW_512 * s2 * Bv + (0 if not args.no_hoist else 145.86e9 * s2 * Bv)
        f_exec_step = F_EXEC_VIEW_512 * s2 * Bv if wino else f_alg_step
        step_s = dev_ms / args.steps * 1e-3
        std_cfg = S == 512 and Bv == 8 and not args.no_hoist and wino
        traffic = {}
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))     # newest round last
        tpath = tfiles[-1] if tfiles else ""
        if tpath and std_cfg:
            with open(tpath) as fh:
                traffic = json.load(fh)
        step_roofline = {
            "bound": "mfma", "achieved": round(f_alg_step / step_s / 1e12, 3), "peak": PEAK_FP32_MFMA / 1e12, "unit": "TFLOP/s",
            "frac": round(f_alg_step / step_s / PEAK_FP32_MFMA, 4),
            "executed": round(f_exec_step / step_s / 1e12, 3), "executed_frac": round(f_exec_step / step_s / PEAK_FP32_MFMA, 4),
            "traffic": (traffic["fetch_bytes_raw_per_step"] + traffic["write_bytes_per_step"]) if traffic else None,
            "note": "whole step per GPU (renders, losses, Adam and host gaps included): algorithmic flops of the direct "
                    "convolutions + Grams (%.1f GF/step, SURVEY 8d) over the HIP-event time of the timed region (%.3f ms/step)"
                    % (f_alg_step / 1e9, dev_ms / args.steps)}
        roofline = step_roofline
        if kernels and kernels.get("conv_fwd", {}).get("ms_per_step", 0) > 0 and kernels.get("conv_dgrad", {}).get("ms_per_step", 0) > 0:
            # the dominant kernel: the Winograd conv launches (forward + input-gradient), priced together
            kms = kernels["conv_fwd"]["ms_per_step"] + kernels["conv_dgrad"]["ms_per_step"]
            kn = kernels["conv_fwd"]["launches_per_step"] + kernels["conv_dgrad"]["launches_per_step"]
            f_conv_alg = 2 * ((189.35e9 - 0.906e9) if wino else 189.35e9) * s2 * Bv
            f_conv_exec = f_conv_alg * (16.0 / 36.0 if wino else 1.0)
            ktraffic = None
            if traffic and kn == traffic.get("wino_kernel_launches_per_step"):
                ktraffic = (traffic["wino_fetch_bytes_raw_per_step"] + traffic["wino_write_bytes_per_step"]) / kn
            roofline = {
                "bound": "mfma", "achieved": round(f_conv_alg / (kms * 1e-3) / 1e12, 3), "peak": PEAK_FP32_MFMA / 1e12,
                "unit": "TFLOP/s", "frac": round(f_conv_alg / (kms * 1e-3) / PEAK_FP32_MFMA, 4),
                "traffic": ktraffic,
                "executed": round(f_conv_exec / (kms * 1e-3) / 1e12, 3),
                "executed_frac": round(f_conv_exec / (kms * 1e-3) / PEAK_FP32_MFMA, 4),
                "kernel": "wino_kernel<MODE,EPI> (csrc/wino.hip)" if wino else "conv3x3_kernel (csrc/conv.hip)",
                "launches_per_step": kn, "avg_launch_ms": round(kms / kn, 4), "ms_per_step": round(kms, 4),
                "share_of_step": round(kms / (dev_ms / args.steps), 4),
                "note": "achieved = ALGORITHMIC flops of the direct 3x3 convolutions these launches replace (2*9*Cin*Cout per "
                        "output pixel, forward + input-gradient of every VGG conv but conv1_1: %.1f GF/step = %.2f GF per "
                        "launch on average) / their HIP-event time, measured on the launch stream over %d steps after the timed "
                        "region.  It exceeds the fp32 MFMA peak because the kernel is Winograd F(2x2,3x3): 16 instead of 36 "
                        "multiplies per 2x2 outputs, still fp32 products + fp32 accumulation.  executed = MFMA flops actually "
                        "issued (16/36 of the algorithmic count) = matrix-pipe utilisation.  traffic = HBM bytes per launch "
                        "(average) from the committed PMC passes %s (FETCH_SIZE raw + WRITE_SIZE; "
                        "gfx950 under-reports 16-B/lane read streams by up to 2x), not re-measured live."
                        % (f_conv_alg / 1e9, f_conv_alg / 1e9 / kn, nprof, os.path.relpath(tpath, ROOT) if tpath else "(none)")}
        tgt = {"texture": "texture-only optimisation", "mesh": "vertex optimisation", "both": "joint vertex + texture optimisation"}[args.target]
        res = {
            "metric": "style-transfer iters/sec (512x512, 8 views, cow_mesh)",
            "value": round(args.steps * world * (Bv / 8.0) / elapsed, 4),
            "unit": "iter/s (one iter = 8 views of %dx%d: render + VGG-19 fwd/bwd + Gram/content loss + Adam)" % (S, S),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic views (seeded cameras) of cow_mesh + Style_1 fixtures; "
                                                            "seeded He-normal VGG-19 weights (pretrained weights need a download)",
            "config": {"workload": "BASELINE.json configs[1]: cow_mesh + Style_1, %dx%d, %d views/GPU/iter, %s "
                                   "(second_approach loop body)" % (S, S, Bv, tgt),
                       "global_views_per_step": global_views, "texture": "%dx%d" % (S, S),
                       "targets_hoisted": not args.no_hoist, "parallelism": "views sharded dp%d, RCCL all-reduce of the texture gradient" % world},
            "final_loss": final_loss,
            "roofline": roofline,
            "step_roofline": step_roofline,
        }
        if kernels:
            res["kernels"] = kernels
        if not args.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline(S, 0)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                res["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
