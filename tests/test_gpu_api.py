"""GPU parity tests at the drop-in API level: the reference's function surface
(style_transfer.py / losses.py / utils.py / the two CLIs) running on libst3d, checked against the
golden vectors produced by the reference's own code, against the CPU oracle, and -- at
BASELINE.json's full size -- through size-independent properties."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available()
    import losses as L
    import style_transfer as ST
    import utils as U
    dev = torch.device("cuda:0")
    U.device = ST.device = L.device = dev
    return ST, L, U, dev


@pytest.fixture(scope="module")
def vgg(mods):
    _, _, U, _ = mods
    return U.get_vgg(seed=0)


@pytest.fixture(scope="module")
def scene(mods, cow):
    _, _, U, dev = mods
    from st3d.render import FoVPerspectiveCameras, MeshRasterizer, MeshRenderer, RasterizationSettings, SoftPhongShader

    def make(S, T, B, seed=0):
        rng = np.random.default_rng(seed)
        tex_np = rng.random((T, T, 3), dtype=np.float32)
        from oracle import render_ref as rr
        g = torch.Generator().manual_seed(seed)
        elev, azim = rr.random_camera_angles(B, lambda k: torch.rand(k, generator=g).numpy())
        R, Tt = rr.look_at_view_transform(2.10, elev, azim, at=(0, 0.10, 0.25))
        mesh = U.build_mesh(torch.from_numpy(cow["verts_uvs"])[None].to(dev),
                            torch.from_numpy(cow["faces_uvs"].astype(np.int64))[None].to(dev),
                            torch.from_numpy(tex_np)[None].to(dev), torch.from_numpy(cow["verts"]).to(dev),
                            torch.from_numpy(cow["faces"].astype(np.int64)).to(dev))
        renderer = MeshRenderer(MeshRasterizer(None, RasterizationSettings(image_size=S)), SoftPhongShader())
        cams = FoVPerspectiveCameras(R=torch.from_numpy(R), T=torch.from_numpy(Tt), device=dev)
        return mesh, renderer, cams, tex_np, R, Tt
    return make


def test_get_features_fused_equals_generic_walk_and_oracle(mods, vgg, golden_dir):
    ST, _, _, dev = mods
    from oracle import perceptual_ref as P
    d = np.load(os.path.join(golden_dir, "g3_perceptual.npz"))
    x = torch.from_numpy(d["cur"]).to(dev)
    fused = ST.get_features(x, vgg)
    assert list(fused) == ["conv1_1", "conv2_1", "conv3_1", "conv4_1", "conv4_2", "conv5_1"]

    class Walk:                     # same modules, but not recognised -> the reference's generic loop
        _modules = vgg._modules
    generic = ST.get_features(x.clone(), Walk)
    ref = P.get_features_ref(torch.from_numpy(d["cur"]), P.make_vgg19_features(seed=0))
    for k in fused:
        assert float(fused[k].min()) == 0.0
        # fused = Winograd kernels, generic walk = direct kernels: same fp32 arithmetic class, different order
        assert float((fused[k] - generic[k]).abs().max()) <= 5e-5 * float(generic[k].abs().max())
        scale = float(ref[k].abs().max())
        assert float((fused[k].cpu() - ref[k]).abs().max()) <= 2e-4 * scale
    # custom layer dict incl. a pool output and the unused tail (module 36)
    f2 = ST.get_features(x, vgg, layers={"4": "pool1", "36": "pool5"})
    assert f2["pool1"].shape == (2, 64, 32, 32) and f2["pool5"].shape == (2, 512, 2, 2)


def test_gram_matrix_autograd(mods):
    ST, _, _, dev = mods
    torch.manual_seed(0)
    t = torch.rand(2, 64, 16, 16, device=dev, requires_grad=True)
    g = ST.gram_matrix(t)
    w = torch.randn_like(g)
    (g * w).sum().backward()
    td = t.detach().double().cpu().requires_grad_(True)
    f = td.reshape(2, 64, 256)
    gr = torch.bmm(f, f.transpose(1, 2))
    (gr * w.double().cpu()).sum().backward()
    assert float((g.detach().cpu().double() - gr.detach()).abs().max()) <= 2e-5 * float(gr.abs().max())
    assert float((t.grad.cpu().double() - td.grad).abs().max()) <= 2e-5 * float(td.grad.abs().max())


@pytest.mark.parametrize("fixture,B,S", [("g3_perceptual.npz", 2, 64), ("g3b_perceptual_96.npz", 1, 96)])
def test_compute_perceptual_loss_matches_reference_golden(mods, vgg, golden_dir, fixture, B, S):
    """loss within 2e-5 relative, d loss/d current within 1e-4 relative L2 of the reference's
    losses.compute_perceptual_loss + autograd (fp32, different summation order; measured 4e-7 / 2e-6,
    tools/parity_margins.py)."""
    _, L, _, dev = mods
    d = np.load(os.path.join(golden_dir, fixture))
    cur = torch.from_numpy(d["cur"]).to(dev).requires_grad_(True)
    kw = {}
    if "style_weight" in d.files:
        kw = {"style_weight": float(d["style_weight"]), "content_weight": float(d["content_weight"])}
    loss = L.compute_perceptual_loss(cur, torch.from_numpy(d["con"]).to(dev), torch.from_numpy(d["sty"]).to(dev), vgg, **kw)
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) <= 2e-5 * float(d["loss"])
    gref = torch.from_numpy(d["grad"])
    assert float((cur.grad.cpu() - gref).norm() / gref.norm()) <= 1e-4
    # the 'texture' branch of compute_second_approach_loss is the same number (losses.py:103-104)
    l2 = L.compute_second_approach_loss(cur.detach(), torch.from_numpy(d["con"]).to(dev), torch.from_numpy(d["sty"]).to(dev),
                                        vgg, kw.get("style_weight", 1e6), kw.get("content_weight", 1), None, None, None, {},
                                        "texture")
    assert l2.item() == loss.item()
    with pytest.raises(AssertionError):
        L.compute_perceptual_loss(cur[:1], torch.from_numpy(d["con"]).to(dev).repeat(2, 1, 1, 1)[:3],
                                  torch.from_numpy(d["sty"]).to(dev), vgg)


def test_first_approach_loss_matches_reference_golden(mods, golden_dir):
    _, L, _, dev = mods
    d = np.load(os.path.join(golden_dir, "g3_perceptual.npz"))
    r = torch.from_numpy(d["cur"]).to(dev).requires_grad_(True)
    loss = L.compute_first_approach_loss(r, torch.from_numpy(d["masks"]).to(dev), torch.from_numpy(d["con"]).to(dev), None,
                                         None, None, {}, "texture")
    loss.backward()
    assert abs(loss.item() - float(d["first_loss"])) <= 1e-6
    rc = torch.from_numpy(d["cur"]).requires_grad_(True)
    m = torch.from_numpy(d["masks"])
    torch.nn.functional.mse_loss(rc * m, torch.from_numpy(d["con"]) * m).backward()
    torch.testing.assert_close(r.grad.cpu(), rc.grad, rtol=1e-5, atol=1e-9)


def test_style_transfer_matches_reference_trajectory(mods, vgg, golden_dir):
    """Six steps of the reference's style_transfer() (golden G4): final pixels within 1e-4 abs
    (Adam's m/sqrt(v) normalisation amplifies tiny gradient differences in the first steps; measured 9e-6)."""
    ST, _, _, dev = mods
    d = np.load(os.path.join(golden_dir, "g4_style_transfer.npz"))
    res = ST.style_transfer(torch.from_numpy(d["init"]).to(dev), torch.from_numpy(d["con"]).to(dev),
                            torch.from_numpy(d["sty"]).to(dev), vgg, steps=int(d["steps"]), style_weight=1e6,
                            content_weight=1, lr=float(d["lr"]))
    assert res.requires_grad and res.is_leaf
    err = (res.detach().cpu() - torch.from_numpy(d["result"])).abs()
    assert float(err.max()) <= 1e-4, float(err.max())
    moved = (torch.from_numpy(d["result"]) - torch.from_numpy(d["init"])).abs().mean()
    assert float(err.mean()) <= 0.01 * float(moved)


def test_render_meshes_pixels_and_texture_gradient_match_oracle(mods, vgg, scene, cow):
    """Rendered pixels within 2e-6 of the CPU restatement; d loss/d texture through render ->
    VGG -> losses within 2e-3 relative L2 of the oracle's chain."""
    _, L, U, dev = mods
    from oracle import perceptual_ref as P
    from oracle import render_ref as rr
    S, T, B = 96, 48, 2
    mesh0, renderer, cams, tex_np, R, Tt = scene(S, T, B, seed=3)
    out = U.setup_optimizations("texture", mesh0, 0.01)
    mesh = U.build_mesh(out["verts_uvs"], out["faces_uvs"], out["texture_map"], out["verts"], out["faces"])
    cur, masks = U.render_meshes(renderer, mesh, [cams[0], cams[1]])          # list of single cameras, as the reference passes
    assert cur.shape == (B, 3, S, S) and masks.shape == (B, 1, S, S) and cur.requires_grad and not masks.requires_grad
    imgs, mref, frags = rr.render_views(cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"], tex_np, R, Tt, S, 8)
    np.testing.assert_allclose(cur.detach().cpu().numpy(), imgs, rtol=0, atol=2e-6)
    np.testing.assert_array_equal(masks.cpu().numpy(), mref)
    rgba = renderer(meshes_world=mesh, cameras=cams[0])
    assert rgba.shape == (1, S, S, 4)
    torch.testing.assert_close(rgba[0, ..., :3].permute(2, 0, 1), cur[0].detach(), rtol=0, atol=0)
    assert torch.equal((rgba[0, ..., 3] > 0).float(), masks[0, 0])
    sty = torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(9))
    con = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(8))
    loss = L.compute_perceptual_loss(cur, con.to(dev), sty.to(dev).expand(B, -1, -1, -1), vgg)
    loss.backward()
    cur_t = torch.from_numpy(imgs).requires_grad_(True)
    ref = P.perceptual_loss_ref(cur_t, con, sty.expand(B, -1, -1, -1), P.make_vgg19_features(seed=0))
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 2e-4 * abs(ref.item())
    gref = np.zeros((T, T, 3), np.float64)
    for b in range(B):
        rr.shade_bwd(cur_t.grad.numpy()[b], frags[b], cow["verts_uvs"], cow["faces_uvs"], tex_np, gref)
    g = out["texture_map"].grad[0].cpu().numpy()
    assert np.linalg.norm(g - gref) <= 2e-3 * np.linalg.norm(gref)
    # fused Adam through the optimizer object the reference's loop drives
    before = out["texture_map"].detach().clone()
    out["optimizer"].step()
    moved = (out["texture_map"].detach() - before).abs()
    seen = torch.from_numpy(gref != 0).to(dev)
    assert float(moved[0][seen].min()) > 0 and float(moved[0][~seen].max()) == 0.0      # unseen texels stay put (notes.txt:12-16)
    out["optimizer"].zero_grad()
    assert out["texture_map"].grad is None


def test_apply_background_modes(mods):
    _, _, U, dev = mods
    torch.manual_seed(0)
    t = torch.rand(2, 3, 32, 32, device=dev, requires_grad=True)
    m = (torch.rand(2, 1, 32, 32, device=dev) > 0.5).float()
    bg = torch.rand(2, 3, 32, 32, device=dev)
    o = U.apply_background(t, m, "style", bg)
    torch.testing.assert_close(o.detach(), t.detach() * m + bg * (1 - m), rtol=0, atol=0)
    o.sum().backward()
    torch.testing.assert_close(t.grad, m.expand_as(t), rtol=0, atol=0)
    n = U.apply_background(t.detach(), m, "noise")
    assert torch.equal(n * m, t.detach() * m) and float(((n - t.detach()).abs() * (1 - m)).sum()) > 0


def _write_cow_assets(tmp, cow, golden_dir, tex_size=64):
    from PIL import Image
    from st3d import io as stio
    tex = torch.from_numpy(cow["texture_u8"][::1024 // tex_size, ::1024 // tex_size].copy()).float() / 255
    obj = os.path.join(tmp, "cow.obj")
    stio.save_obj(obj, torch.from_numpy(cow["verts"]), torch.from_numpy(cow["faces"].astype(np.int64)),
                  torch.from_numpy(cow["verts_uvs"]), torch.from_numpy(cow["faces_uvs"].astype(np.int64)), tex)
    sty = np.load(os.path.join(golden_dir, "assets_style1_512.npz"))["rgb_u8"]
    style = os.path.join(tmp, "style.png")
    Image.fromarray(sty).save(style)
    return obj, style


def test_second_approach_cli_end_to_end(mods, cow, golden_dir, tmp_path):
    """The drop-in CLI (config-1 plumbing at toy size): artefacts, log format, decreasing loss."""
    import second_approach as SA
    obj, style = _write_cow_assets(str(tmp_path), cow, golden_dir)
    outp = str(tmp_path / "out2")
    SA.main(["--obj_path", obj, "--style_path", style, "--size", "64", "--n_views", "3", "--batch_size", "2",
             "--epochs", "4", "--output_path", outp, "--seed", "0", "--lr", "0.02"])
    log = open(os.path.join(outp, "log.txt")).read().splitlines()
    assert log[0] == "Logger:" and len(log) == 5 and log[1].startswith("Epoch 0, Loss ")
    losses = [float(line.split("Loss ")[1]) for line in log[1:]]
    assert losses[-1] < losses[0]
    assert sorted(os.listdir(os.path.join(outp, "current_images"))) == ["view_0.png", "view_1.png", "view_2.png"]
    assert len(os.listdir(os.path.join(outp, "final_render"))) == 12
    for f in ("final.obj", "final.mtl", "final.png"):
        assert os.path.exists(os.path.join(outp, f))


def test_first_approach_cli_end_to_end(mods, cow, golden_dir, tmp_path):
    import first_approach as FA
    obj, style = _write_cow_assets(str(tmp_path), cow, golden_dir)
    outp = str(tmp_path / "out1")
    FA.main(["--obj_path", obj, "--style_path", style, "--size", "64", "--n_views", "2", "--batch_size", "2",
             "--n_style_transfer_steps", "5", "--n_mse_steps", "6", "--output_path", outp, "--seed", "0"])
    log = open(os.path.join(outp, "log.txt")).read().splitlines()
    assert log[0] == "Logger:" and len(log) == 7 and log[1].startswith("Batch 0, Step 0, Loss ")
    losses = [float(line.split("Loss ")[1]) for line in log[1:]]
    assert losses[-1] < losses[0]
    assert sorted(os.listdir(os.path.join(outp, "2d_style_transfer"))) == ["view_0.png", "view_1.png"]
    assert os.path.exists(os.path.join(outp, "final.obj"))


def test_full_size_properties_config2(mods, vgg, scene):
    """BASELINE.json configs[1] (512x512, 8 views, 512^2 texture): size-independent properties --
    the loss is reproducible bit for bit (ordered reductions), batch-linearity of the gradient
    (grad of 8 views == sum of two 4-view halves at batch_denom 8), adjointness of the texture
    scatter at full size, and five optimiser steps reduce the loss."""
    _, L, U, dev = mods
    S, T, B = 512, 512, 8
    mesh0, renderer, cams, tex_np, R, Tt = scene(S, T, B, seed=0)
    out = U.setup_optimizations("texture", mesh0, 0.01)
    tex = out["texture_map"]
    sty = torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        content, _ = U.render_meshes(renderer, mesh0, cams)
    plan = vgg.plan(B, S)

    def grads(cam_slice, denom):
        mesh = U.build_mesh(out["verts_uvs"], out["faces_uvs"], tex, out["verts"], out["faces"])
        cur, _ = U.render_meshes(renderer, mesh, cams[cam_slice])
        n = cur.shape[0]
        loss = L.compute_perceptual_loss(cur, content[cam_slice], sty.expand(n, -1, -1, -1), vgg.__class__ and vgg,
                                         batch_denom=denom)
        tex.grad = None
        loss.backward()
        return loss.detach().clone(), tex.grad.clone(), cur.detach()
    l8, g8, cur8 = grads(slice(0, 8), 8)
    l8b, g8b, _ = grads(slice(0, 8), 8)
    assert l8.item() == l8b.item()                                   # ordered reductions: bitwise reproducible loss
    assert float((g8 - g8b).norm() / g8.norm()) <= 1e-5               # texture scatter uses float atomics (order varies)
    # two plans of batch 4 with batch_denom 8 (what two ranks would compute)
    la, ga, _ = grads(slice(0, 4), 8)
    lb, gb, _ = grads(slice(4, 8), 8)
    assert abs((la + lb).item() - l8.item()) <= 2e-5 * abs(l8.item())
    assert float((ga + gb - g8).norm() / g8.norm()) <= 2e-4
    # adjointness at full size: <g, render(tex + d) - render(tex)> == <scatter(g), d>
    from st3d import ops
    with torch.no_grad():
        d = torch.randn_like(tex) * 0.01
        mesh_d = U.build_mesh(out["verts_uvs"], out["faces_uvs"], tex + d, out["verts"], out["faces"])
        cur_d, _ = U.render_meshes(renderer, mesh_d, cams)
        gimg = torch.randn_like(cur8)
        lhs = float((gimg.double() * (cur_d.double() - cur8.double())).sum())
    mesh = U.build_mesh(out["verts_uvs"], out["faces_uvs"], tex, out["verts"], out["faces"])
    cur, _ = U.render_meshes(renderer, mesh, cams)
    tex.grad = None
    cur.backward(gimg)
    rhs = float((tex.grad.double() * d.double()).sum())
    assert abs(lhs - rhs) <= 1e-3 * max(abs(lhs), abs(rhs))
    # optimisation makes progress
    losses = []
    for _ in range(5):
        out["optimizer"].zero_grad()
        mesh = U.build_mesh(out["verts_uvs"], out["faces_uvs"], tex, out["verts"], out["faces"])
        cur, _ = U.render_meshes(renderer, mesh, cams)
        loss = L.compute_perceptual_loss(cur, content, sty.expand(B, -1, -1, -1), vgg)
        loss.backward()
        out["optimizer"].step()
        losses.append(loss.item())
    assert losses[-1] < losses[0] and all(np.isfinite(losses))


def test_both_target_gradients_match_oracle(mods, vgg, scene, cow):
    """optimization_target 'both' (reference losses.py:117-124): loss = main_w * perceptual + regularisers;
    d loss/d verts and d loss/d texture through render -> VGG -> losses vs the CPU oracle chain."""
    _, L, U, dev = mods
    from oracle import mesh_ref as M
    from oracle import perceptual_ref as P
    from oracle import render_ref as rr
    S, T, B = 64, 32, 2
    mesh0, renderer, cams, tex_np, R, Tt = scene(S, T, B, seed=5)
    out = U.setup_optimizations("both", mesh0, 0.01)
    with torch.no_grad():
        out["verts"].add_(0.003 * torch.randn(out["verts"].shape, generator=torch.Generator().manual_seed(0)).to(dev))
    weights = {"main_loss_weight": 3.0, "mesh_verts_weight": 0.5, "mesh_edge_loss_weight": 1.5,
               "mesh_laplacian_smoothing_weight": 0.8, "mesh_normal_consistency_weight": 1.2}
    sty = torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(2))
    con = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(3))
    mesh = U.build_mesh(out["verts_uvs"], out["faces_uvs"], out["texture_map"], out["verts"], out["faces"])
    cur, _ = U.render_meshes(renderer, mesh, cams)
    target_verts = torch.from_numpy(cow["verts"]).to(dev)
    loss = L.compute_second_approach_loss(cur, con.to(dev), sty.to(dev).expand(B, -1, -1, -1), vgg, 1e6, 1.0, out["verts"],
                                          target_verts, mesh, weights, "both")
    loss.backward()
    # oracle
    v_np = out["verts"].detach().cpu().numpy()
    imgs, _, frags = rr.render_views(v_np, cow["faces"], cow["verts_uvs"], cow["faces_uvs"], tex_np, R, Tt, S, 8)
    np.testing.assert_allclose(cur.detach().cpu().numpy(), imgs, rtol=0, atol=2e-6)
    cur_t = torch.from_numpy(imgs).requires_grad_(True)
    perc = P.perceptual_loss_ref(cur_t, con, sty.expand(B, -1, -1, -1), P.make_vgg19_features(seed=0))
    perc.backward()
    gtex_ref, gverts_ref = rr.render_bwd_views(cur_t.grad.numpy() * 3.0, frags, v_np, cow["faces"], cow["verts_uvs"],
                                               cow["faces_uvs"], tex_np, R, Tt)
    vd = torch.from_numpy(v_np).double().requires_grad_(True)
    faces = torch.from_numpy(cow["faces"].astype(np.int64))
    regs = (0.5 * M.verts_mse_ref(vd, torch.from_numpy(cow["verts"]).double()) + 1.5 * M.mesh_edge_loss_ref(vd, faces)
            + 0.8 * M.mesh_laplacian_smoothing_ref(vd, faces) + 1.2 * M.mesh_normal_consistency_ref(vd, faces))
    regs.backward()
    ref_total = 3.0 * perc.item() + regs.item()
    assert abs(loss.item() - ref_total) <= 2e-4 * abs(ref_total)
    gv_ref = gverts_ref + vd.grad.numpy()
    gv = out["verts"].grad.cpu().numpy()
    assert np.linalg.norm(gv - gv_ref) <= 5e-3 * np.linalg.norm(gv_ref), np.linalg.norm(gv - gv_ref) / np.linalg.norm(gv_ref)
    gt = out["texture_map"].grad[0].cpu().numpy()
    assert np.linalg.norm(gt - gtex_ref) <= 2e-3 * np.linalg.norm(gtex_ref)
    out["optimizer"].step()          # fused Adam over [verts, texture_map]
    assert torch.isfinite(out["verts"]).all() and torch.isfinite(out["texture_map"]).all()
    # the PyTorch3D-named regularisers are callable one by one (reference losses.py:85-87)
    mesh2 = U.build_mesh(out["verts_uvs"], out["faces_uvs"], out["texture_map"], out["verts"], out["faces"])
    e, l, n = L.mesh_edge_loss(mesh2), L.mesh_laplacian_smoothing(mesh2), L.mesh_normal_consistency(mesh2)
    assert e.item() > 0 and l.item() > 0 and n.item() > 0 and e.requires_grad


def test_second_approach_cli_both_target(mods, cow, golden_dir, tmp_path):
    """config-5-style plumbing at toy size: joint vertex + texture optimisation through the CLI."""
    import second_approach as SA
    obj, style = _write_cow_assets(str(tmp_path), cow, golden_dir)
    outp = str(tmp_path / "out_both")
    SA.main(["--obj_path", obj, "--style_path", style, "--size", "64", "--n_views", "2", "--batch_size", "2", "--epochs", "3",
             "--output_path", outp, "--seed", "0", "--optimization_target", "both", "--lr", "0.001", "--save_every", "0"])
    log = open(os.path.join(outp, "log.txt")).read().splitlines()
    assert len(log) == 4 and all(np.isfinite(float(line.split("Loss ")[1])) for line in log[1:])
    assert os.path.exists(os.path.join(outp, "final.obj"))


def test_mesh_without_uvs_gets_synthesised_ones(mods, golden_dir, tmp_path):
    """teapot-style OBJ (`v//vn` faces, no mtllib: SURVEY.md D3, BASELINE config 4): the reference crashes; the
    drop-in synthesises spherical UVs + a texture and runs."""
    import second_approach as SA
    from PIL import Image
    # octahedron
    (tmp_path / "o.obj").write_text("v 1 0 0\nv -1 0 0\nv 0 1 0\nv 0 -1 0\nv 0 0 1\nv 0 0 -1\nvn 0 0 1\n"
                                    "f 1//1 3//1 5//1\nf 3//1 2//1 5//1\nf 2//1 4//1 5//1\nf 4//1 1//1 5//1\n"
                                    "f 3//1 1//1 6//1\nf 2//1 3//1 6//1\nf 4//1 2//1 6//1\nf 1//1 4//1 6//1\n")
    sty = np.load(os.path.join(golden_dir, "assets_style1_512.npz"))["rgb_u8"]
    Image.fromarray(sty).save(tmp_path / "style.png")
    outp = str(tmp_path / "out_t")
    SA.main(["--obj_path", str(tmp_path / "o.obj"), "--style_path", str(tmp_path / "style.png"), "--size", "64", "--n_views", "2",
             "--batch_size", "2", "--epochs", "2", "--output_path", outp, "--seed", "0", "--save_every", "0"])
    log = open(os.path.join(outp, "log.txt")).read().splitlines()
    assert len(log) == 3 and all(np.isfinite(float(line.split("Loss ")[1])) for line in log[1:])


def test_soft_renderer_settings_match_oracle_through_the_api(mods, cow):
    """RasterizationSettings(blur_radius > 0, faces_per_pixel = 4) + BlendParams through MeshRenderer (the PyTorch3D
    configuration space outside the reference's fixed K=1 / blur 0): RGBA against the torch restatement of
    softmax_rgb_blend at the same coverage, autograd d/d texture and d/d verts against fp64 autograd."""
    _, _, U, dev = mods
    from oracle import render_ref as rr
    from oracle import soft_ref as SR
    from st3d.render import (BlendParams, FoVPerspectiveCameras, MeshRasterizer, MeshRenderer, RasterizationSettings,
                             SoftPhongShader)
    S, Tn, B, K = 64, 24, 2, 4
    blur, sigma, gamma, bg = 1e-3, 1e-3, 1e-2, (0.2, 0.5, 0.9)
    rng = np.random.default_rng(3)
    tex_np = rng.random((Tn, Tn, 3), dtype=np.float32)
    g = torch.Generator().manual_seed(11)
    elev, azim = rr.random_camera_angles(B, lambda k: torch.rand(k, generator=g).numpy())
    R, Tt = rr.look_at_view_transform(2.10, elev, azim, at=(0, 0.10, 0.25))
    verts = torch.from_numpy(cow["verts"]).to(dev).requires_grad_(True)
    tex = torch.from_numpy(tex_np)[None].to(dev).requires_grad_(True)
    mesh = U.build_mesh(torch.from_numpy(cow["verts_uvs"])[None].to(dev),
                        torch.from_numpy(cow["faces_uvs"].astype(np.int64))[None].to(dev), tex, verts,
                        torch.from_numpy(cow["faces"].astype(np.int64)).to(dev))
    rs = RasterizationSettings(image_size=S, blur_radius=blur, faces_per_pixel=K)
    assert rs.clip_barycentric_coords and not rs.is_hard
    renderer = MeshRenderer(MeshRasterizer(None, rs), SoftPhongShader(blend_params=BlendParams(sigma, gamma, bg)))
    cams = FoVPerspectiveCameras(R=torch.from_numpy(R), T=torch.from_numpy(Tt), device=dev)
    rgba = renderer(meshes_world=mesh, cameras=cams)
    assert rgba.shape == (B, S, S, 4)
    gimg = rng.standard_normal((B, S, S, 3)).astype(np.float32)
    (rgba[..., :3] * torch.from_numpy(gimg).to(dev)).sum().backward()
    # forward against the fp64 restatement at the GPU's coverage (bit-exact against the C oracle in test_gpu_kernels)
    from st3d import ops
    ndc = ops.project_verts(verts.detach(), cams.R, cams.T)
    frag = ops.raster_soft_fwd(ndc, mesh.faces_i32(), S, K, blur, True)
    p2f = frag[0].cpu().long()
    with torch.no_grad():
        for b in range(B):
            r, a = SR.soft_render(torch.from_numpy(cow["verts"]).double(), torch.from_numpy(R[b]).double(),
                                  torch.from_numpy(Tt[b]).double(), torch.from_numpy(cow["faces"]).long(), p2f[b],
                                  torch.from_numpy(cow["verts_uvs"]).double(), torch.from_numpy(cow["faces_uvs"]).long(),
                                  torch.from_numpy(tex_np).double(), S, True, sigma, gamma, bg)
            np.testing.assert_allclose(rgba[b, ..., :3].detach().cpu().numpy(), r.permute(1, 2, 0).numpy(), atol=2e-5)
            np.testing.assert_allclose(rgba[b, ..., 3].detach().cpu().numpy(), a.numpy(), atol=2e-5)
    # autograd wiring: the same kernels called directly (their gradients are pinned to fp64 autograd in
    # test_gpu_kernels.test_soft_shade_forward_and_backward_match_oracle); atomics reorder, hence not bitwise
    gt, geo = ops.shade_soft_bwd(torch.from_numpy(gimg).to(dev).permute(0, 3, 1, 2).contiguous(), frag,
                                 torch.from_numpy(cow["verts_uvs"]).to(dev), mesh.textures.faces_uvs_i32(),
                                 tex.detach()[0].contiguous(), sigma, gamma, bg)
    gv = ops.project_verts_bwd(verts.detach(), cams.R, cams.T, ops.raster_soft_bwd(geo, frag[0], ndc, mesh.faces_i32(), True))
    assert float((tex.grad[0] - gt).norm() / gt.norm()) <= 1e-5
    assert float((verts.grad - gv).norm() / gv.norm()) <= 1e-5
    # drop-in render_meshes thresholds alpha into the reference's mask (utils.py:72)
    with torch.no_grad():
        cur, masks = U.render_meshes(renderer, mesh, cams)
    assert set(torch.unique(masks).tolist()) <= {0.0, 1.0}
    assert torch.equal(masks[:, 0] > 0, rgba[..., 3].detach() > 0)
    # K=1 / blur 0 with a coloured background runs on the general kernels and agrees with the hard path + background
    hard = MeshRenderer(MeshRasterizer(None, RasterizationSettings(image_size=S)), SoftPhongShader())
    soft1 = MeshRenderer(MeshRasterizer(None, RasterizationSettings(image_size=S)),
                         SoftPhongShader(blend_params=BlendParams(background_color=bg)))
    assert hard.is_hard and not soft1.is_hard
    with torch.no_grad():
        h_rgb, h_mask = hard.render(mesh, cams)
        s_rgb, s_cov = soft1.render(mesh, cams)
    bgt = torch.tensor(bg, device=dev).view(1, 3, 1, 1)
    np.testing.assert_allclose(s_rgb.cpu().numpy(), (h_rgb * h_mask + bgt * (1 - h_mask)).cpu().numpy(), atol=2e-6)
    assert torch.equal(s_cov > 0, h_mask > 0)


def test_fused_regularisers_match_reference_goldens_and_autograd(mods, golden_dir):
    """compute_tv_loss / rgb_range_loss (reference losses.py:48-65) and the L2-to-original-texture term on the HIP path:
    values against the reference's own outputs (G3), gradients against torch autograd of the CPU restatement."""
    _, L, _, dev = mods
    from oracle import perceptual_ref as P
    d = np.load(os.path.join(golden_dir, "g3_perceptual.npz"))
    cur = torch.from_numpy(d["cur"]).to(dev).requires_grad_(True)
    masks = torch.from_numpy(d["masks"]).to(dev)
    tv = L.compute_tv_loss(cur, masks)
    assert abs(float(tv) - float(d["tv_loss"])) <= 2e-6
    (tv * 1.7).backward()
    cr = torch.from_numpy(d["cur"]).double().requires_grad_(True)
    (P.tv_loss_ref(cr, torch.from_numpy(d["masks"]).double()) * 1.7).backward()
    np.testing.assert_allclose(cur.grad.cpu().numpy(), cr.grad.numpy(), atol=1e-8, rtol=1e-5)

    tex = (torch.from_numpy(d["cur"]) * 3 - 1).permute(0, 2, 3, 1).contiguous()

    class M:
        class textures:
            t = tex.to(dev).requires_grad_(True)

            @staticmethod
            def maps_padded():
                return M.textures.t
    rl = L.rgb_range_loss(M)
    assert abs(float(rl) - float(d["rgb_range"])) <= 1e-2
    rl.backward()
    tr = tex.double().requires_grad_(True)
    P.rgb_range_loss_ref(tr).backward()
    assert torch.equal(M.textures.t.grad.cpu().double(), tr.grad)

    M.textures.t.grad = None
    orig = torch.rand(tex.shape, generator=torch.Generator().manual_seed(5))
    l2 = L.texture_l2_loss(M, orig.to(dev))
    ref = ((tex.double() - orig.double()) ** 2).mean()
    assert abs(float(l2) - float(ref)) <= 1e-6 * float(ref)
    l2.backward()
    np.testing.assert_allclose(M.textures.t.grad.cpu().numpy(), (2 * (tex - orig) / tex.numel()).numpy(), rtol=1e-5, atol=1e-10)
    # odd sizes, all-zero mask rows, non-square images
    g = torch.Generator().manual_seed(9)
    img = torch.rand(1, 3, 7, 13, generator=g)
    m = (torch.rand(1, 1, 7, 13, generator=g) > 0.4).float()
    x = img.to(dev).requires_grad_(True)
    t2 = L.compute_tv_loss(x, m.to(dev))
    xr = img.double().requires_grad_(True)
    r2 = P.tv_loss_ref(xr, m.double())
    assert abs(float(t2) - float(r2)) <= 1e-6
    t2.backward(); r2.backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.numpy(), atol=1e-7)


def test_second_approach_checkpoint_resume_and_optional_regularisers(mods, cow, golden_dir, tmp_path):
    """Additions behind default-off flags: TV / RGB-range / L2-to-original regularisers, a separate vertex learning
    rate, and checkpoint -> resume.  A run of 2 epochs + resume for 2 more must land where an uninterrupted 4-epoch run
    lands (same seed, deterministic cameras; float atomics in the texture scatter allow a small tolerance)."""
    import second_approach as SA
    obj, style = _write_cow_assets(str(tmp_path), cow, golden_dir)
    common = ["--obj_path", obj, "--style_path", style, "--size", "64", "--n_views", "2", "--batch_size", "2", "--seed", "0",
              "--lr", "0.02", "--optimization_target", "both", "--verts_lr", "0.0005", "--tv_weight", "0.3",
              "--rgb_range_weight", "0.01", "--texture_l2_weight", "5.0", "--save_every", "0"]
    full, part = str(tmp_path / "full"), str(tmp_path / "part")
    SA.main(common + ["--epochs", "4", "--output_path", full])
    SA.main(common + ["--epochs", "2", "--output_path", part, "--checkpoint_every", "1"])
    ck = os.path.join(part, "checkpoint.pt")
    blob = torch.load(ck, map_location="cpu", weights_only=True)
    assert blob["progress"] == 2 and blob["optimization_target"] == "both" and len(blob["optimizer"]["state"]) == 2
    assert blob["optimizer"]["lrs"] == [0.0005, 0.02]
    SA.main(common + ["--epochs", "4", "--output_path", part, "--resume", ck])
    lf = [float(l.split("Loss ")[1]) for l in open(os.path.join(full, "log.txt")).read().splitlines()[1:]]
    lp = [float(l.split("Loss ")[1]) for l in open(os.path.join(part, "log.txt")).read().splitlines()[1:]]
    assert len(lf) == 4 and len(lp) == 2            # the resumed run logs epochs 2 and 3 into a fresh log
    np.testing.assert_allclose(lp, lf[2:], rtol=2e-3)
    from PIL import Image
    a = np.asarray(Image.open(os.path.join(full, "final.png")), dtype=np.int32)
    b = np.asarray(Image.open(os.path.join(part, "final.png")), dtype=np.int32)
    assert np.abs(a - b).max() <= 2
    with pytest.raises(ValueError):
        SA.main(["--obj_path", obj, "--style_path", style, "--size", "64", "--n_views", "2", "--batch_size", "2", "--epochs", "1",
                 "--output_path", str(tmp_path / "bad"), "--resume", ck])          # texture-only run, 'both' checkpoint


def test_third_approach_cli_end_to_end(mods, cow, golden_dir, tmp_path):
    """notes.txt:38 of the reference ("3rd approach": short alternating rounds over many views), built from the same
    blocks: artefacts, log format, and the texture actually moves towards the stylised renders."""
    import third_approach as TA
    obj, style = _write_cow_assets(str(tmp_path), cow, golden_dir)
    outp = str(tmp_path / "out3")
    TA.main(["--obj_path", obj, "--style_path", style, "--size", "64", "--n_views", "3", "--batch_size", "2", "--n_rounds", "3",
             "--n_style_transfer_steps", "4", "--n_mse_steps", "3", "--output_path", outp, "--seed", "0"])
    log = open(os.path.join(outp, "log.txt")).read().splitlines()
    assert log[0] == "Logger:" and len(log) == 1 + 3 * 2 and log[1].startswith("Round 0, Batch 0, Loss ")
    assert all(np.isfinite(float(line.split("Loss ")[1])) for line in log[1:])
    assert sorted(os.listdir(os.path.join(outp, "2d_style_transfer"))) == ["view_0.png", "view_1.png", "view_2.png"]
    assert len(os.listdir(os.path.join(outp, "final_render"))) == 12 and os.path.exists(os.path.join(outp, "final.obj"))
    from PIL import Image
    tex0 = cow["texture_u8"]
    final = np.asarray(Image.open(os.path.join(outp, "final.png")))
    assert final.shape == (64, 64, 3) and tex0.shape[2] == 3          # resized to --size; it must differ from a plain resize
    assert np.abs(final.astype(np.int32) - np.asarray(Image.fromarray(tex0).resize((64, 64))).astype(np.int32)).max() > 3


def test_full_size_single_view_loss_and_gradient_match_oracle(mods, vgg, scene, golden_dir):
    """BASELINE.json configs[1] resolution (512x512, 512^2 texture, the real Style_1 fixture), ONE view so the CPU oracle
    finishes in seconds: rendered pixels, perceptual loss and d loss / d texture against the torch-CPU restatement of
    the reference step (the same functions the G3 goldens pin to the reference's own code)."""
    _, L, U, dev = mods
    from oracle import perceptual_ref as P
    from oracle import render_ref as rr
    S, T = 512, 512
    mesh0, renderer, cams, tex_np, R, Tt = scene(S, T, 1, seed=3)
    cow = np.load(os.path.join(golden_dir, "assets_cow_mesh.npz"))
    sty_u8 = np.load(os.path.join(golden_dir, "assets_style1_512.npz"))["rgb_u8"]
    style = torch.from_numpy(sty_u8).permute(2, 0, 1).float().div(255.0)[None]
    out = U.setup_optimizations("texture", mesh0, 0.01)
    with torch.no_grad():
        content, _ = U.render_meshes(renderer, mesh0, cams)
        out["texture_map"].add_(0.05 * torch.randn(out["texture_map"].shape, generator=torch.Generator().manual_seed(1)).to(dev))
    mesh = U.build_mesh(out["verts_uvs"], out["faces_uvs"], out["texture_map"], out["verts"], out["faces"])
    cur, _ = U.render_meshes(renderer, mesh, cams)
    loss = L.compute_perceptual_loss(cur, content, style.to(dev), vgg)
    loss.backward()
    # oracle: naive rasteriser + shading, torch-CPU VGG, texture scatter
    tex_cur = out["texture_map"].detach().cpu().numpy()[0]
    img0, _, _ = rr.render_views(cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"], tex_np, R, Tt, S, 16)
    img1, _, frags = rr.render_views(cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"], tex_cur, R, Tt, S, 16)
    np.testing.assert_allclose(cur.detach().cpu().numpy(), img1, rtol=0, atol=2e-6)
    np.testing.assert_allclose(content.cpu().numpy(), img0, rtol=0, atol=2e-6)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cur_t = torch.from_numpy(img1).requires_grad_(True)
    ref = P.perceptual_loss_ref(cur_t, torch.from_numpy(img0), style, P.make_vgg19_features(seed=0))
    ref.backward()
    assert abs(loss.item() - float(ref)) <= 2e-5 * float(ref), (loss.item(), float(ref))
    gt_ref = rr.shade_bwd(cur_t.grad.numpy()[0], frags[0], cow["verts_uvs"], cow["faces_uvs"], tex_cur)
    g = out["texture_map"].grad[0].cpu().double().numpy()
    rel = np.linalg.norm(g - gt_ref) / np.linalg.norm(gt_ref)
    assert rel <= 1e-4, rel


@pytest.mark.parametrize("S,B", [(90, 2), (100, 1), (34, 1)])
def test_sizes_off_the_fast_path_match_oracle(mods, vgg, S, B):
    """Image sizes that are not multiples of 16 (the reference accepts any --size): odd intermediate resolutions fall off
    the Winograd / fused-pool kernels onto the direct ones, MaxPool2d floors.  Loss and gradient vs the CPU oracle."""
    ST, L, _, dev = mods
    from oracle import perceptual_ref as P
    g = torch.Generator().manual_seed(S)
    cur, con = torch.rand(B, 3, S, S, generator=g), torch.rand(B, 3, S, S, generator=g)
    sty = torch.rand(1, 3, S, S, generator=g).repeat(B, 1, 1, 1)
    x = cur.clone().to(dev).requires_grad_(True)
    loss = L.compute_perceptual_loss(x, con.to(dev), sty.to(dev), vgg)
    loss.backward()
    xr = cur.clone().requires_grad_(True)
    model = P.make_vgg19_features(seed=0)
    ref = P.perceptual_loss_ref(xr, con, sty, model)
    ref.backward()
    assert abs(loss.item() - float(ref.detach())) <= 2e-5 * float(ref.detach())
    assert float((x.grad.cpu() - xr.grad).norm() / xr.grad.norm()) <= 1e-4
    # every tap incl. the unused tail (module 36 = pool5) at these sizes
    feats = ST.get_features(cur.to(dev), vgg, layers={"4": "pool1", "28": "conv5_1", "36": "pool5"})
    ref_f, xw = {}, cur.clone()
    with torch.no_grad():
        for name, layer in model._modules.items():
            xw = layer(xw)
            if name in ("4", "36"):
                ref_f["pool1" if name == "4" else "pool5"] = xw.clone()
            if name == "29":                       # the in-place ReLU after conv5_1: what the reference's tap aliases
                ref_f["conv5_1"] = xw.clone()
    for k in feats:
        assert feats[k].shape == ref_f[k].shape
        assert float((feats[k].cpu() - ref_f[k]).abs().max()) <= 2e-4 * float(ref_f[k].abs().max())


def test_async_png_writer_writes_the_pixels_tensor_to_image_would(mods, tmp_path):
    """The per-step dumps go through worker threads (st3d.cli.AsyncImageWriter); the files must hold exactly what the
    reference's tensor_to_image(...).save(...) writes (clamp, x255, truncate)."""
    _, _, U, dev = mods
    from PIL import Image
    from st3d.cli import AsyncImageWriter
    g = torch.Generator().manual_seed(0)
    imgs = (torch.rand(5, 3, 33, 47, generator=g) * 1.4 - 0.2).to(dev)         # values outside [0,1] too
    w = AsyncImageWriter(workers=3, depth=1)
    for rep in range(3):                                                       # more batches than `depth`: back-pressure path
        w.submit(imgs + 0.01 * rep, [str(tmp_path / f"r{rep}_v{k}.png") for k in range(5)])
    w.flush()
    for rep in range(3):
        for k in range(5):
            got = np.asarray(Image.open(tmp_path / f"r{rep}_v{k}.png"))
            want = np.asarray(U.tensor_to_image(imgs[k] + 0.01 * rep))
            np.testing.assert_array_equal(got, want)
    with pytest.raises(Exception):                                             # a failed write surfaces at flush
        w.submit(imgs[:1], [str(tmp_path / "no_such_dir" / "x.png")])
        w.flush()


def test_content_targets_of_alternating_batches_are_kept(mods, vgg, monkeypatch):
    """second_approach.py with n_views > batch_size alternates between a few fixed content batches: the plan keeps each
    batch's conv4_2 target (device copy) instead of recomputing it; modified tensors are recomputed; results are the
    ones a forced recomputation gives, bit for bit."""
    _, L, _, dev = mods
    import st3d.vgg as V
    S, B = 64, 2
    g = torch.Generator().manual_seed(0)
    cur = torch.rand(B, 3, S, S, generator=g).to(dev)
    sty = torch.rand(1, 3, S, S, generator=g).to(dev).expand(B, -1, -1, -1)
    batches = [torch.rand(B, 3, S, S, generator=g).to(dev) for _ in range(3)]
    calls = []
    real = V.call
    monkeypatch.setattr(V, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    plan = vgg.plan(B, S)
    plan.set_content(batches[0], force=True)                      # known starting point whatever ran before
    plan.__dict__["_content_cache"] = {}
    ref = []
    for c in batches:
        plan.set_content(c, force=True)
        ref.append(L.compute_perceptual_loss(cur, c, sty, vgg).item())
    calls.clear()
    seen = [L.compute_perceptual_loss(cur, batches[i % 3], sty, vgg).item() for i in range(9)]
    assert seen == [ref[i % 3] for i in range(9)]
    n_forward = calls.count("st3d_plan_set_content")
    assert n_forward <= 4, calls                                   # at most one forward per batch (+1 before caching starts)
    assert calls.count("st3d_plan_set_content_features") >= 4
    batches[1].mul_(0.5)                                            # in-place change -> new version -> recomputed
    calls.clear()
    changed = L.compute_perceptual_loss(cur, batches[1], sty, vgg).item()
    assert "st3d_plan_set_content" in calls and changed != ref[1]


def test_content_target_is_recomputed_for_a_new_tensor_at_a_recycled_address(mods, vgg, monkeypatch, cow, golden_dir, tmp_path):
    """The target cache must not mistake a NEW content batch for the previous one when the allocator hands it the freed
    batch's address (libst3d kernels write into fresh buffers, so `_version` is 0 for both): every call with fresh
    content pays the conv4_2 forward, as the reference's loop with --content_background noise does."""
    _, L, U, dev = mods
    import st3d.vgg as V
    S, B = 64, 2
    g = torch.Generator().manual_seed(0)
    cur = torch.rand(B, 3, S, S, generator=g).to(dev)
    sty = torch.rand(1, 3, S, S, generator=g).to(dev).expand(B, -1, -1, -1)
    host = [torch.rand(B, 3, S, S, generator=g) for _ in range(4)]
    m = torch.ones(B, 1, S, S, device=dev)
    want = []
    for h in host:
        c = h.to(dev)
        vgg.plan(B, S).set_content(c, force=True)
        want.append(L.compute_perceptual_loss(cur, c, sty, vgg).item())
    assert len(set(want)) == 4
    calls = []
    real = V.call
    monkeypatch.setattr(V, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    got, ptrs = [], []
    for h in host:                                   # same alloc / free sequence every iteration, like the CLI loop
        c = U.apply_background(h.to(dev), m, "style", sty)      # output of a libst3d kernel: fresh buffer, _version 0
        ptrs.append(c.data_ptr())
        got.append(L.compute_perceptual_loss(cur, c, sty, vgg).item())
        del c
    assert got == want
    assert calls.count("st3d_plan_set_content") == 4
    # the shipped CLI with noise backgrounds: one content forward per step
    import second_approach as SA
    obj, style = _write_cow_assets(str(tmp_path), cow, golden_dir)
    calls.clear()
    SA.main(["--obj_path", obj, "--style_path", style, "--size", "64", "--n_views", "2", "--batch_size", "2", "--epochs", "3",
             "--output_path", str(tmp_path / "noise"), "--seed", "0", "--content_background", "noise", "--save_every", "0"])
    assert calls.count("st3d_plan_set_content") == 3, calls


def test_get_vgg_loads_a_local_torchvision_state_dict(mods, tmp_path, golden_dir):
    """The only route to a real stylisation offline: a torchvision-keyed state_dict (`features.<idx>.weight`, what
    models.vgg19(...).state_dict() holds, utils.py:49) saved locally and loaded through get_vgg(weights=...) /
    ST3D_VGG19_WEIGHTS / --vgg_weights.  Checked against the oracle VGG carrying the same tensors."""
    ST, L, U, dev = mods
    from oracle import perceptual_ref as P
    state = P.synthetic_vgg19_state(seed=7, bias_scale=0.1)
    tv_keys = {f"features.{k}": v for k, v in state.items()}
    tv_keys["classifier.0.weight"] = torch.zeros(4, 4)            # the full torchvision dict has classifier entries too
    path = str(tmp_path / "vgg19.pth")
    torch.save(tv_keys, path)
    model = U.get_vgg(weights=path)
    ref = P.make_vgg19_features(state)
    d = np.load(os.path.join(golden_dir, "g3_perceptual.npz"))
    x = torch.from_numpy(d["cur"])
    feats = ST.get_features(x.to(dev), model)
    rf = P.get_features_ref(x, ref)
    for k in rf:
        assert float((feats[k].cpu() - rf[k]).abs().max()) <= 2e-4 * float(rf[k].abs().max()), k
    cur = x.clone().to(dev).requires_grad_(True)
    loss = L.compute_perceptual_loss(cur, torch.from_numpy(d["con"]).to(dev), torch.from_numpy(d["sty"]).to(dev), model)
    loss.backward()
    xr = x.clone().requires_grad_(True)
    lr = P.perceptual_loss_ref(xr, torch.from_numpy(d["con"]), torch.from_numpy(d["sty"]), ref)
    lr.backward()
    assert abs(loss.item() - float(lr.detach())) <= 2e-5 * float(lr.detach())
    assert float((cur.grad.cpu() - xr.grad).norm() / xr.grad.norm()) <= 1e-4
    assert abs(loss.item() - float(d["loss"])) > 1e-3 * float(d["loss"])          # really other weights than seed 0
    # plain '<idx>.weight' keys (a saved `.features` module) and the environment variable
    torch.save(state, str(tmp_path / "features.pth"))
    os.environ["ST3D_VGG19_WEIGHTS"] = str(tmp_path / "features.pth")
    try:
        m2 = U.get_vgg()
    finally:
        del os.environ["ST3D_VGG19_WEIGHTS"]
    f2 = ST.get_features(x.to(dev), m2)
    assert torch.equal(f2["conv5_1"], feats["conv5_1"])
    with pytest.raises(KeyError):
        torch.save({k: v for k, v in state.items() if not k.startswith("28.")}, str(tmp_path / "bad.pth"))
        U.get_vgg(weights=str(tmp_path / "bad.pth"))


def test_reference_style_transfer_body_runs_on_differentiable_get_features(mods, vgg, golden_dir):
    """The reference's own loop body (style_transfer.py:44-83: get_features(optimized_imgs) WITH grad, torch.mean /
    gram_matrix losses written by the caller, total_loss.backward(), torch.optim.Adam) runs unchanged on the st3d VGG:
    get_features is an autograd node whose backward is the plan's input-gradient chain.  Checked against the golden G4
    trajectory produced by the reference's code, and the tap gradients against the fused loss plan."""
    ST, L, _, dev = mods
    d = np.load(os.path.join(golden_dir, "g4_style_transfer.npz"))
    init, con, sty = (torch.from_numpy(d[k]).to(dev) for k in ("init", "con", "sty"))
    model = vgg
    # ---- the reference's body, typed against the drop-in names (get_features / gram_matrix / optim)
    content_features = ST.get_features(con, model)['conv4_2']
    style_features = ST.get_features(sty, model)
    style_grams = {layer: ST.gram_matrix(style_features[layer]) for layer in style_features}
    style_grams.pop('conv4_2')
    optimized_imgs = init.clone().detach().requires_grad_(True)
    optimizer = ST.optim.Adam([optimized_imgs], lr=float(d["lr"]))
    for step in range(int(d["steps"])):
        feats = ST.get_features(optimized_imgs, model)
        assert feats['conv1_1'].requires_grad
        content_loss = torch.mean((feats['conv4_2'] - content_features) ** 2)
        style_loss = 0
        for layer in style_grams:
            f = feats[layer]
            style_loss += torch.mean((ST.gram_matrix(f) - style_grams[layer]) ** 2) / (f.shape[1] ** 2 * f.shape[2] ** 2)
        total_loss = 1 * content_loss + 1e6 * style_loss
        optimizer.zero_grad()
        total_loss.backward()
        if step == 0:      # same gradient as the fused plan delivers for the same loss
            x = init.clone().requires_grad_(True)
            fused = L.compute_perceptual_loss(x, con, sty, model)
            fused.backward()
            assert abs(fused.item() - total_loss.item()) <= 2e-5 * abs(fused.item())
            assert float((optimized_imgs.grad - x.grad).norm() / x.grad.norm()) <= 2e-5
        optimizer.step()
    err = (optimized_imgs.detach().cpu() - torch.from_numpy(d["result"])).abs()
    assert float(err.max()) <= 1e-4, float(err.max())


def test_differentiable_get_features_custom_taps_match_torch_autograd(mods, vgg):
    """Gradients through taps the loss plan never uses -- a conv that feeds a pool ('2' = conv1_2 -> pool1), the pool
    itself ('4'), a deep tap ('25') -- at a size off the Winograd path (S = 40: odd 5x5 maps at the bottom), and a backward
    that happens after ANOTHER forward has reused the plan's buffers: all against torch autograd on the oracle VGG."""
    ST, _, _, dev = mods
    from oracle import perceptual_ref as P
    torch.manual_seed(0)
    model = P.make_vgg19_features(seed=0).double()
    layers = {"2": "conv1_2", "4": "pool1", "10": "conv3_1", "25": "conv4_4"}
    for S, B in ((64, 2), (40, 1)):
        x = torch.rand(B, 3, S, S)
        w = {}
        xg = x.clone().to(dev).requires_grad_(True)
        feats = ST.get_features(xg, vgg, layers=layers)
        ST.get_features(torch.rand(B, 3, S, S).to(dev), vgg)          # clobbers the plan's activations before the backward
        loss = 0
        for k, f in feats.items():
            w[k] = torch.randn(f.shape, generator=torch.Generator().manual_seed(len(k)))
            loss = loss + (f * w[k].to(dev)).sum()
        loss.backward()
        # reference: fp64 autograd through the SAME ReLU gates and max-pool selections the kernels saw.  An activation
        # within rounding of 0, or two window elements within rounding of each other, resolve differently in fp64, and
        # with random-sign tap gradients one such unit moves its whole receptive field by O(1) (observed: one pool2 window)
        with torch.no_grad():
            acts = {n: t.cpu() for n, t in ST.get_features(x.to(dev), vgg, layers={str(i): str(i) for i in range(26)}).items()}
        xr = x.clone().double().requires_grad_(True)
        cur, ref_loss = xr, 0
        for name, layer in model._modules.items():
            if int(name) > 26:
                break
            if isinstance(layer, torch.nn.ReLU):
                cur = cur * (acts[str(int(name) - 1)] > 0).double()       # a conv tap IS its post-ReLU output
                if str(int(name) - 1) in layers:                      # conv taps are post-ReLU (in-place ReLU)
                    ref_loss = ref_loss + (cur * w[layers[str(int(name) - 1)]].double()).sum()
            elif isinstance(layer, torch.nn.MaxPool2d):
                pre, pooled = acts[str(int(name) - 1)], acts[name]
                Hp, Wp = pooled.shape[2:]
                taken, out = torch.zeros_like(pooled, dtype=torch.bool), 0
                for k in range(4):                                    # first maximum in row-major window order
                    dy, dx = k >> 1, k & 1
                    sel = (pre[:, :, dy:2 * Hp:2, dx:2 * Wp:2] == pooled) & ~taken
                    taken |= sel
                    out = out + cur[:, :, dy:2 * Hp:2, dx:2 * Wp:2] * sel.double()
                cur = out
                if name in layers:
                    ref_loss = ref_loss + (cur * w[layers[name]].double()).sum()
            else:
                cur = layer(cur)
        ref_loss.backward()
        assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-4 * abs(float(ref_loss.detach()))
        rel = float((xg.grad.cpu().double() - xr.grad).norm() / xr.grad.norm())
        assert rel <= 1e-4, (S, rel)


def test_graph_replay_of_the_loss_step_is_bitwise_identical(mods, vgg):
    """PerceptualPlan.use_graph(): after one ordinary call the static launch sequence of st3d_plan_loss is captured into a
    HIP graph and replayed.  Same kernels in the same order: loss and gradient must be bit-identical to the plain path --
    across steps, after the content target changed (the graph reads the plan's buffers, not the caller's), and after a
    change of batch size / weights (re-capture)."""
    _, L, _, dev = mods
    S, B = 64, 2
    g = torch.Generator().manual_seed(0)
    imgs = [torch.rand(B, 3, S, S, generator=g).to(dev) for _ in range(4)]
    con = [torch.rand(B, 3, S, S, generator=g).to(dev) for _ in range(2)]
    sty = torch.rand(1, 3, S, S, generator=g).to(dev)
    plan = vgg.plan(B, S)

    def run(graph):
        plan.use_graph(graph)
        out = []
        try:
            for k, (sw, n) in enumerate(((1e6, B), (1e6, B), (1e6, B), (2e5, B), (2e5, 1), (2e5, 1))):
                plan.set_content(con[k % 2][:n], force=True)
                plan.set_style(sty, n, force=True)
                loss, grad = plan.loss(imgs[k % 4][:n], sw, 1.0)
                out.append((loss.clone(), grad.clone()))
        finally:
            plan.use_graph(False)
        return out
    plain, replay = run(False), run(True)
    for (l0, g0), (l1, g1) in zip(plain, replay):
        assert torch.equal(l0, l1) and torch.equal(g0, g1)
    # through the public API too
    x = imgs[0].clone().requires_grad_(True)
    plan.use_graph(True)
    try:
        vals = []
        for _ in range(3):
            x.grad = None
            loss = L.compute_perceptual_loss(x, con[0], sty.expand(B, -1, -1, -1), vgg)
            loss.backward()
            vals.append((loss.item(), x.grad.clone()))
    finally:
        plan.use_graph(False)
    assert vals[0][0] == vals[1][0] == vals[2][0] and torch.equal(vals[0][1], vals[2][1])
    # a style target with a different batch stride (one shared image -> one image per view) after the capture: the
    # captured launch indexed the style Grams with the old stride -- set_style now drops the graph
    sty_b = torch.rand(B, 3, S, S, generator=g).to(dev)

    def seq(graph):
        plan.use_graph(graph)
        out = []
        try:
            plan.set_content(con[0], force=True)
            for style in (sty, sty, sty, sty_b, sty_b, sty_b, sty, sty):
                plan.set_style(style, B, force=True)
                loss, grad = plan.loss(imgs[1], 1e6, 1.0)
                out.append((loss.clone(), grad.clone()))
        finally:
            plan.use_graph(False)
        return out
    for (l0, g0), (l1, g1) in zip(seq(False), seq(True)):
        assert torch.equal(l0, l1) and torch.equal(g0, g1)


def test_plan_cache_is_bounded_and_a_foreign_device_pointer_is_refused(mods, monkeypatch):
    """Vgg19Features.plan keeps at most MAX_PLANS workspaces (least recently used first out); an evicted shape is rebuilt
    with the same results.  st3d._lib.dptr refuses a tensor that does not live on torch's current device (the kernels
    launch there)."""
    _, L, U, dev = mods
    from st3d import _lib
    v = U.get_vgg(seed=0)
    monkeypatch.setattr(type(v), "MAX_PLANS", 2)
    g = torch.Generator().manual_seed(0)
    x = {S: torch.rand(1, 3, S, S, generator=g).to(dev) for S in (32, 48, 64)}
    first = {S: L.compute_perceptual_loss(x[S], x[S].flip(-1), x[S].flip(-2), v).item() for S in (32, 48, 64)}
    assert list(v._plans) == [(1, 48), (1, 64)]
    v.plan(1, 48)                   # touching a shape makes it the most recent one
    again = L.compute_perceptual_loss(x[32], x[32].flip(-1), x[32].flip(-2), v).item()      # rebuilt, (1, 64) goes
    assert list(v._plans) == [(1, 48), (1, 32)] and again == first[32]
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 1)
    with pytest.raises(_lib.St3dError, match="current device"):
        _lib.dptr(x[32])
