"""GPU parity tests of every libst3d kernel against the CPU oracle (oracle/) or a plain
PyTorch fp32 reference of the same op, all through the C ABI.  Tolerances are stated per test.

fp32 MFMA sums in a different order than MKL/oneDNN; for a K-term dot product of O(1) terms the
two fp32 results differ by ~1e-7*sqrt(K)*|terms|, so contractions are checked at rtol 2e-4 of
the output scale (an fp64 reference is used to judge both)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from st3d import _lib, ops as o
    if os.environ.get("ST3D_DIAG_LIB"):          # lab: the same tests against an A/B build of the library (tools/w43_ab_build.sh)
        _lib.SO_PATH = os.path.abspath(os.environ["ST3D_DIAG_LIB"])
    return o


def _cams(n, seed=0):
    from oracle import render_ref as rr
    g = torch.Generator().manual_seed(seed)
    elev, azim = rr.random_camera_angles(n, lambda k: torch.rand(k, generator=g).numpy())
    return rr.look_at_view_transform(2.10, elev, azim, at=(0, 0.10, 0.25))


def _scale_close(got, ref, rtol, name=""):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, f"{name}: max err {err:.3e} > {rtol:.1e} * scale {scale:.3e}"


# ---------------------------------------------------------------------------- render
@pytest.mark.parametrize("S", [64, 200])
def test_raster_matches_oracle(dev, ops, cow, S):
    """pix_to_face bit-exact; zbuf/bary/dists bit-exact (both sides built without FMA contraction)."""
    from oracle import render_ref as rr
    R, T = _cams(3, seed=S)
    verts = torch.from_numpy(cow["verts"]).to(dev)
    faces = torch.from_numpy(cow["faces"]).to(dev)
    ndc = ops.project_verts(verts, torch.from_numpy(R).to(dev), torch.from_numpy(T).to(dev))
    p2f, zbuf, bary, dists = ops.raster_fwd(ndc, faces, S)
    torch.cuda.synchronize()
    for b in range(3):
        ndc_ref = rr.project_verts(cow["verts"], R[b], T[b])
        np.testing.assert_allclose(ndc[b].cpu().numpy(), ndc_ref, rtol=0, atol=0)
        rp, rz, rb, rd = rr.rasterize(ndc_ref, cow["faces"], S, 0.0, nthreads=8)
        assert (rp >= 0).sum() > 0.05 * S * S
        np.testing.assert_array_equal(p2f[b].cpu().numpy(), rp)
        np.testing.assert_array_equal(zbuf[b].cpu().numpy(), rz)
        np.testing.assert_array_equal(bary[b].cpu().numpy(), rb)
        np.testing.assert_array_equal(dists[b].cpu().numpy(), rd)


def test_shade_fwd_bwd_match_oracle(dev, ops, cow):
    """RGB within 2e-6 abs (expf differs by ulps between glibc and the device); texture gradient
    within 1e-5 of the fp64-accumulated oracle relative to its max (float atomics, any order)."""
    from oracle import render_ref as rr
    S, T = 96, 64
    R, Tt = _cams(2, seed=5)
    rng = np.random.default_rng(0)
    tex = rng.random((T, T, 3), dtype=np.float32)
    imgs_ref, masks_ref, frags = rr.render_views(cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"], tex, R,
                                                 Tt, S, nthreads=8)
    verts = torch.from_numpy(cow["verts"]).to(dev)
    faces = torch.from_numpy(cow["faces"]).to(dev)
    uvs = torch.from_numpy(cow["verts_uvs"]).to(dev)
    fuv = torch.from_numpy(cow["faces_uvs"]).to(dev)
    texd = torch.from_numpy(tex).to(dev)
    ndc = ops.project_verts(verts, torch.from_numpy(R).to(dev), torch.from_numpy(Tt).to(dev))
    frag = ops.raster_fwd(ndc, faces, S)
    rgb, mask = ops.shade_fwd(frag, uvs, fuv, texd)
    np.testing.assert_allclose(rgb.cpu().numpy(), imgs_ref, rtol=0, atol=2e-6)
    np.testing.assert_array_equal(mask.cpu().numpy(), masks_ref)
    g = rng.standard_normal((2, 3, S, S)).astype(np.float32)
    gt_ref = np.zeros((T, T, 3), np.float64)
    guv_ref = []
    for b in range(2):
        _, guv = rr.shade_bwd(g[b], frags[b], cow["verts_uvs"], cow["faces_uvs"], tex, gt_ref, want_uv=True)
        guv_ref.append(guv)
    gt, guv = ops.shade_bwd(torch.from_numpy(g).to(dev), frag, uvs, fuv, texd, want_uv=True)
    _scale_close(gt, torch.from_numpy(gt_ref), 1e-5, "grad_texture")
    _scale_close(guv, torch.from_numpy(np.stack(guv_ref)), 1e-4, "grad_uv")


def test_apply_background(dev, ops):
    torch.manual_seed(0)
    img = torch.rand(2, 3, 40, 40, device=dev)
    m = (torch.rand(2, 1, 40, 40, device=dev) > 0.5).float()
    bg = torch.rand(2, 3, 40, 40, device=dev)
    torch.testing.assert_close(ops.apply_background(img, m, bg), img * m + bg * (1 - m), rtol=0, atol=0)
    torch.testing.assert_close(ops.apply_background(img, m, bg[:1]), img * m + bg[:1] * (1 - m), rtol=0, atol=0)


# ---------------------------------------------------------------------------- conv / pool
CONV_CASES = [(2, 3, 64, 64, 64), (1, 64, 64, 48, 40), (2, 64, 128, 32, 32), (1, 128, 256, 24, 56),
              (1, 256, 256, 32, 32), (2, 512, 512, 16, 16), (1, 512, 512, 4, 4), (1, 3, 64, 33, 70)]


@pytest.mark.parametrize("S,B", [(96, 2), (64, 1), (160, 3), (160, 1)])      # 160: 3 x 3 bins per view -- odd counts (list alignment)
def test_raster_does_not_depend_on_stale_workspace_contents(dev, ops, cow, S, B):
    """The rasteriser's workspace (face records, packed tile ranges, coarse bin lists) is a fresh torch.empty per call: it
    must read nothing it has not written.  The caching allocator is primed with blocks full of 0xFF.. / small positive
    integers (plausible face indices and counts) before each call; the fragments must equal the C oracle's bit for bit.
    (Regression: the lanes past the last face carried the empty-range sentinel, which passes a RANGE overlap test with
    the first bin -- phantom face indices in its list, harmless on zeroed memory, a fault on recycled memory.)"""
    from oracle import render_ref as rr
    R, T = _cams(B, seed=5)
    faces = torch.from_numpy(cow["faces"].astype(np.int32)).to(dev)
    ndc = ops.project_verts(torch.from_numpy(cow["verts"]).to(dev), torch.from_numpy(R).to(dev), torch.from_numpy(T).to(dev))
    want = [rr.rasterize(rr.project_verts(cow["verts"], R[b], T[b]), cow["faces"], S, nthreads=8) for b in range(B)]
    for poison in (-1, 7, 0x3fffffff):
        junk = [torch.full((n,), poison, dtype=torch.int32, device=dev) for n in (1 << 22, 1 << 20, 1 << 18, 1 << 16)]
        del junk
        p2f, zbuf, bary, dists = ops.raster_fwd(ndc, faces, S)
        for b in range(B):
            np.testing.assert_array_equal(p2f[b].cpu().numpy(), want[b][0])
            np.testing.assert_array_equal(zbuf[b].cpu().numpy(), want[b][1])


@pytest.mark.parametrize("N,Cin,Cout,H,W", CONV_CASES)
def test_conv3x3_fwd(dev, ops, N, Cin, Cout, H, W):
    torch.manual_seed(Cin * 7 + Cout)
    x = torch.randn(N, Cin, H, W)
    w = torch.randn(Cout, Cin, 3, 3) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.randn(Cout) * 0.1
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    wf, _ = ops.conv3x3_pack(w.to(dev))
    y = ops.conv3x3_fwd(x.to(dev), wf, b.to(dev), Cout, relu=True)
    _scale_close(y, ref, 2e-5, "conv fwd")
    y2 = ops.conv3x3_fwd(x.to(dev), wf, b.to(dev), Cout, relu=False)
    _scale_close(y2, F.conv2d(x.double(), w.double(), b.double(), padding=1), 2e-5, "conv fwd (no relu)")


@pytest.mark.parametrize("N,Cin,Cout,H,W", CONV_CASES)
def test_conv3x3_dgrad(dev, ops, N, Cin, Cout, H, W):
    """gx = d/dx of sum(gy * relu(conv(x))) -- the ReLU gate is fused into the kernel's load."""
    torch.manual_seed(Cin * 3 + Cout)
    x = torch.randn(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3) * (2.0 / (Cin * 9)) ** 0.5).double()
    b = (torch.randn(Cout) * 0.1).double()
    y = F.relu(F.conv2d(x, w, b, padding=1))
    gy = torch.randn_like(y)
    y.backward(gy)
    _, wd = ops.conv3x3_pack(w.float().to(dev))
    gx = ops.conv3x3_dgrad(gy.float().to(dev), y.detach().float().to(dev), wd, Cin)
    _scale_close(gx, x.grad, 2e-5, "conv dgrad")


@pytest.mark.parametrize("N,H,W,has_g,has_d", [(2, 64, 64, True, True), (1, 40, 52, True, True), (2, 17, 22, True, True),
                                                (1, 128, 96, True, False), (1, 32, 32, False, True), (1, 3, 6, True, True)])
def test_conv1_bwd_fused_matches_autograd_and_the_unfused_kernels(dev, ops, N, H, W, has_g, has_d):
    """st3d_conv1_bwd: gx = d/dx [ sum(gy * relu1_1) + coef/2 * <D, Gram(relu1_1)> ]-style gradient, i.e.
    conv1_1^T(gate(gy + coef * D F)) -- against fp64 autograd of the same expression and against the two launches it
    replaces (st3d_gram_bwd(accumulate) + st3d_conv3x3_dgrad).  Sizes cover tails (HW not a multiple of 64 / 256)."""
    torch.manual_seed(H * 7 + W)
    x = torch.randn(N, 3, H, W, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(64, 3, 3, 3) * (2.0 / 27) ** 0.5).double()
    b = (torch.randn(64) * 0.1).double()
    Fm = F.relu(F.conv2d(x, w, b, padding=1))
    gy = torch.randn_like(Fm) if has_g else None
    D = torch.randn(N, 64, 64, dtype=torch.float64) if has_d else None
    if has_d:
        D = 0.5 * (D + D.transpose(1, 2))
    coef = 0.37
    # objective whose gradient w.r.t. F is gy + coef * D F:  <gy, F> + coef/2 * sum_n tr(F^T D F)
    Ff = Fm.reshape(N, 64, H * W)
    obj = 0.0
    if has_g:
        obj = obj + (gy * Fm).sum()
    if has_d:
        obj = obj + 0.5 * coef * torch.einsum("ncp,ncd,ndp->", Ff, D, Ff)
    obj.backward()
    _, wd = ops.conv3x3_pack(w.float().to(dev))
    act = Fm.detach().float().to(dev)
    gyd = gy.float().to(dev) if has_g else None
    Dd = D.float().to(dev).contiguous() if has_d else None
    gx = ops.conv1_bwd(gyd, act, Dd, coef, wd)
    _scale_close(gx, x.grad, 3e-5, "fused conv1_1 backward")
    # the launches it replaces
    if has_g and has_d:
        gtot = ops.gram_bwd(Dd, act, coef, out=gyd.clone())
    elif has_d:
        gtot = ops.gram_bwd(Dd, act, coef)
    else:
        gtot = gyd
    gx2 = ops.conv3x3_dgrad(gtot, act, wd, 3)
    _scale_close(gx, gx2, 3e-5, "fused vs unfused")
    # fixed summation order: bitwise reproducible
    assert torch.equal(gx, ops.conv1_bwd(gyd, act, Dd, coef, wd))


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 64, 64, 32, 64), (1, 128, 128, 16, 16), (1, 256, 256, 8, 40)])
def test_conv3x3_dgrad_unpool(dev, ops, N, Cin, Cout, H, W):
    """gradient through conv -> ReLU -> MaxPool2d(2,2) in one kernel (unpool + gate fused)."""
    torch.manual_seed(H + W)
    x = torch.randn(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3) * (2.0 / (Cin * 9)) ** 0.5).double()
    b = (torch.randn(Cout) * 0.1).double()
    y = F.relu(F.conv2d(x, w, b, padding=1))
    p = F.max_pool2d(y, 2, 2)
    gp = torch.randn_like(p)
    p.backward(gp)
    wf, wd = ops.conv3x3_pack(w.float().to(dev))
    yd = ops.conv3x3_fwd(x.detach().float().to(dev), wf, b.float().to(dev), Cout, relu=True)
    pd, idx = ops.maxpool2x2(yd)
    _scale_close(pd, p, 2e-5, "pool")
    gx = ops.conv3x3_dgrad_unpool(gp.float().to(dev), idx, pd, wd, Cin)
    _scale_close(gx, x.grad, 5e-5, "dgrad_unpool")


@pytest.mark.parametrize("H,W", [(12, 20), (11, 21), (7, 6), (2, 3)])
def test_maxpool_matches_torch(dev, ops, H, W):
    """incl. odd sizes: MaxPool2d floors (last row / column dropped)"""
    torch.manual_seed(0)
    y = torch.randn(2, 5, H, W)
    y[0, 0, 0:2, 0:2] = 0.0                                   # tie: the first element wins
    ref, ridx = F.max_pool2d(y, 2, 2, return_indices=True)
    p, idx = ops.maxpool2x2(y.to(dev))
    torch.testing.assert_close(p.cpu(), ref, rtol=0, atol=0)
    rr_, cc_ = ridx // W, ridx % W
    local = (rr_ % 2) * 2 + (cc_ % 2)
    np.testing.assert_array_equal(idx.cpu().numpy(), local.numpy().astype(np.uint8))


# ---------------------------------------------------------------------------- gram / losses / adam
@pytest.mark.parametrize("B,C,H,W", [(2, 64, 64, 64), (1, 128, 40, 40), (2, 256, 16, 16), (1, 512, 8, 8), (2, 512, 4, 4),
                                     (1, 64, 3, 3), (2, 128, 32, 64), (1, 64, 256, 512), (1, 128, 256, 256)])
def test_gram_fwd_bwd(dev, ops, B, C, H, W):
    torch.manual_seed(C + H)
    f = torch.rand(B, C, H, W)
    fd = f.double().reshape(B, C, H * W)
    ref = torch.bmm(fd, fd.transpose(1, 2))
    g = ops.gram_fwd(f.to(dev))
    _scale_close(g, ref, 2e-5, "gram")
    assert torch.equal(g, g.transpose(1, 2)), "mirrored tiles must make the Gram exactly symmetric"
    D = torch.randn(B, C, C)          # not symmetrised: the kernels must not rely on D == D^T
    refb = 0.37 * torch.bmm(D.double(), fd).reshape(B, C, H, W)
    out = ops.gram_bwd(D.to(dev), f.to(dev), 0.37)
    _scale_close(out, refb, 2e-5, "gram bwd")
    base = torch.randn(B, C, H, W)
    out2 = ops.gram_bwd(D.to(dev), f.to(dev), 0.37, out=base.clone().to(dev))
    _scale_close(out2, refb + base.double(), 2e-5, "gram bwd accumulate")


def test_gram_golden_g1(dev, ops, golden_dir):
    """Golden vector produced by the reference's own gram_matrix (style_transfer.py:31-35)."""
    d = np.load(os.path.join(golden_dir, "g1_gram.npz"))
    g = ops.gram_fwd(torch.from_numpy(d["x"]).to(dev))
    np.testing.assert_allclose(g.cpu().numpy(), d["gram"], rtol=1e-5, atol=1e-5)


def test_sqdiff_and_masked_mse(dev, ops):
    torch.manual_seed(0)
    a, b = torch.randn(3, 1000), torch.randn(1, 1000)
    out, D = ops.sqdiff_sum(a.to(dev), b.to(dev), scale=0.5, want_diff=True)
    ref = 0.5 * ((a.double() - b.double()) ** 2).sum()
    assert abs(out.item() - ref.item()) <= 1e-5 * abs(ref.item())
    torch.testing.assert_close(D.cpu(), a - b, rtol=0, atol=0)
    r = torch.rand(2, 3, 24, 24, dtype=torch.float64, requires_grad=True)
    t = torch.rand(2, 3, 24, 24, dtype=torch.float64)
    m = (torch.rand(2, 1, 24, 24) > 0.4).double()
    loss = F.mse_loss(r * m, t * m)
    loss.backward()
    out, g = ops.masked_mse(r.detach().float().to(dev), t.float().to(dev), m.float().to(dev))
    assert abs(out.item() - loss.item()) <= 1e-5 * loss.item()
    _scale_close(g, r.grad, 1e-5, "masked mse grad")


def test_adam_matches_torch_optim(dev, ops):
    """Five steps of the fused Adam vs torch.optim.Adam (the optimiser the reference uses)."""
    torch.manual_seed(0)
    p0 = torch.randn(5000)
    grads = [torch.randn(5000) * (0.1 + i) for i in range(5)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=0.01)
    p = p0.clone().to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for i, g in enumerate(grads):
        pr.grad = g.clone()
        opt.step()
        ops.adam_step(p, g.to(dev), m, v, i + 1, 0.01)
    torch.testing.assert_close(p.cpu(), pr.detach(), rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------- fused plan vs the reference's golden vectors
def test_plan_matches_reference_golden_g3(dev, golden_dir):
    """compute_perceptual_loss value + gradient produced by the reference's losses.py on the
    seeded VGG (tests/golden/make_golden.py).  Loss within 2e-5 relative, gradient within 1e-4
    relative L2 (fp32, different summation order over K up to 4608 and Gram K = 4096)."""
    from st3d import vgg as V
    d = np.load(os.path.join(golden_dir, "g3_perceptual.npz"))
    model = V.Vgg19Features(V.synthetic_state(0), device=dev)
    plan = model.plan(2, 64)
    cur, con, sty = (torch.from_numpy(d[k]).to(dev) for k in ("cur", "con", "sty"))
    plan.set_content(con)
    plan.set_style(sty, 2)
    loss, grad = plan.loss(cur, 1e6, 1.0)
    torch.cuda.synchronize()
    total = loss[0].item()
    assert abs(total - float(d["loss"])) <= 2e-5 * float(d["loss"]), (total, float(d["loss"]))
    gref = torch.from_numpy(d["grad"])
    rel = (grad.cpu() - gref).norm().item() / gref.norm().item()
    assert rel <= 1e-4, rel
    # features / Grams of the current images against the reference's get_features / gram_matrix
    plan.forward(cur, upto=28)
    f5 = plan.activation(28).cpu().numpy()
    np.testing.assert_allclose(f5, d["feat_conv5_1"], rtol=2e-4, atol=2e-4 * np.abs(d["feat_conv5_1"]).max())
    f1 = plan.activation(0).cpu()
    assert float(f1.min()) == 0.0                                   # taps are post-ReLU
    np.testing.assert_allclose(f1[0, 0].numpy(), d["feat_conv1_1_img0_ch0"], rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------- vertex path (K14) + mesh regularisers (K15)
def test_vertex_path_backward_matches_oracle(dev, ops, cow):
    """shade d/d(bary) -> raster backward -> projection backward vs the C oracle (fp64 accumulation):
    vertex gradient within 2e-4 relative L2 (float atomics, fp32 intermediates)."""
    from oracle import render_ref as rr
    S, T, B = 96, 32, 2
    R, Tt = _cams(B, seed=11)
    rng = np.random.default_rng(3)
    tex = rng.random((T, T, 3), dtype=np.float32)
    imgs, _, frags = rr.render_views(cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"], tex, R, Tt, S, 8)
    g = rng.standard_normal(imgs.shape).astype(np.float32)
    gtex_ref, gverts_ref = rr.render_bwd_views(g, frags, cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"], tex, R, Tt)
    verts = torch.from_numpy(cow["verts"]).to(dev)
    faces = torch.from_numpy(cow["faces"]).to(dev)
    uvs = torch.from_numpy(cow["verts_uvs"]).to(dev)
    fuv = torch.from_numpy(cow["faces_uvs"]).to(dev)
    texd = torch.from_numpy(tex).to(dev)
    Rd, Td = torch.from_numpy(R).to(dev), torch.from_numpy(Tt).to(dev)
    ndc = ops.project_verts(verts, Rd, Td)
    frag = ops.raster_fwd(ndc, faces, S)
    gtex, gbary = ops.shade_bwd(torch.from_numpy(g).to(dev), frag, uvs, fuv, texd, want_bary=True)
    gb_ref = np.stack([rr.uv_to_bary_grad(rr.shade_bwd(g[b], frags[b], cow["verts_uvs"], cow["faces_uvs"], tex, want_uv=True)[1],
                                          frags[b][0], cow["verts_uvs"], cow["faces_uvs"]) for b in range(B)])
    _scale_close(gbary, torch.from_numpy(gb_ref), 1e-4, "grad_bary")
    gndc = ops.raster_bwd(gbary, frag[0], ndc, faces)
    gverts = ops.project_verts_bwd(verts, Rd, Td, gndc)
    rel = np.linalg.norm(gverts.cpu().numpy() - gverts_ref) / np.linalg.norm(gverts_ref)
    assert rel <= 2e-4, rel
    _scale_close(gtex, torch.from_numpy(gtex_ref), 1e-5, "grad_texture")
    # vertices only (no texture gradient requested)
    only = ops.shade_bwd(torch.from_numpy(g).to(dev), frag, uvs, fuv, texd, want_bary=True, want_texture=False)
    assert only[0] is None and torch.equal(only[1], gbary)


def test_gram_forward_of_all_layers_in_one_launch_is_bitwise_the_per_layer_result(dev, ops, monkeypatch):
    """st3d_gram_fwd_multi (round 3: one launch pair for the five style layers of a step instead of ten launches): same
    kernel bodies and the same fixed reduce tree as st3d_gram_fwd, so with the same split counts (ST3D_GRAM_MULTI_SCALE=1,
    set by the fixture below) the results are bit-identical per layer; VGG shapes at 64^2 with 3 images (a batch the split
    counts are not tuned for), plus two shapes outside the whole-tile fast path (C = 96; HW not a multiple of 32), which
    fall back to the per-layer launches inside the same call.  The default (fewer, longer splits: half the slab traffic)
    is checked against fp64 and for run-to-run bitwise reproducibility in the test after this one."""
    monkeypatch.setenv("ST3D_GRAM_MULTI_SCALE", "1")
    g = torch.Generator().manual_seed(5)
    shapes = [(3, 64, 64, 64), (3, 128, 32, 32), (3, 256, 16, 16), (3, 512, 8, 8), (3, 512, 4, 4), (2, 96, 16, 16), (1, 64, 5, 7)]
    feats = [torch.randn(sh, generator=g).clamp_min(0).to(dev) for sh in shapes]
    multi = ops.gram_fwd_multi(feats)
    for f, gm in zip(feats, multi):
        single = ops.gram_fwd(f)
        assert torch.equal(gm, single), f.shape
        ref = torch.bmm(f.double().flatten(2), f.double().flatten(2).transpose(1, 2))
        assert float((gm.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
        assert torch.equal(gm, gm.transpose(1, 2))
    # the five layers alone (what the loss plan launches), full VGG depth at 128^2, batch 2; and a lone relu1_1-like item
    feats = [torch.randn(sh, generator=g).clamp_min(0).to(dev) for sh in
             [(2, 64, 128, 128), (2, 128, 64, 64), (2, 256, 32, 32), (2, 512, 16, 16), (2, 512, 8, 8)]]
    for fs in (feats, feats[:1], feats[1:]):
        for f, gm in zip(fs, ops.gram_fwd_multi(fs)):
            assert torch.equal(gm, ops.gram_fwd(f)), f.shape


def test_gram_forward_multi_default_splits_are_reproducible_and_exact_to_fp32(dev, ops):
    g = torch.Generator().manual_seed(6)
    feats = [torch.randn(sh, generator=g).clamp_min(0).to(dev) for sh in
             [(4, 64, 256, 256), (4, 128, 128, 128), (4, 256, 64, 64), (4, 512, 32, 32), (4, 512, 16, 16)]]
    a = ops.gram_fwd_multi(feats)
    b = ops.gram_fwd_multi(feats)
    for f, ga, gb in zip(feats, a, b):
        assert torch.equal(ga, gb) and torch.equal(ga, ga.transpose(1, 2))
        ref = torch.bmm(f.double().flatten(2), f.double().flatten(2).transpose(1, 2))
        assert float((ga.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_mesh_regularisers_match_oracle(dev, ops, cow):
    """values within 1e-5 relative and gradient within 1e-4 relative L2 of the torch-fp64 restatement
    (autograd) on the cow mesh with perturbed vertices; topology counts checked."""
    from oracle import mesh_ref as M
    from st3d import mesh_losses as ML
    torch.manual_seed(0)
    target = torch.from_numpy(cow["verts"])
    verts = target + 0.01 * torch.randn_like(target)
    faces = torch.from_numpy(cow["faces"].astype(np.int64))
    topo = ML.build_topology(faces.to(dev), verts.shape[0])
    edges_ref, _ = M.unique_edges(faces)
    pairs_ref = M.face_pairs(faces)
    assert topo["edges"].cpu().tolist() == edges_ref.tolist()
    assert topo["pairs"].cpu().tolist() == pairs_ref.tolist()
    assert topo["edges"].shape[0] == 8784 and topo["pairs"].shape[0] == 8784       # closed manifold: E = 3F/2, one pair per edge
    w = [0.7, 1.3, 0.9, 1.1]
    out, grad = ops.mesh_reg(verts.to(dev), target.to(dev), topo, w)
    vd = verts.double().requires_grad_(True)
    terms = [M.verts_mse_ref(vd, target.double()), M.mesh_edge_loss_ref(vd, faces), M.mesh_laplacian_smoothing_ref(vd, faces),
             M.mesh_normal_consistency_ref(vd, faces)]
    total = sum(wi * t for wi, t in zip(w, terms))
    total.backward()
    for k, t in enumerate(terms):
        assert abs(out[1 + k].item() - t.item()) <= 1e-5 * abs(t.item()) + 1e-9, (k, out[1 + k].item(), t.item())
    assert abs(out[0].item() - total.item()) <= 1e-5 * abs(total.item())
    rel = (grad.cpu().double() - vd.grad).norm() / vd.grad.norm()
    assert float(rel) <= 1e-4, float(rel)
    # round 3: every gradient is a fixed-order gather over static CSR lists (no float atomics): bitwise reproducible
    for _ in range(3):
        out2, grad2 = ops.mesh_reg(verts.to(dev), target.to(dev), topo, w)
        assert torch.equal(grad2, grad) and torch.equal(out2, out)
    # the per-vertex inverse of `pairs`: vertex k is named exactly by the entries listed for it, ascending
    po, pr, flat = topo["pair_off"].cpu().numpy(), topo["pair_ref"].cpu().numpy(), topo["pairs"].cpu().numpy().reshape(-1)
    assert po[0] == 0 and po[-1] == flat.size == pr.size
    for k in (0, 1, 17, verts.shape[0] - 1):
        mine = pr[po[k]:po[k + 1]]
        assert (flat[mine] == k).all() and (np.diff(mine) > 0).all() and mine.size == (flat == k).sum()


def test_fixed_point_scatters_propagate_non_finite_gradients(dev, ops, cow):
    """A NaN / Inf in the upstream gradient must come out of the render backward as NaN in BOTH scatter modes (ADVICE r2:
    the fixed-point path used to launder it into a finite number via the integer conversion, so a diverged run kept
    stepping silently; the float-atomic path and the reference propagate it)."""
    S, T = 64, 32
    R, Tt = _cams(1, seed=3)
    verts = torch.from_numpy(cow["verts"]).to(dev)
    faces = torch.from_numpy(cow["faces"]).to(dev).to(torch.int32)
    uvs = torch.from_numpy(cow["verts_uvs"]).to(dev)
    fuv = torch.from_numpy(cow["faces_uvs"]).to(dev).to(torch.int32)
    tex = torch.rand((T, T, 3), generator=torch.Generator().manual_seed(0)).to(dev)
    ndc = ops.project_verts(verts, torch.from_numpy(R).to(dev), torch.from_numpy(Tt).to(dev))
    frag = ops.raster_fwd(ndc, faces, S)
    covered = (frag[0][0] >= 0).nonzero()
    y, x = [int(v) for v in covered[covered.shape[0] // 2]]
    try:
        for det in (True, False):
            ops.set_deterministic(det)
            for bad in (float("nan"), float("inf")):
                g = torch.randn((1, 3, S, S), generator=torch.Generator().manual_seed(1)).to(dev)
                clean_t, clean_b = ops.shade_bwd(g, frag, uvs, fuv, tex, want_bary=True)
                assert torch.isfinite(clean_t).all()
                assert torch.isfinite(ops.raster_bwd(clean_b, frag[0], ndc, faces)).all()
                g[0, 1, y, x] = bad
                gt, gb = ops.shade_bwd(g, frag, uvs, fuv, tex, want_bary=True)
                assert not torch.isfinite(gt).all(), (det, bad)
                gv = ops.raster_bwd(gb, frag[0], ndc, faces)
                assert not torch.isfinite(gv).all(), (det, bad)
    finally:
        ops.set_deterministic(True)


# ---------------------------------------------------------------------------- Winograd F(2x2,3x3) conv
WINO_CASES = [(2, 64, 64, 64, 64), (1, 64, 128, 32, 32), (1, 128, 256, 24, 56), (1, 256, 256, 32, 32), (2, 512, 512, 16, 16),
              (1, 512, 512, 4, 4), (1, 64, 64, 48, 40), (1, 128, 64, 10, 12),
              # more workgroup tiles than CUs: the persistent workgroups walk several tiles (cross-tile prefetch, image
              # change between consecutive tiles, uneven tile counts per workgroup)
              (3, 64, 64, 256, 128), (2, 64, 128, 192, 160), (1, 128, 128, 256, 256)]


@pytest.mark.parametrize("N,Cin,Cout,H,W", WINO_CASES)
def test_wino_fwd_dgrad(dev, ops, N, Cin, Cout, H, W):
    """Winograd conv vs an fp64 direct convolution: within 3e-5 of the output scale (the transforms
    add a few fp32 roundings per product; still fp32 products + fp32 accumulation)."""
    torch.manual_seed(Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3) * (2.0 / (Cin * 9)) ** 0.5).double()
    b = (torch.randn(Cout) * 0.1).double()
    y = F.relu(F.conv2d(x, w, b, padding=1))
    gy = torch.randn_like(y)
    y.backward(gy)
    uf, ud = ops.wino_pack(w.float().to(dev))
    yd = ops.wino_fwd(x.detach().float().to(dev), uf, b.float().to(dev), Cout, relu=True)
    _scale_close(yd, y, 3e-5, "wino fwd")
    y_nr = ops.wino_fwd(x.detach().float().to(dev), uf, b.float().to(dev), Cout, relu=False)
    _scale_close(y_nr, F.conv2d(x.detach(), w, b, padding=1), 3e-5, "wino fwd (no relu)")
    gx = ops.wino_dgrad(gy.float().to(dev), yd, ud, Cin)
    # reference gradient through the SAME gate the kernel saw (the GPU's own activation): an output within rounding
    # of 0 may sit on the other side of the ReLU in fp64, and with millions of outputs some do
    gate = (yd.cpu() > 0).double()
    assert float((gate - (y.detach() > 0).double()).abs().mean()) < 1e-5
    ref_gx = torch.autograd.grad(F.conv2d(x, w, b, padding=1), x, gy * gate)[0]
    _scale_close(gx, ref_gx, 5e-5, "wino dgrad")
    # agreement with the direct MFMA kernel (different algorithm, same inputs)
    wf, wd = ops.conv3x3_pack(w.float().to(dev))
    _scale_close(yd, ops.conv3x3_fwd(x.detach().float().to(dev), wf, b.float().to(dev), Cout, relu=True), 3e-5, "wino vs direct")


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 64, 64, 32, 64), (1, 128, 128, 16, 16), (1, 256, 256, 8, 40),
                                             (3, 64, 64, 256, 128), (2, 128, 64, 160, 192)])
def test_wino_fused_pool_and_unpool(dev, ops, N, Cin, Cout, H, W):
    """conv -> ReLU -> MaxPool2d(2,2) fused in the Winograd epilogue (values + ATen's first-max argmax),
    and the gradient back through pool + ReLU + conv in one kernel."""
    torch.manual_seed(H * 3 + W)
    x = torch.randn(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3) * (2.0 / (Cin * 9)) ** 0.5).double()
    b = (torch.randn(Cout) * 0.1).double()
    y = F.relu(F.conv2d(x, w, b, padding=1))
    p = F.max_pool2d(y, 2, 2)
    gp = torch.randn_like(p)
    uf, ud = ops.wino_pack(w.float().to(dev))
    yd, pd, idx = ops.wino_fwd(x.detach().float().to(dev), uf, b.float().to(dev), Cout, relu=True, pool=True)
    _scale_close(yd, y, 3e-5, "wino fwd")
    p2, idx2 = ops.maxpool2x2(yd)
    assert torch.equal(pd, p2) and torch.equal(idx, idx2)          # fused pool == separate pool kernel on the same values
    none_full, pd3, idx3 = ops.wino_fwd(x.detach().float().to(dev), uf, b.float().to(dev), Cout, relu=True, pool=True,
                                        keep_full=False)
    assert none_full is None and torch.equal(pd3, pd) and torch.equal(idx3, idx)
    gx = ops.wino_dgrad_unpool(gp.float().to(dev), idx, pd, ud, Cin)
    # reference through the SAME argmax / gate the kernel saw (near-ties inside a window and outputs within rounding of
    # 0 can resolve differently in fp64; with millions of windows some do)
    ic = idx.cpu().long()
    up = torch.zeros_like(y)
    gated = gp * (pd.cpu() > 0).double()
    for k in range(4):
        up[:, :, (k >> 1)::2, (k & 1)::2] = gated * (ic == k).double()
    assert float((up != 0).double().mean() - (torch.autograd.grad(p, y, gp, retain_graph=True)[0] != 0).double().mean()) < 1e-4
    ref_gx = torch.autograd.grad(F.conv2d(x, w, b, padding=1), x, up)[0]
    _scale_close(gx, ref_gx, 6e-5, "wino dgrad_unpool")


# ---------------------------------------------------------------------------- Winograd F(4x4,3x3) conv (round 3)
@pytest.mark.parametrize("N,Cin,Cout,H,W", [(1, 64, 64, 4, 64), (2, 128, 64, 8, 64), (1, 256, 256, 16, 128), (1, 512, 512, 64, 64),
                                             (2, 128, 256, 12, 192),
                                             # 8 x 32-pixel workgroup steps (W % 32 == 0, H % 8 == 0): conv5_1 at 512^2, a 96-wide map
                                             (2, 512, 512, 32, 32), (2, 64, 128, 16, 96), (1, 128, 64, 8, 32)])
@pytest.mark.parametrize("slots", ["", "1", "4"])
def test_wino43_fwd_and_chain_dgrad(dev, ops, monkeypatch, N, Cin, Cout, H, W, slots):
    """F(4x4,3x3) kernel (csrc/wino43.hip) against an fp64 direct convolution at the SAME tolerance as the F(2x2,3x3)
    kernel (3e-5 of the output scale forward, 5e-5 input-gradient; measured ~1e-5 at K = 512), the fused pool against the
    separate pool kernel (bitwise), the producer-side output gate / content term and the fused unpool against the plain
    launch on pre-processed operands (bitwise: same arithmetic), and against the F(2x2,3x3) kernel on the same inputs.
    slots: persistent workgroups per cout tile (default: CUs / cout tiles) -- "1" and "4" make one workgroup walk many pixel
    tiles, ragged (18 tiles over 4 slots), so the stream of stages runs across tile boundaries on these small shapes too."""
    if slots:
        monkeypatch.setenv("ST3D_W43_SLOTS", slots)
    torch.manual_seed(Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3) * (2.0 / (Cin * 9)) ** 0.5).double()
    b = (torch.randn(Cout) * 0.1).double()
    pre = F.conv2d(x, w, b, padding=1)
    y = F.relu(pre)
    xd, bd = x.detach().float().to(dev), b.float().to(dev)
    assert ops._lib.load().st3d_wino43_supported(Cin, Cout, H, W) == 1
    uf, ud = ops.wino43_pack(w.float().to(dev))
    yd = ops.wino43_fwd(xd, uf, bd, Cout, relu=True)
    _scale_close(yd, y, 3e-5, "wino43 fwd")
    _scale_close(ops.wino43_fwd(xd, uf, bd, Cout, relu=False), pre, 3e-5, "wino43 fwd (no relu)")
    u2f, u2d = ops.wino_pack(w.float().to(dev))
    _scale_close(yd, ops.wino_fwd(xd, u2f, bd, Cout, relu=True), 3e-5, "wino43 vs wino F(2x2,3x3)")
    # fused pool
    yf, pd, idx = ops.wino43_fwd(xd, uf, bd, Cout, relu=True, pool=True)
    p2, idx2 = ops.maxpool2x2(yd)
    assert torch.equal(yf, yd) and torch.equal(pd, p2) and torch.equal(idx, idx2)
    nf, pd3, idx3 = ops.wino43_fwd(xd, uf, bd, Cout, relu=True, pool=True, keep_full=False)
    assert nf is None and torch.equal(pd3, pd) and torch.equal(idx3, idx)
    # input gradient of a pre-gated gradient
    gy = torch.randn_like(y)
    gate = (yd.cpu() > 0).double()
    gyg = (gy * gate).float().to(dev)
    gx = ops.wino43_dgrad_chain(gyg, ud, Cin)
    ref_gx = torch.autograd.grad(pre, x, gy * gate)[0]
    _scale_close(gx, ref_gx, 5e-5, "wino43 dgrad")
    # producer-side gate of the NEXT link and the content term: bitwise the plain result post-processed
    og = torch.randn(N, Cin, H, W).to(dev)
    tgt = torch.randn(N, Cin, H, W).to(dev)
    assert torch.equal(ops.wino43_dgrad_chain(gyg, ud, Cin, out_gate=og), torch.where(og > 0, gx, torch.zeros_like(gx)))
    want = torch.where(og > 0, gx + 0.37 * (og - tgt), torch.zeros_like(gx))
    assert torch.equal(ops.wino43_dgrad_chain(gyg, ud, Cin, out_gate=og, add_target=tgt, add_coef=0.37), want)
    # fused unpool: pooled-resolution gradient + argmax == the plain launch on the scattered full-resolution gradient
    gp = torch.randn(N, Cout, H // 2, W // 2).to(dev)
    pidx = torch.randint(0, 4, (N, Cout, H // 2, W // 2), dtype=torch.uint8).to(dev)
    up = torch.zeros(N, Cout, H, W, device=dev)
    for k in range(4):
        up[:, :, (k >> 1)::2, (k & 1)::2] = gp * (pidx == k).float()
    assert torch.equal(ops.wino43_dgrad_chain(gp, ud, Cin, pool_idx=pidx), ops.wino43_dgrad_chain(up, ud, Cin))
    assert torch.equal(ops.wino43_dgrad_chain(gp, ud, Cin, pool_idx=pidx, out_gate=og),
                       ops.wino43_dgrad_chain(up, ud, Cin, out_gate=og))
    # shapes outside the tiling are refused, not mis-computed
    assert ops._lib.load().st3d_wino43_supported(Cin, Cout, H, W + 16) == 0 and ops._lib.load().st3d_wino43_supported(48, Cout, H, W) == 0
    assert ops._lib.load().st3d_wino43_supported(Cin, Cout, 12, 96) == 0         # 32-pixel rows need H % 8 == 0


# ---------------------------------------------------------------------------- general soft renderer (K faces / pixel, blur)
@pytest.mark.parametrize("K,blur,clip", [(1, 0.0, False), (3, 0.0, False), (4, 2e-4, True), (8, 1e-3, True)])
def test_soft_raster_matches_oracle(dev, ops, cow, K, blur, clip):
    """(B,S,S,K) fragments bit-exact vs oracle/raster_ref.c:ref_rasterize_k (same operation order, no FMA contraction)."""
    from oracle import render_ref as rr
    S = 72
    R, T = _cams(2, seed=K)
    verts = torch.from_numpy(cow["verts"]).to(dev)
    faces = torch.from_numpy(cow["faces"]).to(dev)
    ndc = ops.project_verts(verts, torch.from_numpy(R).to(dev), torch.from_numpy(T).to(dev))
    p2f, zbuf, bary, dists = ops.raster_soft_fwd(ndc, faces, S, K, blur, clip)
    for b in range(2):
        rp, rz, rb, rd = rr.rasterize_k(rr.project_verts(cow["verts"], R[b], T[b]), cow["faces"], S, K, blur, clip, nthreads=8)
        np.testing.assert_array_equal(p2f[b].cpu().numpy(), rp)
        np.testing.assert_array_equal(zbuf[b].cpu().numpy(), rz)
        np.testing.assert_array_equal(bary[b].cpu().numpy(), rb)
        np.testing.assert_array_equal(dists[b].cpu().numpy(), rd)
        if K > 1:
            assert ((rp[..., 1] >= 0).sum() > 0.02 * S * S)            # second layers exist (back faces of the cow)
    if K == 1 and blur == 0.0:      # the general kernel reproduces the specialised hard rasteriser
        h = ops.raster_fwd(ndc, faces, S)
        assert torch.equal(h[0], p2f[..., 0]) and torch.equal(h[1], zbuf[..., 0]) and torch.equal(h[3], dists[..., 0])


@pytest.mark.parametrize("K,blur,sigma,gamma", [(1, 0.0, 1e-4, 1e-4), (4, 3e-4, 1e-4, 1e-4), (3, 1e-3, 1e-3, 1e-2)])
def test_soft_shade_forward_and_backward_match_oracle(dev, ops, cow, K, blur, sigma, gamma):
    """softmax_rgb_blend over K layers: pixels against the torch restatement; d/d texture, d/d (bary, depth, signed
    edge distance) and d/d vertices against fp64 autograd evaluated on the same fragments."""
    from oracle import render_ref as rr
    from oracle import soft_ref as SR
    S, Tn, B = 64, 24, 2
    bg = (0.2, 0.5, 0.9)
    R, T = _cams(B, seed=7)
    rng = np.random.default_rng(1)
    tex = rng.random((Tn, Tn, 3), dtype=np.float32)
    verts = torch.from_numpy(cow["verts"]).to(dev)
    faces = torch.from_numpy(cow["faces"]).to(dev)
    uvs = torch.from_numpy(cow["verts_uvs"]).to(dev)
    fuv = torch.from_numpy(cow["faces_uvs"]).to(dev)
    texd = torch.from_numpy(tex).to(dev)
    Rd, Td = torch.from_numpy(R).to(dev), torch.from_numpy(T).to(dev)
    clip = blur > 0
    ndc = ops.project_verts(verts, Rd, Td)
    frag = ops.raster_soft_fwd(ndc, faces, S, K, blur, clip)
    rgb, alpha = ops.shade_soft_fwd(frag, uvs, fuv, texd, sigma, gamma, bg)
    g = rng.standard_normal((B, 3, S, S)).astype(np.float32)
    gt, geo = ops.shade_soft_bwd(torch.from_numpy(g).to(dev), frag, uvs, fuv, texd, sigma, gamma, bg)
    gndc = ops.raster_soft_bwd(geo, frag[0], ndc, faces, clip)
    gverts = ops.project_verts_bwd(verts, Rd, Td, gndc)
    # fp64 autograd at the same coverage, stage by stage on the SAME fp32 fragments (an end-to-end fp64 recomputation
    # lands a handful of fragments in the neighbouring texel cell / on the other side of a clamp, where the
    # gradient is discontinuous; those outliers say nothing about the kernels)
    fc = torch.from_numpy(cow["faces"]).long()
    uv64, fuv64 = torch.from_numpy(cow["verts_uvs"]).double(), torch.from_numpy(cow["faces_uvs"]).long()
    tt = torch.from_numpy(tex).double().requires_grad_(True)
    vt = torch.from_numpy(cow["verts"]).double().requires_grad_(True)
    # the fp32 depth term (zfar - z)/(zfar - znear) carries ~6e-8 of rounding which the softmax divides by gamma:
    # with K > 1 layers the fp32 pixel can sit 6e-8/gamma away from the fp64 restatement (6e-4 at gamma = 1e-4)
    atol = 2e-5 if K == 1 else max(2e-5, 3e-8 / gamma)
    gtol = 5e-5 if K == 1 else max(5e-5, 2e-7 / gamma)
    for b in range(B):
        p2f = frag[0][b].cpu().long()
        mask = p2f >= 0
        bl, zl, dl = (frag[i][b].cpu().double().requires_grad_(True) for i in (2, 1, 3))
        colors = SR.sample_texture(bl, p2f, uv64, fuv64, tt)
        r, a_ = SR.softmax_rgb_blend(colors, zl, dl, mask, sigma, gamma, bg)
        np.testing.assert_allclose(rgb[b].cpu().numpy(), r.detach().permute(2, 0, 1).numpy(), atol=atol)
        np.testing.assert_allclose(alpha[b, 0].cpu().numpy(), a_.detach().numpy(), atol=2e-5)
        (r.permute(2, 0, 1) * torch.from_numpy(g[b]).double()).sum().backward()
        # d/d depth and d/d distance multiply (colour_k - pixel), an fp32 difference with ~1e-7 of rounding, by
        # 1/(99 gamma) resp. 1/(4 sigma): that absolute floor is what is left when the true gradient vanishes (K = 1)
        gnorm = float(np.linalg.norm(g[b]))
        for ref_g, got_g, floor in ((bl.grad, geo[0][b], 0.0), (zl.grad, geo[1][b], 4e-7 * gnorm / (99 * gamma)),
                                    (dl.grad, geo[2][b], 4e-7 * gnorm / (4 * sigma))):
            m = mask.double() if ref_g.dim() == 3 else mask.double().unsqueeze(-1)
            err = float(((got_g.cpu().double() - ref_g) * m).norm())
            assert err <= gtol * float((ref_g * m).norm()) + floor, (err, float((ref_g * m).norm()), floor)
        # raster + projection backward with the GPU's own upstream gradients
        ndc_b = SR.project(vt, torch.from_numpy(R[b]).double(), torch.from_numpy(T[b]).double())
        # (the fp32 ndc the kernels consumed, with the fp64 projection's graph attached)
        ndc_b = ndc_b + (ndc[b].cpu().double() - ndc_b).detach()
        bary64, pz64, sd64, m64 = SR.soft_geometry(ndc_b, fc, p2f, S, clip)
        md = m64.double()
        ((bary64 * geo[0][b].cpu().double() * md.unsqueeze(-1)).sum() + (pz64 * geo[1][b].cpu().double() * md).sum()
         + (sd64 * geo[2][b].cpu().double() * md).sum()).backward()
    rel_t = float((gt.cpu().double() - tt.grad).norm() / tt.grad.norm())
    rel_v = float((gverts.cpu().double() - vt.grad).norm() / vt.grad.norm())
    assert rel_t <= (2e-5 if K == 1 else max(2e-5, 2e-7 / gamma)), rel_t
    assert rel_v <= 5e-5, rel_v


@pytest.mark.parametrize("K,blur,near", [(1, 0.0, True), (4, 3e-4, False), (8, 1e-3, True)])
def test_soft_backward_scatters_fixed_point_vs_float_atomics(dev, ops, cow, K, blur, near):
    """Round 3: the general path's two scatters (texture: per-tile texel table; vertices: per-tile face table) in both
    modes.  Fixed point (the default): bitwise reproducible; float atomics: equal to it within their own run-to-run spread.
    128^2 with up to 8 layers overflows the 2048-texel / 512-face tables of a tile, so the direct-to-global path of a
    contribution that finds no slot is exercised; `near` puts the camera inside the cow's bounding sphere (clipped faces,
    faces that cover many tiles -- the shape of BASELINE config 5 after its vertices reach the camera).  A NaN upstream
    gradient comes out as NaN in both modes."""
    from oracle import render_ref as rr
    S, Tn, B = 128, 64, 2
    if near:
        R, T = rr.look_at_view_transform(0.8, [10.0, -20.0], [35.0, 140.0], at=(0, 0.10, 0.25))
    else:
        R, T = _cams(B, seed=11)
    rng = np.random.default_rng(2)
    verts = torch.from_numpy(cow["verts"]).to(dev)
    faces = torch.from_numpy(cow["faces"]).to(dev)
    uvs = torch.from_numpy(cow["verts_uvs"]).to(dev)
    fuv = torch.from_numpy(cow["faces_uvs"]).to(dev)
    texd = torch.from_numpy(rng.random((Tn, Tn, 3), dtype=np.float32)).to(dev)
    ndc = ops.project_verts(verts, torch.from_numpy(R).to(dev), torch.from_numpy(T).to(dev))
    clip = blur > 0
    out = ops.raster_soft_fwd(ndc, faces, S, K, blur, clip, z_clip=0.5)
    frag, slots = out[:4], out[4]
    assert float((frag[0] >= 0).float().mean()) > 0.05
    g = torch.from_numpy(rng.standard_normal((B, 3, S, S)).astype(np.float32)).to(dev)

    def both():
        gt, geo = ops.shade_soft_bwd(g, frag, uvs, fuv, texd, 1e-4, 1e-4, (1.0, 1.0, 1.0))
        gv = ops.raster_soft_bwd(geo, frag[0], ndc, faces, clip, slots=slots, z_clip=0.5)
        return gt, gv
    try:
        assert ops.is_deterministic()
        t1, v1 = both()
        t2, v2 = both()
        assert torch.equal(t1, t2) and torch.equal(v1, v2)
        ops.set_deterministic(False)
        ta, va = both()
        tb, vb = both()
        spread_t = float((ta - tb).norm() / ta.norm())
        spread_v = float((va - vb).norm() / va.norm())
        assert float((t1 - ta).norm() / ta.norm()) <= max(3 * spread_t, 2e-6)
        assert float((v1 - va).norm() / va.norm()) <= max(3 * spread_v, 2e-5)
        gn = g.clone()
        covered = (frag[0][0, ..., 0] >= 0).nonzero()
        y, x = [int(v) for v in covered[covered.shape[0] // 2]]
        gn[0, 2, y, x] = float("nan")
        for det in (True, False):
            ops.set_deterministic(det)
            gt, geo = ops.shade_soft_bwd(gn, frag, uvs, fuv, texd, 1e-4, 1e-4, (1.0, 1.0, 1.0))
            assert not torch.isfinite(gt).all()
            assert not torch.isfinite(ops.raster_soft_bwd(geo, frag[0], ndc, faces, clip, slots=slots, z_clip=0.5)).all()
    finally:
        ops.set_deterministic(True)


@pytest.mark.parametrize("K,blur,cull,persp", [(1, 0.0, True, True), (2, 0.0, False, False), (4, 5e-4, True, False)])
def test_soft_raster_cull_backfaces_and_perspective_correct_flags(dev, ops, cow, K, blur, cull, persp):
    """RasterizationSettings.cull_backfaces / perspective_correct=False on the general kernels: fragments bit-exact vs
    the C oracle; the vertex gradient without perspective correction against fp64 autograd at the same coverage."""
    from oracle import render_ref as rr
    from oracle import soft_ref as SR
    S, B = 72, 2
    R, T = _cams(B, seed=11 + K)
    clip = blur > 0
    verts = torch.from_numpy(cow["verts"]).to(dev)
    faces = torch.from_numpy(cow["faces"]).to(dev)
    Rd, Td = torch.from_numpy(R).to(dev), torch.from_numpy(T).to(dev)
    ndc = ops.project_verts(verts, Rd, Td)
    frag = ops.raster_soft_fwd(ndc, faces, S, K, blur, clip, cull_backfaces=cull, perspective_correct=persp)
    for b in range(B):
        ref = rr.rasterize_k(rr.project_verts(cow["verts"], R[b], T[b]), cow["faces"], S, K, blur, clip, nthreads=8,
                             cull_backfaces=cull, perspective_correct=persp)
        for got, want in zip(frag, ref):
            np.testing.assert_array_equal(got[b].cpu().numpy(), want)
    plain = ops.raster_soft_fwd(ndc, faces, S, K, blur, clip)
    if cull and K == 1:
        assert torch.equal(plain[0], frag[0])                   # closed mesh seen from outside: nearest faces are front faces
    if not persp:
        assert not torch.equal(plain[2], frag[2])
        rng = np.random.default_rng(0)
        gb = torch.from_numpy(rng.standard_normal((B, S, S, K, 3)).astype(np.float32)).to(dev)
        gz = torch.from_numpy(rng.standard_normal((B, S, S, K)).astype(np.float32)).to(dev)
        gd = torch.zeros_like(gz)
        gndc = ops.raster_soft_bwd((gb, gz, gd), frag[0], ndc, faces, clip, perspective_correct=False)
        fc = torch.from_numpy(cow["faces"]).long()
        for b in range(B):
            nd = ndc[b].cpu().double().requires_grad_(True)
            p2f = frag[0][b].cpu().long()
            bary64, pz64, _, m64 = SR.soft_geometry(nd, fc, p2f, S, clip, perspective_correct=False)
            md = m64.double()
            ((bary64 * gb[b].cpu().double() * md.unsqueeze(-1)).sum() + (pz64 * gz[b].cpu().double() * md).sum()).backward()
            rel = float((gndc[b].cpu().double() - nd.grad).norm() / nd.grad.norm())
            assert rel <= 5e-5, rel


def _near_scenes(cow):
    """(name, ndc (V,3), faces): the analytic straddling triangles of tests/test_oracle_soft.py and the cow seen from
    INSIDE its bounding sphere (camera 0.75 from the look-at point: hundreds of faces cross z = 0.5 or lie behind it)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_soft import _straddling_scene
    from oracle import render_ref as rr
    out = [("one behind", *_straddling_scene(1)), ("two behind", *_straddling_scene(2))]
    R, T = rr.look_at_view_transform(0.75, [10.0], [35.0], at=(0, 0.10, 0.25))
    out.append(("cow from inside", rr.project_verts(cow["verts"], R[0], T[0]), cow["faces"]))
    return out


@pytest.mark.parametrize("K,blur,persp", [(1, 0.0, True), (3, 2e-3, True), (2, 0.0, False)])
def test_soft_raster_near_plane_clipping_matches_oracle(dev, ops, cow, K, blur, persp):
    """z_clip (PyTorch3D clips meshes at znear / 2 before rasterising): clipped coverage, ORIGINAL face indices, converted
    barycentrics, depth, distances and the fragments' record slots bit-exact against oracle/raster_ref.c:ref_rasterize_k3."""
    from oracle import render_ref as rr
    S = 80
    for name, ndc_np, faces_np in _near_scenes(cow):
        ndc = torch.from_numpy(ndc_np)[None].to(dev)
        faces = torch.from_numpy(faces_np).to(dev)
        got = ops.raster_soft_fwd(ndc, faces, S, K, blur, blur > 0, perspective_correct=persp, z_clip=0.5)
        ref = rr.rasterize_k(ndc_np, faces_np, S, K, blur, blur > 0, nthreads=8, perspective_correct=persp, z_clip=0.5,
                             return_slots=True)
        assert len(got) == 5
        for g, r in zip(got, ref):
            np.testing.assert_array_equal(g[0].cpu().numpy(), r, err_msg=name)
        slots = ref[4]
        if name != "cow from inside":
            assert ((slots >= 0) & (slots & 1 == 1)).any() == (name == "one behind")
        else:
            unclipped = rr.rasterize_k(ndc_np, faces_np, S, K, blur, blur > 0, nthreads=8, perspective_correct=persp)
            assert (unclipped[0] != ref[0]).mean() > 0.02            # the plane really cuts this view
            assert ((slots >= 0) & (slots & 1 == 1)).sum() > 20       # second halves of split quadrilaterals are hit


def test_soft_raster_backward_through_clipped_faces_matches_fp64_autograd(dev, ops, cow):
    """d loss / d projected vertices through fragments that live on clipped sub-triangles (the cut points and the
    barycentric conversion depend on the vertices) against torch autograd of oracle/soft_ref.py:clipped_geometry."""
    from oracle import soft_ref as SR
    S = 40
    rng = np.random.default_rng(0)
    for name, ndc_np, faces_np in _near_scenes(cow)[:2]:
        for persp, K, blur in ((True, 1, 0.0), (False, 1, 0.0), (True, 2, 3e-3)):
            clip = blur > 0
            ndc = torch.from_numpy(ndc_np)[None].to(dev)
            faces = torch.from_numpy(faces_np).to(dev)
            p2f, zbuf, bary, dists, slots = ops.raster_soft_fwd(ndc, faces, S, K, blur, clip, perspective_correct=persp, z_clip=0.5)
            assert (slots >= 0).sum() > 15
            gb = torch.from_numpy(rng.standard_normal((1, S, S, K, 3)).astype(np.float32)).to(dev)
            gz = torch.from_numpy(rng.standard_normal((1, S, S, K)).astype(np.float32)).to(dev)
            gd = torch.from_numpy(rng.standard_normal((1, S, S, K)).astype(np.float32)).to(dev)
            g = ops.raster_soft_bwd((gb, gz, gd), p2f, ndc, faces, clip, perspective_correct=persp, slots=slots, z_clip=0.5)
            nd = torch.from_numpy(ndc_np).double().requires_grad_(True)
            b64, z64, d64, m64 = SR.clipped_geometry(nd, torch.from_numpy(faces_np).long(), slots[0].cpu().long(), S, clip, persp, 0.5)
            np.testing.assert_allclose(b64.detach().numpy()[m64.numpy()], bary[0].cpu().numpy()[m64.numpy()], atol=5e-6)
            md = m64.double()
            ((b64 * gb[0].cpu().double() * md.unsqueeze(-1)).sum() + (z64 * gz[0].cpu().double() * md).sum()
             + (d64 * gd[0].cpu().double() * md).sum()).backward()
            rel = float((g[0].cpu().double() - nd.grad).norm() / nd.grad.norm())
            assert rel <= 1e-4, (name, persp, K, rel)


def test_renderer_near_plane_policy(dev, cow, monkeypatch):
    """PyTorch3D clips at z_clip_value = znear / 2.  Through MeshRenderer: an explicit z_clip_value renders on the general
    kernels WITH clipping (pixels against the oracle's clipped fragments, gradients flow); the specialised K = 1 path does
    not clip, so the renderer asks for the vertices' nearest depth BEFORE it renders: a batch that reaches the plane goes to the
    clipping kernels (one warning; 0/1 mask as the hard settings promise), under ST3D_NEAR_PLANE=raise it fails; the
    reference's own views (nothing nearer than 0.78) stay on the specialised kernels."""
    import utils as U
    from oracle import render_ref as rr
    from st3d import ops as O
    from st3d.render import FoVPerspectiveCameras, MeshRasterizer, MeshRenderer, RasterizationSettings, SoftPhongShader
    U.device = dev
    S = 64
    rng = np.random.default_rng(2)
    tex_np = rng.random((32, 32, 3), dtype=np.float32)
    tex = torch.from_numpy(tex_np)[None].to(dev).requires_grad_(True)
    verts = torch.from_numpy(cow["verts"]).to(dev).requires_grad_(True)
    mesh = U.build_mesh(torch.from_numpy(cow["verts_uvs"])[None].to(dev), torch.from_numpy(cow["faces_uvs"].astype(np.int64))[None].to(dev),
                        tex, verts, torch.from_numpy(cow["faces"].astype(np.int64)).to(dev))
    Rn, Tn = rr.look_at_view_transform(0.75, [10.0], [35.0], at=(0, 0.10, 0.25))          # camera inside the cow's bounding sphere
    near = FoVPerspectiveCameras(R=torch.from_numpy(Rn), T=torch.from_numpy(Tn), device=dev)
    Rf, Tf = _cams(1, seed=1)
    far = FoVPerspectiveCameras(R=torch.from_numpy(Rf), T=torch.from_numpy(Tf), device=dev)
    clipping = MeshRenderer(MeshRasterizer(None, RasterizationSettings(image_size=S, z_clip_value=0.5)), SoftPhongShader())
    plain = MeshRenderer(MeshRasterizer(None, RasterizationSettings(image_size=S)), SoftPhongShader())
    assert plain.is_hard and not clipping.is_hard
    O.check_near_plane(block=True)
    rgb, cov = clipping.render(mesh, near)
    frag = rr.rasterize_k(rr.project_verts(cow["verts"], Rn[0], Tn[0]), cow["faces"], S, 1, 0.0, nthreads=8, z_clip=0.5)
    np.testing.assert_array_equal((cov[0, 0] > 0).cpu().numpy(), frag[0][..., 0] >= 0)
    ref_rgb, _ = rr.shade_fwd(tuple(a[..., 0] if a.ndim == 3 else a[..., 0, :] for a in frag), cow["verts_uvs"], cow["faces_uvs"], tex_np)
    np.testing.assert_allclose(rgb[0].detach().cpu().numpy(), ref_rgb, atol=3e-6)
    rgb.sum().backward()
    assert torch.isfinite(verts.grad).all() and float(verts.grad.abs().sum()) > 0
    # far view: both paths agree and the watch stays silent
    with torch.no_grad():
        a, _ = plain.render(mesh, far)
        b, _ = clipping.render(mesh, far)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-6)
    O.check_near_plane(block=True)
    # near view under the reference's hard settings, policy "raise": loud, BEFORE anything is rendered
    monkeypatch.setattr(O, "NEAR_PLANE_POLICY", "raise")
    with torch.no_grad(), pytest.raises(RuntimeError, match="z_clip_value"):
        plain.render(mesh, near)
    # default policy "clip": the FIRST such frame already goes through the clipping kernels (one warning per process) and the
    # renderer still hands out what its hard settings promise: the 0/1 coverage mask (alpha in [0.5, 1) leaked)
    monkeypatch.setattr(O, "NEAR_PLANE_POLICY", "clip")
    try:
        with torch.no_grad():
            with pytest.warns(UserWarning, match="near clipping plane"):
                c, ccov = plain.render(mesh, near)
            assert not O.near_plane_triggered()         # decided per render from the vertices' depths, nothing sticky
            c2, m2 = U.render_meshes(plain, mesh, near)
            a2, _ = plain.render(mesh, far)             # and the next far view is back on the specialised kernels
        covered = frag[0][..., 0] >= 0
        for cov_t in (ccov, m2):
            assert set(np.unique(cov_t.cpu().numpy()).tolist()) <= {0.0, 1.0}
            np.testing.assert_array_equal(cov_t[0, 0].cpu().numpy() == 1.0, covered)
        np.testing.assert_allclose(c[0].cpu().numpy(), ref_rgb, atol=3e-6)
        np.testing.assert_allclose(c2[0].cpu().numpy(), ref_rgb, atol=3e-6)
        np.testing.assert_array_equal(a2.cpu().numpy(), a.cpu().numpy())
        # vertices under optimisation are asked every step (no cache): moving the SAME tensor into the plane is noticed
        from st3d.render import reaches_near_plane
        R, T = far.R, far.T
        vv = torch.from_numpy(cow["verts"]).to(dev)
        assert not reaches_near_plane(vv, R, T, 0.5) and not reaches_near_plane(vv, R, T, 0.5)
        vv += (torch.tensor([0.0, 0.0, -1.7], device=dev) @ R[0].t())          # in place: same address, new version
        assert reaches_near_plane(vv, R, T, 0.5)
        # the asynchronous watch of DIRECT st3d_raster_fwd callers (no renderer in between) still raises its flag
        ndc = O.project_verts(torch.from_numpy(cow["verts"]).to(dev), near.R, near.T)
        O.raster_fwd(ndc, mesh.faces_i32(), S, z_clip=0.5)
        O.check_near_plane(block=True)
        assert O.near_plane_triggered()
    finally:
        O.reset_near_plane()
    assert not O.near_plane_triggered()


def test_renderer_routes_non_default_raster_flags_to_the_general_kernels(dev, cow):
    """MeshRenderer with cull_backfaces=True or perspective_correct=False (K = 1, blur 0) leaves the specialised path; the
    culled render of the closed cow equals the default render, the uncorrected one differs and back-propagates."""
    import utils as U
    from st3d.render import FoVPerspectiveCameras, MeshRasterizer, MeshRenderer, RasterizationSettings, SoftPhongShader
    U.device = dev
    S = 64
    R, T = _cams(2, seed=3)
    rng = np.random.default_rng(5)
    tex = torch.from_numpy(rng.random((1, 32, 32, 3), dtype=np.float32)).to(dev).requires_grad_(True)
    verts = torch.from_numpy(cow["verts"]).to(dev).requires_grad_(True)
    mesh = U.build_mesh(torch.from_numpy(cow["verts_uvs"])[None].to(dev), torch.from_numpy(cow["faces_uvs"].astype(np.int64))[None].to(dev),
                        tex, verts, torch.from_numpy(cow["faces"].astype(np.int64)).to(dev))
    cams = FoVPerspectiveCameras(R=torch.from_numpy(R), T=torch.from_numpy(T), device=dev)
    mk = lambda **kw: MeshRenderer(MeshRasterizer(None, RasterizationSettings(image_size=S, **kw)), SoftPhongShader())
    base, culled, flat = mk(), mk(cull_backfaces=True), mk(perspective_correct=False)
    assert base.is_hard and not culled.is_hard and not flat.is_hard
    with torch.no_grad():
        a, ma = base.render(mesh, cams)
        c, mc = culled.render(mesh, cams)
    np.testing.assert_allclose(c.cpu().numpy(), a.cpu().numpy(), atol=2e-6)
    assert torch.equal(mc > 0, ma > 0)
    f, _ = flat.render(mesh, cams)
    assert float((f.detach() - a).abs().max()) > 1e-3
    f.sum().backward()
    assert torch.isfinite(verts.grad).all() and float(verts.grad.abs().sum()) > 0 and float(tex.grad.abs().sum()) > 0


def _crop_ref(x, y0, x0, h, halo, fn):
    """fn (a stack of pad-1 convolutions needing `halo` pixels of context) on the crop [y0,y0+h) x [x0,x0+h) of the
    image x (1,C,S,S): evaluated on the crop + halo clipped to the image -- zero padding at the region edge is the true
    padding on image borders and only pollutes the halo ring elsewhere."""
    S = x.shape[-1]
    ya, xa, yb, xb = max(y0 - halo, 0), max(x0 - halo, 0), min(y0 + h + halo, S), min(x0 + h + halo, S)
    out = fn(x[:, :, ya:yb, xa:xb].cpu().double())
    return out[:, :, y0 - ya:y0 - ya + h, x0 - xa:x0 - xa + h]


def test_wino_32bit_offset_guard_and_large_images(dev, ops):
    """The Winograd kernels address one image of the input with 32-bit BYTE offsets: st3d_wino_supported must refuse
    Cin*H*W*4 >= 2^31 (the plan then runs the direct kernels), and the largest admitted class -- offsets above 2^30
    bytes -- must still be exact.  64 -> 64 channels at 2048^2 (1 GiB per operand) against an fp64 convolution on
    crops at the corners, an edge, the middle and the far end of the buffer."""
    from st3d import _lib
    lib = _lib.load()
    assert lib.st3d_wino_supported(64, 64, 2048, 2048) == 1
    assert lib.st3d_wino_supported(64, 64, 2896, 2896) == 1            # 64*2896^2*4 = 2 147 024 896 < 2^31
    assert lib.st3d_wino_supported(64, 64, 2900, 2900) == 0
    assert lib.st3d_wino_supported(64, 64, 4096, 4096) == 0 and lib.st3d_wino_supported(512, 64, 1024, 1024) == 0
    assert lib.st3d_wino_supported(128, 128, 2048, 2048) == 0           # 2 GiB exactly
    S, C, h = 2048, 64, 40
    x = torch.randn((1, C, S, S), generator=torch.Generator(device=dev).manual_seed(0), device=dev)
    w = torch.randn((C, C, 3, 3), generator=torch.Generator().manual_seed(1)) * 0.05
    b = torch.randn((C,), generator=torch.Generator().manual_seed(2))
    uf, ud = ops.wino_pack(w.to(dev))
    y = ops.wino_fwd(x, uf, b.to(dev), C, relu=False)
    gx = ops.wino_dgrad(x, None, ud, C)
    wd = w.double()
    fwd = lambda t: F.conv2d(t, wd, b.double(), padding=1)
    bwd = lambda t: F.conv_transpose2d(t, wd, padding=1)               # input-gradient of conv2d(., w, pad 1)
    for y0, x0 in ((0, 0), (0, S - h), (S - h, 0), (S - h, S - h), (1000, 1004), (S - h, 900)):
        ref = _crop_ref(x, y0, x0, h, 1, fwd)
        got = y[:, :, y0:y0 + h, x0:x0 + h].cpu().double()
        assert float((got - ref).abs().max()) <= 3e-5 * float(ref.abs().max()), (y0, x0)
        refg = _crop_ref(x, y0, x0, h, 1, bwd)
        gotg = gx[:, :, y0:y0 + h, x0:x0 + h].cpu().double()
        assert float((gotg - refg).abs().max()) <= 3e-5 * float(refg.abs().max()), (y0, x0)
    del y, gx
    # above the guard the entry points refuse loudly instead of reading garbage
    with pytest.raises(_lib.St3dError):
        ops.wino_fwd(torch.empty((1, 64, 2900, 2900), device=dev), uf, None, C)
    # the same for the F(4x4,3x3) kernel (what the plan runs at this size): guard, then the 1 GiB operands
    assert lib.st3d_wino43_supported(64, 64, 2048, 2048) == 1 and lib.st3d_wino43_supported(64, 64, 2880, 2880) == 1
    assert lib.st3d_wino43_supported(64, 64, 2944, 2944) == 0 and lib.st3d_wino43_supported(128, 128, 2048, 2048) == 0
    u6f, u6d = ops.wino43_pack(w.to(dev))
    y = ops.wino43_fwd(x, u6f, b.to(dev), C, relu=False)
    gx = ops.wino43_dgrad_chain(x, u6d, C)
    for y0, x0 in ((0, 0), (0, S - h), (S - h, 0), (S - h, S - h), (1000, 1004), (S - h, 900)):
        ref = _crop_ref(x, y0, x0, h, 1, fwd)
        got = y[:, :, y0:y0 + h, x0:x0 + h].cpu().double()
        assert float((got - ref).abs().max()) <= 3e-5 * float(ref.abs().max()), (y0, x0)
        refg = _crop_ref(x, y0, x0, h, 1, bwd)
        gotg = gx[:, :, y0:y0 + h, x0:x0 + h].cpu().double()
        assert float((gotg - refg).abs().max()) <= 5e-5 * float(refg.abs().max()), (y0, x0)
    del y, gx
    with pytest.raises(_lib.St3dError):
        ops.wino43_fwd(torch.empty((1, 64, 2944, 2944), device=dev), u6f, None, C)


def test_plan_falls_back_to_direct_kernels_above_the_wino_guard(dev):
    """VGG forward at 2912^2 (conv1_2's input is 64*2912^2*4 B > 2^31): the plan must take the direct kernels for that
    layer and still match the CPU convolution (crops of conv1_2's post-ReLU tap)."""
    import style_transfer as ST
    from st3d import vgg as V
    S, h = 2912, 20
    model = V.get_vgg(seed=0, device=dev)
    x = torch.rand((1, 3, S, S), generator=torch.Generator().manual_seed(0)).to(dev)
    c12 = ST.get_features(x, model, layers={"2": "c12"})["c12"]
    assert c12.shape == (1, 64, S, S)
    st = {k: v.double() for k, v in V.synthetic_state(0).items()}
    net = lambda t: torch.relu(F.conv2d(torch.relu(F.conv2d(t, st["0.weight"], st["0.bias"], padding=1)), st["2.weight"],
                                        st["2.bias"], padding=1))
    for y0, x0 in ((0, 0), (S - h, S - h), (1500, 24), (S - h, 1200)):
        ref = _crop_ref(x, y0, x0, h, 2, net)
        got = c12[:, :, y0:y0 + h, x0:x0 + h].cpu().double()
        assert float((got - ref).abs().max()) <= 5e-5 * float(ref.abs().max()), (y0, x0)
    del model


def _random_soup(seed, F, with_traps=True):
    """Random triangle soup in NDC: sizes from sub-pixel to half the screen, depths on both sides of the image plane and
    of the near clipping plane, plus the traps a mesh file never guarantees against: zero-area faces, exact duplicates
    (depth ties), vertices exactly on pixel centres (edge-function zeros), faces wholly off screen."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1.1, 1.1, (F, 1, 2))
    size = 10 ** rng.uniform(-2.5, -0.2, (F, 1, 1))
    xy = c + size * rng.standard_normal((F, 3, 2))
    z = rng.uniform(0.3, 3.0, (F, 1)) + rng.uniform(-0.4, 0.4, (F, 3)) * (rng.random((F, 1)) < 0.5)
    z[rng.random(F) < 0.03] *= -1.0                                   # behind the camera
    v = np.concatenate([xy, z[..., None]], 2).astype(np.float32)
    if with_traps:
        v[1] = v[0]                                                   # duplicate face: equal depth everywhere
        v[2, 2] = v[2, 0]                                             # two coincident vertices: zero area
        v[3, :, :2] = np.float32(1 - (2 * 10 + 1) / 64)               # all three vertices on one pixel centre (S = 64)
        v[4, 0, :2] = [np.float32(1 - (2 * 20 + 1) / 64), np.float32(1 - (2 * 30 + 1) / 64)]    # a vertex on a pixel centre
        v[5, :, 0] += 5.0                                             # off screen
    verts = v.reshape(-1, 3)
    faces = np.arange(3 * F, dtype=np.int32).reshape(F, 3)
    return verts, faces


@pytest.mark.parametrize("seed,F,S", [(0, 300, 64), (1, 2500, 64), (2, 700, 100), (3, 50, 17)])
def test_rasterisers_match_oracle_on_random_triangle_soups(dev, ops, seed, F, S):
    """Bit-exact fragments against the C oracle on adversarial soups, for the specialised K = 1 kernel and for the general
    kernel under every combination of the settings it implements (K, blur + barycentric clipping, back-face culling,
    perspective correction, near-plane clipping)."""
    from oracle import render_ref as rr
    verts, faces = _random_soup(seed, F)
    ndc = torch.from_numpy(verts)[None].to(dev)
    fd = torch.from_numpy(faces).to(dev)
    hard = ops.raster_fwd(ndc, fd, S)
    ref = rr.rasterize(verts, faces, S, 0.0, 8)
    for g, r in zip(hard, ref):
        np.testing.assert_array_equal(g[0].cpu().numpy(), r)
    assert (ref[0] >= 0).mean() > 0.03
    combos = [(1, 0.0, False, True, None), (3, 0.0, True, True, None), (4, 1e-3, False, False, None), (2, 0.0, False, True, 0.5),
              (8, 2e-3, True, True, 0.5), (3, 5e-4, False, False, 0.7)]
    for K, blur, cull, persp, zc in combos:
        got = ops.raster_soft_fwd(ndc, fd, S, K, blur, blur > 0, cull_backfaces=cull, perspective_correct=persp, z_clip=zc)
        want = rr.rasterize_k(verts, faces, S, K, blur, blur > 0, nthreads=8, cull_backfaces=cull, perspective_correct=persp,
                              z_clip=zc, return_slots=zc is not None)
        for g, r in zip(got, want):
            np.testing.assert_array_equal(g[0].cpu().numpy(), r, err_msg=str((K, blur, cull, persp, zc)))


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 64, 64, 32, 64), (1, 128, 256, 16, 24), (1, 64, 128, 160, 96)])
def test_producer_gated_backward_chain_is_bitwise_the_consumer_gated_one(dev, ops, N, Cin, Cout, H, W):
    """st3d_wino_dgrad_chain moves the ReLU gates to the producer of each gradient.  Zeros are zeros: every variant must
    be BITWISE what the consumer-gated launches (st3d_wino_dgrad / st3d_wino_dgrad_unpool, themselves checked against
    autograd above) return on the same data.
      input side : gy pre-multiplied by its gate + act=None            == gy gated by act in the kernel
                   pooled gy pre-gated by pooled > 0 + pooled=None     == gated by pooled in the kernel
      output side: out_gate                                            == the plain launch, then zeroed where gate <= 0"""
    g = torch.Generator().manual_seed(H + W + Cin)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dev)
    _, ud = ops.wino_pack(w)
    act = torch.relu(torch.randn(N, Cout, H, W, generator=g)).to(dev)           # about half the gates closed
    gy = torch.randn(N, Cout, H, W, generator=g).to(dev)
    out_gate = torch.relu(torch.randn(N, Cin, H, W, generator=g)).to(dev)
    ref = ops.wino_dgrad(gy, act, ud, Cin)
    pre = torch.where(act > 0, gy, torch.zeros_like(gy))
    assert torch.equal(ops.wino_dgrad_chain(pre, ud, Cin), ref)
    assert torch.equal(ops.wino_dgrad_chain(gy, ud, Cin, act=act), ref)
    want = torch.where(out_gate > 0, ref, torch.zeros_like(ref))
    assert torch.equal(ops.wino_dgrad_chain(gy, ud, Cin, act=act, out_gate=out_gate), want)
    assert torch.equal(ops.wino_dgrad_chain(pre, ud, Cin, out_gate=out_gate), want)
    # + the content-loss term of the output tensor (what st3d_axpy_diff adds), before the gate
    target = torch.randn(N, Cin, H, W, generator=g).to(dev)
    want3 = torch.where(out_gate > 0, ref + 0.25 * (out_gate - target), torch.zeros_like(ref))
    assert torch.equal(ops.wino_dgrad_chain(pre, ud, Cin, out_gate=out_gate, add_target=target, add_coef=0.25), want3)
    # pooled input
    full = torch.relu(torch.randn(N, Cout, H, W, generator=g)).to(dev)
    pooled, idx = ops.maxpool2x2(full)
    gp = torch.randn(N, Cout, H // 2, W // 2, generator=g).to(dev)
    ref2 = ops.wino_dgrad_unpool(gp, idx, pooled, ud, Cin)
    gp_pre = torch.where(pooled > 0, gp, torch.zeros_like(gp))
    assert torch.equal(ops.wino_dgrad_chain(gp, ud, Cin, pool_idx=idx, pooled=pooled), ref2)
    assert torch.equal(ops.wino_dgrad_chain(gp_pre, ud, Cin, pool_idx=idx), ref2)
    want2 = torch.where(out_gate > 0, ref2, torch.zeros_like(ref2))
    assert torch.equal(ops.wino_dgrad_chain(gp_pre, ud, Cin, pool_idx=idx, out_gate=out_gate), want2)


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 64, 64), (1, 128, 40, 40), (2, 256, 16, 16), (1, 512, 8, 8), (1, 512, 32, 32),
                                     (1, 128, 256, 256), (1, 96, 20, 12)])
def test_gram_bwd_gated_zeroes_exactly_the_closed_gates(dev, ops, B, C, H, W):
    """st3d_gram_bwd_gated == st3d_gram_bwd followed by `where(feat > 0, ., 0)`, bitwise, with and without accumulation
    (the gate bits come from the operand tile in LDS; every tile shape the launcher picks is covered)."""
    g = torch.Generator().manual_seed(C + H)
    feat = torch.relu(torch.randn(B, C, H, W, generator=g)).to(dev)
    D = torch.randn(B, C, C, generator=g)
    D = (0.5 * (D + D.transpose(1, 2))).to(dev).contiguous()
    base = torch.randn(B, C, H, W, generator=g).to(dev)
    zero = torch.zeros_like(feat)
    assert torch.equal(ops.gram_bwd(D, feat, 0.3, gated=True), torch.where(feat > 0, ops.gram_bwd(D, feat, 0.3), zero))
    assert torch.equal(ops.gram_bwd(D, feat, 0.3, out=base.clone(), gated=True),
                       torch.where(feat > 0, ops.gram_bwd(D, feat, 0.3, out=base.clone()), zero))


def test_plan_backward_with_producer_side_gates_is_bitwise_the_plain_one(dev, golden_dir, monkeypatch):
    """The loss plan's backward with the gates at the producers + the fused relu1_1/conv1_1 kernel (defaults) against
    ST3D_PREGATE=0 (bitwise: zeros are zeros) and against ST3D_TAP0_FUSED=0 (another summation order: 1e-5).  The bitwise
    comparison holds between runs on the SAME conv kernels: the un-gated chain has no F(4x4,3x3) path, so it is made with
    ST3D_WINO43=0; the default plan (F(4x4,3x3) at conv1_2 here) is compared with it at the kernels' tolerance."""
    from st3d import vgg as V
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 64, 64, generator=g).to(dev)
    con = torch.rand(2, 3, 64, 64, generator=g).to(dev)
    sty = torch.rand(1, 3, 64, 64, generator=g).to(dev)

    def run(env):
        for k in ("ST3D_PREGATE", "ST3D_TAP0_FUSED", "ST3D_WINO43"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = V.Vgg19Features(V.synthetic_state(0), device=dev)        # the switches are read when the VGG handle is made
        plan = net.plan(2, 64)
        plan.set_content(con)
        plan.set_style(sty, 2)
        loss, grad = plan.loss(x, 1e6, 1.0)
        return loss.clone(), grad.clone()
    l0, g0 = run({"ST3D_WINO43": "0"})
    l1, g1 = run({"ST3D_WINO43": "0", "ST3D_PREGATE": "0"})
    l2, g2 = run({"ST3D_WINO43": "0", "ST3D_PREGATE": "0", "ST3D_TAP0_FUSED": "0"})
    assert torch.equal(l0, l1) and torch.equal(g0, g1)
    assert torch.equal(l0, l2)
    _scale_close(g0, g2, 1e-5, "fused vs unfused bottom of the backward")
    l3, g3 = run({})
    assert float((l3 - l0).abs().max() / l0.abs().max()) < 1e-5
    _scale_close(g3, g0, 5e-5, "F(4x4,3x3) vs F(2x2,3x3) plan")
