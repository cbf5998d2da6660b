#!/usr/bin/env python3
"""Generates the golden fixtures in this directory.  Runs ONLY in the build container
(needs /root/reference); the fixtures it writes are committed, the reference never ships.

What it does
  * imports the reference's own ``style_transfer.py`` and ``losses.py`` from
    /root/reference (``losses.py`` needs three names from the absent ``pytorch3d.loss``;
    a dummy module exposing them is pre-registered -- they are only reached by the
    'mesh'/'both' branches, which are therefore NOT covered by these vectors),
  * drives them with seeded inputs and a VGG-19-shaped ``nn.Sequential`` with seeded
    weights (``oracle.perceptual_ref.make_vgg19_features`` -- torchvision's pretrained
    weights need a download and are unavailable offline),
  * stores inputs + the reference's outputs as small ``.npz`` files (G1-G4 of SURVEY.md 8c),
  * converts the reference's DATA files used by the benchmark configs (cow mesh + texture,
    Style_1.jpg at 512x512) into ``assets_*.npz`` so the GPU box, which has no
    /root/reference, can run configs 1/2 on the real mesh.

Usage:  python tests/golden/make_golden.py          (G1-G4 + assets, ~1 min)
        python tests/golden/make_golden.py g5       (G5: 200 reference iterations at 256^2, ~5 min)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "2d-to-3d-style-transfer_amd"))

from oracle import perceptual_ref as P  # noqa: E402


def import_reference():
    stub = types.ModuleType("pytorch3d.loss")
    for n in ("mesh_edge_loss", "mesh_laplacian_smoothing", "mesh_normal_consistency"):
        setattr(stub, n, lambda *a, **k: (_ for _ in ()).throw(RuntimeError("pytorch3d absent")))
    sys.modules.setdefault("pytorch3d", types.ModuleType("pytorch3d"))
    sys.modules["pytorch3d.loss"] = stub
    sys.path.insert(0, REF)
    mods = {}
    for name in ("style_transfer", "losses"):
        spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        if name == "style_transfer":
            sys.modules["style_transfer"] = m          # losses.py does `from style_transfer import *`
        spec.loader.exec_module(m)
        mods[name] = m
    sys.path.remove(REF)
    return mods["style_transfer"], mods["losses"]


def main():
    torch.set_num_threads(8)
    ST, L = import_reference()
    out = {}

    # ---- G1: gram_matrix (style_transfer.py:31-35)
    torch.manual_seed(0)
    x = torch.randn(2, 4, 3, 5)
    g = ST.gram_matrix(x)
    np.savez(os.path.join(HERE, "g1_gram.npz"), x=x.numpy(), gram=g.numpy(), gram_sum=np.float32(g.sum().item()))
    out["g1 gram.sum"] = g.sum().item()

    # ---- G2: tap semantics (in-place ReLU => taps are post-ReLU), style_transfer.py:10-27
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(inplace=True),
                              torch.nn.Conv2d(4, 4, 3, padding=1), torch.nn.ReLU(inplace=True))
    xin = torch.randn(1, 3, 8, 8)
    feats = ST.get_features(xin, net, layers={"0": "a", "2": "b"})
    np.savez(os.path.join(HERE, "g2_taps.npz"), x=xin.numpy(),
             w0=net[0].weight.detach().numpy(), b0=net[0].bias.detach().numpy(),
             w2=net[2].weight.detach().numpy(), b2=net[2].bias.detach().numpy(),
             a=feats["a"].detach().numpy(), b=feats["b"].detach().numpy())
    out["g2 min a/b"] = (feats["a"].min().item(), feats["b"].min().item())

    # ---- G3: perceptual loss + gradient through the reference's losses.py on the seeded VGG
    vgg = P.make_vgg19_features(seed=0)
    torch.manual_seed(0)
    cur = torch.rand(2, 3, 64, 64)
    con = torch.rand(2, 3, 64, 64)
    sty = torch.rand(1, 3, 64, 64).repeat(2, 1, 1, 1)
    masks = (torch.rand(2, 1, 64, 64) > 0.3).float()
    cur_g = cur.clone().requires_grad_(True)
    loss = L.compute_perceptual_loss(cur_g, con, sty, vgg)
    loss.backward()
    loss2 = L.compute_second_approach_loss(cur, con, sty, vgg, 1e6, 1, None, None, None, {}, "texture")
    feats = ST.get_features(cur, vgg)
    fstats = {}
    for k, v in feats.items():
        fstats[f"feat_{k}_sum"] = np.float64(v.double().sum().item())
        fstats[f"feat_{k}_abs"] = np.float64(v.double().abs().sum().item())
        fstats[f"feat_{k}_min"] = np.float32(v.min().item())
        gm = ST.gram_matrix(v)
        fstats[f"gram_{k}_sum"] = np.float64(gm.double().sum().item())
        if v.shape[1] <= 128:
            fstats[f"gram_{k}"] = gm.numpy()
    fstats["feat_conv5_1"] = feats["conv5_1"].numpy()
    fstats["feat_conv1_1_img0_ch0"] = feats["conv1_1"][0, 0].numpy()
    fa = L.compute_first_approach_loss(cur, masks, con, None, None, None, {}, "texture")
    tv = L.compute_tv_loss(cur, masks)

    class _M:                               # rgb_range_loss only touches mesh.textures.maps_padded()
        class textures:
            @staticmethod
            def maps_padded():
                return (cur * 3 - 1).permute(0, 2, 3, 1)
    rr = L.rgb_range_loss(_M)
    np.savez_compressed(os.path.join(HERE, "g3_perceptual.npz"), cur=cur.numpy(), con=con.numpy(), sty=sty.numpy(),
                        masks=masks.numpy(), loss=np.float32(loss.item()), loss_second=np.float32(loss2.item()),
                        grad=cur_g.grad.numpy(), grad_l2=np.float32(cur_g.grad.norm().item()),
                        first_loss=np.float32(fa.item()), tv_loss=np.float32(tv.item()),
                        rgb_range=np.float32(rr.item()), **fstats)
    out["g3 loss"] = loss.item()
    out["g3 |grad|"] = cur_g.grad.norm().item()
    out["g3 first/tv/rr"] = (fa.item(), tv.item(), rr.item())

    # ---- G3b: style/content weights other than the defaults, batch of 1, non-square-free 96x96
    torch.manual_seed(3)
    cur1 = torch.rand(1, 3, 96, 96)
    con1 = torch.rand(1, 3, 96, 96)
    sty1 = torch.rand(1, 3, 96, 96)
    c1 = cur1.clone().requires_grad_(True)
    l1 = L.compute_perceptual_loss(c1, con1, sty1, vgg, style_weight=2.5e5, content_weight=3.0)
    l1.backward()
    np.savez_compressed(os.path.join(HERE, "g3b_perceptual_96.npz"), cur=cur1.numpy(), con=con1.numpy(),
                        sty=sty1.numpy(), loss=np.float32(l1.item()), grad=c1.grad.numpy(),
                        style_weight=np.float32(2.5e5), content_weight=np.float32(3.0))
    out["g3b loss"] = l1.item()

    # ---- G4: style_transfer() trajectory (style_transfer.py:38-85): Adam + the whole VGG loop
    ST.tqdm = lambda it, **k: it
    torch.manual_seed(4)
    init = torch.rand(2, 3, 32, 32)
    con4 = torch.rand(2, 3, 32, 32)
    sty4 = torch.rand(1, 3, 32, 32).repeat(2, 1, 1, 1)
    res = ST.style_transfer(init, con4, sty4, vgg, steps=6, style_weight=1e6, content_weight=1, lr=0.01)
    np.savez_compressed(os.path.join(HERE, "g4_style_transfer.npz"), init=init.numpy(), con=con4.numpy(),
                        sty=sty4.numpy(), result=res.detach().numpy(), steps=6, lr=np.float32(0.01))
    out["g4 result mean"] = res.mean().item()

    # ---- assets: the reference's DATA files used by configs 1/2 (cow mesh + Style_1)
    from st3d import io as stio
    from PIL import Image
    verts, faces, aux = stio.load_obj(os.path.join(REF, "objects/cow_mesh/cow.obj"))
    tex = list(aux.texture_images.values())[0]
    np.savez_compressed(os.path.join(HERE, "assets_cow_mesh.npz"),
                        verts=verts.numpy(), faces=faces.verts_idx.numpy().astype(np.int32),
                        verts_uvs=aux.verts_uvs.numpy(), faces_uvs=faces.textures_idx.numpy().astype(np.int32),
                        texture_u8=(tex * 255.0).round().to(torch.uint8).numpy())
    out["cow V/F/VT/tex"] = (tuple(verts.shape), tuple(faces.verts_idx.shape), tuple(aux.verts_uvs.shape), tuple(tex.shape))
    # load_as_tensor (utils.py:34-44): PIL RGB -> Resize((S,S)) bilinear(antialias) -> /255
    im = Image.open(os.path.join(REF, "imgs/Style_1.jpg")).convert("RGB").resize((512, 512), Image.BILINEAR)
    np.savez_compressed(os.path.join(HERE, "assets_style1_512.npz"), rgb_u8=np.asarray(im, dtype=np.uint8))

    # ---- assets for BASELINE configs[3] / configs[4] (teapot: verts/faces only, it ships without UVs or a texture;
    # bob: 5344 quads -> 10688 fan-triangulated faces, its 2048^2 texture stored at 512^2 -- every run resizes it to
    # --size anyway, second_approach.py:84-94) and the other style images (Style_3/5 are RGBA: .convert('RGB'))
    verts, faces, aux = stio.load_obj(os.path.join(REF, "objects/bob_mesh/bob.obj"))
    tex = list(aux.texture_images.values())[0]
    t512 = torch.nn.functional.interpolate(tex.permute(2, 0, 1)[None], size=512, mode="bilinear", align_corners=False,
                                           antialias=True)[0].permute(1, 2, 0)
    np.savez_compressed(os.path.join(HERE, "assets_bob_mesh.npz"),
                        verts=verts.numpy(), faces=faces.verts_idx.numpy().astype(np.int32),
                        verts_uvs=aux.verts_uvs.numpy(), faces_uvs=faces.textures_idx.numpy().astype(np.int32),
                        texture_u8=(t512.clamp(0, 1) * 255.0).round().to(torch.uint8).numpy())
    out["bob V/F/VT/tex"] = (tuple(verts.shape), tuple(faces.verts_idx.shape), tuple(aux.verts_uvs.shape), tuple(tex.shape))
    verts, faces, aux = stio.load_obj(os.path.join(REF, "objects/teapot_mesh/teapot.obj"))
    assert aux.verts_uvs is None and faces.textures_idx is None and not aux.texture_images
    np.savez_compressed(os.path.join(HERE, "assets_teapot_mesh.npz"), verts=verts.numpy(),
                        faces=faces.verts_idx.numpy().astype(np.int32))
    out["teapot V/F"] = (tuple(verts.shape), tuple(faces.verts_idx.shape))
    for k, fn in ((3, "Style_3.png"), (4, "Style_4.jpeg"), (5, "Style_5.png")):
        im = Image.open(os.path.join(REF, "imgs", fn)).convert("RGB").resize((512, 512), Image.BILINEAR)
        np.savez_compressed(os.path.join(HERE, f"assets_style{k}_512.npz"), rgb_u8=np.asarray(im, dtype=np.uint8))

    for k, v in out.items():
        print(k, v)


def g5_config1():
    """G5 = BASELINE.json configs[0]: the reference's own style_transfer() (style_transfer.py:38-85)
    on cow_mesh + Style_1 at 256x256, 4 views, 200 iterations on the CPU, lr 0.01 (first_approach.py's
    --style_transfer_lr default), started from the content images (--style_transfer_init content).  Inputs are stored
    as uint8 so both sides start from bit-identical pixels: the content images are the CPU oracle's renders of the
    cow (4 seeded random cameras, seed 0) quantised to 8 bits, the style image is Style_1 through PIL at 256x256.
    The reference's per-step `total_loss` is captured by watching Tensor.backward (the function does not return it).
    ~5 minutes on 8 cores."""
    import time
    from PIL import Image
    from oracle import render_ref as RR
    torch.set_num_threads(8)
    ST, L = import_reference()
    ST.tqdm = lambda it, **k: it
    S, B, steps, lr = 256, 4, 200, 0.01
    cow = np.load(os.path.join(HERE, "assets_cow_mesh.npz"))
    tex = torch.from_numpy(cow["texture_u8"]).float().div(255.0)
    tex = torch.nn.functional.interpolate(tex.permute(2, 0, 1)[None], size=S, mode="bilinear",
                                          align_corners=False)[0].permute(1, 2, 0).contiguous().numpy()
    g = torch.Generator().manual_seed(0)
    elev, azim = RR.random_camera_angles(B, lambda k: torch.rand(k, generator=g).numpy())
    R, T = RR.look_at_view_transform(2.10, elev, azim, at=(0, 0.10, 0.25))
    imgs, _, _ = RR.render_views(cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"], tex, R, T, S, 8)
    content_u8 = (torch.from_numpy(imgs).clamp(0, 1) * 255.0).round().to(torch.uint8)
    content = content_u8.float().div(255.0)
    im = Image.open(os.path.join(REF, "imgs/Style_1.jpg")).convert("RGB").resize((S, S), Image.BILINEAR)
    style_u8 = torch.from_numpy(np.asarray(im, dtype=np.uint8)).permute(2, 0, 1).contiguous()
    style = style_u8.float().div(255.0)[None].repeat(B, 1, 1, 1)
    vgg = P.make_vgg19_features(seed=0)
    losses = []
    real_backward = torch.Tensor.backward

    def spy(self, *a, **k):
        if self.dim() == 0:
            losses.append(float(self.detach()))
        return real_backward(self, *a, **k)
    torch.Tensor.backward = spy
    t0 = time.time()
    try:
        res = ST.style_transfer(content, content, style, vgg, steps=steps, style_weight=1e6, content_weight=1, lr=lr)
    finally:
        torch.Tensor.backward = real_backward
    final = L.compute_perceptual_loss(res.detach(), content, style, vgg)
    print("G5: %d steps in %.0f s; loss %.6g -> %.6g (after the last update %.6g)" % (steps, time.time() - t0, losses[0],
                                                                                  losses[-1], float(final)))
    np.savez_compressed(os.path.join(HERE, "g5_config1_style_transfer.npz"), content_u8=content_u8.numpy(),
                        style_u8=style_u8.numpy(), R=R, T=T, steps=steps, lr=np.float32(lr),
                        losses=np.asarray(losses, np.float64), final_loss=np.float64(float(final)),
                        result_f16=res.detach().numpy().astype(np.float16))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g5":
        g5_config1()
    else:
        main()
