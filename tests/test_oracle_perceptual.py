"""Pins the oracle's VGG / Gram / loss restatement (oracle/perceptual_ref.py) against golden
vectors produced by the REFERENCE's own style_transfer.py / losses.py (tests/golden/make_golden.py).
CPU only.  Both sides are torch-CPU fp32 built from the same ops, so tolerances are tight."""
import os

import numpy as np
import pytest
import torch

from oracle import perceptual_ref as P


@pytest.fixture(scope="module")
def vgg():
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    return P.make_vgg19_features(seed=0)


def test_vgg_layout_matches_torchvision_numbering(vgg):
    mods = list(vgg._modules.items())
    assert len(mods) == 37
    convs = [int(k) for k, m in mods if isinstance(m, torch.nn.Conv2d)]
    assert convs == [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32, 34]
    pools = [int(k) for k, m in mods if isinstance(m, torch.nn.MaxPool2d)]
    assert pools == [4, 9, 18, 27, 36]
    assert all(m.inplace for _, m in mods if isinstance(m, torch.nn.ReLU))
    assert sum(p.numel() for p in vgg.parameters()) == 20024384


def test_g1_gram(golden_dir):
    d = np.load(os.path.join(golden_dir, "g1_gram.npz"))
    g = P.gram_ref(torch.from_numpy(d["x"]))
    np.testing.assert_allclose(g.numpy(), d["gram"], rtol=1e-6, atol=1e-6)
    assert abs(float(g.sum()) - 209.80531311035156) < 1e-3       # value observed in SURVEY.md 8c (G1)


def test_g2_taps_are_post_relu(golden_dir):
    """The reference's feature dict holds POST-ReLU activations (in-place ReLU aliasing)."""
    d = np.load(os.path.join(golden_dir, "g2_taps.npz"))
    assert d["a"].min() == 0.0 and d["b"].min() == 0.0
    x = torch.from_numpy(d["x"])
    a = torch.relu(torch.nn.functional.conv2d(x, torch.from_numpy(d["w0"]), torch.from_numpy(d["b0"]), padding=1))
    b = torch.relu(torch.nn.functional.conv2d(a, torch.from_numpy(d["w2"]), torch.from_numpy(d["b2"]), padding=1))
    np.testing.assert_allclose(a.numpy(), d["a"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(b.numpy(), d["b"], rtol=1e-6, atol=1e-6)


def test_g3_perceptual_loss_and_gradient(golden_dir, vgg):
    d = np.load(os.path.join(golden_dir, "g3_perceptual.npz"))
    cur = torch.from_numpy(d["cur"]).requires_grad_(True)
    loss = P.perceptual_loss_ref(cur, torch.from_numpy(d["con"]), torch.from_numpy(d["sty"]), vgg)
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) <= 1e-5 * float(d["loss"])
    assert abs(float(d["loss_second"]) - float(d["loss"])) <= 1e-6 * float(d["loss"])    # 'texture' adds nothing
    rel = (cur.grad - torch.from_numpy(d["grad"])).norm() / torch.from_numpy(d["grad"]).norm()
    assert float(rel) <= 1e-5
    feats = P.get_features_ref(cur.detach(), vgg)
    for k, f in feats.items():
        assert float(f.min()) == float(d[f"feat_{k}_min"]) == 0.0
        assert abs(float(f.double().sum()) - float(d[f"feat_{k}_sum"])) <= 1e-5 * abs(float(d[f"feat_{k}_sum"]))
        g = P.gram_ref(f)
        assert abs(float(g.double().sum()) - float(d[f"gram_{k}_sum"])) <= 1e-5 * abs(float(d[f"gram_{k}_sum"]))
    np.testing.assert_allclose(feats["conv5_1"].numpy(), d["feat_conv5_1"], rtol=1e-5, atol=1e-6)


def test_g3_auxiliary_losses(golden_dir):
    d = np.load(os.path.join(golden_dir, "g3_perceptual.npz"))
    cur, con, masks = (torch.from_numpy(d[k]) for k in ("cur", "con", "masks"))
    assert abs(float(P.first_approach_loss_texture_ref(cur, masks, con)) - float(d["first_loss"])) <= 1e-7
    assert abs(float(P.tv_loss_ref(cur, masks)) - float(d["tv_loss"])) <= 1e-6
    assert abs(float(P.rgb_range_loss_ref((cur * 3 - 1).permute(0, 2, 3, 1))) - float(d["rgb_range"])) <= 1e-2


def test_g3b_weights_and_96px(golden_dir, vgg):
    d = np.load(os.path.join(golden_dir, "g3b_perceptual_96.npz"))
    cur = torch.from_numpy(d["cur"]).requires_grad_(True)
    loss = P.perceptual_loss_ref(cur, torch.from_numpy(d["con"]), torch.from_numpy(d["sty"]), vgg,
                                 style_weight=float(d["style_weight"]), content_weight=float(d["content_weight"]))
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) <= 1e-5 * float(d["loss"])
    rel = (cur.grad - torch.from_numpy(d["grad"])).norm() / torch.from_numpy(d["grad"]).norm()
    assert float(rel) <= 1e-5


def test_g4_style_transfer_trajectory(golden_dir, vgg):
    """Six Adam steps of the reference's style_transfer() (targets once, Adam on the pixels)."""
    d = np.load(os.path.join(golden_dir, "g4_style_transfer.npz"))
    res, losses = P.style_transfer_ref(torch.from_numpy(d["init"]), torch.from_numpy(d["con"]), torch.from_numpy(d["sty"]),
                                       vgg, steps=int(d["steps"]), lr=float(d["lr"]))
    np.testing.assert_allclose(res.numpy(), d["result"], rtol=0, atol=2e-5)
    assert losses[-1] < losses[0]
