"""CPU checks of oracle/loop_ref.py (the restated second_approach.py:145-190 step the GPU trajectory tests compare
with) and of the scene helpers the config tests use."""
import numpy as np
import torch

import _scenes as SC


def _tiny(target="texture", S=32, B=2, hoist=True):
    from oracle import loop_ref as LR
    cow = SC.load_asset("cow")
    R, T = SC.random_cameras(B, seed=0)
    torch.set_num_threads(4)
    return LR.SecondApproachRef(cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"], SC.texture_at(cow, S), R, T, S,
                                SC.style_at(1, S), target=target, lr=0.01 if target == "texture" else 0.001, nthreads=4,
                                hoist=hoist), cow


def test_texture_loop_reduces_the_loss_and_leaves_unseen_texels_alone():
    ref, _ = _tiny()
    losses = [ref.step() for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    g = ref.last["grad_texture"]
    unseen = (g == 0).all(axis=2)
    assert unseen.any() and (~unseen).any()
    assert np.array_equal(ref.tex[unseen], ref.tex0[unseen])          # notes.txt:12-16: no gradient, no Adam update
    assert np.abs(ref.tex - ref.tex0)[~unseen].max() <= 4 * 0.01 * 1.001           # |Adam step| <= lr (up to rounding)


def test_first_step_is_compute_perceptual_loss_of_the_renders():
    """step() = render -> perceptual_loss_ref (pinned to the reference by G3) -> shade_bwd -> Adam: its loss is the
    perceptual loss of its own renders, and with current == content the content term vanishes."""
    from oracle import perceptual_ref as P
    ref, _ = _tiny()
    loss, gtex, gverts = ref.loss_and_grads()
    cur = torch.from_numpy(ref.last["current"])
    direct = P.perceptual_loss_ref(cur, torch.from_numpy(ref.content()), ref.style.expand(2, -1, -1, -1), ref.model)
    assert loss == float(direct) and gverts is None
    assert ref.last["content_loss"] == 0.0 and ref.last["style_loss"] > 0


def test_both_target_moves_vertices_and_counts_the_regularisers():
    ref, cow = _tiny("both")
    l0 = ref.step()
    assert abs(l0 - (3.0 * ref.last["perceptual"] + ref.last["regs"])) <= 1e-6 * abs(l0)      # losses.py:117-124
    assert ref.last["regs"] > 0                        # edge / laplacian / normal terms of the undeformed cow
    assert np.abs(ref.verts - cow["verts"]).max() > 0 and np.abs(ref.verts - cow["verts"]).max() <= 0.001 + 1e-7


def test_midpoint_subdivision_keeps_the_rendered_surface():
    """tests/_scenes.subdivide (the bunny substitute of config 3) splits every triangle 1 -> 4 in position and UV space:
    the surface and its texture mapping are unchanged, so the renders must agree except on pixels whose centre sits
    within rounding of an edge."""
    from oracle import render_ref as rr
    cow = SC.load_asset("cow")
    v, f, uv, fuv = SC.subdivide(cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"])
    assert f.shape == (4 * 5856, 3) and v.shape[0] > cow["verts"].shape[0]
    tex = SC.texture_at(cow, 64)
    R, T = SC.random_cameras(1, seed=2)
    a, ma, _ = rr.render_views(cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"], tex, R, T, 96, 4)
    b, mb, _ = rr.render_views(v, f, uv, fuv, tex, R, T, 96, 4)
    same = ma == mb
    assert same.mean() > 0.995
    d = np.abs(a - b).max(axis=1, keepdims=True)
    assert np.median(d[(ma > 0) & same]) <= 1e-5 and (d > 1e-3).mean() <= 0.02


def test_synthesised_uvs_follow_the_documented_parametrisation():
    from oracle import loop_ref as LR
    from st3d import io
    tea = SC.load_asset("teapot")
    got = io.synthesize_uvs(torch.from_numpy(tea["verts"])).numpy()
    np.testing.assert_allclose(got, LR.synth_uvs_ref(tea["verts"]), atol=2e-6)
    assert got.min() >= 0 and got.max() <= 1
