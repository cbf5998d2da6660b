"""The general soft-rasteriser restatement (oracle/raster_ref.c: ref_rasterize_k, oracle/soft_ref.py).
PARITY UNPINNED w.r.t. PyTorch3D (absent): hand-computable cases + consistency with the K=1 path.  CPU only."""
import numpy as np
import torch

from oracle import render_ref as rr
from oracle import soft_ref as SR


def test_k1_blur0_equals_the_hard_rasteriser(cow):
    R, T = rr.look_at_view_transform(2.10, [15.0], [40.0], at=(0, 0.10, 0.25))
    ndc = rr.project_verts(cow["verts"], R[0], T[0])
    a = rr.rasterize(ndc, cow["faces"], 48, 0.0, 4)
    b = rr.rasterize_k(ndc, cow["faces"], 48, 1, 0.0, clip_bary=False, nthreads=4)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y[..., 0] if y.ndim == 3 else y[..., 0, :])


def test_k_faces_are_depth_sorted_and_blur_adds_a_halo():
    S = 16
    tri = [[-0.5, -0.5], [0.5, -0.5], [0.0, 0.5]]
    v = np.array([t + [3.0] for t in tri] + [t + [1.0] for t in tri] + [t + [2.0] for t in tri], np.float32)
    f = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8]], np.int32)
    p2f, zbuf, bary, dists = rr.rasterize_k(v, f, S, 2, 0.0)
    cov = p2f[..., 0] >= 0
    assert cov.sum() > 10
    assert np.all(p2f[cov][:, 0] == 1) and np.all(p2f[cov][:, 1] == 2)          # two nearest of three, ascending z
    assert np.allclose(zbuf[cov], [1.0, 2.0])
    p2f3, *_ = rr.rasterize_k(v, f, S, 4, 0.0)
    assert np.all(p2f3[cov][:, :3] == [1, 2, 0]) and np.all(p2f3[cov][:, 3] == -1)
    # blur: pixels just outside the triangle are kept with a positive distance < blur_radius
    blur = 0.02
    p2fb, zb, bb, db = rr.rasterize_k(v[3:6], np.array([[0, 1, 2]], np.int32), S, 1, blur)
    halo = (p2fb[..., 0] >= 0) & ~cov
    assert halo.sum() > 0 and np.all(db[..., 0][halo] > 0) and np.all(db[..., 0][halo] < blur)
    assert np.all(db[..., 0][cov] < 0)
    # clipped barycentrics are in [0,1] and sum to 1 on halo pixels
    assert np.all(bb[..., 0, :][halo] >= 0) and np.allclose(bb[..., 0, :][halo].sum(-1), 1.0, atol=1e-6)


def test_soft_blend_reduces_to_the_k1_formula(cow):
    """softmax_rgb_blend with K = 1 equals the (prob*texel + 1e-10)/(prob + 1e-10) of the hard path."""
    S, Tn = 40, 16
    R, T = rr.look_at_view_transform(2.10, [10.0], [25.0], at=(0, 0.10, 0.25))
    tex = np.random.default_rng(0).random((Tn, Tn, 3), dtype=np.float32)
    imgs, masks, frags = rr.render_views(cow["verts"], cow["faces"], cow["verts_uvs"], cow["faces_uvs"], tex, R, T, S, 4)
    p2f = torch.from_numpy(frags[0][0]).long()[..., None]
    rgb, alpha = SR.soft_render(torch.from_numpy(cow["verts"]), torch.from_numpy(R[0]), torch.from_numpy(T[0]),
                                torch.from_numpy(cow["faces"]).long(), p2f, torch.from_numpy(cow["verts_uvs"]),
                                torch.from_numpy(cow["faces_uvs"]).long(), torch.from_numpy(tex), S, False, 1e-4, 1e-4)
    np.testing.assert_allclose(rgb.numpy(), imgs[0], atol=3e-5)
    np.testing.assert_array_equal((alpha > 0).float().numpy(), masks[0][0])


def test_blend_weights_two_layers():
    """Two layers at equal depth and equal edge distance blend 50/50; gamma -> small makes the nearer one win."""
    colors = torch.tensor([[[[1.0, 0, 0], [0, 0, 1.0]]]])                       # (1,1,K=2,3)
    z = torch.tensor([[[2.0, 2.0]]])
    d = torch.tensor([[[-0.01, -0.01]]])
    m = torch.ones(1, 1, 2, dtype=torch.bool)
    rgb, a = SR.softmax_rgb_blend(colors, z, d, m, sigma=1e-2, gamma=1e-2)
    assert abs(float(rgb[0, 0, 0]) - float(rgb[0, 0, 2])) < 1e-6 and float(a) > 0.9
    z2 = torch.tensor([[[1.0, 2.0]]])
    rgb2, _ = SR.softmax_rgb_blend(colors, z2, d, m, sigma=1e-2, gamma=1e-4)
    assert float(rgb2[0, 0, 0]) > 0.99 and float(rgb2[0, 0, 2]) < 0.01


def test_cull_backfaces_sign_convention_and_closed_mesh(cow):
    """RasterizationSettings.cull_backfaces skips faces whose signed NDC area is negative.  (a) one triangle, both
    windings: exactly one of them survives culling, and it is the one whose vertices run counter-clockwise as SEEN in the
    image (NDC +X points left, SURVEY.md A.1, so that is the winding with positive edge-function area); (b) the cow
    -- a closed, outward-wound mesh viewed from outside -- loses only hidden layers: the nearest face of every pixel is
    front-facing, so the K = 1 result is identical with and without culling, while the second layer (the far side of the
    body) disappears."""
    S = 24
    v = np.array([[-0.5, -0.5, 2.0], [0.5, -0.5, 2.0], [0.0, 0.6, 2.0]], np.float32)
    ccw, cw = np.array([[0, 1, 2]], np.int32), np.array([[0, 2, 1]], np.int32)
    area = (v[2, 0] - v[0, 0]) * (v[1, 1] - v[0, 1]) - (v[2, 1] - v[0, 1]) * (v[1, 0] - v[0, 0])     # edge_fn(v2; v0, v1)
    assert area < 0                                                      # this order is "back-facing" by the rule
    keep_a = rr.rasterize_k(v, ccw, S, 1, 0.0, cull_backfaces=True)[0]
    keep_b = rr.rasterize_k(v, cw, S, 1, 0.0, cull_backfaces=True)[0]
    both = rr.rasterize_k(v, ccw, S, 1, 0.0)[0]
    assert (both >= 0).sum() > 30 and (keep_a >= 0).sum() == 0
    np.testing.assert_array_equal(keep_b, rr.rasterize_k(v, cw, S, 1, 0.0)[0])
    R, T = rr.look_at_view_transform(2.10, [20.0, -35.0], [30.0, 200.0], at=(0, 0.10, 0.25))
    for b in range(2):
        ndc = rr.project_verts(cow["verts"], R[b], T[b])
        off = rr.rasterize_k(ndc, cow["faces"], 64, 2, 0.0, nthreads=4)
        on = rr.rasterize_k(ndc, cow["faces"], 64, 2, 0.0, nthreads=4, cull_backfaces=True)
        for x, y in zip(off, on):
            np.testing.assert_array_equal(x[:, :, 0], y[:, :, 0])
        assert (off[0][..., 1] >= 0).sum() > 5 * max((on[0][..., 1] >= 0).sum(), 1)


def test_perspective_correct_off_gives_screen_space_barycentrics():
    """A triangle with vertex depths 1, 2, 4: at the pixel nearest its screen-space centroid the uncorrected barycentrics
    are the edge-function weights (~1/3 each, depth ~ their plain mean); with perspective correction they are weighted
    by 1/z_i (b_i ~ (1/z_i) / sum 1/z_j, depth = harmonic interpolation)."""
    S = 33
    v = np.array([[-0.6, -0.5, 1.0], [0.6, -0.5, 2.0], [0.0, 0.7, 4.0]], np.float32)
    f = np.array([[0, 2, 1]], np.int32)
    yi, xi = np.unravel_index(np.argmin([[(1 - (2 * x + 1) / S - 0.0) ** 2 + (1 - (2 * y + 1) / S + 0.1) ** 2 for x in range(S)]
                                         for y in range(S)]), (S, S))
    px, py = 1 - (2 * xi + 1) / S, 1 - (2 * yi + 1) / S
    p2f0, z0, b0, _ = rr.rasterize_k(v, f, S, 1, 0.0, perspective_correct=False)
    p2f1, z1, b1, _ = rr.rasterize_k(v, f, S, 1, 0.0, perspective_correct=True)
    np.testing.assert_array_equal(p2f0, p2f1)                            # coverage does not depend on the correction
    w = b0[yi, xi, 0]
    verts2 = v[f[0]]
    np.testing.assert_allclose((w[:, None] * verts2[:, :2]).sum(0), [px, py], atol=1e-6)      # screen-space interpolation
    assert abs(z0[yi, xi, 0] - float((w * verts2[:, 2]).sum())) < 1e-6
    want = w / verts2[:, 2]
    want /= want.sum()
    np.testing.assert_allclose(b1[yi, xi, 0], want, atol=1e-6)
    assert abs(z1[yi, xi, 0] - 1.0 / float((w / verts2[:, 2]).sum())) < 1e-5


def _straddling_scene(n_behind):
    """One big triangle through the near clipping plane z = 0.5 (PyTorch3D z_clip_value = znear / 2) seen by an identity
    camera: `n_behind` of its vertices have depth 0.25 < 0.5 (still in front of the image plane, so the unclipped
    rasteriser -- which knows nothing of the plane -- serves as the reference for barycentrics and depth)."""
    s = 1.7320508
    view = np.array([[-0.25, -0.2, 1.6], [0.3, -0.15, 1.1], [0.02, 0.12, 0.25]], np.float32)
    if n_behind == 2:
        view[1, 2] = 0.25
        view[1, :2] *= 0.2
    ndc = np.stack([s * view[:, 0] / view[:, 2], s * view[:, 1] / view[:, 2], view[:, 2]], 1).astype(np.float32)
    return ndc, np.array([[0, 1, 2]], np.int32)


def test_near_plane_clipping_keeps_the_part_in_front_and_the_original_barycentrics():
    """z_clip: the face is cut at depth 0.5 into a quadrilateral (two triangles) or a smaller triangle; covered pixels are
    exactly those of the uncut face whose interpolated depth is >= 0.5, pix_to_face names the ORIGINAL face and the
    converted barycentrics / depth equal the uncut face's (to rounding), with and without perspective correction; a face
    wholly behind the plane disappears."""
    S = 96
    for n_behind in (1, 2):
        ndc, f = _straddling_scene(n_behind)
        for persp in (True, False):
            full = rr.rasterize_k(ndc, f, S, 1, 0.0, perspective_correct=persp)
            p2f, z, b, d, slots = rr.rasterize_k(ndc, f, S, 1, 0.0, perspective_correct=persp, z_clip=0.5, return_slots=True)
            cov_full, cov = full[0][..., 0] >= 0, p2f[..., 0] >= 0
            assert cov.sum() > 50 and cov_full.sum() > cov.sum() + 20
            # depth along the uncut face AS THIS MODE INTERPOLATES IT decides what is in front of the plane: the cut points
            # are placed with view-space interpolation when perspective_correct, linearly in NDC otherwise
            zt = full[1][..., 0]
            want = cov_full & (zt >= 0.5)
            border = np.abs(zt - 0.5) < 0.02
            assert np.array_equal(cov[~border], want[~border])
            assert set(np.unique(slots[cov])) == ({0, 1} if n_behind == 1 else {0})
            assert np.all(p2f[cov] == 0)
            np.testing.assert_allclose(b[cov, 0], full[2][cov, 0], atol=2e-5)
            np.testing.assert_allclose(z[cov, 0], full[1][cov, 0], atol=2e-5)
            assert np.all(z[cov, 0] >= 0.5 - 1e-5) and np.all(d[cov, 0] < 0)
    ndc, f = _straddling_scene(1)
    ndc[:, 2] = 0.3                                                     # every vertex behind the plane
    assert (rr.rasterize_k(ndc, f, S, 1, 0.0, z_clip=0.5)[0] >= 0).sum() == 0
    assert (rr.rasterize_k(ndc, f, S, 1, 0.0)[0] >= 0).sum() > 0


def test_the_two_halves_of_a_clipped_quad_never_share_a_pixel():
    """With blur the halo of t1 overlaps t2 along their shared diagonal; PyTorch3D keeps, per pixel, only the half whose
    edge is nearer -- so a face appears at most once in a pixel's K list, as without clipping."""
    S = 64
    ndc, f = _straddling_scene(1)
    p2f, z, b, d, slots = rr.rasterize_k(ndc, f, S, 3, 4e-3, z_clip=0.5, return_slots=True)
    assert ((slots == 0).any(-1) & (slots == 1).any(-1)).sum() == 0
    assert (slots[..., 1] >= 0).sum() == 0                              # one face, one layer
    both = (slots[..., 0] == 0).sum() > 10 and (slots[..., 0] == 1).sum() > 10
    assert both
    nb = rr.rasterize_k(ndc, f, S, 3, 4e-3)                             # the unclipped face covers the clipped one's pixels
    assert np.all(nb[0][..., 0][p2f[..., 0] >= 0] == 0)


def test_clip_face_restatements_agree():
    """oracle/soft_ref.py:clip_face (torch, differentiable) against the C restatement, all cases, both interpolations."""
    for n_behind in (1, 2):
        ndc, f = _straddling_scene(n_behind)
        for persp in (True, False):
            S = 48
            p2f, z, b, d, slots = rr.rasterize_k(ndc, f, S, 1, 0.0, perspective_correct=persp, z_clip=0.5, return_slots=True)
            bary, pz, sd, mask = SR.clipped_geometry(torch.from_numpy(ndc).double(), torch.from_numpy(f).long(),
                                                     torch.from_numpy(slots).long(), S, False, persp, 0.5)
            m = mask.numpy()
            np.testing.assert_allclose(bary.numpy()[m], b[m], atol=3e-6)
            np.testing.assert_allclose(pz.numpy()[m], z[m], atol=3e-6)
            np.testing.assert_allclose(sd.numpy()[m], d[m], atol=1e-6)
