"""Host-side logic that needs no GPU: OBJ/MTL reader + writer, cameras, containers, view
sharding, CLI flag surface, error behaviour of the drop-in modules."""
import os

import numpy as np
import pytest
import torch

from oracle import render_ref as rr


def test_obj_reader_tokens_fan_triangulation_and_negative_indices(tmp_path):
    from st3d import io as stio
    from PIL import Image
    Image.fromarray(np.full((4, 4, 3), 128, np.uint8)).save(tmp_path / "t.png")
    (tmp_path / "m.mtl").write_text("newmtl mat\nmap_Kd t.png\nKd 1 1 1\n")
    (tmp_path / "m.obj").write_text(
        "mtllib m.mtl\nusemtl mat\n"
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\n"
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
        "vn 0 0 1\n"
        "f 1/1/1 2/2/1 3/3/1 4/4/1\n"        # quad -> fan (0,1,2),(0,2,3)
        "f -1/1 -5/2 -4/3\n")                  # negative = relative to the end
    verts, faces, aux = stio.load_obj(str(tmp_path / "m.obj"))
    assert verts.shape == (5, 3)
    assert faces.verts_idx.tolist() == [[0, 1, 2], [0, 2, 3], [4, 0, 1]]
    assert faces.textures_idx.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2]]
    assert aux.verts_uvs.shape == (4, 2)
    tex = list(aux.texture_images.values())[0]
    assert tex.shape == (4, 4, 3) and abs(float(tex[0, 0, 0]) - 128 / 255) < 1e-6
    # round trip through the writer
    stio.save_obj(str(tmp_path / "out.obj"), verts, faces.verts_idx, aux.verts_uvs, faces.textures_idx, tex)
    v2, f2, a2 = stio.load_obj(str(tmp_path / "out.obj"))
    assert torch.allclose(v2, verts) and f2.verts_idx.tolist() == faces.verts_idx.tolist()
    assert os.path.exists(tmp_path / "out.mtl") and os.path.exists(tmp_path / "out.png")


def test_obj_without_uvs_reports_none(tmp_path):
    """teapot-style faces `v//vn`: no UVs, and synthesize_uvs gives a usable set."""
    from st3d import io as stio
    (tmp_path / "t.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    verts, faces, aux = stio.load_obj(str(tmp_path / "t.obj"))
    assert faces.textures_idx is None and aux.verts_uvs is None and aux.texture_images is None
    uv = stio.synthesize_uvs(verts)
    assert uv.shape == (3, 2) and float(uv.min()) >= 0 and float(uv.max()) <= 1


def test_cameras_match_the_oracle_restatement():
    from st3d import render as R
    elev = torch.tensor([12.0, -40.0, 89.0])
    azim = torch.tensor([30.0, -150.0, 5.0])
    Rt, Tt = R.look_at_view_transform(dist=2.10, elev=elev, azim=azim, at=((0, 0.10, 0.25),))
    Rn, Tn = rr.look_at_view_transform(2.10, elev.numpy(), azim.numpy(), at=(0, 0.10, 0.25))
    np.testing.assert_allclose(Rt.numpy(), Rn, atol=2e-6)
    np.testing.assert_allclose(Tt.numpy(), Tn, atol=2e-6)
    m = R.RotateAxisAngle(37.0, axis="X").get_matrix()[..., :3, :3].squeeze(0)
    np.testing.assert_allclose(m.numpy(), rr.rotate_axis_angle(37.0, "X"), atol=1e-6)
    cams = R.FoVPerspectiveCameras(R=Rt, T=Tt)
    assert len(cams) == 3 and cams[1].R.shape == (1, 3, 3)
    Rj, Tj = R.join_cameras([cams[0], cams[2]])
    assert torch.equal(Rj, Rt[[0, 2]]) and torch.equal(Tj, Tt[[0, 2]])


def test_fixed_cameras_and_random_cameras_shapes():
    import utils as U
    U.device = torch.device("cpu")
    c = U.build_fixed_cameras(12, shuffle=False)
    Rn, Tn = rr.fixed_cameras(12)
    np.testing.assert_allclose(c.R.numpy(), Rn, atol=1e-6)
    np.testing.assert_allclose(c.T.numpy(), Tn, atol=1e-6)
    g = torch.Generator().manual_seed(0)
    c2 = U.build_random_cameras(5, generator=g)
    g = torch.Generator().manual_seed(0)
    elev, azim = rr.random_camera_angles(5, lambda k: torch.rand(k, generator=g).numpy())
    Rn, Tn = rr.look_at_view_transform(2.10, elev, azim, at=(0, 0.10, 0.25))
    np.testing.assert_allclose(c2.R.numpy(), Rn, atol=5e-6)
    np.testing.assert_allclose(c2.T.numpy(), Tn, atol=5e-6)


def test_containers_and_setup_optimizations_keys():
    import utils as U
    verts = torch.rand(5, 3)
    faces = torch.tensor([[0, 1, 2], [2, 3, 4]])
    mesh = U.build_mesh(torch.rand(1, 6, 2), torch.tensor([[[0, 1, 2], [3, 4, 5]]]), torch.rand(1, 8, 8, 3), verts, faces)
    assert mesh.verts_padded().shape == (1, 5, 3) and mesh.faces_packed().shape == (2, 3)
    assert mesh.textures.maps_padded().shape == (1, 8, 8, 3)
    with pytest.raises(ValueError):
        U.build_mesh(torch.rand(1, 6, 2), torch.tensor([[[0, 1, 9]]]), torch.rand(1, 8, 8, 3), verts, faces).textures.faces_uvs_i32()
    for target, leaves in (("texture", {"texture_map"}), ("mesh", {"verts"}), ("both", {"texture_map", "verts"})):
        out = U.setup_optimizations(target, mesh, 0.01)
        assert set(out) == {"optimizable_mesh", "optimizer", "texture_map", "verts", "faces", "verts_uvs", "faces_uvs"}
        assert {k for k in ("texture_map", "verts") if out[k].requires_grad} == leaves
        assert out["texture_map"].data_ptr() != mesh.textures.maps_padded().data_ptr()      # a clone is optimised
    fin = U.finalize_mesh(U.build_mesh(torch.rand(1, 6, 2), torch.tensor([[[0, 1, 2]]]), torch.rand(1, 4, 4, 3) * 3 - 1, verts, faces))
    t = fin.textures.maps_padded()
    assert float(t.min()) >= 0 and float(t.max()) <= 1 and not t.requires_grad


def test_apply_background_white_is_identity_and_tensor_to_image():
    import utils as U
    t = torch.rand(2, 3, 4, 4)
    assert U.apply_background(t, torch.ones(2, 1, 4, 4), background_type="white") is t
    img = U.tensor_to_image(torch.tensor([[[1.5]], [[0.5]], [[-1.0]]]))
    assert img.getpixel((0, 0)) == (255, 127, 0)


def test_shard_views_partitions_every_batch():
    from st3d.optim import shard_views
    for n in (1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            spans = [shard_views(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_cli_flags_and_defaults_match_the_reference():
    import first_approach as FA
    import second_approach as SA
    a = SA.build_parser().parse_args([])
    assert (a.n_views, a.epochs, a.size, a.batch_size, a.lr, a.style_weight, a.content_weight) == (6, 3000, 768, 4, 0.01, 1e6, 1.0)
    assert (a.content_background, a.current_background, a.optimization_target, a.main_loss_weight) == ("white", "white", "texture", 3.0)
    assert a.output_path == "/content/output_second" and a.resize_texture is True and a.randomize_views is True
    b = FA.build_parser().parse_args([])
    assert (b.n_mse_steps, b.n_style_transfer_steps, b.style_transfer_init, b.mse_lr, b.style_transfer_lr) == (100, 3000, "content", 0.01, 0.01)
    assert b.output_path == "/content/output_first"
    # type=bool gotcha preserved: any non-empty string is True
    assert SA.build_parser().parse_args(["--resize_texture", "False"]).resize_texture is True


def test_unknown_opt_type_raises_unbound_local_error():
    """losses.py:98,126 of the reference fall through to `return loss` with loss unbound."""
    import losses as L
    with pytest.raises(UnboundLocalError):
        L.compute_first_approach_loss(None, None, None, None, None, None, {}, "nonsense")
    with pytest.raises(UnboundLocalError):
        L.compute_second_approach_loss(None, None, None, None, 1.0, 1.0, None, None, None, {}, "nonsense")


def test_api_only_losses_oracle_matches_reference_goldens(golden_dir):
    """compute_tv_loss / rgb_range_loss (reference losses.py:48-65): the CPU restatement reproduces the values the
    reference itself produced (G3); the HIP versions are checked against both in test_gpu_api.py.  On CPU tensors
    the product functions refuse to run (no CPU fallback)."""
    import losses as L
    from oracle import perceptual_ref as P
    d = np.load(os.path.join(golden_dir, "g3_perceptual.npz"))
    cur, masks = torch.from_numpy(d["cur"]), torch.from_numpy(d["masks"])
    assert abs(float(P.tv_loss_ref(cur, masks)) - float(d["tv_loss"])) <= 1e-6
    assert abs(float(P.rgb_range_loss_ref((cur * 3 - 1).permute(0, 2, 3, 1))) - float(d["rgb_range"])) <= 1e-2
    with pytest.raises(RuntimeError):
        L.compute_tv_loss(cur, masks)


def test_raster_settings_and_blend_params_validation():
    """PyTorch3D defaults: clip_barycentric_coords follows blur_radius > 0; unsupported switches fail loudly."""
    from st3d.render import BlendParams, RasterizationSettings, uses_hard_path
    rs = RasterizationSettings(image_size=512, blur_radius=0.0, faces_per_pixel=1)       # first_approach.py:107
    assert rs.is_hard and not rs.clip_barycentric_coords and uses_hard_path(rs, None) and uses_hard_path(rs, BlendParams())
    soft = RasterizationSettings(image_size=(64, 64), blur_radius=1e-4, faces_per_pixel=8, bin_size=0)
    assert soft.image_size == 64 and soft.clip_barycentric_coords and not soft.is_hard
    assert not RasterizationSettings(blur_radius=1e-4, clip_barycentric_coords=False).clip_barycentric_coords
    assert not uses_hard_path(rs, BlendParams(sigma=1e-3)) and not uses_hard_path(rs, BlendParams(background_color=(0, 0, 0)))
    for general in (dict(cull_backfaces=True), dict(perspective_correct=False)):      # supported, on the general kernels
        assert not RasterizationSettings(**general).is_hard and not uses_hard_path(RasterizationSettings(**general), None)
    assert RasterizationSettings(perspective_correct=None).perspective_correct          # PyTorch3D: None -> True for perspective cameras
    for bad in (dict(faces_per_pixel=0), dict(faces_per_pixel=9), dict(blur_radius=-1.0), dict(image_size=(64, 32))):
        with pytest.raises((NotImplementedError, ValueError)):
            RasterizationSettings(**bad)
    with pytest.raises(ValueError):
        BlendParams(gamma=0.0)


def test_bench_accounting_helpers():
    """bench.py's roofline arithmetic (no GPU): SURVEY.md 8d flop counts, the Gram forward's ISSUED fractions as gram.hip
    computes them (blocks on / above the diagonal only: 3/4, 10/16 of 32x32 blocks; 10/16, 36/64 of 64x64 wave tiles --
    the round-2 line priced single-tile layers at 100 % and printed a fraction of 1.08), and the guard that refuses to
    print any fraction above 1."""
    import bench
    S, B = 512, 8
    fwd = sum(bench.conv_alg_flops(m, S, 1) for m, *_ in bench.CONVS)
    assert abs(fwd / 1e9 - 189.35) < 0.05                                    # forward to conv5_1 per view
    gram = sum(bench.gram_alg_flops(m, S, 1) for m in bench.STYLE_TAPS)
    assert abs(gram / 1e9 - 9.13) < 0.01
    assert [bench.gram_fwd_issued_fraction(m) for m in (0, 5, 10, 19, 28)] == [0.75, 0.625, 0.625, 0.5625, 0.5625]
    # the one launch that holds all five layers (module tag 99): flop-weighted sums
    assert bench.gram_alg_flops(99, S, B) == sum(bench.gram_alg_flops(m, S, B) for m in bench.STYLE_TAPS)
    issued = bench.gram_fwd_issued_flops(99, S, B)
    assert abs(issued / 1e9 - (12.885 + 10.737 + 10.737 + 9.664 + 2.416)) < 0.01     # = the PMC-counted 46.44 GF of a step
    bench.check_fractions({"roofline": {"frac": 0.79}, "kernels": {"a": {"mfma_frac": 1.0, "hbm_frac": None}}, "layers": [{"mfma_frac": 0.5}]})
    with pytest.raises(AssertionError, match="mfma_frac"):
        bench.check_fractions({"layers": [{"mfma_frac": 1.08}]})
