"""N > 1 rehearsal on ONE GPU: two ranks (gloo collectives, both on cuda:0 -- RCCL itself needs one card per rank, the
driver measures that on an 8-GPU node) run the drop-in CLIs with the view batch sharded, and must reproduce the 1-rank
run: logged losses, final vertices, final texture.  Covers the scaling rules of SURVEY.md 8e on the real kernels:
image terms divided by the GLOBAL batch, view-independent mesh regularisers counted once, uneven shards (3 views over
2 ranks), a rank without views, cameras broadcast from rank 0."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "2d-to-3d-style-transfer_amd")


def _assets(tmp, cow, golden_dir):
    from PIL import Image
    from st3d import io as stio
    tex = torch.from_numpy(cow["texture_u8"][::16, ::16].copy()).float() / 255
    obj = os.path.join(tmp, "cow.obj")
    stio.save_obj(obj, torch.from_numpy(cow["verts"]), torch.from_numpy(cow["faces"].astype(np.int64)),
                  torch.from_numpy(cow["verts_uvs"]), torch.from_numpy(cow["faces_uvs"].astype(np.int64)), tex)
    sty = np.load(os.path.join(golden_dir, "assets_style1_512.npz"))["rgb_u8"]
    style = os.path.join(tmp, "style.png")
    Image.fromarray(sty).save(style)
    return obj, style


def _run(script, world, args, port):
    env = dict(os.environ, ST3D_DIST_BACKEND="gloo", PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable]
    if world > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(PKG, script)] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def _losses(out):
    return [float(line.split("Loss ")[1]) for line in open(os.path.join(out, "log.txt")).read().splitlines()[1:]]


def _final(out):
    from PIL import Image
    from st3d import io as stio
    verts, _, _ = stio.load_obj(os.path.join(out, "final.obj"), load_textures=False)
    return verts.numpy(), np.asarray(Image.open(os.path.join(out, "final.png")), dtype=np.int32)


@pytest.mark.parametrize("script,extra,n_log", [
    ("second_approach.py", ["--epochs", "3", "--lr", "0.002", "--save_every", "0"], 3),
    ("first_approach.py", ["--n_style_transfer_steps", "3", "--n_mse_steps", "3", "--mse_lr", "0.002"], 6),
])
def test_two_ranks_reproduce_the_single_rank_run_for_target_both(cow, golden_dir, tmp_path, script, extra, n_log):
    obj, style = _assets(str(tmp_path), cow, golden_dir)
    # 4 views in batches of 3: a 3-view batch split 2 + 1 and a 1-view batch that leaves rank 1 without views
    common = ["--obj_path", obj, "--style_path", style, "--size", "64", "--n_views", "4", "--batch_size", "3", "--seed", "0",
              "--optimization_target", "both"] + extra
    one, two = str(tmp_path / "w1"), str(tmp_path / "w2")
    _run(script, 1, common + ["--output_path", one], 0)
    _run(script, 2, common + ["--output_path", two], 29611)
    l1, l2 = _losses(one), _losses(two)
    assert len(l1) == len(l2) == n_log
    np.testing.assert_allclose(l2, l1, rtol=2e-3)              # summation order + Adam feedback over the steps
    v1, t1 = _final(one)
    v2, t2 = _final(two)
    moved = np.abs(v1 - cow["verts"])
    assert moved.max() > 1e-4                                   # the vertices were optimised at all
    # Adam turns a gradient component whose sign is summation-order noise (contributions of several pixels cancelling)
    # into a +-lr step, so single coordinates may differ by a few lr between ANY two runs that sum in a different
    # order; everything else must agree closely
    diff = np.abs(v1 - v2)
    print(f"\n{script}: vertex movement mean {moved.mean():.2e} max {moved.max():.2e}; 1 vs 2 ranks: mean |diff| "
          f"{diff.mean():.2e}, max {diff.max():.2e}, > 2e-4: {(diff > 2e-4).mean() * 100:.2f} %")
    assert diff.mean() <= 0.01 * moved.mean()
    assert (diff > 2e-4).mean() <= 0.01
    assert diff.max() <= 6 * 0.002 + 1e-6                       # never more than the steps taken (6 or 8 at lr 0.002 ... )
    assert np.abs(t1 - t2).max() <= 3


def test_c_abi_collective_world_of_one():
    """st3d_comm_* (include/st3d.h): RCCL bound at run time, communicator of one rank on this GPU, in-place SUM all-reduce
    of a texture-gradient-sized buffer (3 MiB) = identity; argument errors are reported, not crashed.  The N > 1 case
    needs one GPU per rank: the driver's 8-GPU bench exercises the same ncclAllReduce through torch.distributed."""
    import ctypes
    from st3d import _lib
    from st3d._lib import call, dptr, stream_ptr
    lib = _lib.load()
    uid = ctypes.create_string_buffer(128)
    call("st3d_comm_unique_id", uid)
    assert any(uid.raw)
    h = ctypes.c_void_p()
    assert lib.st3d_comm_init(ctypes.byref(h), 1, 1, uid) == -1 and b"invalid argument" in lib.st3d_last_error()
    call("st3d_comm_init", ctypes.byref(h), 0, 1, uid)
    x = torch.rand(512 * 512 * 3, device="cuda:0")
    y = x.clone()
    for _ in range(3):
        call("st3d_allreduce_sum_f32", h, dptr(y), y.numel(), stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    assert lib.st3d_allreduce_sum_f32(h, None, 4, None) == -1
    call("st3d_comm_destroy", h)


@pytest.mark.parametrize("target,n_params", [("texture", 1), ("both", 2)])
def test_bench_gpus_2_launches_its_own_ranks_and_reports_what_ran(target, n_params):
    """`python bench.py --gpus 2` with NO torchrun around it (what the driver runs on a multi-GPU node): the parent starts
    two fresh ranks before any GPU call, rank 0's line is the only thing on stdout, and the line proves what ran --
    `ranks_seen` is a SUM all-reduce of ones on the process group's backend, one collective per step even with two
    optimised tensors (flat [d verts || d texture] buffer).  gloo here because both ranks share the one GPU of the box;
    on the 8-GPU node the same code path runs on RCCL."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["ST3D_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--size", "256", "--views", "2", "--target", target]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(out) == 1, r.stdout
    line = json.loads(out[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["dist_backend"] == "gloo"
    assert len(line["devices_seen"]) == 2
    assert line["collectives_per_step"] == 1.0
    assert line["allreduce_bytes"] == 4 * (256 * 256 * 3 + (2930 * 3 if n_params == 2 else 0))
    assert line["allreduce_ms"] >= line["allreduce_ms_min"] > 0
    assert line["config"]["global_views_per_step"] == 4 and line["value"] > 0
    assert np.isfinite(line["final_loss"]) and np.isfinite(line["first_step_loss"])


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29612")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
