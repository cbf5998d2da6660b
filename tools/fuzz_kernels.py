#!/usr/bin/env python3
"""Randomised differential run of the VGG-side kernels (GPU): random shapes within the supported ranges, each launch
checked against fp64 torch autograd or -- where two code paths must agree bitwise -- against the other path.

    python tools/fuzz_kernels.py [--seconds 120] [--seed 0]

Prints one line per failure and a summary; exit code 1 if anything failed.  Not part of the test suite (the suite holds
the fixed cases); this is the tool that picks the odd shapes nobody thought of."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2d-to-3d-style-transfer_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from st3d import ops  # noqa: E402

dev = torch.device("cuda:0")


def close(got, ref, rtol):
    ref = ref.double().cpu()
    err = (got.double().cpu() - ref).abs().max().item()
    return err <= rtol * (ref.abs().max().item() + 1e-30), err


def case_wino(rng, g):
    Cin = int(rng.choice([8, 16, 24, 64, 128, 200, 256]))
    Cout = int(rng.choice([64, 128, 192, 256]))
    H = 2 * int(rng.integers(1, 40))
    W = 4 * int(rng.integers(1, 30))
    N = int(rng.integers(1, 4))
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).double()
    b = (torch.randn(Cout, generator=g) * 0.1).double()
    y = F.relu(F.conv2d(x, w, b, padding=1))
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    uf, ud = ops.wino_pack(w.float().to(dev))
    yd = ops.wino_fwd(x.detach().float().to(dev), uf, b.float().to(dev), Cout, relu=True)
    ok1, e1 = close(yd, y, 5e-5)
    if Cin % 64:                 # the input gradient writes Cin channels in tiles of 64: forward only for these
        return ok1, f"wino N{N} {Cin}->{Cout} {H}x{W}: fwd {e1:.2e}"
    gate = (yd > 0)
    ref_gx = torch.autograd.grad(F.conv2d(x, w, b, padding=1), x, gy * gate.cpu().double())[0]
    gx = ops.wino_dgrad(gy.float().to(dev), yd, ud, Cin)
    ok2, e2 = close(gx, ref_gx, 1e-4)
    # producer-gated variants must be bitwise the consumer-gated launch
    pre = torch.where(gate, gy.float().to(dev), torch.zeros_like(yd))
    og = torch.relu(torch.randn(N, Cin, H, W, generator=g)).to(dev)
    ok3 = torch.equal(ops.wino_dgrad_chain(pre, ud, Cin), gx)
    ok4 = torch.equal(ops.wino_dgrad_chain(pre, ud, Cin, out_gate=og), torch.where(og > 0, gx, torch.zeros_like(gx)))
    # pooled input
    pooled, idx = ops.maxpool2x2(yd)
    gp = torch.randn(pooled.shape, generator=g).to(dev)
    r2 = ops.wino_dgrad_unpool(gp, idx, pooled, ud, Cin)
    ok5 = torch.equal(ops.wino_dgrad_chain(torch.where(pooled > 0, gp, torch.zeros_like(gp)), ud, Cin, pool_idx=idx), r2)
    return ok1 and ok2 and ok3 and ok4 and ok5, f"wino N{N} {Cin}->{Cout} {H}x{W}: fwd {e1:.2e} dgrad {e2:.2e} chain {ok3} {ok4} {ok5}"


def case_wino43(rng, g):
    """F(4x4,3x3): random supported shape (both tile geometries), random persistent-slot count and slot numbering -- the
    stream of stages crosses tile boundaries, ragged -- against fp64 and, bitwise, against its own separate-launch forms."""
    Cin = int(rng.choice([64, 128, 192, 256, 320]))          # (the pack serves both directions: both channel counts % 64)
    Cout = int(rng.choice([64, 128, 192]))
    if rng.integers(0, 2):
        H, W = 4 * int(rng.integers(1, 12)), 64 * int(rng.integers(1, 4))
    else:
        H, W = 8 * int(rng.integers(1, 6)), 32 * int(rng.integers(1, 6))
    N = int(rng.integers(1, 4))
    os.environ["ST3D_W43_SLOTS"] = str(int(rng.choice([0, 1, 2, 3, 4, 7, 16, 1000])))
    os.environ["ST3D_W43_XCD"] = str(int(rng.integers(0, 2)))
    if os.environ["ST3D_W43_SLOTS"] == "1000":
        del os.environ["ST3D_W43_SLOTS"]
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).double()
    b = (torch.randn(Cout, generator=g) * 0.1).double()
    pre = F.conv2d(x, w, b, padding=1)
    uf, ud = ops.wino43_pack(w.float().to(dev))
    xd = x.detach().float().to(dev)
    yd = ops.wino43_fwd(xd, uf, b.float().to(dev), Cout, relu=True)
    ok1, e1 = close(yd, F.relu(pre), 3e-5)
    yf, pd, idx = ops.wino43_fwd(xd, uf, b.float().to(dev), Cout, relu=True, pool=True)
    p2, idx2 = ops.maxpool2x2(yd)
    ok2 = torch.equal(yf, yd) and torch.equal(pd, p2) and torch.equal(idx, idx2)
    gy = torch.randn(pre.shape, generator=g, dtype=torch.float64)
    gate = (yd.cpu() > 0).double()
    gyg = (gy * gate).float().to(dev)
    gx = ops.wino43_dgrad_chain(gyg, ud, Cin)
    ok3, e3 = close(gx, torch.autograd.grad(pre, x, gy * gate)[0], 5e-5)
    og = torch.randn(N, Cin, H, W, generator=g).to(dev)
    ok4 = torch.equal(ops.wino43_dgrad_chain(gyg, ud, Cin, out_gate=og), torch.where(og > 0, gx, torch.zeros_like(gx)))
    gp = torch.randn(N, Cout, H // 2, W // 2, generator=g).to(dev)
    pidx = torch.randint(0, 4, (N, Cout, H // 2, W // 2), generator=g, dtype=torch.uint8).to(dev)
    up = torch.zeros(N, Cout, H, W, device=dev)
    for k in range(4):
        up[:, :, (k >> 1)::2, (k & 1)::2] = gp * (pidx == k).float()
    ok5 = torch.equal(ops.wino43_dgrad_chain(gp, ud, Cin, pool_idx=pidx), ops.wino43_dgrad_chain(up, ud, Cin))
    tag = f"wino43 N{N} {Cin}->{Cout} {H}x{W} slots {os.environ.get('ST3D_W43_SLOTS', 'default')} xcd {os.environ['ST3D_W43_XCD']}"
    os.environ.pop("ST3D_W43_SLOTS", None); os.environ.pop("ST3D_W43_XCD", None)
    return ok1 and ok2 and ok3 and ok4 and ok5, f"{tag}: fwd {e1:.2e} pool {ok2} dgrad {e3:.2e} gate {ok4} unpool {ok5}"


def case_gram(rng, g):
    C = 32 * int(rng.integers(1, 17))
    H, W = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    B = int(rng.integers(1, 4))
    f = torch.relu(torch.randn(B, C, H, W, generator=g)).to(dev)
    G = ops.gram_fwd(f)
    ff = f.double().reshape(B, C, -1)
    ok1, e1 = close(G, ff @ ff.transpose(1, 2), 2e-4)
    D = torch.randn(B, C, C, generator=g)
    D = (0.5 * (D + D.transpose(1, 2))).to(dev).contiguous()
    base = torch.randn(B, C, H, W, generator=g).to(dev)
    out = ops.gram_bwd(D, f, 0.7, out=base.clone())
    ok2, e2 = close(out, base.double() + 0.7 * (D.double() @ ff).reshape(B, C, H, W), 2e-4)
    ok3 = torch.equal(ops.gram_bwd(D, f, 0.7, out=base.clone(), gated=True), torch.where(f > 0, out, torch.zeros_like(out)))
    return ok1 and ok2 and ok3, f"gram B{B} C{C} {H}x{W}: fwd {e1:.2e} bwd {e2:.2e} gated {ok3}"


def case_tap0(rng, g):
    H, W = int(rng.integers(1, 90)), 2 * int(rng.integers(1, 60))
    N = int(rng.integers(1, 4))
    has_g, has_d = bool(rng.integers(0, 4)), bool(rng.integers(0, 4))
    if not (has_g or has_d):
        has_g = True
    act = torch.relu(torch.randn(N, 64, H, W, generator=g)).to(dev)
    gy = torch.randn(N, 64, H, W, generator=g).to(dev) if has_g else None
    D = torch.randn(N, 64, 64, generator=g).to(dev) if has_d else None
    w = (torch.randn(64, 3, 3, 3, generator=g) * 0.3).to(dev)
    _, wd = ops.conv3x3_pack(w)
    got = ops.conv1_bwd(gy, act, D, 0.4, wd)
    tot = (gy.double() if has_g else 0) + (0.4 * (D.double() @ act.double().reshape(N, 64, -1)).reshape(N, 64, H, W) if has_d else 0)
    gated = tot * (act > 0).double()
    ref = F.conv_transpose2d(gated.cpu(), w.double().cpu(), padding=1)
    ok, e = close(got, ref, 5e-5)
    return ok, f"tap0 N{N} {H}x{W} g{int(has_g)} D{int(has_d)}: {e:.2e}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", default="", help="substring of the case names to run (e.g. wino43)")
    a = ap.parse_args()
    import numpy as np
    rng = np.random.default_rng(a.seed)
    g = torch.Generator().manual_seed(a.seed)
    cases = [case_wino, case_wino43, case_gram, case_tap0]
    if a.only:
        cases = [c for c in cases if a.only in c.__name__]
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < a.seconds:
        fn = cases[n % len(cases)]
        ok, msg = fn(rng, g)
        n += 1
        if not ok:
            bad += 1
            print("FAIL", msg, flush=True)
        elif n % 25 == 0:
            print(f"[{time.time() - t0:5.0f} s] {n} cases, last: {msg}", flush=True)
    print(f"{n} cases, {bad} failures")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
