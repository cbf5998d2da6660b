#!/usr/bin/env python3
"""Prints the actual deviations behind the tolerance-based parity tests (golden vectors of the reference)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2d-to-3d-style-transfer_amd")]
import numpy as np, torch
import losses as L, style_transfer as ST, utils as U
dev = torch.device("cuda:0"); U.device = ST.device = L.device = dev
vgg = U.get_vgg(seed=0)
G = os.path.join(ROOT, "tests", "golden")
for fx in ("g3_perceptual.npz", "g3b_perceptual_96.npz"):
    d = np.load(os.path.join(G, fx))
    cur = torch.from_numpy(d["cur"]).to(dev).requires_grad_(True)
    kw = {"style_weight": float(d["style_weight"]), "content_weight": float(d["content_weight"])} if "style_weight" in d.files else {}
    loss = L.compute_perceptual_loss(cur, torch.from_numpy(d["con"]).to(dev), torch.from_numpy(d["sty"]).to(dev), vgg, **kw)
    loss.backward()
    gref = torch.from_numpy(d["grad"])
    print(f"{fx}: loss rel err {abs(loss.item()-float(d['loss']))/float(d['loss']):.2e} (tol 2e-4), grad rel L2 {float((cur.grad.cpu()-gref).norm()/gref.norm()):.2e} (tol 1e-3)")
d = np.load(os.path.join(G, "g4_style_transfer.npz"))
ST.tqdm = lambda it, **k: it
res = ST.style_transfer(torch.from_numpy(d["init"]).to(dev), torch.from_numpy(d["con"]).to(dev), torch.from_numpy(d["sty"]).to(dev), vgg, steps=int(d["steps"]), lr=float(d["lr"]))
err = (res.detach().cpu() - torch.from_numpy(d["result"])).abs()
print(f"g4 style_transfer 6 steps: max abs err {float(err.max()):.2e} (tol 5e-4), mean {float(err.mean()):.2e}")
