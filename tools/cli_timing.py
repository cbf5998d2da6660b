#!/usr/bin/env python3
"""Wall-clock of the drop-in second_approach CLI at config 2 (cow fixture, 512^2, 8 views) with and without the
per-step PNG dumps the reference does (second_approach.py:183-185)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2d-to-3d-style-transfer_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from PIL import Image
import second_approach as SA
import st3d.io as sio

cow = np.load(os.path.join(ROOT, "tests/golden/assets_cow_mesh.npz"))
tmp = tempfile.mkdtemp()
obj = os.path.join(tmp, "cow.obj")
sio.save_obj(obj, torch.from_numpy(cow["verts"]), torch.from_numpy(cow["faces"].astype(np.int64)), torch.from_numpy(cow["verts_uvs"]),
             torch.from_numpy(cow["faces_uvs"].astype(np.int64)), torch.from_numpy(cow["texture_u8"]).float() / 255)
sty = os.path.join(tmp, "style.png")
Image.fromarray(np.load(os.path.join(ROOT, "tests/golden/assets_style1_512.npz"))["rgb_u8"]).save(sty)
for save_every, epochs in ((0, 20), (1, 20)):
    out = os.path.join(tmp, f"out{save_every}")
    t0 = time.time()
    SA.main(["--obj_path", obj, "--style_path", sty, "--size", "512", "--n_views", "8", "--batch_size", "8", "--epochs", str(epochs),
             "--output_path", out, "--seed", "0", "--save_every", str(save_every)])
    torch.cuda.synchronize()
    print(f"save_every={save_every}: {epochs} epochs in {time.time() - t0:.2f} s (incl. setup + final export)", flush=True)
