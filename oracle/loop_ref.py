"""oracle/loop_ref.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of ONE optimiser step of the reference's 3-D loop, ``second_approach.py:145-190``
(zero_grad -> content render -> build_mesh -> current render -> compute_second_approach_loss ->
backward -> Adam), assembled from the other oracle modules:

  render (utils.py:65-77)                 oracle/render_ref.py  (C restatement, PARITY UNPINNED)
  compute_perceptual_loss (losses.py:12)  oracle/perceptual_ref.py (pinned to the reference's own code, G1-G5)
  mesh regularisers (losses.py:112-124)   oracle/mesh_ref.py    (PARITY UNPINNED)
  torch.optim.Adam                        oracle/raster_ref.c:ref_adam_step (pinned against torch.optim.Adam)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
Backgrounds: 'white' (the reference's default: the render background is already white, utils.py:29-30).
"""
import numpy as np
import torch

from . import mesh_ref as M
from . import perceptual_ref as P
from . import render_ref as RR

DEFAULT_WEIGHTS = {"main_loss_weight": 3.0, "mesh_verts_weight": 1.0, "mesh_edge_loss_weight": 1.0,
                   "mesh_laplacian_smoothing_weight": 1.0, "mesh_normal_consistency_weight": 1.0}   # second_approach.py:33-37


class SecondApproachRef:
    """State of the reference loop between steps: the optimised texture (T,T,3) / verts (V,3), Adam moments and
    step count (one Adam over all leaves, utils.py:183-195), the fixed content mesh and cameras."""

    def __init__(self, verts, faces, verts_uvs, faces_uvs, texture, R, T, S, style, model=None, target="texture",
                 lr=0.01, style_weight=1e6, content_weight=1.0, weights=None, nthreads=8, hoist=True):
        self.verts0 = np.ascontiguousarray(verts, np.float32)
        self.faces = np.ascontiguousarray(faces, np.int32)
        self.verts_uvs = np.ascontiguousarray(verts_uvs, np.float32)
        self.faces_uvs = np.ascontiguousarray(faces_uvs, np.int32)
        self.tex0 = np.ascontiguousarray(texture, np.float32)
        self.R, self.T, self.S = np.asarray(R, np.float32), np.asarray(T, np.float32), int(S)
        self.style = style if style.dim() == 4 else style[None]          # (1,3,S,S) torch
        self.model = model if model is not None else P.make_vgg19_features(seed=0)
        self.target, self.lr = target, float(lr)
        self.sw, self.cw = float(style_weight), float(content_weight)
        self.weights = dict(DEFAULT_WEIGHTS if weights is None else weights)
        self.nthreads, self.hoist = int(nthreads), hoist
        # setup_optimizations (utils.py:173-204): clones of the mesh tensors become the leaves
        self.tex = self.tex0.copy()
        self.verts = self.verts0.copy()
        self.state = {k: [np.zeros_like(a), np.zeros_like(a)] for k, a in (("tex", self.tex), ("verts", self.verts))}
        self.t = 0
        self._content = None
        self._faces_t = torch.from_numpy(self.faces.astype(np.int64))
        self.last = {}

    def content(self):
        """second_approach.py:160 renders the ORIGINAL mesh every step; it never changes, so once is the same."""
        if self._content is None or not self.hoist:
            self._content, _, _ = RR.render_views(self.verts0, self.faces, self.verts_uvs, self.faces_uvs, self.tex0,
                                                  self.R, self.T, self.S, self.nthreads)
        return self._content

    def loss_and_grads(self):
        B = self.R.shape[0]
        content = torch.from_numpy(self.content())
        cur, masks, frags = RR.render_views(self.verts, self.faces, self.verts_uvs, self.faces_uvs, self.tex, self.R,
                                            self.T, self.S, self.nthreads)
        cur_t = torch.from_numpy(cur).requires_grad_(True)
        perc, closs, sloss = P.perceptual_loss_ref(cur_t, content, self.style.expand(B, -1, -1, -1), self.model,
                                                   self.sw, self.cw, return_parts=True)
        perc.backward()
        gimg = cur_t.grad.numpy()
        self.last = {"current": cur, "masks": masks, "perceptual": float(perc.detach()), "content_loss": float(closs.detach()),
                     "style_loss": float(sloss.detach())}
        if self.target == "texture":                                     # losses.py:103-104
            gtex = np.zeros(self.tex.shape, np.float64)
            for b in range(B):
                RR.shade_bwd(gimg[b], frags[b], self.verts_uvs, self.faces_uvs, self.tex, gtex)
            return float(perc.detach()), gtex, None
        mw = self.weights["main_loss_weight"]                            # losses.py:108-124
        gtex, gverts = RR.render_bwd_views(gimg * mw, frags, self.verts, self.faces, self.verts_uvs, self.faces_uvs,
                                           self.tex, self.R, self.T)
        vd = torch.from_numpy(self.verts).double().requires_grad_(True)
        regs = (self.weights["mesh_verts_weight"] * M.verts_mse_ref(vd, torch.from_numpy(self.verts0).double())
                + self.weights["mesh_edge_loss_weight"] * M.mesh_edge_loss_ref(vd, self._faces_t)
                + self.weights["mesh_laplacian_smoothing_weight"] * M.mesh_laplacian_smoothing_ref(vd, self._faces_t)
                + self.weights["mesh_normal_consistency_weight"] * M.mesh_normal_consistency_ref(vd, self._faces_t))
        regs.backward()
        self.last["regs"] = float(regs.detach())
        gverts = gverts + vd.grad.numpy()
        if self.target == "mesh":
            gtex = None
        return mw * float(perc.detach()) + float(regs.detach()), gtex, gverts

    def step(self):
        """-> the loss the reference would log for this step (before the update)."""
        loss, gtex, gverts = self.loss_and_grads()
        self.t += 1
        if gtex is not None:
            g = np.ascontiguousarray(gtex, np.float32)
            RR.adam_step(self.tex, g, self.state["tex"][0], self.state["tex"][1], self.t, self.lr)
        if gverts is not None:
            g = np.ascontiguousarray(gverts, np.float32)
            RR.adam_step(self.verts, g, self.state["verts"][0], self.state["verts"][1], self.t, self.lr)
        self.last["grad_texture"], self.last["grad_verts"] = gtex, gverts
        return loss


def synth_uvs_ref(verts):
    """Independent numpy (float64 -> float32) restatement of the parametrisation the drop-in documents for meshes
    without UVs (teapot, SURVEY.md D3: the reference crashes there, so there is no reference behaviour):
    spherical about the centroid, u = atan2(x, z) / 2pi + 0.5, v = acos(-y / r) / pi, faces_uvs == faces."""
    v = np.asarray(verts, np.float64)
    v = v - v.astype(np.float32).mean(axis=0, dtype=np.float32).astype(np.float64)
    r = np.maximum(np.linalg.norm(v, axis=1), 1e-8)
    u = np.arctan2(v[:, 0], v[:, 2]) / (2 * np.pi) + 0.5
    w = np.arccos(np.clip(-v[:, 1] / r, -1, 1)) / np.pi
    return np.stack([u, w], axis=1).astype(np.float32)
