"""Frozen VGG-19 feature extractor on libst3d (utils.py:48-52 get_vgg, style_transfer.py:10-27).

``Vgg19Features`` keeps the duck type the reference's ``get_features`` walks -- an ordered
``_modules`` dict with keys '0'..'36' (torchvision ``vgg19().features`` numbering, SURVEY.md
A.7) whose values are callables -- so the reference loop ``for name, layer in
model._modules.items(): x = layer(x)`` still works on it (each module is one HIP launch),
while ``get_features`` / ``compute_perceptual_loss`` recognise the object and dispatch to the
fused plan (``PerceptualPlan``: st3d_plan_* in include/st3d.h).

Weights: the reference downloads IMAGENET1K_V1 (utils.py:49), which is impossible offline.
``get_vgg()`` loads a local state_dict when ``ST3D_VGG19_WEIGHTS`` (or ``weights=``) names one
(torchvision key layout ``features.<idx>.weight`` or ``<idx>.weight``), otherwise seeded
He-normal weights -- the same generator sequence as the oracle's
``synthetic_vgg19_state`` so both sides can be driven with identical parameters.
"""
import ctypes
import math
import os
from collections import OrderedDict

import torch

from . import _lib, ops
from ._lib import call, dptr, stream_ptr

VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
DEFAULT_TAPS = {"0": "conv1_1", "5": "conv2_1", "10": "conv3_1", "19": "conv4_1", "21": "conv4_2", "28": "conv5_1"}
STYLE_TAP_MODULES = (0, 5, 10, 19, 28)
CONTENT_TAP_MODULE = 21


def synthetic_state(seed=0, bias_scale=0.05):
    g = torch.Generator().manual_seed(seed)
    state, cin, idx = {}, 3, 0
    for v in VGG19_CFG:
        if v == "M":
            idx += 1
            continue
        state[f"{idx}.weight"] = torch.randn((v, cin, 3, 3), generator=g) * math.sqrt(2.0 / (cin * 9))
        state[f"{idx}.bias"] = torch.randn((v,), generator=g) * bias_scale
        cin = v
        idx += 2
    return state


class _Conv:
    """conv3x3(pad 1); its ReLU is the next module.  When called from the generic
    ``_modules`` walk the pre-ReLU value is materialised (one launch with relu=0)."""
    kind = "conv"

    def __init__(self, owner, idx, cin, cout):
        self.owner, self.idx, self.cin, self.cout = owner, idx, cin, cout
        self.wf = self.wd = self.bias = None

    def __call__(self, x):
        return ops.conv3x3_fwd(x, self.wf, self.bias, self.cout, relu=False)


class _ReLU:
    kind = "relu"
    inplace = True

    def __call__(self, x):
        return x.relu_()        # in place, as torchvision builds it (the tap tensor is overwritten)


class _Pool:
    kind = "pool"

    def __call__(self, x):
        return ops.maxpool2x2(x, want_idx=False)


class Vgg19Features:
    def __init__(self, state=None, device="cuda"):
        self.device = torch.device(device)
        self._modules = OrderedDict()
        cin, idx = 3, 0
        for v in VGG19_CFG:
            if v == "M":
                self._modules[str(idx)] = _Pool()
                idx += 1
            else:
                self._modules[str(idx)] = _Conv(self, idx, cin, v)
                self._modules[str(idx + 1)] = _ReLU()
                cin = v
                idx += 2
        h = ctypes.c_void_p()
        call("st3d_vgg_create", ctypes.byref(h))
        self._h = h
        self._plans = OrderedDict()     # (B, S) -> PerceptualPlan, least recently used first
        self.load_state_dict(state if state is not None else synthetic_state(0))

    def load_state_dict(self, state):
        for name, mod in self._modules.items():
            if mod.kind != "conv":
                continue
            w = state.get(f"{name}.weight", state.get(f"features.{name}.weight"))
            b = state.get(f"{name}.bias", state.get(f"features.{name}.bias"))
            if w is None or b is None:
                raise KeyError(f"state_dict lacks conv {name}")
            w = w.detach().to(self.device, torch.float32).contiguous()
            b = b.detach().to(self.device, torch.float32).contiguous()
            assert tuple(w.shape) == (mod.cout, mod.cin, 3, 3), (name, tuple(w.shape))
            mod.wf, mod.wd = ops.conv3x3_pack(w)
            mod.bias = b
            call("st3d_vgg_set_conv", self._h, int(name), dptr(w), dptr(b), stream_ptr())
        torch.cuda.synchronize()

    def parameters(self):
        return iter(())          # frozen (utils.py:50-51)

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    MAX_PLANS = int(os.environ.get("ST3D_MAX_PLANS", "4"))

    def plan(self, B, S):
        """The (cached) workspace for batch B at SxS.  A plan holds every activation of its shape (4.4 GB for 8 views at
        512^2, 35 GB for 16 at 1024^2: st3d_plan_bytes), so at most MAX_PLANS shapes stay resident, least recently used first out; a
        plan that does not fit is retried once after the others are released.  An evicted shape is rebuilt on its next
        use (targets are re-derived from the tensors the caller passes, see set_content / set_style)."""
        key = (int(B), int(S))
        p = self._plans.get(key)
        if p is not None:
            self._plans.move_to_end(key)
            return p
        while len(self._plans) >= max(self.MAX_PLANS, 1):
            self._plans.popitem(last=False)[1].close()
        try:
            p = PerceptualPlan(self, *key)
        except _lib.St3dError:
            if not self._plans:
                raise
            while self._plans:
                self._plans.popitem(last=False)[1].close()
            torch.cuda.empty_cache()
            p = PerceptualPlan(self, *key)
        self._plans[key] = p
        return p

    def __del__(self):
        try:
            for p in list(getattr(self, "_plans", {}).values()):
                p.close()
            if getattr(self, "_h", None):
                _lib.load().st3d_vgg_destroy(self._h)
                self._h = None
        except Exception:
            pass


class PerceptualPlan:
    """Workspace + fixed launch sequence for batch B at SxS (st3d_plan in include/st3d.h)."""

    def __init__(self, vgg, B, S):
        self.vgg, self.B, self.S = vgg, B, S
        h = ctypes.c_void_p()
        call("st3d_plan_create", ctypes.byref(h), vgg._h, B, S)
        self._h = h
        self._content_key = self._style_key = None
        self._content_ref = self._style_ref = None      # the keyed tensors themselves (see _same)
        self.generation = 0         # bumped by every call that runs a VGG forward in this plan's buffers
        self.loss_buf = torch.zeros((3,), dtype=torch.float32, device=vgg.device)
        if os.environ.get("ST3D_GRAPH", "0") not in ("", "0"):
            self.use_graph(True)

    def use_graph(self, on=True):
        """Replay the loss step as one HIP graph (st3d_plan_graph); off by default."""
        call("st3d_plan_graph", self._h, 1 if on else 0)

    def close(self):
        if self._h:
            _lib.load().st3d_plan_destroy(self._h)
            self._h = None

    def bytes(self):
        return _lib.load().st3d_plan_bytes(self._h)

    def forward(self, imgs, upto=28):
        imgs = imgs.detach().to(torch.float32).contiguous()
        call("st3d_plan_forward", self._h, dptr(imgs), imgs.shape[0], int(upto), stream_ptr())
        self._n = imgs.shape[0]
        self.generation += 1

    def backward(self, grads, upto):
        """{module index: d loss / d (that module's output)} for the images of the LAST forward -> d loss / d images
        (st3d_plan_backward: the dgrad chain of the loss plan seeded with the caller's tap gradients)."""
        ptrs = (ctypes.c_void_p * 37)()
        keep = []
        for m, g in grads.items():
            g = g.detach().to(torch.float32).contiguous()
            keep.append(g)
            ptrs[int(m)] = g.data_ptr()
        out = torch.empty((self._n, 3, self.S, self.S), dtype=torch.float32, device=self.vgg.device)
        call("st3d_plan_backward", self._h, self._n, int(upto), ptrs, dptr(out), stream_ptr())
        return out

    def activation(self, module_idx, n=None):
        """Tensor VIEW of the plan's buffer after `module_idx` (valid until the next forward)."""
        p = ctypes.c_void_p()
        C, H, W = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        call("st3d_plan_activation", self._h, int(module_idx), ctypes.byref(p), ctypes.byref(C), ctypes.byref(H),
             ctypes.byref(W))
        n = n or self._n
        return _wrap_device_ptr(p.value, (n, C.value, H.value, W.value), self.vgg.device)

    @staticmethod
    def _key(t):
        return (t.data_ptr(), t._version, tuple(t.shape))

    @staticmethod
    def _same(key, ref, k, t):
        """Is `t` the tensor a target was computed from, unmodified?  (address, version, shape) alone is unsound:
        libst3d kernels write into fresh ``torch.empty`` buffers (version stays 0) and the allocator hands a freed
        address to the next batch of the same shape, so the keyed tensor is held (`ref`) -- its address cannot be
        reused while it is the current target -- and compared by identity of storage + version."""
        return key is not None and ref is not None and k == key and ref.data_ptr() == t.data_ptr()

    CONTENT_CACHE = 8      # targets kept (67 MB each for 8 views at 512^2); the reference alternates ceil(n_views / batch) batches

    def set_content(self, content, force=False):
        """conv4_2 of `content` becomes the plan's content target.  Keyed on the tensor's identity/version: a caller that
        cycles through a few fixed content batches (second_approach.py:145-160) pays the VGG forward once per batch, a
        device copy afterwards; new or modified tensors (``--content_background noise``) are recomputed."""
        k = self._key(content)
        if not force and self._same(self._content_key, self._content_ref, k, content):
            return
        n = content.shape[0]
        cache = self.__dict__.setdefault("_content_cache", {})
        hit = None if force else cache.get(k)
        if hit is not None:         # (features, the tensor: holding it keeps its address from being reused while cached)
            call("st3d_plan_set_content_features", self._h, dptr(hit[0]), n, stream_ptr())
        else:
            c = content.detach().to(torch.float32).contiguous()
            call("st3d_plan_set_content", self._h, dptr(c), n, stream_ptr())
            self.generation += 1
            if self._content_key is not None and self.CONTENT_CACHE > 0:
                # a second distinct batch showed up: start keeping targets (single-batch runs never pay the copy)
                feats = torch.empty((n, 512, self.S // 8, self.S // 8), dtype=torch.float32, device=self.vgg.device)
                call("st3d_plan_get_content_features", self._h, dptr(feats), n, stream_ptr())
                if len(cache) >= self.CONTENT_CACHE:
                    cache.pop(next(iter(cache)))
                cache[k] = (feats, content)
        self._content_key, self._content_ref = k, content

    def set_style(self, style, n, force=False):
        k = self._key(style) + (n,)
        if force or not self._same(self._style_key, self._style_ref, k, style):
            s = style.detach().to(torch.float32).contiguous()
            sb = s.shape[0]
            if sb > 1 and s.stride(0) == 0:
                sb = 1
            if sb > 1 and bool((s[0:1] == s).all()):      # the reference repeats one image (second_approach.py:157)
                sb = 1
            s = s[:1].contiguous() if sb == 1 else s
            call("st3d_plan_set_style", self._h, dptr(s), sb, n, stream_ptr())
            self.generation += 1
            self._style_key, self._style_ref = k, style

    def loss(self, current, style_weight, content_weight, batch_denom=None, want_grad=True):
        """-> (loss_buf view [total, content, style], grad (n,3,S,S) or None)."""
        cur = current.detach().to(torch.float32).contiguous()
        n = cur.shape[0]
        grad = torch.empty_like(cur) if want_grad else None
        call("st3d_plan_loss", self._h, dptr(cur), n, int(batch_denom or n), float(style_weight), float(content_weight),
             dptr(self.loss_buf), dptr(grad), stream_ptr())
        self.generation += 1
        return self.loss_buf, grad

    def profile(self, enable):
        call("st3d_plan_profile", self._h, 1 if enable else 0)

    def profile_read(self):
        ms = (ctypes.c_float * 10)()
        nl = (ctypes.c_int * 10)()
        call("st3d_plan_profile_read", self._h, ms, nl)
        return {k: {"ms": ms[i], "launches": nl[i]} for i, k in enumerate(self.FAMILIES)}


    # conv_*: Winograd F(2x2,3x3) launches; conv43_*: Winograd F(4x4,3x3); convx_*: direct / vector-ALU kernels
    FAMILIES = ("conv_fwd", "conv_dgrad", "pool", "gram_fwd", "gram_bwd", "elementwise", "convx_fwd", "convx_dgrad", "conv43_fwd",
                "conv43_dgrad")

    def profile_launches(self):
        """[(family, VGG module index, ms)] for every launch bracket since the last read (profiling on)."""
        cap = 4096
        tags, ms, n = (ctypes.c_int * cap)(), (ctypes.c_float * cap)(), ctypes.c_int(0)
        call("st3d_plan_profile_launches", self._h, tags, ms, cap, ctypes.byref(n))
        return [(self.FAMILIES[tags[i] // 100], tags[i] % 100, ms[i]) for i in range(min(n.value, cap))]


class _DevArray:
    """__cuda_array_interface__ shim so torch can view a raw device pointer without a copy."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": tuple(shape), "typestr": "<f4", "version": 3,
                                         "strides": None}


def _wrap_device_ptr(ptr, shape, device):
    return torch.as_tensor(_DevArray(ptr, shape), device=device)


def get_vgg(weights=None, device="cuda", seed=0):
    """Drop-in for utils.py:48-52.  `weights`: path to a local state_dict (or env
    ST3D_VGG19_WEIGHTS); default = seeded synthetic weights (no download is attempted)."""
    path = weights or os.environ.get("ST3D_VGG19_WEIGHTS")
    state = None
    if path:
        state = torch.load(path, map_location="cpu", weights_only=True)
    else:
        state = synthetic_state(seed)
    return Vgg19Features(state, device=device)
