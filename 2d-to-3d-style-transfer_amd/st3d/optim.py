"""Fused Adam + view-sharded data parallelism.

``Adam`` mirrors the part of ``torch.optim.Adam`` the reference touches (utils.py:185-195,
style_transfer.py:57: default betas/eps, no weight decay, ``zero_grad()`` / ``step()``) and runs
one fused HIP launch per parameter (st3d_adam_step).

Multi-GPU: one process per GPU, each rank renders/VGGs its slice of the view
batch with the loss means divided by the GLOBAL batch; ``step()`` all-reduces (SUM) the flat
gradient of every parameter over RCCL (torch.distributed backend "nccl" == RCCL on ROCm,
xGMI inside a node) and then every rank applies the identical Adam update -- parameters stay
replicated without a broadcast.  The message is 3 MiB at 512^2 (one launch-latency-bound
collective per step); view-independent terms (mesh regularisers) must be added after the
reduce or scaled by 1/world.
"""
import os

import torch
import torch.distributed as dist

from . import ops


def dist_info():
    """(rank, world, local_rank) from the torchrun environment (1 process per GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_distributed(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* when WORLD_SIZE > 1."""
    rank, world, local = dist_info()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # ST3D_DIST_BACKEND=gloo: rehearse the N>1 path with several ranks sharing one GPU (RCCL refuses that)
            backend = os.environ.get("ST3D_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_views(n_views, rank, world):
    """Contiguous slice [lo, hi) of the view batch owned by `rank` (uneven tails allowed)."""
    base, rem = divmod(n_views, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_reduce_sum_(t):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class Adam:
    """``Adam(params, lr)`` as the reference uses it, or ``Adam([{"params": [...], "lr": ...}, ...])`` for one
    learning rate per tensor group (the reference's notes.txt:29 idea; torch.optim's param_groups convention).
    ``state_dict()`` / ``load_state_dict()`` carry step counts and both moment buffers for checkpoint / resume."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, reduce_grads=True):
        params = list(params)
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        if params and isinstance(params[0], dict):
            self.param_groups = [{"params": list(g["params"]), "lr": float(g.get("lr", lr)), "betas": self.betas,
                                  "eps": self.eps} for g in params]
        else:
            self.param_groups = [{"params": params, "lr": float(lr), "betas": self.betas, "eps": self.eps}]
        self.params = [p for g in self.param_groups for p in g["params"]]
        self.lr = self.param_groups[0]["lr"]
        self.state = {}
        self.reduce_grads = reduce_grads

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def _state_of(self, p):
        st = self.state.get(id(p))
        if st is None:
            st = {"step": 0, "exp_avg": torch.zeros_like(p, memory_format=torch.contiguous_format),
                  "exp_avg_sq": torch.zeros_like(p, memory_format=torch.contiguous_format)}
            self.state[id(p)] = st
        return st

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad.contiguous()
                if self.reduce_grads:
                    all_reduce_sum_(g)
                st = self._state_of(p)
                st["step"] += 1
                if not p.is_contiguous():
                    raise RuntimeError("st3d Adam needs contiguous parameters")
                ops.adam_step(p.data, g, st["exp_avg"], st["exp_avg_sq"], st["step"], group["lr"], self.betas[0],
                              self.betas[1], self.eps)

    def state_dict(self):
        """Positional (like torch.optim): entry k belongs to the k-th parameter in construction order."""
        out = []
        for p in self.params:
            st = self._state_of(p)
            out.append({"step": st["step"], "exp_avg": st["exp_avg"].detach().cpu(), "exp_avg_sq": st["exp_avg_sq"].detach().cpu()})
        return {"state": out, "lrs": [g["lr"] for g in self.param_groups]}

    def load_state_dict(self, sd):
        if len(sd["state"]) != len(self.params):
            raise ValueError("optimizer state holds %d tensors, this optimizer has %d" % (len(sd["state"]), len(self.params)))
        for p, e in zip(self.params, sd["state"]):
            if tuple(e["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("optimizer state shape %s does not match parameter %s" % (tuple(e["exp_avg"].shape), tuple(p.shape)))
            st = self._state_of(p)
            st["step"] = int(e["step"])
            st["exp_avg"].copy_(e["exp_avg"])
            st["exp_avg_sq"].copy_(e["exp_avg_sq"])
