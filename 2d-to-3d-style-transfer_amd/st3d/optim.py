"""Fused Adam + view-sharded data parallelism.

``Adam`` mirrors the part of ``torch.optim.Adam`` the reference touches (utils.py:185-195,
style_transfer.py:57: default betas/eps, no weight decay, ``zero_grad()`` / ``step()``) and runs
one fused HIP launch per parameter (st3d_adam_step).

Multi-GPU: one process per GPU, each rank renders/VGGs its slice of the view
batch with the loss means divided by the GLOBAL batch; ``step()`` all-reduces (SUM) the flat
gradient of every parameter over RCCL (torch.distributed backend "nccl" == RCCL on ROCm,
xGMI inside a node) and then every rank applies the identical Adam update -- parameters stay
replicated without a broadcast.  The gradients of ALL parameters travel as one flat buffer
[d verts || d texture], so a step has exactly ONE collective: 3 MiB (+ 12*V bytes)
at 512^2, launch-latency bound; view-independent terms (mesh regularisers) must be added after the
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


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def all_reduce_sum_(t):
    if _world() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def comm_info(device=None):
    """What the process group really is, for bench / log lines: backend name, the number of ranks that answered a SUM
    all-reduce of ones ON THAT BACKEND (not WORLD_SIZE read back from the environment), the collective library version."""
    info = {"dist_backend": None, "ranks_seen": 1, "rccl_version": None}
    if _world() > 1:
        info["dist_backend"] = dist.get_backend()
        one = torch.ones(1, device=device if info["dist_backend"] == "nccl" or device is not None else "cpu")
        dist.all_reduce(one)
        info["ranks_seen"] = int(one.item())
        if info["dist_backend"] == "nccl":
            try:
                info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:              # the version is a report, not a dependency
                info["rccl_version"] = "unavailable: %r" % (e,)
    return info


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
        self._flat = None                   # [grad of param 0 || grad of param 1 || ...]: the one message of a step
        self.collectives = 0                # all-reduces issued so far (tests: one per step whatever the parameter count)

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

    def _reduced_grads(self):
        """{id(p): gradient summed over ranks}.  One parameter: its gradient is reduced in place.  Several: they are
        packed into one persistent flat buffer in construction order (for setup_optimizations('both'): vertices, then
        texture), reduced by ONE collective, and handed on as views of that buffer."""
        live = [p for p in self.params if p.grad is not None]
        grads = {id(p): p.grad.contiguous() for p in live}
        if not (self.reduce_grads and _world() > 1) or not live:
            return grads
        if len(live) < len(self.params):
            # a collective must have the same length on every rank; a rank that skipped a parameter would desynchronise it
            raise RuntimeError("st3d Adam: with several ranks every parameter needs a gradient on every rank "
                               "(ranks without views contribute zeros: cli.Run.zero_contribution)")
        if len(live) == 1:
            all_reduce_sum_(grads[id(live[0])])
        else:
            n = sum(g.numel() for g in grads.values())
            if self._flat is None or self._flat.numel() != n or self._flat.device != live[0].device:
                self._flat = torch.empty(n, dtype=torch.float32, device=live[0].device)
            views, off = {}, 0
            for p in live:
                g = grads[id(p)]
                v = self._flat[off:off + g.numel()]
                v.copy_(g.reshape(-1))
                views[id(p)] = v.view(g.shape)
                off += g.numel()
            all_reduce_sum_(self._flat)
            grads = views
            for p in live:                  # like the in-place single-parameter case: p.grad holds the summed gradient
                p.grad = views[id(p)]
        self.collectives += 1
        return grads

    @torch.no_grad()
    def step(self):
        with ops.trace("allreduce"):
            grads = self._reduced_grads()
        with ops.trace("adam"):
            for group in self.param_groups:
                for p in group["params"]:
                    g = grads.get(id(p))
                    if g is None:
                        continue
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
