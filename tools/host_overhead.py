#!/usr/bin/env python3
"""How launch/host-bound are small workloads? wall time per step vs device time per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-to-3d-style-transfer_amd")]
import torch
from st3d import vgg as V, optim as O
dev = torch.device("cuda:0")
model = V.Vgg19Features(V.synthetic_state(0), device=dev)
for B, S, graph in ((4, 256, 0), (4, 256, 1), (2, 128, 0), (2, 128, 1), (8, 512, 0), (8, 512, 1)):
    plan = model.plan(B, S)
    plan.use_graph(bool(graph))
    x = torch.rand(B, 3, S, S, device=dev).requires_grad_(True)
    plan.set_content(torch.rand(B, 3, S, S, device=dev)); plan.set_style(torch.rand(1, 3, S, S, device=dev), B)
    opt = O.Adam([x], lr=0.01, reduce_grads=False)
    def step():
        _, g = plan.loss(x, 1e6, 1.0); x.grad = g; opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize()
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): step()
    t_enq = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    print(f"B={B} S={S} graph={graph}: enqueue {t_enq/n*1e3:.3f} ms/step (host), wall {t_wall/n*1e3:.3f} ms/step, device {e0.elapsed_time(e1)/n:.3f} ms/step")
