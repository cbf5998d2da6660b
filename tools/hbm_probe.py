#!/usr/bin/env python3
"""Practical HBM rates on the box (copy, read-modify-write, read-only sum) at the sizes the HBM-bound kernels move:
the yardstick for shade/raster/Gram-backward/conv1_1 numbers in DESIGN.md."""
import torch
dev = torch.device("cuda:0")
n = 8 * 64 * 512 * 512          # conv1_1 activation of config 2: 537 MB
x = torch.rand(n, device=dev); y = torch.rand(n, device=dev); z = torch.empty_like(x)


def timed(fn, bytes_moved, name, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:34s} {ms * 1e3:8.1f} us  {bytes_moved / ms / 1e9:6.2f} TB/s")


b = n * 4
timed(lambda: z.copy_(x), 2 * b, "copy (1 read + 1 write)")
timed(lambda: y.add_(x), 3 * b, "y += x (2 reads + 1 write)")
timed(lambda: torch.add(x, y, out=z), 3 * b, "z = x + y (2 reads + 1 write)")
timed(lambda: x.sum(), b, "sum (1 read)")
timed(lambda: z.fill_(1.0), b, "fill (1 write)")
