#!/usr/bin/env python3
"""Time the GO read-out (node-wise linear + BatchNorm over nodes + ReLU) forward and backward at the bench shape."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import ops  # noqa: E402

dev = "cuda"
B, F, N, D, G = 512, 5, 400, 32, 2
x = torch.randn(B, F, N, device=dev, requires_grad=True)
w = torch.randn(D, F, device=dev, requires_grad=True)
ga, be = torch.ones(N, device=dev, requires_grad=True), torch.zeros(N, device=dev, requires_grad=True)
rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
SIDE = torch.cuda.Stream()


def fwd():
    return ops.NodeLinearBN.apply(x, w, ga, be, rm, rv, True, 0.1, 1e-5, G)


def timeit(fn, iters=20):
    torch.cuda.synchronize()
    with torch.cuda.stream(SIDE):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=SIDE):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


with torch.no_grad():
    t_f = timeit(fwd)


def fb():
    y = fwd()
    torch.autograd.grad(y, (x, w, ga, be), dy)


with torch.cuda.stream(SIDE):
    dy = torch.randn(B, N, D, device=dev)
t_fb = timeit(fb)
print(f"read-out [{B},{F},{N}] -> [{B},{N},{D}], {G} groups: forward {t_f:.2f} us, forward+backward {t_fb:.2f} us")
