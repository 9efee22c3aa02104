#!/usr/bin/env python3
"""Time the GO attention / decoder kernels (forward and backward) on the structures and shapes of the bench step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from igcn_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
model, _ = bench.build_model(dev)
go = model.go_network
B = int(os.environ.get("B", "512"))


SIDE = torch.cuda.Stream()      # autograd replays backward on the forward's stream: forward, warm-up and capture share it


def timeit(fn, iters=int(os.environ.get("ITERS", "20"))):
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


def timeit_eager(fn, iters=50):
    """Events around an eager loop (the backward closures replay saved autograd graphs; hipGraph capture of those
    crashed in hipStreamEndCapture on this stack).  Valid while the device time per call exceeds the host's."""
    with torch.cuda.stream(SIDE):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for j, csr in enumerate(go.enc_csr):
    w = (go.w_inc[j].weight, go.w_s_loop[j].weight, go.w_att_in[j].weight.view(-1), go.w_att_s[j].weight.view(-1))
    fin = w[0].shape[1]
    x = torch.randn(B, fin, csr.n_rows, device=dev, requires_grad=True)
    xd, wd = x.detach(), [t.detach() for t in w]
    t_f = timeit(lambda: ops.GoAttention.apply(xd, *wd, csr))
    with torch.cuda.stream(SIDE):
        y = ops.GoAttention.apply(x, *w, csr)
    dy = torch.randn_like(y)
    t_b = timeit_eager(lambda: torch.autograd.grad(y, (x,) + w, dy, retain_graph=True))
    print(f"attn   layer {j}: N={csr.n_rows} {fin}->{y.shape[1]}  fwd {t_f:7.2f} us   bwd {t_b:7.2f} us", flush=True)
for j, csr in enumerate(go.dec_csr):
    w = (go.w_out[j].weight, go.w_s_loop_out[j].weight)
    fin = w[0].shape[1]
    x = torch.randn(B, fin, csr.n_cols, device=dev, requires_grad=True)
    xd, wd = x.detach(), [t.detach() for t in w]
    t_f = timeit(lambda: ops.GoDecode.apply(xd, *wd, csr))
    with torch.cuda.stream(SIDE):
        y = ops.GoDecode.apply(x, *w, csr)
    dy = torch.randn_like(y)
    t_b = timeit_eager(lambda: torch.autograd.grad(y, (x,) + w, dy, retain_graph=True))
    print(f"decode layer {j}: {csr.n_cols}->{csr.n_rows} {fin}->{y.shape[1]}  fwd {t_f:7.2f} us   bwd {t_b:7.2f} us",
          flush=True)
