#!/usr/bin/env python3
"""Cost model of hipGraph replay on this stack: per-node time of chains of tiny dependent kernels."""
import sys
import time

import torch

dev = "cuda"
x = torch.zeros(64, device=dev)
big = torch.zeros(1 << 22, device=dev)


def bench(fn, n_nodes, label, reps=50):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:50s} {n_nodes:4d} nodes  {dt*1e6:9.1f} us/replay  {dt*1e6/n_nodes:6.2f} us/node", flush=True)


def chain(n):
    def f():
        for _ in range(n):
            x.add_(1.0)
    return f


def chain_big(n):
    def f():
        for _ in range(n):
            big.add_(1.0)
    return f


def mixed(n):
    def f():
        for _ in range(n):
            big.add_(1.0)
            for _ in range(9):
                x.add_(1.0)
    return f


for n in (50, 300):
    bench(chain(n), n, "tiny dependent add_ (64 floats)")
bench(chain_big(50), 50, "16 MB add_ (HBM-bound, ~8 us each)")
bench(mixed(30), 300, "1 big + 9 tiny, repeated")
