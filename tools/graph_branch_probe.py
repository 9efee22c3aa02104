#!/usr/bin/env python3
"""Does a captured hipGraph run independent branches concurrently?  Two chains of small kernels (each kernel well under
one CU-filling grid), captured (a) back to back on one stream, (b) forked onto a second stream and joined — replay times.
usage: graph_branch_probe.py [chain_len=20]"""
import sys

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
a = torch.randn(64, 4096, device=dev)
b = torch.randn(64, 4096, device=dev)
big1 = torch.randn(32 << 20, device=dev)
big2 = torch.randn(32 << 20, device=dev)


def chain(t):
    for _ in range(n):
        t = t * 1.0001 + 0.5
    return t


def chain_big(t):
    for _ in range(n):
        t.mul_(1.0001)
    return t


def timed(fn, iters=50):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


side = torch.cuda.Stream()


def serial(f, x, y):
    f(x)
    f(y)


def forked(f, x, y):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        f(y)
    f(x)
    main.wait_stream(side)


for name, f, x, y in (("small (launch-bound) kernels", chain, a, b), ("128 MB streaming kernels", chain_big, big1, big2)):
    t1 = timed(lambda: serial(f, x, y))
    t2 = timed(lambda: forked(f, x, y))
    print(f"{name}: 2 x {n} kernels  one stream {t1:8.1f} us   forked {t2:8.1f} us   ratio {t2 / t1:.2f}", flush=True)
