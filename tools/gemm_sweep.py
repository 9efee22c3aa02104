#!/usr/bin/env python3
"""Sweep (tile width cap, split-K) of igcn_gemm_f32 over the GEMM shapes of the bench train step: which launch
configuration is fastest per shape (the host heuristics in ops._split_k / gemm.hip are fitted to this table)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import ops  # noqa: E402
from igcn_amd._lib import call, ptr, stream_ptr  # noqa: E402
from gemm_step_shapes import SHAPES  # noqa: E402

dev = "cuda"


def run(form, a, b, m, n, k, sk, out, scratch):
    if form == "nt":
        call("igcn_gemm_f32", m, n, k, ptr(a), k, 1, ptr(b), k, 1, None, ptr(out), n, 0, sk, ptr(scratch), stream_ptr())
    elif form == "nn":
        call("igcn_gemm_f32", m, n, k, ptr(a), k, 1, ptr(b), 1, n, None, ptr(out), n, 0, sk, ptr(scratch), stream_ptr())
    else:
        call("igcn_gemm_f32", m, n, k, ptr(a), 1, m, ptr(b), 1, n, None, ptr(out), n, 0, sk, ptr(scratch), stream_ptr())


def timeit(fn, iters=20):
    for _ in range(2):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
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


tot_cur = tot_best = 0.0
for cnt, m, n, k, form in SHAPES:
    if form == "nt":
        a, b = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)
    elif form == "nn":
        a, b = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev)
    else:
        a, b = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev)
    out = torch.empty(m, n, device=dev)
    res = {}
    splits = sorted({1, 2, 4, 8, 16, 32, 64, 128, 256, 512, ops._split_k(m, n, k)})
    for cap in (64, 32, 16):
        os.environ["IGCN_GEMM_BN"] = str(cap)
        for sk in splits:
            if sk > max(1, k // 32) or sk * m * n > (1 << 27):
                continue
            scratch = torch.empty(sk * m * n, device=dev) if sk > 1 else None
            res[(cap, sk)] = timeit(lambda: run(form, a, b, m, n, k, sk, out, scratch))
    cur = res.get((64, ops._split_k(m, n, k)))
    best = min(res, key=res.get)
    tot_cur += cnt * cur
    tot_best += cnt * res[best]
    top = sorted(res.items(), key=lambda kv: kv[1])[:4]
    print(f"{cnt}x {form} M={m:7d} N={n:5d} K={k:7d}  cur(split {ops._split_k(m, n, k):3d}) {cur:6.2f}  best "
          + "  ".join(f"bn{c}/s{s}:{t:5.2f}" for (c, s), t in top), flush=True)
print(f"total current {tot_cur:.1f} us/step, best-of-sweep {tot_best:.1f} us/step")
