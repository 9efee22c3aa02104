#!/usr/bin/env python3
"""igcn_gemm_f32 (exact-fp32 MFMA) on the shapes of the train step: time, TFLOP/s against the 157 TFLOP/s fp32
matrix peak, and GB/s of compulsory traffic against 8 TB/s — which of the two bounds each shape sits on."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import ops  # noqa: E402

SHAPES = [  # (what, M, N, K, form)
    ("GCN layer 2 transform  X W^T", 46080, 16, 16, "nt"),
    ("key|value projection", 204800, 64, 32, "nt"),
    ("key|value projection dX", 204800, 32, 64, "nn"),
    ("key|value projection dW", 64, 32, 204800, "tn"),
    ("lin1  [2B,2912] -> 64", 512, 64, 2912, "nt"),
    ("lin1 dW", 64, 2912, 512, "tn"),
    ("Gram matrix s s^T (one pass)", 256, 256, 2880, "nt"),
    ("gene map  x T^T", 512, 6000, 54, "nt"),
    ("gene map dx = dy T", 512, 54, 6000, "nn"),
    ("lin1_regr [2B,3182] -> 64", 512, 64, 3182, "nt"),
    ("lin1 dX", 512, 2912, 64, "nn"),
    ("Gram backward  S s", 256, 2880, 256, "nn"),
    ("latent 0  [2B,400] -> 32", 512, 32, 400, "nt"),
    ("latent 0 dW", 32, 400, 512, "tn"),
    ("q projection", 46080, 32, 32, "nt"),
    ("q projection dW", 32, 32, 46080, "tn"),
    ("square 4096 (reference point)", 4096, 4096, 4096, "nt"),
]
dev = "cuda"
out = []
for what, m, n, k, form in SHAPES:
    if form == "nt":
        a, b = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)
        fn = lambda: ops.gemm_nt(a, b)                      # noqa: E731
    elif form == "nn":
        a, b = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev)
        fn = lambda: ops.gemm_nn(a, b)                      # noqa: E731
    else:
        a, b = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev)
        fn = lambda: ops.gemm_tn(a, b)                      # noqa: E731
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    flops, byts = 2.0 * m * n * k, 4.0 * (m * k + n * k + m * n)
    row = {"shape": what, "M": m, "N": n, "K": k, "us": round(us, 2), "tflops": round(flops / us / 1e6, 2),
           "mfma_frac": round(flops / us / 1e6 / 157.3, 4), "gbs": round(byts / us / 1e3, 1),
           "hbm_frac": round(byts / us / 1e3 / 8000.0, 4)}
    out.append(row)
    print(f"{what:34s} M={m:7d} N={n:5d} K={k:7d} {us:8.2f} us  {row['tflops']:7.2f} TFLOP/s ({100*row['mfma_frac']:5.1f}% MFMA)"
          f"  {row['gbs']:7.1f} GB/s ({100*row['hbm_frac']:5.1f}% HBM)", flush=True)
print(json.dumps(out))
