#!/usr/bin/env python3
"""Time igcn_gemm_f32 on every (M, N, K, split) the bench train step issues (list traced from one eager step),
one shape at a time in a hipGraph of 20 calls; prints us per call and the share of the step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import ops  # noqa: E402

# (count per step, M, N, K, form) — form guessed from the operand roles: nt = activations x weight^T,
# nn = grad x weight, tn = weight gradient (reduction over the long axis)
SHAPES = [
    (4, 46080, 32, 32, "nt"), (2, 512, 64, 3, "nt"), (2, 512, 32, 32, "nt"), (2, 512, 3, 64, "nt"),
    (2, 46080, 16, 16, "nt"), (2, 32, 32, 46080, "tn"), (2, 3, 64, 512, "tn"), (2, 256, 2880, 256, "nn"),
    (2, 256, 256, 2880, "nt"), (1, 64, 32, 204800, "tn"), (1, 64, 3182, 512, "tn"), (1, 64, 2912, 512, "tn"),
    (1, 6000, 54, 512, "tn"), (1, 54, 3000, 512, "tn"), (1, 512, 64, 3182, "nt"), (1, 512, 64, 2912, "nt"),
    (1, 512, 6000, 54, "nt"), (1, 512, 54, 6000, "nt"), (1, 512, 54, 3000, "nn"), (1, 512, 400, 32, "nt"),
    (1, 512, 32, 400, "nn"), (1, 512, 3182, 64, "nn"), (1, 512, 3000, 54, "nt"), (1, 512, 2912, 64, "nn"),
    (1, 46080, 3, 16, "nt"), (1, 46080, 16, 3, "nn"), (1, 32, 400, 512, "tn"), (1, 32, 32, 512, "tn"),
    (1, 204800, 64, 32, "nt"), (1, 204800, 32, 64, "nn"), (1, 16, 3, 46080, "tn"), (1, 16, 16, 46080, "tn"),
]
dev = "cuda"
tot = 0.0
rows = []
for cnt, m, n, k, form in SHAPES:
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
    byts = 4.0 * (m * k + n * k + m * n)
    tot += cnt * us
    rows.append((cnt * us, cnt, m, n, k, form, us, byts / us / 1e3, ops._split_k(m, n, k)))
for t, cnt, m, n, k, form, us, gbs, sk in sorted(rows, reverse=True):
    print(f"{cnt}x {form} M={m:7d} N={n:5d} K={k:7d} split={sk:4d} {us:7.2f} us  {gbs:7.1f} GB/s   {t:7.1f} us/step", flush=True)
print(f"total {tot:.1f} us/step")
