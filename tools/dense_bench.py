#!/usr/bin/env python3
"""The dense-block SGCN op (ops.DenseSgcn) on the configs[4] shape, forward + backward, for per-kernel timing:
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 <repo>/tools/dense_bench.py [mode] [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import ops, synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rois, g = 512, 32
b = synth.brain_batch(g, seed=1, rois=rois, tsne_dim=8, dense=True).to("cuda")
plan = ops.plan_for(b)
assert plan.dense_blocks
torch.manual_seed(0)
dev = "cuda"
prob = torch.randn(rois, 3, device=dev, requires_grad=True)
pb = torch.randn(6, 1, device=dev, requires_grad=True)
sp = torch.randn(1, 54, device=dev, requires_grad=True)
w0 = (torch.randn(16, 3, device=dev) * 0.5).requires_grad_(True)
b0 = torch.zeros(16, device=dev, requires_grad=True)
w1 = (torch.randn(16, 16, device=dev) * 0.3).requires_grad_(True)
b1 = torch.zeros(16, device=dev, requires_grad=True)
x = b.x.clone().requires_grad_(True)
hp = (0.1, 0.1, 0.1, 0.1, 1e-6)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(iters + 3):
    if it == 3:
        torch.cuda.synchronize()
        e0.record()
    xcat, regp = ops.DenseSgcn.apply(x, b.edge_attr, prob, pb, sp, mode, rois, hp, None, w0, b0, w1, b1)
    loss = xcat.sum() + (regp.sum() if regp.numel() else 0.0)
    loss.backward()
e1.record()
torch.cuda.synchronize()
print(f"{mode}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us per forward + backward (eager, incl. torch glue)")
