#!/usr/bin/env python3
"""Time the scatter-aggregate entry points at the stress shape (64 dense 512-ROI graphs = both passes of a step):
forward, backward (dh + dbias + coefficient gradients), hot replays between HIP events.  IGCN_PROPAGATE_NO_LDS=1
selects the wave-per-target kernels for an A/B."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from igcn_amd import ops, synth  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402

dev = torch.device("cuda", 0)
rois, g, f = 512, int(os.environ.get("GRAPHS", "32")), 16
data = Batch.from_data_list(synth.brain_graph_list(g, seed=1, rois=rois, tsne_dim=8, dense=True)).to(dev)
plan = ops.plan_for(data).replicate(2)
n = 2 * data.x.shape[0]
ew = torch.cat([data.edge_attr, data.edge_attr]).requires_grad_(True)
coef = ops.GcnNorm.apply(ew, plan)
h = torch.randn(n, f, device=dev, requires_grad=True)
bias = torch.zeros(f, device=dev, requires_grad=True)
cot = torch.randn(n, f, device=dev)
iters = 10


def fwd():
    for _ in range(iters):
        ops.GcnPropagate.apply(h.detach(), coef[0].detach(), coef[1].detach(), bias.detach(), plan, True, coef[2],
                               coef[3])


out = ops.GcnPropagate.apply(h, coef[0], coef[1], bias, plan, True, coef[2], coef[3])


def bwd():
    for _ in range(iters):
        torch.autograd.grad(out, (h, coef[0], coef[1], bias), cot, retain_graph=True)


e = plan.n_edges
kern_bytes = 8 * e + 8 * n * f + 8 * n
us = bench._time_graph(fwd) / iters
print(f"fwd: {us:.1f} us per call; fwd-kernel bytes {kern_bytes / 1e6:.0f} MB -> {kern_bytes / us / 1e6:.2f} TB/s")
bwd()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
bwd()
e1.record()
torch.cuda.synchronize()
print(f"bwd (mask + dh + dbias + dw, eager): {e0.elapsed_time(e1) * 1e3 / iters:.1f} us per call")
