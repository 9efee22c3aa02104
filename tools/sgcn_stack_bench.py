#!/usr/bin/env python3
"""Time the LDS-resident SGCN stack (igcn_sgcn_stack_fwd / _bwd) at the bench shape: 512 graphs (both passes of a
step) x 90 ROIs x 270 edges, H0=3, F=16, L=2.  Direct C-ABI calls, hot replays between HIP events."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from igcn_amd import _lib, ops, synth  # noqa: E402
from igcn_amd._lib import call, ptr, stream_ptr  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402

dev = torch.device("cuda", 0)
rois, g, h0, f, layers = 90, 256, 3, 16, 2
data = Batch.from_data_list(synth.brain_graph_list(g, seed=1, rois=rois, tsne_dim=8)).to(dev)
plan = ops.plan_for(data).replicate(2)
n, e = 2 * data.x.shape[0], plan.n_edges
x = torch.rand(n, h0, device=dev)
ew = torch.cat([data.edge_attr, data.edge_attr])
ws = [torch.randn(f, h0 if l == 0 else f, device=dev) * 0.3 for l in range(layers)]
bs = [torch.randn(f, device=dev) * 0.1 for _ in range(layers)]
wp = (ctypes.c_void_p * layers)(*[w.data_ptr() for w in ws])
bp = (ctypes.c_void_p * layers)(*[b.data_ptr() for b in bs])
xcat = torch.empty(n, layers * f, device=dev)
dxcat = torch.randn(n, layers * f, device=dev)
dx, dew = torch.empty_like(x), torch.empty_like(ew)
npar = int(_lib.load().igcn_sgcn_stack_param_floats(h0, f, layers))
dpar = torch.empty(npar, device=dev)
scratch = torch.empty(2 * g * npar, device=dev)
emax = plan._stack_dims[1]
iters = 20


def fwd():
    for _ in range(iters):
        call("igcn_sgcn_stack_fwd", n // rois, rois, emax, h0, f, layers, ptr(x), ptr(ew), ptr(plan.src32), ptr(plan.dst32),
             ptr(plan.tgt_ptr), ptr(plan.tgt_perm), ptr(plan.loop_edge), wp, bp, ptr(xcat), None, stream_ptr())


def bwd():
    for _ in range(iters):
        call("igcn_sgcn_stack_bwd", n // rois, rois, emax, h0, f, layers, ptr(x), ptr(ew), ptr(plan.src32), ptr(plan.dst32),
             ptr(plan.tgt_ptr), ptr(plan.tgt_perm), ptr(plan.src_ptr), ptr(plan.src_perm), ptr(plan.loop_edge), wp, bp,
             ptr(dxcat), None, ptr(dx), ptr(dew), ptr(dpar), ptr(scratch), None, stream_ptr())


alg = (n // rois) * (4 * rois * h0 + 20 * (e // (n // rois)) + 4 * rois * layers * f)
for name, fn in (("fwd", fwd), ("bwd (+ its reduce)", bwd)):
    us = bench._time_graph(fn) / iters
    print(f"{name}: {us:.2f} us per launch; SURVEY 8d fused lower bound {alg / 1e6:.1f} MB -> {alg / us / 1e6:.2f} TB/s")
