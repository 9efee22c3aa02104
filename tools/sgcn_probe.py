#!/usr/bin/env python3
"""Phase timestamps of k_sgcn_stack_fwd (library built with IGCN_HIPCC_EXTRA=-DSF_PROBE_ON): wall_clock64 (100 MHz) of
thread 0 of the first 8 workgroups at the phase boundaries, at the bench shape (512 graphs x 90 ROIs x 270 edges)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from igcn_amd import _lib, ops, synth  # noqa: E402
from igcn_amd._lib import call, ptr, stream_ptr  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402

dev = torch.device("cuda", 0)
rois, g, h0, f, layers = 90, 256, 3, 16, 2
data = Batch.from_data_list(synth.brain_graph_list(g, seed=1, rois=rois, tsne_dim=8)).to(dev)
plan = ops.plan_for(data).replicate(2)
n = 2 * data.x.shape[0]
x = torch.rand(n, h0, device=dev)
ew = torch.cat([data.edge_attr, data.edge_attr])
ws = [torch.randn(f, h0 if l == 0 else f, device=dev) * 0.3 for l in range(layers)]
bs = [torch.randn(f, device=dev) * 0.1 for _ in range(layers)]
wp = (ctypes.c_void_p * layers)(*[w.data_ptr() for w in ws])
bp = (ctypes.c_void_p * layers)(*[b.data_ptr() for b in bs])
xcat = torch.empty(n, layers * f, device=dev)
emax = plan._stack_dims[1]
for _ in range(5):
    call("igcn_sgcn_stack_fwd", n // rois, rois, emax, h0, f, layers, ptr(x), ptr(ew), ptr(plan.src32), ptr(plan.dst32),
         ptr(plan.tgt_ptr), ptr(plan.tgt_perm), ptr(plan.loop_edge), wp, bp, ptr(xcat), None, stream_ptr())
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 128)()
print("rc", raw.igcn_debug_sf_probe(buf))
names = ["stage loads", "gcn_norm a", "norm b", "layer 0", "layer 1", "store"]
idx = [0, 1, 2, 3, 4, 5, 8]
for wg in range(8):
    t = [buf[wg * 16 + i] for i in idx]
    print(f"  wg {wg}: start {(t[0] - buf[0]) * 10:6d} ns  " +
          "  ".join(f"{nm} {(t[i + 1] - t[i]) * 10:5d}" for i, nm in enumerate(names)) + f"  total {(t[-1] - t[0]) * 10} ns")

# ---- backward
dxcat = torch.randn(n, layers * f, device=dev)
dx, dew = torch.empty_like(x), torch.empty_like(ew)
npar = int(_lib.load().igcn_sgcn_stack_param_floats(h0, f, layers))
dpar = torch.empty(npar, device=dev)
scratch = torch.empty((n // rois) * npar, device=dev)
for _ in range(5):
    call("igcn_sgcn_stack_bwd", n // rois, rois, emax, h0, f, layers, ptr(x), ptr(ew), ptr(plan.src32), ptr(plan.dst32),
         ptr(plan.tgt_ptr), ptr(plan.tgt_perm), ptr(plan.src_ptr), ptr(plan.src_perm), ptr(plan.loop_edge), wp, bp,
         ptr(dxcat), None, ptr(dx), ptr(dew), ptr(dpar), ptr(scratch), None, stream_ptr())
torch.cuda.synchronize()
raw.igcn_debug_sf_probe(buf)
names = ["staging+norm", "forward", "bwd layer 1", "bwd layer 0", "norm bwd + stores"]
idx = [8, 9, 10, 11, 12, 13, 14]
print("backward: stamps 8 start, 9 staged, 10 forward done, 11/12 backward layers begin, 13 norm backward begins, 14 end")
for wg in range(8):
    t = [buf[wg * 16 + i] for i in idx]
    print(f"  wg {wg}: " + "  ".join(f"{i}->{j} {(t[b2 + 1] - t[b2]) * 10:5d}" for b2, (i, j) in enumerate(zip(idx[:-1], idx[1:]))) +
          f"  total {(t[-1] - t[0]) * 10} ns")
