#!/usr/bin/env python3
"""Phase timestamps of k_sgcn_front_fwd (library built with IGCN_HIPCC_EXTRA=-DSF_PROBE_ON): wall_clock64 (100 MHz) of
thread 0 of the first 8 workgroups at the phase boundaries, at the bench shape (2 x 256 graphs x 90 ROIs x 270 edges)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from igcn_amd import _lib, ops, synth  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402

dev = torch.device("cuda", 0)
rois, g, h0, f, layers = 90, 256, 3, 16, 2
data = Batch.from_data_list(synth.brain_graph_list(g, seed=1, rois=rois, tsne_dim=8)).to(dev)
plan = ops.plan_for(data)
prob, pb, snps = torch.randn(rois, h0, device=dev), torch.randn(2 * h0, 1, device=dev), torch.randn(1, 54, device=dev)
ws = [torch.randn(f, h0 if l == 0 else f, device=dev) * 0.3 for l in range(layers)]
bs = [torch.randn(f, device=dev) * 0.1 for _ in range(layers)]
wb = [t for pair in zip(ws, bs) for t in pair]
big = torch.empty(64 * 1024 * 1024, device=dev)
for it in range(6):
    big.fill_(float(it))                                  # evict the batch from the caches: the in-step condition
    ops.SgcnFront.apply(data.x, prob, pb, data.edge_attr, plan, rois, snps, (0.1, 0.1, 0.1, 0.1, 1e-6), data.snps_feat,
                        data.edge_index, *wb)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 128)()
print("rc", raw.igcn_debug_sf_probe(buf))
names = ["stage", "histogram+scan", "placement+mask", "ew*e, lists", "layers", "regulariser", "stores"]
for wg in range(8):
    t = [buf[wg * 16 + i] for i in range(8)]
    print(f"  wg {wg}: start {(t[0] - buf[0]) * 10:6d} ns  " +
          "  ".join(f"{nm} {(t[i + 1] - t[i]) * 10:5d}" for i, nm in enumerate(names)) + f"  total {(t[-1] - t[0]) * 10} ns")
