#!/usr/bin/env python3
"""Phase times inside k_ds_agg (dense-block aggregation, configs[4] shape): build with IGCN_HIPCC_EXTRA=-DDS_PROBE_ON.
Stamps per wave of the 8 workgroups of graph 0 (ns after the wave's start): 3 walk done, 4 barrier passed, 5 reduced +
stored, 6 next layer's operands written (only when a layer follows)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import _lib, ops, synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
rois, g = 512, 32
b = synth.brain_batch(g, seed=1, rois=rois, tsne_dim=8, dense=True).to("cuda")
plan = ops.plan_for(b)
dev = "cuda"
torch.manual_seed(0)
prob = torch.randn(rois, 3, device=dev)
pb = torch.randn(6, 1, device=dev)
sp = torch.randn(1, 54, device=dev)
w0, b0 = torch.randn(16, 3, device=dev) * 0.5, torch.zeros(16, device=dev)
hp = (0.1, 0.1, 0.1, 0.1, 1e-6)
for _ in range(5):
    ops.DenseSgcn.apply(b.x, b.edge_attr, prob, pb, sp, mode, rois, hp, None, w0, b0)        # ONE layer: one k_ds_agg launch
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 512)()
print("rc", raw.igcn_debug_ds_probe(buf))
t0 = min(buf[(wg * 8 + w) * 8] for wg in range(8) for w in range(8))
names = {3: "walk done", 4: "barrier passed", 5: "reduced + stored", 6: "next layer's operands"}
for wg in (0, 5):
    for w in range(8):
        t = [buf[(wg * 8 + w) * 8 + i] for i in range(7)]
        print(f"wg {wg} wave {w}: start +{(t[0] - t0) * 10:5d} ns  " +
              "  ".join(f"{nm} +{(t[i] - t[0]) * 10:5d}" for i, nm in names.items() if t[i] >= t[0]))
