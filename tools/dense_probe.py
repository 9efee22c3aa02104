#!/usr/bin/env python3
"""Phase times inside k_ds_agg (dense-block aggregation, configs[4] shape): build with IGCN_HIPCC_EXTRA=-DDS_PROBE_ON.
Stamps per wave of the 8 workgroups of graph 0 (ns after the wave's start): 1 the first step's products issued (its
data has arrived), 2 the first chunk's (8 of 16 steps), 3 walk done, 4 barrier passed, 5 reduced +
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
if mode == "step":
    # the stamps of the LAST k_ds_agg launch of a captured configs[4] train step after 20 back-to-back replays: the kernel
    # as the step runs it (clocks, caches and neighbours of the replay, not of an idle chip)
    sys.path.insert(0, ROOT)
    import bench  # noqa: E402
    from igcn_amd.data import Batch  # noqa: E402
    from igcn_amd.train import FlatAdam, GraphedTrainStep  # noqa: E402
    wl = bench.WORKLOADS["stress"]
    model, _ = bench.build_model(torch.device("cuda", 0), wl, wl["bf16"])
    opt = FlatAdam(model.parameters(), lr=1e-3)
    data = Batch.from_data_list(synth.brain_graph_list(wl["graphs"], seed=1000, rois=wl["rois"], tsne_dim=90,
                                                       dense=wl["dense"])).to("cuda")
    data.x.requires_grad_(True)
    step = GraphedTrainStep(model, opt, data)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
cold = torch.zeros(160 * 1024 * 1024, device=dev) if os.environ.get("DS_PROBE_COLD", "1") == "1" else None   # 640 MB
for _ in range(5 if mode != "step" else 0):
    if cold is not None:
        cold.add_(1.0)                    # evicts the 256 MB Infinity Cache: ew then comes from HBM, as inside the step
    ops.DenseSgcn.apply(b.x, b.edge_attr, prob, pb, sp, mode, rois, hp, None, w0, b0)        # ONE layer: one k_ds_agg launch
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 1024)()
print("rc", raw.igcn_debug_ds_probe(buf))
t0 = min(buf[(wg * 8 + w) * 16] for wg in range(8) for w in range(8))
names = {1: "step 0 done", 2: "chunk 0 done", 3: "walk done", 4: "barrier passed", 5: "reduced + stored", 6: "next layer's operands"}
for wg in (0, 5):
    for w in range(8):
        t = [buf[(wg * 8 + w) * 16 + i] for i in range(7)]
        c = [buf[(wg * 8 + w) * 16 + 8 + i] for i in range(7)]
        print(f"wg {wg} wave {w}: start +{(t[0] - t0) * 10:5d} ns  " +
              "  ".join(f"{nm} +{(t[i] - t[0]) * 10:5d}" for i, nm in names.items() if t[i] >= t[0]) +
              f"   shader clock over the walk {(c[3] - c[0]) / max(1, (t[3] - t[0]) * 10):.2f} GHz")
