#!/usr/bin/env python3
"""Where does a graphed train step spend its time: host-side hipGraphLaunch vs device execution."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from igcn_amd import synth  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402
from igcn_amd.train import FlatAdam, GraphedTrainStep  # noqa: E402

dev = torch.device("cuda", 0)
model, _ = bench.build_model(dev)
opt = FlatAdam(model.parameters(), lr=1e-3)
data = Batch.from_data_list(synth.brain_graph_list(256, seed=1000, rois=90, tsne_dim=90)).to(dev)
data.x.requires_grad_(True)
step = GraphedTrainStep(model, opt, data)
for _ in range(5):
    step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue of {n} replays: {(t1-t0)/n*1e3:.3f} ms each; until device idle: {(t2-t0)/n*1e3:.3f} ms each")
# one replay at a time, device time by events
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    e0.record()
    step()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("single replay, event-timed (ms):", " ".join(f"{t:.3f}" for t in ts))
