#!/usr/bin/env python3
"""Print every libigcn entry point of ONE eager train step with its integer arguments (IGCN_DEBUG_SYNC tracing)."""
import os
import sys

os.environ["IGCN_DEBUG_SYNC"] = "0"
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from igcn_amd import _lib, synth  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402
from igcn_amd.train import FlatAdam, train_step  # noqa: E402

dev = torch.device("cuda", 0)
model, _ = bench.build_model(dev)
opt = FlatAdam(model.parameters(), lr=1e-3)
data = Batch.from_data_list(synth.brain_graph_list(256, seed=1000, rois=90, tsne_dim=90)).to(dev)
for _ in range(2):
    train_step(model, opt, data)
torch.cuda.synchronize()
orig = _lib.call


def traced(name, *args):
    ints = [a for a in args if isinstance(a, int) and not isinstance(a, bool) and abs(a) < (1 << 31)]
    print(name, ints)
    return orig(name, *args)


import igcn_amd.ops as ops  # noqa: E402
import igcn_amd.train as train  # noqa: E402
for mod in (ops, train, _lib):
    if hasattr(mod, "call"):
        mod.call = traced
train_step(model, opt, data)
torch.cuda.synchronize()
