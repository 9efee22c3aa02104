#!/usr/bin/env python3
"""List the torch (non-libigcn) operator calls of one eager train step with their shapes and the repo line that
issued them (forward) or 'backward' (autograd thread).  View/metadata ops are skipped."""
import os
import sys
import traceback
from collections import Counter

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from igcn_amd import synth  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402
from igcn_amd.train import FlatAdam, train_step  # noqa: E402

SKIP = ("view", "reshape", "expand", "permute", "transpose", "aten.t.default", "detach", "alias", "as_strided", "select",
        "slice", "unsqueeze", "squeeze", "empty", "size", "stride", "is_", "_unsafe_view", "unbind", "split", "narrow",
        "record_stream", "lift_fresh", "_local_scalar", "set_", "resize_", "chunk", "unfold", "new_empty", "sym_")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in SKIP):
            shapes = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))
            where = "backward"
            for fr in reversed(traceback.extract_stack()):
                if "ig-gcn_amd" in fr.filename or "igcn_amd" in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            self.cnt[(where, name, shapes)] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda", 0)
model, _ = bench.build_model(dev)
opt = FlatAdam(model.parameters(), lr=1e-3)
data = Batch.from_data_list(synth.brain_graph_list(256, seed=1000, rois=90, tsne_dim=90)).to(dev)
for _ in range(2):
    train_step(model, opt, data)
torch.cuda.synchronize()
with Log() as log:
    train_step(model, opt, data)
torch.cuda.synchronize()
for (where, name, shapes), c in sorted(log.cnt.items()):
    print(f"{c:3d} {where:28s} {name:40s} {shapes}")
print("total:", sum(log.cnt.values()))
