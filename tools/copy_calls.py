#!/usr/bin/env python3
"""Every device-side aten copy/clone/contiguous of one eager train step (forward AND backward), with shapes and the
package frame that issued it — device-to-device copies become memcpy nodes in the captured step."""
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

dev = torch.device("cuda", 0)
model, _ = bench.build_model(dev)
opt = FlatAdam(model.parameters(), lr=1e-3)
data = Batch.from_data_list(synth.brain_graph_list(256, seed=1000, rois=90, tsne_dim=90)).to(dev)
for _ in range(2):
    data._igcn_plan = None
    train_step(model, opt, data)
torch.cuda.synchronize()
cnt = Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        views = ("view", "_unsafe_view", "reshape", "expand", "t", "transpose", "permute", "slice", "select", "detach",
                 "alias", "as_strided", "unsqueeze", "squeeze", "split", "split_with_sizes", "unbind", "empty",
                 "empty_like", "empty_strided", "new_empty", "narrow", "chunk", "is_same_size", "sym_size", "stride",
                 "size", "numel", "dim", "_local_scalar_dense", "item", "lift_fresh", "unfold", "flatten", "view_as",
                 "record_stream", "is_pinned", "storage_offset", "sym_numel", "sym_stride", "sym_storage_offset",
                 "is_contiguous", "result_type", "movedim")
        if name not in views:
            t = next((a for a in args if isinstance(a, torch.Tensor)), None)
            if t is not None and t.is_cuda:
                fr = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()
                      if "ig-gcn_amd" in f.filename or "igcn_amd" in f.filename][-2:]
                cnt[(name, tuple(t.shape), str(t.dtype).replace("torch.", ""), " <- ".join(fr) or "autograd engine")] += 1
        return func(*args, **(kwargs or {}))


with Log():
    data._igcn_plan = None
    train_step(model, opt, data)
torch.cuda.synchronize()
for (name, shp, dt, where), c in sorted(cnt.items(), key=lambda kv: (kv[0][3], kv[0][0])):
    print(f"{c:3d} {name:10s} {str(shp):22s} {dt:8s} {where}")
print("total:", sum(cnt.values()))
