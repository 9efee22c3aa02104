#!/usr/bin/env python3
"""List the torch (non-libigcn) device launches of one eager train step with the Python line that issued them."""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

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
for _ in range(3):
    data._igcn_plan = None
    train_step(model, opt, data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    data._igcn_plan = None
    train_step(model, opt, data)
    torch.cuda.synchronize()
cnt = Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and len(ev.kernels) > 0 \
            and not any(len(c.kernels) > 0 for c in ev.cpu_children):
        where = "?"
        for fr in ev.stack:
            if "ig-gcn_amd" in fr or "igcn_amd" in fr:
                where = fr.split("/")[-1]
                break
        if where == "?" and ev.stack:
            where = "autograd:" + ev.stack[0].split("/")[-1][:60]
        cnt[(ev.name, where, tuple(ev.input_shapes[0]) if getattr(ev, "input_shapes", None) else ())] += 1
for (name, where, shp), c in sorted(cnt.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"{c:3d} {name:28s} {where}")
print("total aten launches:", sum(cnt.values()))

# device-to-device copies are memcpy activities, not kernels: list every aten::copy_ with the frames that issued it
cp = Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name == "aten::copy_":
        frames = [fr.split("/")[-1][:70] for fr in ev.stack if "ig-gcn_amd" in fr or "igcn_amd" in fr][:2]
        cp[" <- ".join(frames) if frames else ("autograd:" + (ev.stack[0].split("/")[-1][:60] if ev.stack else "?"))] += 1
print("aten::copy_ calls (kernel or memcpy):")
for k, c in sorted(cp.items(), key=lambda kv: -kv[1]):
    print(f"{c:3d} {k}")
