#!/usr/bin/env python3
"""List every igcn_gemm_* call of one eager train step of the bench model: shape, operand form, split-K the launch
heuristic picks, and whether the output is a final gradient (its split-K sum deferred)."""
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from igcn_amd import _lib, ops, synth  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402
from igcn_amd.train import FlatAdam, train_step  # noqa: E402

dev = torch.device("cuda", 0)
model, go = bench.build_model(dev)
opt = FlatAdam(model.parameters(), lr=1e-3)
data = Batch.from_data_list(synth.brain_graph_list(256, seed=1000, rois=90, tsne_dim=90)).to(dev)
data.x.requires_grad_(True)
train_step(model, opt, data)
log = []
orig = _lib.call


def spy(name, *args):
    if name.startswith("igcn_gemm_f32") or name.startswith("igcn_gemm_bf16"):
        m, n, k = args[0], args[1], args[2]
        if name.endswith("batched_sum"):
            log.append((name, m, n, k, "batch=%d" % args[3], "", ""))
        else:
            sam, sak, sbn, sbk = args[4], args[5], args[7], args[8]
            form = ("N" if sak == 1 else "T") + ("T" if sbk == 1 else "N")
            log.append((name, m, n, k, form, "split=%d" % args[13], "final" if args[12] & 0x100 else ""))
    return orig(name, *args)


_lib.call = spy
ops.call = spy
train_step(model, opt, data)
torch.cuda.synchronize()
for row, cnt in Counter(log).items():
    print(cnt, *row)
print(len(log), "gemm calls")
