#!/usr/bin/env python3
"""How far the bf16-transform path (igcn_gemm_bf16 behind bf16_transforms=True) sits from the fp32 path at the
configs[4] shape: max scale-relative difference of every forward output and every gradient, B=2 (the numbers quoted
beside BF16_TOL / BF16_GTOL in tests/test_gpu_stress.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import igcn_amd  # noqa: E402,F401
from igcn_amd import synth  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402
import test_gpu_stress as T  # noqa: E402

go = synth.go_hierarchy(T.POOL, seed=1)
graphs = synth.brain_graph_list(2, seed=77, rois=T.ROIS, tsne_dim=16, dense=True)
res = {}
for bf in (False, True):
    model, _ = T._model(go, bf)
    data = Batch.from_data_list(graphs).to("cuda")
    outs = model(data, None, "cuda", isExplain=True)
    cot = T._probe(outs, 9)
    sum((o * c.cuda()).sum() for o, c in zip(outs, cot)).backward()
    res[bf] = ([o.detach() for o in outs], {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
               data.x.grad.clone())
rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))      # noqa: E731
print("outputs:", {n: f"{rel(a, b):.2e}" for n, a, b in zip(T.NAMES, res[True][0], res[False][0])})
g = {k: rel(res[True][1][k], v) for k, v in res[False][1].items() if float(v.abs().max()) > 1e-6}
print("grad data.x:", f"{rel(res[True][2], res[False][2]):.2e}", " worst parameter gradients:",
      [(f"{v:.2e}", k) for v, k in sorted(((v, k) for k, v in g.items()), reverse=True)[:5]])
