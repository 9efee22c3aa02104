#!/usr/bin/env python3
"""Is a gradient difference of the model under the split-bf16 attention core a ReLU decision or an inaccuracy?
Runs tests/test_gpu_model.py::test_full_model_vs_oracle_larger's computation (eval mode, B = 32, isExplain=True) with the
attention core's forward / backward routed to the exact-fp32 or the split kernels in all four combinations and prints
the worst parameter-gradient errors against the fp64 oracle, and the smallest |pre-activation| of the ReLU behind the
attention's output projection."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import igcn_amd  # noqa: E402,F401
from igcn_amd import ops, synth  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402
from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP  # noqa: E402
from oracle import go_network as OG, sgcn_img_snp as OS  # noqa: E402
from _weights import seeded_state  # noqa: E402

bsz, pool, explain = 32, (300, 120, 60, 19, 1), True
go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=1)
a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
graphs = synth.brain_graph_list(bsz, seed=77, rois=90, tsne_dim=16)
rng = np.random.default_rng(9)
real_call = ops.call
relu_out = {}


def run(fwd, bwd):
    def routed(name, *args):
        if name == "igcn_attn_core_fwd":
            name = fwd
        elif name == "igcn_attn_core_bwd":
            name = bwd
        return real_call(name, *args)
    ops.call = routed
    model = SGCN_GCN_IMGSNP(2, 16, a_g, a, pool_dim, 32, "cuda", rois=90, H_0=3, num_classes=3, isSoftSimilarity=True,
                            rbf_gamma=0.01, isCrossAtten=True, num_regr=3, isuseProb4Regr=True, isImageOnly=False,
                            isSNPsOnly=False).cuda().eval()
    sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, 5)
    model.load_state_dict(sd)
    data = Batch.from_data_list(graphs).to("cuda")
    seen = []
    real_ca = model._cross_attention

    def spy(qq, mm):
        r_ = real_ca(qq, mm)
        seen.append(r_.detach().cpu())
        return r_
    model._cross_attention = spy
    outs = model(data, None, "cuda", isExplain=explain)
    relu_out[fwd] = seen[0]
    r = np.random.default_rng(9)
    cot = [torch.from_numpy(r.standard_normal(tuple(o.shape))).float() for o in outs]
    sum((o * c.cuda()).sum() for o, c in zip(outs, cot)).backward()
    ops.call = real_call
    acts = {}
    return sd, cot, {k: p.grad.detach().cpu().double() for k, p in model.named_parameters() if p.grad is not None}, \
        [o.detach().cpu().double() for o in outs]


sd, cot, g_exact, _ = run("igcn_attn_core_fwd", "igcn_attn_core_bwd") if os.environ.get("IGCN_ATTN_EXACT_FP32") else (None,) * 4
a_g_c, a_c = synth.go_sparse_inputs(go_snps, adj)
idx = OG.go_index_sets(a_g_c, a_c, list(pool), 2)
res, outs_ = {}, {}
for tag, (f, b) in {"split/exact": ("igcn_attn_core_split_fwd", "igcn_attn_core_bwd"),
                    "exact/exact": ("igcn_attn_core_fwd", "igcn_attn_core_bwd")}.items():
    sd, cot, res[tag], outs_[tag] = run(f, b)
sdo = OS.make_leaf_state(sd, dtype=torch.float64)
dcpu = Batch.from_data_list(graphs)
dcpu.x = dcpu.x.double().requires_grad_(True)
dcpu.edge_attr, dcpu.snps_feat = dcpu.edge_attr.double(), dcpu.snps_feat.double()
cfg = SimpleNamespace(num_layers=2, rois=90, image_only=False, rbf_gamma=0.01)
ref = OS.model_forward(sdo, cfg, idx, dcpu, explain, training=False)
sum((o * c.double()).sum() for o, c in zip(ref, cot)).backward()
for tag, g in res.items():
    errs = []
    for k in OS.trainable_keys(sdo):
        if sdo[k].grad is None or k not in g:
            continue
        w = sdo[k].grad
        errs.append((float((g[k] - w).abs().max() / max(float(w.abs().max()), 1e-6)), k))
    errs.sort(reverse=True)
    print(tag, "  ".join(f"{k}={e:.1e}" for e, k in errs[:4]), flush=True)

a, b_ = outs_["split/exact"], outs_["exact/exact"]
for i, (x, y) in enumerate(zip(a, b_)):
    print("output", i, "split vs exact rel", float((x - y).abs().max() / y.abs().max()), " vs oracle",
          float((x - ref[i].detach()).abs().max() / ref[i].detach().abs().max()), float((y - ref[i].detach()).abs().max() / ref[i].detach().abs().max()))
# which elements of the gradient differ: a single spike (one decision) or everywhere (precision)?
k = "go_network.G_B.0.bias"
d = (res["split/exact"][k] - res["exact/exact"][k]).abs()
print(k, "diff: max", float(d.max()), "median", float(d.median()), "n > 10% of max:", int((d > 0.1 * d.max()).sum()), "of", d.numel())

x, y = relu_out["igcn_attn_core_split_fwd"], relu_out["igcn_attn_core_fwd"]
flip = (x > 0) != (y > 0)
print("relu(out_proj) decisions that differ:", int(flip.sum()), "of", flip.numel(), "; largest value among them:",
      float(torch.maximum(x, y)[flip].max()) if flip.any() else 0.0, "; scale", float(y.max()))
