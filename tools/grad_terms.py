#!/usr/bin/env python3
"""Which loss term carries a gradient discrepancy?  Train-mode step losses with ONE lambda switched on at a time, HIP
vs the fp64 oracle, worst parameters per term.  usage: grad_terms.py [bsz] [full|small]"""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import igcn_amd  # noqa: E402,F401
from _weights import seeded_state  # noqa: E402
from igcn_amd import synth  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402
from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP  # noqa: E402
from igcn_amd.train import losses  # noqa: E402
from oracle import go_network as OG, sgcn_img_snp as OS  # noqa: E402

bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pool = (1800, 800, 300, 99, 1) if (len(sys.argv) > 2 and sys.argv[2] == "full") else (300, 120, 60, 19, 1)
go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=1)
a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
model = SGCN_GCN_IMGSNP(2, 16, a_g, a, pool_dim, 32, "cuda", rois=90, H_0=3, num_classes=3, isSoftSimilarity=True,
                        rbf_gamma=0.01, isCrossAtten=True, num_regr=3, isuseProb4Regr=True, isImageOnly=False,
                        isSNPsOnly=False).cuda().train()
sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, 5)
for m in (model, model.go_network):
    m._dropout_enabled = False
graphs = synth.brain_graph_list(bsz, seed=78, rois=90, tsne_dim=16)
a_g_c, a_c = synth.go_sparse_inputs(go_snps, adj)
idx = OG.go_index_sets(a_g_c, a_c, list(pool), 2)
cfg = SimpleNamespace(num_layers=2, rois=90, image_only=False, rbf_gamma=0.01)
names = ["ce/mi", "reg", "prob", "recon", "cluster", "orth"]
base = [1.0, 1.0, 0.5, 1.5e-6, 0.1, 0.2]
for t in range(7):
    lam = list(base) if t == 6 else [base[i] if i == t else 0.0 for i in range(6)]
    st = OS.make_leaf_state(sd, dtype=torch.float64)
    dd = Batch.from_data_list(graphs)
    dd.x = dd.x.double().requires_grad_(True)
    dd.edge_attr, dd.snps_feat = dd.edge_attr.double(), dd.snps_feat.double()
    dd.tsne_fdim, dd.clini_score = dd.tsne_fdim.double(), dd.clini_score.double()
    lo, _, _ = OS.train_losses(st, cfg, idx, dd, lam, dropout=False)
    lo.backward()
    model.load_state_dict(sd)
    model.zero_grad()
    data = Batch.from_data_list(graphs).to("cuda")
    loss, _, _ = losses(model, data, lam)
    loss.backward()
    params = dict(model.named_parameters())
    rows = []
    for k in OS.trainable_keys(st):
        g = st[k].grad
        if g is None or params[k].grad is None:
            continue
        sc = float(g.abs().max())
        if sc == 0:
            continue
        rows.append((float((params[k].grad.cpu().double() - g).abs().max()) / sc, sc, k))
    rows.sort(reverse=True)
    print(f"--- {'all terms' if t == 6 else names[t]}: loss {float(loss):.6f} vs {float(lo):.6f}")
    for e, sc, k in rows[:5]:
        print(f"    {k:45s} rel {e:.2e}  (scale {sc:.2e})")
