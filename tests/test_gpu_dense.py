"""The dense-block SGCN path (csrc/sgcn_dense.hip, ops.DenseSgcn): batches of COMPLETE graphs — a dense adjacency as
COO, BASELINE configs[4] — whose masks, gcn_norm, GCNConv layers and mask regulariser run on ``edge_attr`` as the dense
matrix it is.  Against the fp64 oracle (cal_probability kernel/sgcn_img_snp.py:133-151, PyG GCNConv, loss_probability
:153-181) for one pass (plain / masked) and for both passes of a train step, forward and every gradient; plus the
structure check that guards the path."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import assert_matches

pytestmark = pytest.mark.gpu
HP = SimpleNamespace(lamda_x_l1=0.1, lamda_e_l1=0.2, lamda_x_ent=0.3, lamda_e_ent=0.15)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from igcn_amd import _lib
    _lib.load()


def _batch(n_graphs, rois, seed):
    from igcn_amd import synth
    return synth.brain_batch(n_graphs, seed=seed, rois=rois, tsne_dim=8, dense=True)


def test_complete_graphs_are_detected_and_anything_else_is_not():
    from igcn_amd import _lib, ops
    b = _batch(3, 64, 1).to("cuda")
    plan = ops.plan_for(b)
    assert plan.dense_blocks
    plan.check()
    # the same edges in another order: not row-major -> general kernels
    ei = b.edge_index.clone()
    ei[:, [5, 6]] = ei[:, [6, 5]]
    other = ops.GraphPlan(ei, b.x.shape[0], b.ptr, b.edge_ptr, b._max_nodes, b._max_edges)
    assert not other.dense_blocks
    # a sparse batch of the same node count: not complete
    from igcn_amd import synth
    sparse = synth.brain_batch(3, seed=1, rois=64).to("cuda")
    assert not ops.plan_for(sparse).dense_blocks
    # a dense plan re-verifies every new batch on the device
    plan.rebuild(ei)
    with pytest.raises(_lib.IgcnError, match="complete graphs"):
        plan.check()
    plan.status.zero_()
    plan.rebuild(b.edge_index)
    plan.check()


def _oracle(b, sd, layers, mode, cot, reg_w):
    """fp64: xcat of the requested passes, loss_probability, and the gradients of sum(xcat * cot) + reg_w * loss."""
    from oracle import sgcn_img_snp as OS
    from oracle.pyg_ops import gcn_conv
    st = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    x = b.x.double().clone().requires_grad_(True)
    ew, ei = b.edge_attr.double(), b.edge_index
    rois = b._max_nodes
    outs = []
    for masked in ([False] if mode == "plain" else [True] if mode == "masked" else [False, True]):
        if masked:
            xm, ewm, _ = OS.edge_and_region_masks(st, x, ei, ew, rois)
        else:
            xm, ewm = x, ew
        h, hs = xm, []
        for l in range(layers):
            w, bb = (st["conv1.lin.weight"], st["conv1.bias"]) if l == 0 else (st[f"convs.{l - 1}.lin.weight"],
                                                                                st[f"convs.{l - 1}.bias"])
            h = torch.relu(gcn_conv(h, ei, ewm, w, bb))
            hs.append(h)
        outs.append(torch.cat(hs, dim=1))
    xcat = torch.cat(outs, dim=0)
    total = (xcat * cot.double()).sum()
    reg = None
    if mode != "plain":
        reg = OS.loss_probability(st, x, ei, ew, rois, HP, eps=1e-6)
        total = total + reg_w * reg
    total.backward()
    return xcat.detach(), (reg.detach() if reg is not None else None), x.grad, {k: v.grad for k, v in st.items()}


@pytest.mark.parametrize("mode", ["plain", "masked", "both"])
@pytest.mark.parametrize("rois,n_graphs,layers", [(64, 3, 2), (128, 2, 3), (64, 2, 1)])
def test_dense_sgcn_vs_fp64_oracle(mode, rois, n_graphs, layers):
    from igcn_amd import ops
    rng = np.random.default_rng(rois + layers)
    b = _batch(n_graphs, rois, seed=rois)
    f, h0 = 16, 3
    t = lambda *s: torch.from_numpy(rng.standard_normal(s)).float()                        # noqa: E731
    sd = {"prob": t(rois, h0) * 0.7, "prob_bias": t(2 * h0, 1) * 0.8, "snps_prob": t(1, 54),
          "conv1.lin.weight": t(f, h0) * 0.6, "conv1.bias": t(f) * 0.2}
    for l in range(1, layers):
        sd[f"convs.{l - 1}.lin.weight"] = t(f, f) * 0.4
        sd[f"convs.{l - 1}.bias"] = t(f) * 0.2
    copies = 2 if mode == "both" else 1
    cot = t(copies * n_graphs * rois, layers * f)
    reg_w = 0.37
    want_xcat, want_reg, want_dx, want = _oracle(b, sd, layers, mode, cot, reg_w)

    bd = b.to("cuda")
    plan = ops.plan_for(bd)
    assert plan.dense_blocks and ops.dense_sgcn_supported(plan, rois, h0, f, layers)
    dev = {k: v.cuda().requires_grad_(True) for k, v in sd.items()}
    xg = bd.x.clone().requires_grad_(True)
    wb = []
    for l in range(layers):
        wb += [dev["conv1.lin.weight"], dev["conv1.bias"]] if l == 0 else [dev[f"convs.{l - 1}.lin.weight"],
                                                                            dev[f"convs.{l - 1}.bias"]]
    reg_hp = (HP.lamda_x_l1, HP.lamda_x_ent, HP.lamda_e_l1, HP.lamda_e_ent, 1e-6)
    xcat, regp = ops.DenseSgcn.apply(xg, bd.edge_attr, dev["prob"], dev["prob_bias"], dev["snps_prob"], mode, rois, reg_hp, None,
                                     *wb)
    assert_matches(xcat, want_xcat.numpy(), 1e-4, "xcat")
    total = (xcat * cot.cuda()).sum()
    if mode != "plain":
        assert abs(float(regp.sum()) - float(want_reg)) <= 1e-5 * abs(float(want_reg))
        total = total + reg_w * regp.sum()
    else:
        assert regp.numel() == 0
    total.backward()
    assert_matches(xg.grad, want_dx.numpy(), 1e-3, "dx")
    for k, g in want.items():
        if mode == "plain" and k in ("prob", "prob_bias", "snps_prob"):
            assert dev[k].grad is None
            continue
        assert_matches(dev[k].grad, g.numpy(), 1e-3, "grad " + k, floor=1e-6)


def test_model_takes_the_dense_path_and_matches_the_general_kernels(monkeypatch):
    """The whole model's train losses on a batch of complete graphs: dense-block path vs the general (sorted plan, record
    streams) kernels of csrc/sgcn.hip on the same weights — every loss term and gradient (both are checked against the
    fp64 oracle elsewhere; this pins the integration: pass stacking, regulariser partials, SNP mask, data.x.grad)."""
    from igcn_amd import ops, synth
    from igcn_amd.data import Batch
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    from igcn_amd.train import losses
    from _weights import seeded_state
    pool, rois = (40, 20, 10, 4, 1), 128
    go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=3)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
    model = SGCN_GCN_IMGSNP(2, 16, a_g, a, pool_dim, 32, "cuda", rois=rois, H_0=3, num_classes=3, isSoftSimilarity=True,
                            rbf_gamma=0.01, isCrossAtten=True, num_regr=3, isuseProb4Regr=True, isImageOnly=False,
                            isSNPsOnly=False).cuda().train()
    sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, 7)
    for m in (model, model.go_network):
        m._dropout_enabled = False
    graphs = synth.brain_graph_list(6, seed=4, rois=rois, tsne_dim=8, dense=True)
    lam = [1.0, 1.0, 0.5, 1.5e-6, 0.1, 0.2]
    res = {}
    for tag in ("dense", "general"):
        if tag == "general":
            monkeypatch.setenv("IGCN_NO_DENSE_BLOCKS", "1")
        for batched in (True, False):
            model.load_state_dict(sd)
            model.zero_grad()
            model.batched_passes = batched
            data = Batch.from_data_list(graphs).to("cuda")
            assert ops.plan_for(data).dense_blocks == (tag == "dense")
            loss, terms, _ = losses(model, data, lam)
            loss.backward()
            res[tag, batched] = (float(loss), {k: float(v) for k, v in terms.items()}, data.x.grad.clone(),
                                 {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    for batched in (True, False):
        ld, td, gxd, gd = res["dense", batched]
        lg, tg, gxg, gg = res["general", batched]
        assert abs(ld - lg) <= 2e-5 * max(1.0, abs(lg)), (ld, lg)
        for k in tg:
            assert abs(td[k] - tg[k]) <= 2e-5 * max(1.0, abs(tg[k])), (k, td[k], tg[k])
        assert_matches(gxd, gxg.cpu().numpy(), 2e-4, "data.x.grad")
        assert set(gd) == set(gg)
        for k in gg:
            assert_matches(gd[k], gg[k].cpu().numpy(), 2e-4, "grad " + k, floor=1e-6)


def _poison_fixture(rois, bsz):
    import copy
    from igcn_amd import synth
    from igcn_amd.data import Batch
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    go_snps, adj, pool_dim = synth.go_hierarchy((40, 20, 10, 4, 1), seed=3)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
    torch.manual_seed(0)
    model = SGCN_GCN_IMGSNP(2, 16, a_g, a, pool_dim, 32, "cuda", rois=rois, H_0=3, num_classes=3, isSoftSimilarity=True,
                            rbf_gamma=0.01, isCrossAtten=True, num_regr=3, isuseProb4Regr=True, isImageOnly=False,
                            isSNPsOnly=False).cuda().train()
    for m in (model, model.go_network):
        m._dropout_enabled = False
    make = lambda: Batch.from_data_list(synth.brain_graph_list(bsz, seed=5, rois=rois, tsne_dim=16, dense=True)).to("cuda")  # noqa: E731
    good = make()
    # the same shapes, but two edges of graph 1 swapped: not the row-major complete graph any more
    bad = copy.copy(good)
    bad.edge_index = good.edge_index.clone()
    k = rois * rois + 7
    bad.edge_index[:, [k, k + 1]] = bad.edge_index[:, [k + 1, k]]
    bad._igcn_plan = None
    return model, good, bad, make


def test_a_batch_that_is_not_row_major_complete_poisons_the_loss():
    """the dense-block kernels never read ``edge_index``; ``plan.rebuild`` verifies it on the device
    (igcn_dense_blocks_check) and a failed check must not train silently — the next forward turns the degrees, hence
    the outputs and the loss, into NaN, for that batch only (the flag is consumed)."""
    from igcn_amd import ops
    from igcn_amd.train import losses
    model, good, bad, _ = _poison_fixture(64, 4)
    plan = ops.plan_for(good)
    assert plan.dense_blocks
    loss, _, _ = losses(model, good)
    assert bool(torch.isfinite(loss))
    plan.rebuild(bad.edge_index)                                   # what every step does: the device-side check
    bad._igcn_plan = plan
    loss_bad, _, _ = losses(model, bad)
    assert bool(torch.isnan(loss_bad))
    with torch.no_grad():                                          # the plain (eval-style) pass shows it in its outputs
        plan.rebuild(bad.edge_index)
        out = model(bad, None, "cuda")
    assert bool(torch.isnan(out[2]).any())
    with pytest.raises(Exception):
        plan.check()                                               # (the sticky word for host-side checks)
    plan.status.zero_()
    plan.rebuild(good.edge_index)
    good._igcn_plan = plan
    loss2, _, _ = losses(model, good)
    assert bool(torch.isfinite(loss2)) and abs(float(loss2) - float(loss)) <= 1e-5 * abs(float(loss))


def test_a_corrupted_batch_poisons_the_captured_step_too():
    """The same guard inside the hipGraph: ``load()`` of a corrupted batch -> NaN loss from the replay (the check and
    the kernels that consume its flag are both captured); a good batch afterwards trains normally."""
    from igcn_amd.train import FlatAdam, GraphedTrainStep
    model, good, bad, make = _poison_fixture(128, 4)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    static = make()
    static.x.requires_grad_(True)
    step = GraphedTrainStep(model, opt, static, warmup=1)
    assert step.plan.dense_blocks and bool(torch.isfinite(step()))
    snap = opt.flat.clone()
    step.load(bad)
    assert bool(torch.isnan(step()))
    with torch.no_grad():                                          # (the NaN step has poisoned the parameters: restore)
        opt.flat.copy_(snap)
        opt.exp_avg.zero_()
        opt.exp_avg_sq.zero_()
    step.plan.status.zero_()
    step.load(good)
    assert bool(torch.isfinite(step()))
