"""GPU parity of the drop-in model classes: the HIP-backed SGCN_GCN_IMGSNP / Gene_ontology_network against
(a) the golden vectors captured from the reference itself and (b) the CPU oracle on larger seeded inputs.
Everything goes through libigcn.so (the C ABI); tolerance is scale-relative 1e-4 in eval mode.  Training
mode normalises with BatchNorm over 3-8 samples, which amplifies fp32 rounding (the fp64 oracle is just as
far from the reference's own fp32 numbers, see test_oracle_golden.py), hence the looser bound there.
"""
import ast
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import assert_matches, golden_group
from _weights import seeded_state

pytestmark = pytest.mark.gpu

TOL = {"eval": 1e-4, "train": 1e-3}
GTOL = {"eval": 1e-3, "train": 3e-2}
NAMES = ["logp", "x_hat", "out_z", "out_lin", "lin_f", "reg"]
FULL = ["full_tiny", "full_r90", "full_l3", "full_b32", "var_image_only", "var_image_only_noprob", "var_snps_only",
        "var_fusion_noprob", "var_graph_pool", "var_multifusion_l3h2", "var_multifusion_l3h10"]
# the *_b32 fixtures were captured from the reference at B=32, where training-mode BatchNorm no longer amplifies fp32
# rounding: there the north-star bounds hold in TRAINING mode too — 1e-4 on outputs, 1e-3 on gradients
B32 = ("go_b32", "full_b32", "var_multifusion_l3h10")
# var_multifusion_l3h10 (B = 32, rois = 270, hidden 10): outputs at 1e-4 like the other B = 32 fixtures; gradients at
# 3e-3 — one GO read-out pre-activation of that fixture lies within fp32 rounding of zero and the HIP path decides it
# unlike the reference's CPU run (d w_att_in.1.weight 1.2e-3 away; seeds 53 / 54 of the same shape: 2.5e-3 / 1.2e-3,
# seeds 52 / 55: the reference's own fp32 run is 1e-2 from its fp64 evaluation).  The same shape holds 1e-3 against the
# fp64 oracle once those decisions are imposed: test_train_mode_multifusion_vs_oracle
GT_FIXTURE = {"var_multifusion_l3h10": 3e-3}


def tol(name, mode):
    return 1e-4 if name in B32 else TOL[mode]


def gtol(name, mode):
    return GT_FIXTURE.get(name, 1e-3) if name in B32 else GTOL[mode]


def grad_floor(wg, k, floor):
    """See tests/test_oracle_golden.py: a shift with a (nearly) zero exact gradient is judged on its layer's scale."""
    sib = wg.get(k[:-5] + ".weight") if k.endswith(".bias") else None
    if sib is not None and not isinstance(sib, tuple):
        floor = max(floor, 0.5 * float(np.abs(sib).max()))
    return floor


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from igcn_amd import _lib
    _lib.load()


def _probe(outs, seed):
    rng = np.random.default_rng(seed)
    return [torch.from_numpy(rng.standard_normal(tuple(o.shape))).float() for o in outs]


def _go_model(store):
    from igcn_amd import synth
    from igcn_amd.go_model import Gene_ontology_network
    a_g, a = synth.go_sparse_inputs(store["go_snps"], store["adj"], "cuda")
    net = Gene_ontology_network(a_g, a, 2, 2, [5, 5], [store["pool"].tolist()], int(store["l_dim"]), "cuda",
                                dim_snps_atten=int(store["d_att"])).cuda()
    sd = seeded_state({k: v.shape for k, v in net.state_dict().items()}, int(store["seed"]), net.state_dict())
    net.load_state_dict(sd)
    net._dropout_enabled = False
    return net


@pytest.mark.parametrize("maps", ["default", "csr"])
@pytest.mark.parametrize("name", ["go_tiny", "go_small", "go_b32"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_go_network_vs_reference_golden(golden, monkeypatch, name, mode, maps):
    """``maps``: the SNP <-> GO maps as the batch size selects them (these fixtures: dense image + GEMMs) and forced
    onto the LDS-tiled CSR kernels a 256-graph step runs on (IGCN_SPARSE_MAPS=1)."""
    if maps == "csr":
        monkeypatch.setenv("IGCN_SPARSE_MAPS", "1")
    store = golden(name)
    net = _go_model(store)
    net.train(mode == "train")
    snps = torch.from_numpy(store["snps"]).cuda().requires_grad_(True)
    latent, x_d, _, att = net(snps, None, "cuda")
    want = golden_group(store, f"{mode}/out")
    assert_matches(latent, want["latent"], tol(name, mode), "latent")
    assert_matches(x_d, want["x_D"], tol(name, mode), "x_D")
    assert_matches(att, want["atten_out"], tol(name, mode), "atten_out")
    cot = _probe([latent, x_d, att], int(store["seed"]) + 2)
    sum((o * c.cuda()).sum() for o, c in zip([latent, x_d, att], cot)).backward()
    wg = golden_group(store, f"{mode}/grad")
    assert_matches(snps.grad, wg.pop("snps"), gtol(name, mode), "grad snps")
    params = dict(net.named_parameters())
    for k, w in wg.items():
        assert params[k].grad is not None, k
        assert_matches(params[k].grad, w, gtol(name, mode), "grad " + k, floor=1e-4)
    if mode == "train":
        bufs = net.state_dict()
        for k, w in golden_group(store, "train/buffers_after").items():
            assert_matches(bufs[k], w, tol(name, mode), "buffer " + k, floor=1e-2)


def _full_model(store):
    from igcn_amd import synth
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    rois, hidden, layers, bsz, seed, top_k = [int(v) for v in store["cfg"]]
    pool = store["pool"].tolist()
    go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=seed)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
    flags = dict(isSoftSimilarity=True, rbf_gamma=0.01, isCrossAtten=True, num_regr=3, isuseProb4Regr=True,
                 isImageOnly=False, isSNPsOnly=False)
    if "variant" in store:              # var_* fixtures: the other heads of forward() (sgcn_img_snp.py:257-285)
        flags.update(dict(ast.literal_eval(str(store["variant"]))))
    h0 = int(store["h0"]) if "h0" in store else 3       # 1 under --isMultiFusion (var_multifusion_*: row i1)
    model = SGCN_GCN_IMGSNP(layers, hidden, a_g, a, pool_dim, 32, "cuda", rois=rois, H_0=h0, num_classes=3,
                            **flags).cuda()
    ref_keys = sorted(store["state_keys"].tolist())
    assert sorted(model.state_dict().keys()) == ref_keys          # checkpoints interchange with the reference
    sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, seed, model.state_dict())
    model.load_state_dict(sd)
    model._dropout_enabled = False
    model.go_network._dropout_enabled = False
    graphs = synth.brain_graph_list(bsz, seed=seed + 10, rois=rois, h0=h0, top_k=top_k, tsne_dim=16)
    return model, graphs, seed


@pytest.mark.parametrize("name", FULL)
@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("explain", [False, True])
def test_full_model_vs_reference_golden(golden, name, mode, explain):
    from igcn_amd.data import Batch
    store = golden(name)
    model, graphs, seed = _full_model(store)
    model.train(mode == "train")
    data = Batch.from_data_list(graphs).to("cuda")
    outs = model(data, None, "cuda", isExplain=explain)
    tag = f"{mode}/explain{int(explain)}"
    want = golden_group(store, tag + "/out")
    for n, o in zip(NAMES, outs):
        assert_matches(o, want[n], tol(name, mode), n)
    cot = _probe(outs, seed + 3)
    sum((o * c.cuda()).sum() for o, c in zip(outs, cot)).backward()
    wg = golden_group(store, tag + "/grad")
    if "data.x" in wg:
        assert_matches(data.x.grad, wg.pop("data.x"), gtol(name, mode), "grad data.x")
    else:                                   # SNP-only head, plain pass: the image branch is not on the path
        assert data.x.grad is None or not bool(data.x.grad.abs().max() > 0)
    params = dict(model.named_parameters())
    for k, w in wg.items():
        assert params[k].grad is not None, k
        assert_matches(params[k].grad, w, gtol(name, mode), "grad " + k, floor=grad_floor(wg, k, 1e-4))
    for k, p in params.items():             # nothing the reference leaves without a gradient gets one here
        if k not in wg and p.grad is not None:
            assert not bool(p.grad.abs().max() > 0), "unexpected grad " + k


@pytest.mark.parametrize("name", FULL)
@pytest.mark.parametrize("batched", [True, False])
def test_train_step_vs_reference_golden(golden, name, batched):
    from igcn_amd.data import Batch
    from igcn_amd.train import FlatAdam, losses
    store = golden(name)
    model, graphs, seed = _full_model(store)
    model.train(True)
    model.batched_passes = batched      # True: both passes as one 2B-sample sweep; False: two forward() calls
    data = Batch.from_data_list(graphs).to("cuda")
    opt = FlatAdam(model.parameters(), lr=1e-3)
    opt.zero_grad()
    lam = store["lam"].tolist()
    loss, terms, outs = losses(model, data, lam)
    ref_loss = float(store["step/loss"])
    # OrthogonalConstraint: the reference sums the squares of an (R*D) x (R*D) fp32 matrix (:198-205), which at
    # R*D = 2880 is itself ~1e-3 away from the exact value; the Gram form here is exact to 1e-7 (checked against an
    # fp64 evaluation of the reference's own formula below), so this one term gets the reference's rounding as slack
    # (--isMultiFusion: R*D = 8100, the reference's fp32 sum is 9e-3 from its exact value there)
    ref_orth = float(store["step/term/orth"])
    rd = int(store["cfg"][0]) * int(store["cfg"][1]) * int(store["cfg"][2])
    slack = (2e-3 if rd <= 3000 else 2e-2) * abs(ref_orth)
    assert abs(float(loss) - ref_loss) <= 2e-4 * max(1.0, abs(ref_loss)) + slack
    for k, v in terms.items():
        ref = float(store[f"step/term/{k}"])
        assert abs(float(v) - ref) <= 2e-4 * max(1.0, abs(ref)) + (slack if k == "orth" else 0.0), (k, float(v), ref)
    if lam[5] != 0:
        out_z = (outs[2] if len(outs) == 6 else outs[0][2]).detach().double().cpu()
        w = out_z[:len(graphs)]
        wn = w / w.norm(dim=1)[:, None]
        exact = float(torch.norm(wn.T @ wn - torch.eye(wn.shape[1], dtype=torch.float64)) ** 2 / w.shape[0] ** 2)
        assert abs(float(terms["orth"]) - lam[5] * exact) <= 1e-5 * max(1.0, abs(lam[5] * exact))
    loss.backward()
    params = dict(model.named_parameters())
    wg = golden_group(store, "step/grad")
    gt = GT_FIXTURE.get(name, 1e-3) if name in B32 else 1e-2
    assert_matches(data.x.grad, wg.pop("data.x"), gt, "grad data.x")
    grads = {}
    for k, w in wg.items():
        if isinstance(w, tuple) or np.any(w):
            assert_matches(params[k].grad, w, gt, "grad " + k, floor=grad_floor(wg, k, 1e-5))
        else:
            g = params[k].grad                                        # untouched parameters: no (or zero) grad
            assert g is None or not bool(g.abs().max() > 0), k
        grads[k] = w
    opt.step()
    lr = 1e-3
    bufs = model.state_dict()
    for k, w in golden_group(store, "step/buffers_after").items():      # running stats: plain pass, then masked
        assert_matches(bufs[k], w, 1e-3, "buffer " + k, floor=1e-2)
    for k, w in golden_group(store, "step/param_after").items():
        p = params[k].detach().cpu()
        if isinstance(w, tuple) or k not in grads or isinstance(grads[k], tuple):
            assert_matches(p, w, 2.5 * lr, "param " + k, floor=1.0)
            continue
        g = torch.from_numpy(grads[k])
        diff = (p - torch.from_numpy(w)).abs()
        solid = g.abs() > 5e-2 * g.abs().max() if g.abs().max() > 0 else torch.zeros_like(g, dtype=torch.bool)
        sib = grads.get(k[:-5] + ".weight") if k.endswith(".bias") else None
        if sib is not None and not isinstance(sib, tuple) and float(g.abs().max()) < 1e-2 * float(np.abs(sib).max()):
            solid = torch.zeros_like(solid)                            # an all-noise gradient
        assert float(diff[solid].max() if solid.any() else 0.0) <= 5e-5, "param " + k
        assert float(diff.max()) <= 2.01 * lr, "param (noise-level grads) " + k


@pytest.mark.parametrize("bsz,pool,explain", [(32, (300, 120, 60, 19, 1), False), (32, (300, 120, 60, 19, 1), True),
                                              (16, (1800, 800, 300, 99, 1), True)])
def test_full_model_vs_oracle_larger(bsz, pool, explain):
    """R=90, L=2, h=16 (the benchmark model) against the CPU oracle, eval mode, 1e-4."""
    from igcn_amd import ops, synth
    from igcn_amd.data import Batch
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    from oracle import go_network as OG, sgcn_img_snp as OS
    go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=1)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
    model = SGCN_GCN_IMGSNP(2, 16, a_g, a, pool_dim, 32, "cuda", rois=90, H_0=3, num_classes=3,
                            isSoftSimilarity=True, rbf_gamma=0.01, isCrossAtten=True, num_regr=3,
                            isuseProb4Regr=True, isImageOnly=False, isSNPsOnly=False).cuda().eval()
    sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, 5)
    model.load_state_dict(sd)
    graphs = synth.brain_graph_list(bsz, seed=77, rois=90, tsne_dim=16)
    data = Batch.from_data_list(graphs).to("cuda")
    # relu(out_proj(attention)) (:242) has pre-activations that are zero to rounding (at B = 32 / the 500-node DAG one of
    # 92 160 sits at 4e-8 of a 0.35 scale): which side of zero it falls on decides one summand of every gradient
    # upstream (1 % of d out_proj.weight's largest entry), and the split-bf16 attention core — 6e-6 from the exact one
    # on the attention output — decides it the other way.  The HIP path's decisions at that site are observed and
    # imposed on the oracle INSIDE a 2e-5 band, agreement is required outside it (tools/attn_flip_check.py)
    seen = []
    real_ca = model._cross_attention

    def spy(query, memory, **kw):
        out = real_ca(query, memory, **kw)
        if not kw.get("defer_out_proj"):
            seen.append(out.detach().cpu() > 0)
        return out
    model._cross_attention = spy
    real_fused = ops.OutProjHeadInputs.apply        # relu(out_proj) inside the head-input launch: its 4th output is the layer's

    def spy_fused(*a):
        out = real_fused(*a)
        seen.append(out[3].detach().cpu().view(a[0].shape[0], -1, a[1].shape[0]) > 0)
        return out
    ops.OutProjHeadInputs.apply = spy_fused
    try:
        outs = model(data, None, "cuda", isExplain=explain)
    finally:
        ops.OutProjHeadInputs.apply = real_fused
    cot = _probe(outs, 9)
    sum((o * c.cuda()).sum() for o, c in zip(outs, cot)).backward()
    # oracle, fp64
    from conftest import relu_forced
    a_g_c, a_c = synth.go_sparse_inputs(go_snps, adj)
    idx = OG.go_index_sets(a_g_c, a_c, list(pool), 2)
    sdo = OS.make_leaf_state(sd, dtype=torch.float64)
    dcpu = Batch.from_data_list(graphs)
    dcpu.x = dcpu.x.double().requires_grad_(True)
    dcpu.edge_attr, dcpu.snps_feat = dcpu.edge_attr.double(), dcpu.snps_feat.double()
    cfg = SimpleNamespace(num_layers=2, rois=90, image_only=False, rbf_gamma=0.01)
    with relu_forced({11: seen[0]}, band=2e-5) as rf:           # (site 11 of a pass with two GCNConv layers: out_proj)
        ref = OS.model_forward(sdo, cfg, idx, dcpu, explain, training=False)
    assert rf.mismatch_outside == 0 and rf.flips <= 4, (rf.mismatch_outside, rf.flips)
    sum((o * c.double()).sum() for o, c in zip(ref, cot)).backward()
    for n, o, r in zip(NAMES, outs, ref):
        assert_matches(o, r.detach().numpy(), 1e-4, n)
    # gradients at the stated 1e-3.  (Round 2 allowed 3e-3 / 5e-3 here "for ReLU flips"; measured with
    # tools/relu_margin.py at these very shapes: 9 of ~0.5 M pre-activations lie within 1e-6 of zero, and the worst
    # parameter gradient is 2e-5 away from the fp64 oracle — the slack was never used.)
    assert_matches(data.x.grad, dcpu.x.grad.numpy(), 1e-3, "grad data.x")
    params = dict(model.named_parameters())
    for k in OS.trainable_keys(sdo):
        if sdo[k].grad is None:
            continue
        assert_matches(params[k].grad, sdo[k].grad.numpy(), 1e-3, "grad " + k, floor=1e-6)


def train_mode_vs_oracle(monkeypatch, rois, pool, bsz, dense=False, bf16=False, maps=("sparse", "default"),
                         graph_seed=78, go_seed=1, tol=1e-4, gtol=1e-3, max_flips=40, band=2e-5, formulations=(True, False),
                         layers=2, hidden=16, h0=3):
    """TRAINING mode (batch statistics in every BatchNorm, dropout off) of the HIP model against the fp64 oracle: the
    seven loss terms of train() at ``tol`` and every gradient at ``gtol``, for the step formulations ``formulations``
    (True: both passes as one 2B-sample sweep; False: two forward() calls).  Returns the number of imposed ReLU
    decisions.

    ReLU decisions.  tools/relu_margin.py at the default shapes: B.0's smallest pre-activation is 1.5e-6 of its layer's
    scale; fp32 puts it on the other side of zero, and d B.0.bias[node] — a 32-term sum of scale 7e-4 — moves by one
    3e-5 summand (5 % of that tensor's scale; d conc.weight, a cancelling sum behind a training-mode BatchNorm, by
    2 %), while everything else stays below 4e-4.  Instead of a blanket allowance the test OBSERVES the HIP path's
    decisions (the post-ReLU activations its fused kernels return), imposes them on the fp64 oracle where the
    pre-activation is within ``band`` of zero, requires agreement everywhere else, and holds every gradient to ``gtol``."""
    from conftest import relu_forced
    from igcn_amd import ops, synth
    from igcn_amd.data import Batch
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    from igcn_amd.train import losses
    from oracle import go_network as OG, sgcn_img_snp as OS
    lam = [1.0, 1.0, 0.5, 1.5e-6, 0.1, 0.2]
    go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=go_seed)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
    model = SGCN_GCN_IMGSNP(layers, hidden, a_g, a, pool_dim, 32, "cuda", rois=rois, H_0=h0, num_classes=3,
                            isSoftSimilarity=True, rbf_gamma=0.01, isCrossAtten=True, num_regr=3,
                            isuseProb4Regr=True, isImageOnly=False, isSNPsOnly=False, bf16_transforms=bf16).cuda().train()
    sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, 5)
    model.load_state_dict(sd)
    for m in (model, model.go_network):
        m._dropout_enabled = False
    graphs = synth.brain_graph_list(bsz, seed=graph_seed, rois=rois, h0=h0, tsne_dim=16, dense=dense)
    # oracle, fp64, training mode
    a_g_c, a_c = synth.go_sparse_inputs(go_snps, adj)
    idx = OG.go_index_sets(a_g_c, a_c, list(pool), 2)
    cfg = SimpleNamespace(num_layers=layers, rois=rois, image_only=False, rbf_gamma=0.01)
    n0, n1 = sum(pool), sum(pool[1:])

    def hip_run(batched):
        """(loss, terms, grads, forced decisions per oracle ReLU site, ignore masks) of one HIP evaluation."""
        seen = {}
        classes = (ops.SgcnStack, ops.SgcnFront, ops.DenseSgcn, ops.GcnPropagate, ops.GoAttentionLN, ops.GoDecodeLN, ops.NodeLinearBNPair,
                   ops.NodeLinearBN, ops.BatchNorm1dGrouped, ops.LinearBN1d, ops.Linear, ops.LinearPair)
        model.load_state_dict(sd)                                     # running statistics back to the start
        model.zero_grad()
        model.batched_passes = batched
        data = Batch.from_data_list(graphs).to("cuda")
        with monkeypatch.context() as mp:
            for cls in classes:
                def wrapped(*a, _cls=cls, _apply=cls.apply):
                    out = _apply(*a)
                    seen.setdefault(_cls.__name__, []).append(out)
                    return out
                mp.setattr(cls, "apply", wrapped)
            loss, terms, _ = losses(model, data, lam)
        loss.backward()
        params = dict(model.named_parameters())
        grads = {k: params[k].grad for k in params if params[k].grad is not None}
        grads["data.x"] = data.x.grad
        forced, ignore = {}, {}

        def halves(t):                      # stacked sweep: rows [0,B) plain pass, [B,2B) masked pass; else one pass each
            t = t.detach().cpu()
            return [t[:t.shape[0] // 2], t[t.shape[0] // 2:]] if batched else None

        def per_pass(name, k_th, pick=lambda o: o):
            """The k-th call's output per pass: batched -> the two halves of call k; else calls k (plain), k + n (masked)."""
            calls = seen[name]
            if batched:
                return halves(pick(calls[k_th]))
            n = len(calls) // 2
            return [pick(calls[k_th]).detach().cpu(), pick(calls[n + k_th]).detach().cpu()]
        # complete graphs run on the dense blocks; the batched sweep on small graphs through the one-launch front
        stack = "DenseSgcn" if "DenseSgcn" in seen else "SgcnFront" if "SgcnFront" in seen else "SgcnStack"
        if stack in seen:
            xc = per_pass(stack, 0, lambda o: o[0] if isinstance(o, tuple) else o)
        else:       # a graph too large for the LDS-resident stack (rois = 270 at three 16-wide layers): one launch per layer
            cols = [per_pass("GcnPropagate", l_) for l_ in range(layers)]
            xc = [torch.cat([c[p_] for c in cols], dim=1) for p_ in range(2)]
        first = lambda o: o[0] if isinstance(o, tuple) else o        # noqa: E731 — the last encoder layer returns three aliases
        ln = [per_pass("GoAttentionLN", k, first) for k in range(2)] + [per_pass("GoDecodeLN", k) for k in range(2)]
        if "NodeLinearBNPair" in seen:
            att = per_pass("NodeLinearBNPair", 0, lambda o: o[0])
            inp = per_pass("NodeLinearBNPair", 0, lambda o: o[1])
            outd = per_pass("NodeLinearBN", 0)
        else:                                # read-outs as three single launches: attention, input, gene decoding
            att, inp, outd = (per_pass("NodeLinearBN", k) for k in range(3))
        # the latent MLP's two BatchNorm layers: the wide one through ops.LinearBN1d when its product is split
        hb = ([per_pass("LinearBN1d", 0), per_pass("BatchNorm1dGrouped", 0)] if "LinearBN1d" in seen
              else [per_pass("BatchNorm1dGrouped", k) for k in range(2)])
        lp = [per_pass("LinearPair", 0, lambda o, i=i: o[i]) for i in range(2)]
        for p_ in range(2):
            # oracle ReLU sites per pass: ``layers`` GCNConv layers, then twelve more (2 encoder LayerNorms, two read-outs,
            # 2 decoder LayerNorms, gene decoding, 2 latent BatchNorms, out_proj, 2 head layers)
            base = (12 + layers) * p_
            fp_ = xc[p_].shape[1] // layers               # hidden widths off the kernel grid come back zero-padded
            for l_ in range(layers):
                forced[base + l_] = xc[p_][:, l_ * fp_:l_ * fp_ + hidden] > 0
            base += layers - 2
            # encoder LayerNorm sites: the HIP kernel returns the POOLED activations (nodes >= pool[j]); the nodes it
            # drops feed nothing (go_model.py:251), so their decisions are ignored
            for j, (n_in, drop) in enumerate(((n0, pool[0]), (n1, pool[1]))):
                z = ln[j][p_].permute(0, 2, 1) > 0                    # [B, N - drop, f]
                full = torch.zeros(z.shape[0], n_in, z.shape[2], dtype=torch.bool)
                full[:, drop:, :] = z
                ig = torch.zeros_like(full)
                ig[:, :drop, :] = True
                forced[base + 2 + j], ignore[base + 2 + j] = full, ig
            forced[base + 4] = att[p_] > 0
            forced[base + 5] = inp[p_].reshape(inp[p_].shape[0], -1) > 0
            forced[base + 6] = ln[2][p_].permute(0, 2, 1) > 0
            forced[base + 7] = ln[3][p_].permute(0, 2, 1) > 0
            forced[base + 8] = outd[p_].reshape(outd[p_].shape[0], -1) > 0
            forced[base + 9], forced[base + 10] = hb[0][p_] > 0, hb[1][p_] > 0
            forced[base + 11] = None                                  # out_proj + ReLU: fused in the GEMM epilogue
            forced[base + 12], forced[base + 13] = lp[0][p_] > 0, lp[1][p_] > 0
        return loss, terms, grads, forced, ignore

    flips_seen = 0
    for mp_ in maps:
        if mp_ == "sparse":
            monkeypatch.setenv("IGCN_SPARSE_MAPS", "1")
        else:
            monkeypatch.delenv("IGCN_SPARSE_MAPS", raising=False)
        for batched in formulations:
            loss, terms, got, forced, ignore = hip_run(batched)
            st = OS.make_leaf_state(sd, dtype=torch.float64)
            dd = Batch.from_data_list(graphs)
            dd.x = dd.x.double().requires_grad_(True)
            dd.edge_attr, dd.snps_feat = dd.edge_attr.double(), dd.snps_feat.double()
            dd.tsne_fdim, dd.clini_score = dd.tsne_fdim.double(), dd.clini_score.double()
            with relu_forced(forced, band=band, ignore=ignore) as rf:
                ref_loss, ref_terms, _ = OS.train_losses(st, cfg, idx, dd, lam, dropout=False)
            ref_loss.backward()
            assert rf.mismatch_outside == 0, (mp_, batched, rf.mismatch_outside)     # decisions agree outside the band
            assert rf.flips <= max_flips, rf.flips
            flips_seen += rf.flips
            assert abs(float(loss) - float(ref_loss)) <= tol * max(1.0, abs(float(ref_loss))), (batched, float(loss))
            for k, v in terms.items():
                r = float(ref_terms[k])
                assert abs(float(v) - r) <= tol * max(1.0, abs(r)), (batched, k, float(v), r)
            want = {k: st[k].grad for k in OS.trainable_keys(st) if st[k].grad is not None}
            want["data.x"] = dd.x.grad
            for k, w in want.items():
                assert_matches(got[k], w.numpy(), gtol, f"grad {k} (maps={mp_}, batched={batched}, "
                                                        f"{rf.flips} imposed decisions)", floor=1e-6)
    return flips_seen


def test_train_mode_losses_and_gradients_at_default_dims_vs_oracle(monkeypatch):
    """TRAINING mode at the benchmark's own dimensions — R = 90, L = 2, h = 16, the 3000-node GO DAG (LDS-resident GO
    attention backward, LDS decoder), B = 32 — with the CSR SNP <-> GO maps the 256-graph step uses (``sparse``; the
    default at B = 32 is the dense-image form): the seven loss terms of train() at 1e-4 and every gradient at 1e-3
    against the fp64 oracle, both step formulations."""
    flips = train_mode_vs_oracle(monkeypatch, 90, (1800, 800, 300, 99, 1), 32)
    assert flips > 0                          # the mechanism is exercised: B.0 has a node that close to zero (of ~3.3 M)


@pytest.mark.parametrize("layers,hidden", [(3, 2), (3, 10)])
def test_train_mode_multifusion_vs_oracle(monkeypatch, layers, hidden):
    """Row i1 in TRAINING mode: rois = 270, H_0 = 1 (kernel/train_eval_sgcn_img_snps.py:63-67) with the first and the
    last entry of the --isMultiFusion sweep (main.py:147-150), B = 32, a 500-node GO DAG: the seven loss terms at 1e-4,
    every gradient at 1e-3 against the fp64 oracle with the HIP path's ReLU decisions imposed inside the 2e-5 band (the
    reference-captured fixture of the same shape, var_multifusion_l3h10, sits 1.2e-3 away on one GO gradient because
    fp32 decides such a pre-activation differently — see test_full_model_vs_reference_golden's bound for it)."""
    train_mode_vs_oracle(monkeypatch, 270, (300, 120, 60, 19, 1), 32, maps=("default",), layers=layers, hidden=hidden,
                         h0=1)


# ---- the image-only sibling SGCN_GCN (kernel/sgcn.py:272-388; BASELINE configs[0]/[1]) -------------------------
def _sgcn_model(store):
    from igcn_amd import synth
    from igcn_amd.sgcn import SGCN_GCN
    rois, hidden, layers, bsz, seed, top_k = [int(v) for v in store["cfg"]]
    model = SGCN_GCN(None, layers, hidden, rois=rois, H_0=3, num_features=3, num_classes=2).cuda()
    assert sorted(model.state_dict().keys()) == sorted(store["state_keys"].tolist())
    sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, seed, model.state_dict())
    model.load_state_dict(sd)
    model._dropout_enabled = False
    graphs = synth.brain_graph_list(bsz, seed=seed + 10, rois=rois, top_k=top_k, tsne_dim=16, num_classes=2)
    return model, graphs, seed


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("explain", [False, True])
def test_sgcn_only_vs_reference_golden(golden, mode, explain):
    from igcn_amd.data import Batch
    store = golden("sgcn_only")
    model, graphs, seed = _sgcn_model(store)
    model.train(mode == "train")
    data = Batch.from_data_list(graphs).to("cuda")
    out = model(data, explain)
    tag = f"{mode}/explain{int(explain)}"
    assert_matches(out, golden_group(store, tag + "/out")["logp"], 1e-4, "logp")
    (out * _probe([out], seed + 3)[0].cuda()).sum().backward()
    wg = golden_group(store, tag + "/grad")
    assert_matches(data.x.grad, wg.pop("data.x"), 1e-3, "grad data.x")
    params = dict(model.named_parameters())
    for k, w in wg.items():
        assert_matches(params[k].grad, w, 1e-3, "grad " + k, floor=1e-4)


@pytest.mark.parametrize("batched", [True, False])
def test_sgcn_only_train_step_vs_reference_golden(golden, batched):
    """train() of kernel/train_eval_sgcn.py:296-314: loss terms, gradients and the post-Adam parameters."""
    from igcn_amd.data import Batch
    from igcn_amd.train import FlatAdam, losses
    store = golden("sgcn_only")
    model, graphs, seed = _sgcn_model(store)
    model.train(True)
    model.batched_passes = batched
    data = Batch.from_data_list(graphs).to("cuda")
    opt = FlatAdam(model.parameters(), lr=1e-3)
    opt.zero_grad()
    loss, terms, _ = losses(model, data)
    assert abs(float(loss) - float(store["step/loss"])) <= 1e-4 * max(1.0, abs(float(store["step/loss"])))
    for k, v in terms.items():
        assert abs(float(v) - float(store[f"step/term/{k}"])) <= 1e-4, k
    loss.backward()
    params = dict(model.named_parameters())
    wg = golden_group(store, "step/grad")
    assert_matches(data.x.grad, wg.pop("data.x"), 1e-3, "grad data.x")
    for k, w in wg.items():
        assert_matches(params[k].grad, w, 1e-3, "grad " + k, floor=1e-5)
    opt.step()
    for k, w in golden_group(store, "step/param_after").items():
        p = params[k].detach().cpu()
        if isinstance(w, tuple):
            assert_matches(p, w, 2.5e-3, "param " + k, floor=1.0)
            continue
        assert float((p - torch.from_numpy(w)).abs().max()) <= 2.01e-3, "param " + k
        if k in wg and not isinstance(wg[k], tuple):
            g = torch.from_numpy(wg[k])
            solid = g.abs() > 5e-2 * g.abs().max()
            if solid.any():
                assert float((p - torch.from_numpy(w)).abs()[solid].max()) <= 5e-5, "param " + k


@pytest.mark.parametrize("bsz", [32, 256])
def test_sgcn_only_vs_oracle_config_sizes(bsz):
    """BASELINE configs[0] (B=32) and configs[1] (B=256): SGCN-only forward/backward against the fp64 oracle."""
    from igcn_amd import synth
    from igcn_amd.data import Batch
    from igcn_amd.sgcn import SGCN_GCN
    from oracle import sgcn as OSG, sgcn_img_snp as OS
    model = SGCN_GCN(None, 2, 16, rois=90, H_0=3, num_features=3, num_classes=3).cuda().eval()
    sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, 6)
    model.load_state_dict(sd)
    graphs = synth.brain_graph_list(bsz, seed=1000, rois=90, tsne_dim=16)
    for explain in (False, True):
        data = Batch.from_data_list(graphs).to("cuda")
        model.zero_grad()
        out = model(data, explain)
        cot = _probe([out], 4)[0]
        (out * cot.cuda()).sum().backward()
        sdo = OS.make_leaf_state(sd, dtype=torch.float64)
        dcpu = Batch.from_data_list(graphs)
        dcpu.x = dcpu.x.double().requires_grad_(True)
        dcpu.edge_attr = dcpu.edge_attr.double()
        ref = OSG.model_forward(sdo, 90, dcpu, explain)
        (ref * cot.double()).sum().backward()
        assert_matches(out, ref.detach().numpy(), 1e-4, "logp")
        assert_matches(data.x.grad, dcpu.x.grad.numpy(), 3e-3, "grad data.x")
        params = dict(model.named_parameters())
        for k in OS.trainable_keys(sdo):
            if sdo[k].grad is not None:
                assert_matches(params[k].grad, sdo[k].grad.numpy(), 5e-3, "grad " + k, floor=1e-6)


def test_eval_passes_vs_oracle():
    """eval_loss / eval_acc / eval_outputs (kernel/train_eval_sgcn_img_snps.py:551-631) over a two-batch loader."""
    from igcn_amd import synth
    from igcn_amd.data import DataLoader
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    from igcn_amd.train import eval_acc, eval_loss, eval_outputs, output_importance
    from igcn_amd.data import Batch
    from oracle import go_network as OG, sgcn_img_snp as OS
    pool = (60, 30, 20, 9, 1)
    go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=2)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
    model = SGCN_GCN_IMGSNP(2, 8, a_g, a, pool_dim, 32, "cuda", rois=90, H_0=3, num_classes=3,
                            isSoftSimilarity=True, rbf_gamma=0.01, isCrossAtten=True, num_regr=3,
                            isuseProb4Regr=True, isImageOnly=False, isSNPsOnly=False).cuda()
    sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, 8)
    model.load_state_dict(sd)
    graphs = synth.brain_graph_list(12, seed=5, rois=90, tsne_dim=16)
    lam = [1.0, 1.0, 0.5, 1.5e-6, 0.1, 0.2]
    got = eval_loss(model, DataLoader(graphs, batch_size=8), lam, device="cuda")
    acc = eval_acc(model, DataLoader(graphs, batch_size=8), device="cuda")
    outs = eval_outputs(model, DataLoader(graphs, batch_size=8), device="cuda")
    a_g_c, a_c = synth.go_sparse_inputs(go_snps, adj)
    idx = OG.go_index_sets(a_g_c, a_c, list(pool), 2)
    cfg = SimpleNamespace(num_layers=2, rois=90, image_only=False, rbf_gamma=0.01)
    want, hits, logps = 0.0, 0, []
    sd = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}     # fp64 oracle
    with torch.no_grad():
        for lo, hi in ((0, 8), (8, 12)):
            d = Batch.from_data_list(graphs[lo:hi])
            for k in ("x", "edge_attr", "snps_feat", "clini_score", "tsne_fdim"):
                setattr(d, k, getattr(d, k).double())
            o1 = OS.model_forward(sd, cfg, idx, d, False, training=False)
            o2 = OS.model_forward(sd, cfg, idx, d, True, training=False)
            y, clin = d.y.view(-1), d.clini_score.view(-1)
            import torch.nn.functional as F
            loss = lam[0] * F.nll_loss(o1[0], y) + lam[0] * F.nll_loss(o2[0], y) \
                + lam[1] * (F.mse_loss(o1[5].view(-1), clin) + F.mse_loss(o2[5].view(-1), clin)) / 2 \
                + lam[2] * OS.loss_probability(sd, d.x, d.edge_index, d.edge_attr, 90) \
                + lam[3] * (((o1[1] - d.snps_feat) ** 2).sum() + ((o2[1] - d.snps_feat) ** 2).sum()) / 2 \
                + lam[4] * (OS.consist_loss(o1[2], d.tsne_fdim, 0.01) + OS.consist_loss(o2[2], d.tsne_fdim, 0.01)) / 2 \
                + lam[5] * OS.orthogonal_constraint(o1[2])
            want += float(loss) * (hi - lo)
            hits += int(o1[0].max(1)[1].eq(y).sum())
            logps.append(o1[0])
    assert abs(got - want / 12) <= 1e-4 * max(1.0, abs(want / 12)), (got, want / 12)
    assert acc == hits / 12
    assert_matches(outs["logp"], torch.cat(logps).numpy(), 1e-4, "logp")
    assert outs["reg"].shape == (12, 3) and outs["out_lin"].shape[0] == 12 and outs["linear_outf"].shape == (12, 64)
    imp = output_importance(model)
    assert imp["node_importance"].shape == (90, 3) and imp["snps_importance"].shape == (1, 54) \
        and imp["prob_bias"].shape == (6, 1)


@pytest.mark.parametrize("rois,bsz", [(96, 4), (300, 2)])
def test_full_model_dense_graphs_vs_oracle(rois, bsz):
    """The stress shape of BASELINE configs[4] in small: dense brain graphs (E = R^2 per graph -> radix-sort graph
    plan, wave-per-target scatter-aggregate), more ROIs than the attention core's 256-query limit at R=300."""
    from igcn_amd import synth
    from igcn_amd.data import Batch
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    from oracle import go_network as OG, sgcn_img_snp as OS
    pool = (300, 120, 60, 19, 1)
    go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=1)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
    model = SGCN_GCN_IMGSNP(2, 16, a_g, a, pool_dim, 32, "cuda", rois=rois, H_0=3, num_classes=3,
                            isSoftSimilarity=True, rbf_gamma=0.01, isCrossAtten=True, num_regr=3,
                            isuseProb4Regr=True, isImageOnly=False, isSNPsOnly=False).cuda().eval()
    sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, 5)
    model.load_state_dict(sd)
    graphs = synth.brain_graph_list(bsz, seed=78, rois=rois, tsne_dim=16, dense=True)
    data = Batch.from_data_list(graphs).to("cuda")
    outs = model(data, None, "cuda", isExplain=True)
    cot = _probe(outs, 9)
    sum((o * c.cuda()).sum() for o, c in zip(outs, cot)).backward()
    a_g_c, a_c = synth.go_sparse_inputs(go_snps, adj)
    idx = OG.go_index_sets(a_g_c, a_c, list(pool), 2)
    sdo = OS.make_leaf_state(sd, dtype=torch.float64)
    dcpu = Batch.from_data_list(graphs)
    dcpu.x = dcpu.x.double().requires_grad_(True)
    dcpu.edge_attr, dcpu.snps_feat = dcpu.edge_attr.double(), dcpu.snps_feat.double()
    cfg = SimpleNamespace(num_layers=2, rois=rois, image_only=False, rbf_gamma=0.01)
    ref = OS.model_forward(sdo, cfg, idx, dcpu, True, training=False)
    sum((o * c.double()).sum() for o, c in zip(ref, cot)).backward()
    for n, o, r in zip(NAMES, outs, ref):
        assert_matches(o, r.detach().numpy(), 1e-4, n)
    assert_matches(data.x.grad, dcpu.x.grad.numpy(), 3e-3, "grad data.x")
    params = dict(model.named_parameters())
    for k in OS.trainable_keys(sdo):
        if sdo[k].grad is not None:
            assert_matches(params[k].grad, sdo[k].grad.numpy(), 5e-3, "grad " + k, floor=1e-6)


@pytest.mark.parametrize("rois,dense", [(90, False), (72, True)])
def test_graphed_train_step_matches_eager(rois, dense):
    """GraphedTrainStep (whole step replayed from one hipGraph) against the eager train_step on a twin model, three
    steps on changing batches.  k=3 graphs take the LDS plan build, dense 72-ROI graphs (5184 edges per graph) the
    tiled counting-sort build — both hand-written and captured inside the step graph.  The warm-up steps of the
    constructor are rolled back, so no state is restored by hand here."""
    import copy
    from igcn_amd import synth
    from igcn_amd.data import Batch
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    from igcn_amd.train import FlatAdam, GraphedTrainStep, train_step
    pool = (60, 30, 20, 9, 1)
    go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=2)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
    torch.manual_seed(3)
    m1 = SGCN_GCN_IMGSNP(2, 8, a_g, a, pool_dim, 32, "cuda", rois=rois, H_0=3, num_classes=3,
                         isSoftSimilarity=True, rbf_gamma=0.01, isCrossAtten=True, num_regr=3,
                         isuseProb4Regr=True, isImageOnly=False, isSNPsOnly=False).cuda().train()
    for m in (m1, m1.go_network):
        m._dropout_enabled = False
    m2 = copy.deepcopy(m1)
    batches = [Batch.from_data_list(synth.brain_graph_list(6, seed=50 + i, rois=rois, tsne_dim=16, dense=dense)).to("cuda")
               for i in range(3)]
    lam = [1.0, 1.0, 0.5, 1.5e-6, 0.1, 0.2]
    o1, o2 = FlatAdam(m1.parameters(), lr=1e-3), FlatAdam(m2.parameters(), lr=1e-3)
    static = Batch.from_data_list(synth.brain_graph_list(6, seed=50, rois=rois, tsne_dim=16, dense=dense)).to("cuda")
    static.x.requires_grad_(True)
    snap = {k: v.detach().clone() for k, v in m1.state_dict().items()}
    step = GraphedTrainStep(m1, o1, static, lam, warmup=2)
    assert step.plan_in_graph and step.plan._tiled == dense
    for k, v in m1.state_dict().items():                 # the constructor's warm-up steps left no trace
        assert torch.equal(v, snap[k]), k
    assert int(o1.step_count.item()) == 0 and not bool(o1.exp_avg.any())
    for b in batches:
        step.load(b)
        l1 = float(step())
        l2 = float(train_step(m2, o2, b, lam))
        assert abs(l1 - l2) <= 1e-4 * max(1.0, abs(l2)), (l1, l2)
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        # Adam normalises: an ELEMENT whose gradient is rounding noise (< 1e-6) moves by up to lr per step in a
        # direction the noise picks, and the noise depends on buffer alignment (vector vs scalar code paths)
        d = (p1.detach() - p2.detach()).abs()
        tol = torch.full_like(d, 2e-4) if p2.grad is None else torch.where(p2.grad.abs() > 1e-6, 2e-4, 3.5e-3)
        assert bool((d <= tol).all()), (k, float(d.max()))


_DIST_SCRIPT = r"""
import copy, os, sys, torch
sys.path.insert(0, os.environ["IGCN_ROOT"])
import igcn_amd  # noqa: F401
from igcn_amd import synth
from igcn_amd.data import Batch
from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
from igcn_amd.train import FlatAdam, GraphedTrainStep, train_step
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
pool = (60, 30, 20, 9, 1)
go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=2)
a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
torch.manual_seed(3)
m1 = SGCN_GCN_IMGSNP(2, 8, a_g, a, pool_dim, 32, "cuda", rois=90, H_0=3, num_classes=3, isSoftSimilarity=True,
                     rbf_gamma=0.01, isCrossAtten=True, num_regr=3, isuseProb4Regr=True, isImageOnly=False,
                     isSNPsOnly=False).cuda().train()
for m in (m1, m1.go_network):
    m._dropout_enabled = False
m2, m3, m4 = copy.deepcopy(m1), copy.deepcopy(m1), copy.deepcopy(m1)
batches = [Batch.from_data_list(synth.brain_graph_list(6, seed=50 + i, rois=90, tsne_dim=16)).to("cuda")
           for i in range(3)]
lam = [1.0, 1.0, 0.5, 1.5e-6, 0.1, 0.2]
o1, o2, o3, o4 = (FlatAdam(m.parameters(), lr=1e-3) for m in (m1, m2, m3, m4))
def static_batch():
    b = Batch.from_data_list(synth.brain_graph_list(6, seed=50, rois=90, tsne_dim=16)).to("cuda")
    b.x.requires_grad_(True)
    return b
# (1) torch.distributed's RCCL group between two graphs
step = GraphedTrainStep(m1, o1, static_batch(), lam, warmup=2, distributed=True)
assert step.g_opt is not None
# (2) libigcn's own communicator (igcn_comm_*): the all-reduce captured INSIDE the one step graph
from igcn_amd.comm import Comm
comm = Comm()
step3 = GraphedTrainStep(m3, o3, static_batch(), lam, warmup=2, comm=comm, comm_in_graph=True)   # opt-in
print("comm_in_graph", step3.comm_in_graph)
assert step3.comm_in_graph and step3.g_opt is None
# (3) the two-bucket exchange on that communicator: three graphs, the heads' all-reduce (RCCL, side stream) beside the second
step4 = GraphedTrainStep(m4, o4, static_batch(), lam, warmup=2, comm=comm, two_buckets=True)
assert step4.two is not None and step4.g_rest is not None and step4.g_opt is not None and not step4.comm_in_graph
for b in batches:
    step.load(b)
    step3.load(b)
    step4.load(b)
    l1 = float(step())
    l3 = float(step3())
    l4 = float(step4())
    l2 = float(train_step(m2, o2, b, lam))
    assert abs(l1 - l2) <= 1e-4 * max(1.0, abs(l2)), (l1, l2)
    assert abs(l3 - l2) <= 1e-4 * max(1.0, abs(l2)), (l3, l2)
    assert l4 == l3, (l4, l3)                              # same kernels, same arithmetic as the one-graph form
for (k, p3), (_, p4) in zip(m3.named_parameters(), m4.named_parameters()):
    assert torch.equal(p3.detach(), p4.detach()), k        # ... and the same parameters, bit for bit, after three steps
for ma in (m1, m3):
    for (k, p1), (_, p2) in zip(ma.named_parameters(), m2.named_parameters()):
        # Adam normalises: an ELEMENT whose gradient is rounding noise (< 1e-6) moves by up to lr per step in a
        # direction the noise picks, and the noise depends on buffer alignment (vector vs scalar code paths)
        d = (p1.detach() - p2.detach()).abs()
        tol = torch.full_like(d, 2e-4) if p2.grad is None else torch.where(p2.grad.abs() > 1e-6, 2e-4, 3.5e-3)
        assert bool((d <= tol).all()), (k, float(d.max()))
comm.close()
torch.distributed.destroy_process_group()
print("DIST-OK")
"""


def test_graphed_step_with_live_rccl_group():
    """The N>1 control flow of GraphedTrainStep on a single-rank RCCL group — the part of the multi-GPU path a one-GPU
    box can run — against the eager step of a twin model: (1) [forward..backward, pack] graph -> torch.distributed
    all-reduce on the flat gradient -> [Adam] graph, captured in thread-local mode while the process group's threads
    are alive; (2) libigcn's own communicator (igcn_comm_*), its all-reduce captured inside ONE step graph; (3) the
    two-bucket form on that communicator (three graphs, the heads' all-reduce on a side stream), bit-identical to (2).  In
    a child process, so that the process group does not outlive the test."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, IGCN_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", _DIST_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIST-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_model_on_builder_output_vs_oracle(golden, tmp_path):
    """The GO-DAG builder's output (PANTHER-format JSON -> go_snps, adj, pool_dim) drives the HIP model; eval forward
    and input gradient against the fp64 oracle on the same structure."""
    from igcn_amd import go_builder, synth
    from igcn_amd.data import Batch
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    from oracle import go_network as OG, sgcn_img_snp as OS
    store = golden("go_builder")
    (tmp_path / "a.json").write_text(str(store["json"]))
    (tmp_path / "c.txt").write_text(str(store["connection"]))
    (tmp_path / "s.txt").write_text(str(store["snps_to_gene"]))
    go_snps, adj, pool_dim, _, _, _, _ = go_builder.parse_go_json(str(tmp_path / "a.json"), str(tmp_path / "c.txt"),
                                                                   str(tmp_path / "s.txt"))
    pool = [[int(v) for v in pool_dim[0]]]
    a_g, a = go_builder.model_inputs(go_snps, adj, "cuda")
    model = SGCN_GCN_IMGSNP(2, 8, a_g, a, pool, 32, "cuda", rois=90, H_0=3, num_classes=3, isSoftSimilarity=True,
                            rbf_gamma=0.01, isCrossAtten=True, num_regr=3, isuseProb4Regr=True, isImageOnly=False,
                            isSNPsOnly=False).cuda().eval()
    sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, 13)
    model.load_state_dict(sd)
    graphs = synth.brain_graph_list(6, seed=9, rois=90, tsne_dim=16)
    data = Batch.from_data_list(graphs).to("cuda")
    outs = model(data, None, "cuda", isExplain=True)
    cot = _probe(outs, 2)
    sum((o * c.cuda()).sum() for o, c in zip(outs, cot)).backward()
    a_g_c, a_c = go_builder.model_inputs(go_snps, adj)
    idx = OG.go_index_sets(a_g_c, a_c, pool[0], 2)
    sdo = OS.make_leaf_state(sd, dtype=torch.float64)
    dcpu = Batch.from_data_list(graphs)
    dcpu.x = dcpu.x.double().requires_grad_(True)
    dcpu.edge_attr, dcpu.snps_feat = dcpu.edge_attr.double(), dcpu.snps_feat.double()
    cfg = SimpleNamespace(num_layers=2, rois=90, image_only=False, rbf_gamma=0.01)
    ref = OS.model_forward(sdo, cfg, idx, dcpu, True, training=False)
    sum((o * c.double()).sum() for o, c in zip(ref, cot)).backward()
    for n, o, r in zip(NAMES, outs, ref):
        assert_matches(o, r.detach().numpy(), 1e-4, n)
    assert_matches(data.x.grad, dcpu.x.grad.numpy(), 3e-3, "grad data.x")


def test_deferred_reductions_give_the_same_gradients(golden):
    """backward_to_grads(defer=True) queues the ~30 final "sum the block partials" launches of the backward and runs
    them as ONE launch (igcn_reduce_flush): same arithmetic, same summation order => bit-identical gradients."""
    from igcn_amd import _lib
    from igcn_amd.data import Batch
    from igcn_amd.train import FlatAdam, backward_to_grads, losses, _single_use_parameters
    store = golden("full_b32")
    model, graphs, seed = _full_model(store)
    model.train(True)
    assert _single_use_parameters(model)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    got = []
    for defer in (False, True):
        data = Batch.from_data_list(graphs).to("cuda")
        opt.zero_grad()
        loss, _, _ = losses(model, data, store["lam"].tolist())
        backward_to_grads(loss, opt, data, defer=defer)
        assert _lib.load().igcn_reduce_pending() == 0
        got.append({k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()})
        got[-1]["data.x"] = data.x.grad.clone()
    n_grads = 0
    for k, g in got[0].items():
        if g is None:
            assert got[1][k] is None, k
            continue
        n_grads += 1
        assert torch.equal(g, got[1][k]), k
    assert n_grads > 40


@pytest.mark.parametrize("switch", ["IGCN_NO_FUSED_SGCN", "IGCN_NO_DEFER", "IGCN_NO_GEMM_GROUPS", "IGCN_NO_READOUT_PAIR",
                                    "IGCN_LN_AFFINE_NOW", "IGCN_SPMM_DVAL_NOW", "IGCN_NO_PROJ_FUSED", "IGCN_NO_HEAD_FUSED",
                                    "IGCN_NO_MASK_REG_FUSED", "IGCN_NO_GRAD_FAN", "IGCN_NO_LN_FUSED",
                                    "IGCN_NO_LOSS_HEAD_FUSED", "IGCN_SPARSE_MAPS", "IGCN_NO_LINEAR_BN_FUSED",
                                    "IGCN_NO_FRONT_FUSED", "IGCN_NO_HEAD_LOSS_FUSED",
                                    "IGCN_NO_GRAM_LOSS_PAIRED", "IGCN_NO_RELU_OWED", "IGCN_NO_OUTPROJ_FUSED"])
def test_every_host_side_switch_gives_the_default_train_step(golden, monkeypatch, switch):
    """every A/B switch that the Python layer reads selects a second code path — each of them must give
    the default path's train step (loss, every gradient) on the ``full_b32`` model, so a losing variant cannot rot
    unnoticed.  (The switches the library reads at load time are covered by test_alternative_kernel_variants_agree.)"""
    from igcn_amd.data import Batch
    from igcn_amd.train import FlatAdam, backward_to_grads, losses, _single_use_parameters
    store = golden("full_b32")
    lam = store["lam"].tolist()

    def run():
        model, graphs, _ = _full_model(store)
        model.train(True)
        opt = FlatAdam(model.parameters(), lr=1e-3)
        data = Batch.from_data_list(graphs).to("cuda")
        opt.zero_grad()
        loss, terms, _ = losses(model, data, lam)
        backward_to_grads(loss, opt, data, defer=_single_use_parameters(model))
        g = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        g["data.x"] = data.x.grad.clone()
        return float(loss), g

    want_loss, want = run()
    monkeypatch.setenv(switch, "1")
    got_loss, got = run()
    assert abs(got_loss - want_loss) <= 2e-5 * max(1.0, abs(want_loss)), (switch, got_loss, want_loss)
    assert set(got) == set(want)
    for k, w in want.items():
        assert_matches(got[k], w.cpu().numpy(), 2e-4, f"{switch}: grad {k}", floor=1e-6)


def test_deferred_reductions_with_frozen_parameters(golden):
    """Parameters with ``requires_grad=False``: autograd drops their gradients as soon as a backward node returns them, so
    an op must not leave a QUEUED sum pointing at such a buffer (it reduces on the spot instead: ``ops._leaves``).  The
    deferred backward with a handful of frozen tensors gives every remaining gradient bit for bit as the immediate one."""
    from igcn_amd.data import Batch
    from igcn_amd.train import FlatAdam, backward_to_grads, losses
    store = golden("full_b32")
    model, graphs, _ = _full_model(store)
    model.train(True)
    frozen = ("lin2.weight", "lin2_regr.bias", "lin1.weight", "multihead_attn.out_proj.weight", "go_network.t.0",
              "go_network.w_inc.0.weight", "conv1.bias", "go_network.latent.0.weight")
    params = dict(model.named_parameters())
    for k in frozen:
        params[k].requires_grad_(False)
    opt = FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    got = []
    for defer in (False, True):
        data = Batch.from_data_list(graphs).to("cuda")
        opt.zero_grad()
        loss, _, _ = losses(model, data, store["lam"].tolist())
        backward_to_grads(loss, opt, data, defer=defer)
        torch.cuda.synchronize()
        got.append({k: (p.grad.clone() if p.grad is not None else None) for k, p in params.items()})
    assert all(got[1][k] is None for k in frozen)
    n = 0
    for k, g in got[0].items():
        if g is None:
            assert got[1][k] is None, k
            continue
        n += 1
        assert torch.equal(g, got[1][k]), k
    assert n > 35


def test_a_failed_deferred_block_leaves_no_queue_behind(golden, monkeypatch):
    """an exception while the block exits (here: the queued LayerNorm-affine pass fails) must not leave the
    library in defer mode with entries that point at freed partial buffers — the queue is emptied while the buffers are
    alive, and the next backward (plain or deferred) gives the usual gradients."""
    from igcn_amd import _lib, ops
    from igcn_amd.data import Batch
    from igcn_amd.train import FlatAdam, backward_to_grads, losses
    store = golden("full_b32")
    model, graphs, _ = _full_model(store)
    model.train(True)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    lam = store["lam"].tolist()

    def grads(defer):
        data = Batch.from_data_list(graphs).to("cuda")
        opt.zero_grad()
        loss, _, _ = losses(model, data, lam)
        backward_to_grads(loss, opt, data, defer=defer)
        return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    want = grads(True)

    def boom():
        raise RuntimeError("injected")
    with monkeypatch.context() as mp:
        mp.setattr(ops, "_flush_spmm_dval", boom)
        with pytest.raises(RuntimeError, match="injected"):
            grads(True)
    lib = _lib.load()
    assert lib.igcn_reduce_pending() == 0 and not ops._DEFER["on"] and not ops._DEFER["keep"]
    torch.cuda.synchronize()
    for defer in (False, True):
        got = grads(defer)
        assert set(got) == set(want)
        for k in want:
            assert torch.equal(got[k], want[k]), (defer, k)


def test_nothing_stays_queued_after_a_step_even_one_that_raises_half_way(golden, monkeypatch):
    """igcn_stream_pending(stream) — deferred reductions + dropout rider + product riders still waiting — is
    0 after a train step, eager or graphed, and also after a step that raised between queueing its riders and the
    launches that would have carried them (the Gram products are queued inside the forward; here the heads' launch that
    carries them never comes); the masks drawn ahead for that forward are dropped with them and the next step
    runs as if nothing had happened."""
    from igcn_amd import _lib, ops, train
    from igcn_amd.data import Batch
    from igcn_amd.train import FlatAdam, GraphedTrainStep, stream_pending, train_step
    store = golden("full_b32")
    model, graphs, _ = _full_model(store)
    model.train(True)
    model._dropout_enabled = model.go_network._dropout_enabled = True
    opt = FlatAdam(model.parameters(), lr=1e-3)
    lam = store["lam"].tolist()
    data = Batch.from_data_list(graphs).to("cuda")
    assert stream_pending() == 0
    train_step(model, opt, data, lam)
    assert stream_pending() == 0
    # riders queued, then the step dies before their carriers: the Gram riders inside the forward ...
    seen = {}

    def boom(*a, **k):
        seen["pending"] = int(_lib.load().igcn_stream_pending(_lib.stream_ptr()))
        raise RuntimeError("injected")
    with monkeypatch.context() as mp:
        mp.setattr(ops, "linear_pair", boom)
        with pytest.raises(RuntimeError, match="injected"):
            train_step(model, opt, data, lam)
    assert seen["pending"] >= 1                      # (the products really were waiting when the step died)
    assert stream_pending() == 0
    # ... and the dropout rider of a captured step's plan build
    step = GraphedTrainStep(model, opt, data, lam, warmup=1)
    assert stream_pending() == 0
    with monkeypatch.context() as mp:
        mp.setattr(step.plan, "rebuild", boom)
        with pytest.raises(RuntimeError, match="injected"):
            step._fwd_bwd()
    assert seen["pending"] >= 1 and stream_pending() == 0
    assert model.go_network._predrawn is None        # masks drawn ahead for the forward that never came: dropped
    loss = train_step(model, opt, data, lam)         # an eager step of the same shape does not inherit them
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss)) and stream_pending() == 0
    step()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(step.loss)) and stream_pending() == 0


@pytest.mark.parametrize("layers,hidden", [(2, 10), (3, 10), (4, 5)])
@pytest.mark.parametrize("fused", [True, False])
def test_sweep_widths_off_the_kernel_grid_vs_oracle(layers, hidden, fused):
    """The (layers, hidden) entries of the reference's sweep (main.py:152-158) whose width is not a multiple of 4 —
    (2,10), (3,10), (4,5) — run zero-PADDED to 16 / 8 columns through the 16-byte kernels (the LDS-resident stack when
    ``fused``, gcn_norm + MFMA transform + scatter-aggregate otherwise): eval forward (isExplain=True) and every
    gradient against the fp64 oracle at R=90."""
    from igcn_amd import synth
    from igcn_amd.data import Batch
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    from oracle import go_network as OG, sgcn_img_snp as OS
    pool = (60, 30, 20, 9, 1)
    go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=1)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
    model = SGCN_GCN_IMGSNP(layers, hidden, a_g, a, pool_dim, 32, "cuda", rois=90, H_0=3, num_classes=3,
                            isSoftSimilarity=True, rbf_gamma=0.01, isCrossAtten=True, num_regr=3,
                            isuseProb4Regr=True, isImageOnly=False, isSNPsOnly=False).cuda().eval()
    model.fused_sgcn_stack = fused
    sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, 6)
    model.load_state_dict(sd)
    graphs = synth.brain_graph_list(8, seed=78, rois=90, tsne_dim=16)
    data = Batch.from_data_list(graphs).to("cuda")
    outs = model(data, None, "cuda", isExplain=True)
    assert outs[2].shape == (8, 90 * layers * hidden)
    cot = _probe(outs, 9)
    sum((o * c.cuda()).sum() for o, c in zip(outs, cot)).backward()
    a_g_c, a_c = synth.go_sparse_inputs(go_snps, adj)
    idx = OG.go_index_sets(a_g_c, a_c, list(pool), 2)
    sdo = OS.make_leaf_state(sd, dtype=torch.float64)
    dcpu = Batch.from_data_list(graphs)
    dcpu.x = dcpu.x.double().requires_grad_(True)
    dcpu.edge_attr, dcpu.snps_feat = dcpu.edge_attr.double(), dcpu.snps_feat.double()
    cfg = SimpleNamespace(num_layers=layers, rois=90, image_only=False, rbf_gamma=0.01)
    ref = OS.model_forward(sdo, cfg, idx, dcpu, True, training=False)
    sum((o * c.double()).sum() for o, c in zip(ref, cot)).backward()
    for n, o, r in zip(NAMES, outs, ref):
        assert_matches(o, r.detach().numpy(), 1e-4, n)
    assert_matches(data.x.grad, dcpu.x.grad.numpy(), 3e-3, "grad data.x")
    params = dict(model.named_parameters())
    for k in OS.trainable_keys(sdo):
        if sdo[k].grad is None:
            continue
        assert_matches(params[k].grad, sdo[k].grad.numpy(), 5e-3, "grad " + k, floor=1e-6)


MULTIFUSION = [(3, 2), (2, 3), (4, 3), (2, 5), (3, 10)]        # main.py:147-150 (layers, hiddens)


@pytest.mark.parametrize("layers,hidden", MULTIFUSION)
def test_multifusion_sweep_vs_oracle(layers, hidden):
    """Row i1: every (layers, hidden) entry of the --isMultiFusion sweep (main.py:147-150) at the shapes that flag gives
    the trainer (kernel/train_eval_sgcn_img_snps.py:63-67: rois = 270, H_0 = 1) — hidden 2 / 3 padded to 4, 5 to 8, 10
    to 16; head_dim 3 / 6 / 5 / 15; 270 queries — eval forward (isExplain=True) and every gradient against the fp64
    oracle, then one captured train step (both passes batched) against the eager step."""
    from igcn_amd import synth
    from igcn_amd.data import Batch
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    from igcn_amd.train import FlatAdam, GraphedTrainStep, train_step
    from oracle import go_network as OG, sgcn_img_snp as OS
    pool = (60, 30, 20, 9, 1)
    rois = 270
    go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=1)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")

    def build():
        m = SGCN_GCN_IMGSNP(layers, hidden, a_g, a, pool_dim, 32, "cuda", rois=rois, H_0=1, num_classes=3,
                            isSoftSimilarity=True, rbf_gamma=0.01, isCrossAtten=True, num_regr=3, isuseProb4Regr=True,
                            isImageOnly=False, isSNPsOnly=False, isMultiFusion=True).cuda()
        m.load_state_dict(seeded_state({k: v.shape for k, v in m.state_dict().items()}, 6))
        return m
    model = build().eval()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    graphs = synth.brain_graph_list(8, seed=78, rois=rois, h0=1, tsne_dim=16)
    data = Batch.from_data_list(graphs).to("cuda")
    outs = model(data, None, "cuda", isExplain=True)
    assert outs[2].shape == (8, rois * layers * hidden)
    cot = _probe(outs, 9)
    sum((o * c.cuda()).sum() for o, c in zip(outs, cot)).backward()
    a_g_c, a_c = synth.go_sparse_inputs(go_snps, adj)
    idx = OG.go_index_sets(a_g_c, a_c, list(pool), 2)
    sdo = OS.make_leaf_state(sd, dtype=torch.float64)
    dcpu = Batch.from_data_list(graphs)
    dcpu.x = dcpu.x.double().requires_grad_(True)
    dcpu.edge_attr, dcpu.snps_feat = dcpu.edge_attr.double(), dcpu.snps_feat.double()
    cfg = SimpleNamespace(num_layers=layers, rois=rois, image_only=False, rbf_gamma=0.01)
    ref = OS.model_forward(sdo, cfg, idx, dcpu, True, training=False)
    sum((o * c.double()).sum() for o, c in zip(ref, cot)).backward()
    for n, o, r in zip(NAMES, outs, ref):
        assert_matches(o, r.detach().numpy(), 1e-4, n)
    assert_matches(data.x.grad, dcpu.x.grad.numpy(), 3e-3, "grad data.x")
    params = dict(model.named_parameters())
    for k in OS.trainable_keys(sdo):
        if sdo[k].grad is None:
            continue
        assert_matches(params[k].grad, sdo[k].grad.numpy(), 5e-3, "grad " + k, floor=1e-6)
    # the captured step on these shapes = the eager step (dropout off: the two draw different masks otherwise)
    lam = [1.0, 1.0, 0.5, 1.5e-6, 0.1, 0.2]
    res = []
    for graphed in (False, True):
        m = build().train()
        m._dropout_enabled = m.go_network._dropout_enabled = False
        opt = FlatAdam(m.parameters(), lr=1e-3)
        d = Batch.from_data_list(graphs).to("cuda")
        if graphed:
            loss = GraphedTrainStep(m, opt, d, lam)()
        else:
            loss = train_step(m, opt, d, lam)
        torch.cuda.synchronize()
        res.append((float(loss), {k: v.detach().clone() for k, v in m.state_dict().items()}))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * max(1.0, abs(res[0][0]))
    for k, v in res[0][1].items():
        if v.dtype.is_floating_point:
            assert_matches(res[1][1][k], v.cpu().numpy(), 2.5e-3, "after step " + k, floor=1.0)


def test_graphed_step_load_takes_the_new_batch_s_graph_offsets():
    """GraphedTrainStep.load copies ptr / edge_ptr too: a batch with the same totals but other per-graph edge counts
    must be grouped on ITS segments (the per-graph plan builders read them), not on the construction batch's."""
    import copy
    from igcn_amd import synth
    from igcn_amd.data import Batch, Data
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    from igcn_amd.train import FlatAdam, GraphedTrainStep, train_step
    rois, pool = 12, (16, 8, 5, 2, 1)
    go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=2)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
    torch.manual_seed(4)
    m1 = SGCN_GCN_IMGSNP(2, 8, a_g, a, pool_dim, 32, "cuda", rois=rois, H_0=3, num_classes=3, isSoftSimilarity=True,
                         rbf_gamma=0.01, isCrossAtten=True, num_regr=3, isuseProb4Regr=True, isImageOnly=False,
                         isSNPsOnly=False).cuda().train()
    for m in (m1, m1.go_network):
        m._dropout_enabled = False
    m2, m3, m4 = copy.deepcopy(m1), copy.deepcopy(m1), copy.deepcopy(m1)
    rng = np.random.default_rng(9)

    def graph(e):
        src, dst = rng.integers(0, rois, e), rng.integers(0, rois, e)
        return Data(x=torch.from_numpy(rng.random((rois, 3))).float(),
                    edge_index=torch.from_numpy(np.vstack([src, dst])).long(),
                    edge_attr=torch.from_numpy(rng.random(e) + 0.05).float(),
                    y=torch.tensor([int(rng.integers(3))]), clust_y=torch.tensor([0]),
                    snps_feat=torch.from_numpy(rng.random((1, 54))).float(),
                    tsne_fdim=torch.from_numpy(rng.random((1, 6))).float(),
                    clini_score=torch.from_numpy(rng.random(3)).float())
    counts_a, counts_b = (30, 10, 25, 15), (10, 30, 15, 25)           # same total, other segments
    batch_a = Batch.from_data_list([graph(e) for e in counts_a]).to("cuda")
    batch_b = Batch.from_data_list([graph(e) for e in counts_b]).to("cuda")
    assert batch_a.edge_index.shape == batch_b.edge_index.shape and not torch.equal(batch_a.edge_ptr, batch_b.edge_ptr)
    lam = [1.0, 1.0, 0.5, 1.5e-6, 0.1, 0.2]
    o1, o2 = FlatAdam(m1.parameters(), lr=1e-3), FlatAdam(m2.parameters(), lr=1e-3)
    batch_a.x.requires_grad_(True)
    step = GraphedTrainStep(m1, o1, batch_a, lam, warmup=1)
    step.load(batch_b)
    l1 = float(step())
    step.plan.check()
    l2 = float(train_step(m2, o2, batch_b, lam))
    assert abs(l1 - l2) <= 1e-5 * max(1.0, abs(l2)), (l1, l2)
    with pytest.raises(ValueError):                                   # a batch without the offsets is refused
        step.load(SimpleNamespace(x=batch_b.x, edge_index=batch_b.edge_index))
    # same total edge count, but a graph LARGER than any of the construction batch (35 > 30): the captured fused stack
    # was sized for 30 edges per graph — refused by load(); a step built with max_edges=40 takes it and agrees with
    # the eager step; and the kernel itself flags (never silently skips) a graph beyond its launch size
    batch_c = Batch.from_data_list([graph(e) for e in (35, 5, 20, 20)]).to("cuda")
    with pytest.raises(ValueError, match="more edges"):
        step.load(batch_c)
    o3, o4 = FlatAdam(m3.parameters(), lr=1e-3), FlatAdam(m4.parameters(), lr=1e-3)
    batch_a2 = Batch.from_data_list([graph(e) for e in counts_a]).to("cuda")
    batch_a2.x.requires_grad_(True)
    step3 = GraphedTrainStep(m3, o3, batch_a2, lam, warmup=1, max_edges=40)
    step3.load(batch_c)
    l3 = float(step3())
    step3.plan.check()
    l4 = float(train_step(m4, o4, batch_c, lam))
    assert abs(l3 - l4) <= 1e-5 * max(1.0, abs(l4)), (l3, l4)
    from igcn_amd import _lib, ops
    plan_c = ops.plan_for(batch_c)
    assert plan_c._stack_dims == (rois, 35)
    plan_c._stack_dims = (rois, 30)                                   # lie about the launch size
    convs = [m4.conv1, *m4.convs]
    wb = [t for c in convs for t in (c.lin.weight.detach(), c.bias.detach())]
    ops.SgcnStack.apply(batch_c.x.detach(), batch_c.edge_attr, plan_c, rois, *wb)
    with pytest.raises(_lib.IgcnError, match="more edges"):
        plan_c.check()


def test_loss_head_bad_labels_and_disabled_class_terms():
    """igcn_loss_head_fwd: a label outside [0, C) poisons the loss (NaN) instead of reading out of bounds, and with
    lam[0] == 0 (main.py's default) the class scores are not read at all — a non-finite log-probability cannot reach
    the loss through 0 * inf (the reference sets loss_ce = loss_mi = 0.0 outright, :540-542)."""
    from igcn_amd import ops
    b, c, nr, s = 8, 3, 3, 54
    dev = "cuda"
    logp = torch.log_softmax(torch.randn(2 * b, c, device=dev), dim=1)
    y = torch.randint(0, c, (b,), device=dev)
    reg, clin = torch.randn(2 * b, nr, device=dev), torch.randn(b * nr, device=dev)
    x_hat, snps = torch.randn(2 * b, s, device=dev), torch.randn(b, s, device=dev)
    gram, prob = torch.rand(2, 2, device=dev), torch.rand((), device=dev)
    lam = [1.0, 1.0, 0.5, 1.5e-6, 0.1, 0.2]
    good, _ = ops.LossHead.apply(logp, y, reg, clin, x_hat, snps, gram, prob, lam, 1.0, 1.0)
    assert bool(torch.isfinite(good))
    bad = y.clone()
    bad[3] = c                                                         # out of range
    loss, _ = ops.LossHead.apply(logp, bad, reg, clin, x_hat, snps, gram, prob, lam, 1.0, 1.0)
    assert bool(torch.isnan(loss))
    logp_inf = logp.clone()
    logp_inf[0, int(y[0])] = float("-inf")
    lam0 = [0.0] + lam[1:]
    loss0, terms0 = ops.LossHead.apply(logp_inf, y, reg, clin, x_hat, snps, gram, prob, lam0, 1.0, 1.0)
    assert bool(torch.isfinite(loss0)) and float(terms0[0]) == 0.0 and float(terms0[1]) == 0.0
