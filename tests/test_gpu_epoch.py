"""The fast path under the reference's EPOCH loop (kernel/train_eval_sgcn_img_snps.py): the optimiser object the loop
decays the learning rate through (:169-171), the ragged last batch of ``DataLoader(train_dataset, batch_size,
shuffle=True)`` (:96-97), and a multi-step trajectory pinned by the reference itself (``train_traj`` golden: reference
model + torch.optim.Adam, four batches, lr halved after step 2)."""
import copy

import numpy as np
import pytest
import torch

from conftest import assert_matches, golden_group
from test_gpu_model import _full_model

pytestmark = pytest.mark.gpu

LAM = [1.0, 1.0, 0.5, 1.5e-6, 0.1, 0.2]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from igcn_amd import _lib
    _lib.load()


def _sampled(t, cap=4096):
    flat = t.detach().reshape(-1)
    stride = -(-flat.numel() // cap)
    return flat[::stride] if stride > 1 else t.detach()


def _traj_batches(store):
    from igcn_amd import synth
    from igcn_amd.data import Batch
    rois, _, _, bsz, seed, top_k = [int(v) for v in store["cfg"]]
    return [Batch.from_data_list(synth.brain_graph_list(bsz, seed=seed + 10 + k, rois=rois, top_k=top_k, tsne_dim=16))
            .to("cuda") for k in range(int(store["n_steps"]))]


def _check_trajectory_end(store, model, lr_sum):
    params = dict(model.named_parameters())
    for k, w in golden_group(store, "traj/param_after").items():
        p = _sampled(params[k]).cpu().reshape(-1)
        w = torch.from_numpy(w).reshape(-1)
        solid = torch.from_numpy(np.unpackbits(store[f"traj/solid/{k}"])[:w.numel()].astype(bool))
        d = (p - w).abs()
        if f"traj/untouched/{k}" in store:                 # the reference never steps it (no gradient): same here
            assert float(d.max()) == 0.0, k
            continue
        # elements whose value is determined beyond fp32 rounding (see make_golden.capture_traj): the full_b32 bound of
        # one step (5e-5), four times over; everything else: Adam moves a noise-gradient element by at most lr per step
        assert float(d[solid].max() if solid.any() else 0.0) <= 1e-4, ("param " + k, float(d[solid].max()))
        assert float(d.max()) <= 2.01 * lr_sum, ("param (noise-level grads) " + k, float(d.max()))
    bufs = model.state_dict()
    for k, w in golden_group(store, "traj/buffers_after").items():
        assert_matches(_sampled(bufs[k]), w, 1e-3, "buffer " + k, floor=1e-2)


@pytest.mark.parametrize("route", ["graphed", "eager", "epoch_trainer"])
def test_trajectory_with_lr_decay_vs_reference_golden(golden, route):
    """Four optimisation steps on four different batches with ``param_group['lr'] *= 0.5`` after the second — through
    the captured step (the rate is a device scalar the replayed Adam kernel reads), the eager step and the EpochTrainer
    (eager first sighting, then capture + replays) — against the reference's own trajectory: loss per step at the
    full_b32 bound, final parameters, running statistics."""
    from igcn_amd.train import EpochTrainer, FlatAdam, GraphedTrainStep, train_step
    store = golden("train_traj")
    model, _, _ = _full_model(store)
    model.train(True)
    batches = _traj_batches(store)
    lam = store["lam"].tolist()
    opt = FlatAdam(model.parameters(), lr=1e-3)
    if route == "graphed":
        static = copy.copy(batches[0])
        for k in ("x", "edge_index", "edge_attr", "snps_feat", "y", "clini_score", "tsne_fdim", "clust_y", "ptr",
                  "edge_ptr"):
            setattr(static, k, getattr(batches[0], k).clone())
        static._igcn_plan = None
        step = GraphedTrainStep(model, opt, static, lam, warmup=2)
    elif route == "epoch_trainer":
        trainer = EpochTrainer(model, opt, lam)
    ref_loss, ref_terms, ref_lr = store["traj/loss"], store["traj/terms"], store["traj/lr"]
    for k, b in enumerate(batches):
        assert opt.param_groups[0]["lr"] == pytest.approx(float(ref_lr[k]), rel=1e-12)
        assert float(opt.lr_dev.item()) == pytest.approx(float(ref_lr[k]), rel=1e-6)
        if route == "graphed":
            step.load(b)
            loss = float(step())
        elif route == "eager":
            loss = float(train_step(model, opt, b, lam))
        else:
            loss = float(trainer.step(b))
        # (the orth term carries the reference's own fp32 rounding of a 2880 x 2880 sum: tests/test_gpu_model.py)
        slack = 2e-3 * abs(float(ref_terms[k][6]))
        assert abs(loss - float(ref_loss[k])) <= 2e-4 * max(1.0, abs(float(ref_loss[k]))) + slack, (k, loss, ref_loss[k])
        if k == 1:
            for group in opt.param_groups:                  # kernel/train_eval_sgcn_img_snps.py:169-171, verbatim
                group['lr'] = 0.5 * group['lr']
    if route == "epoch_trainer":
        assert trainer.counts == {"eager": 1, "captured": 1, "replayed": 3}
    assert int(opt.step_count.item()) == len(batches)
    _check_trajectory_end(store, model, float(ref_lr.sum()))


def test_ignoring_the_decay_would_fail_the_trajectory(golden):
    """The golden is sensitive to what it is there to catch: the same four steps at a constant rate miss it."""
    from igcn_amd.train import FlatAdam, train_step
    store = golden("train_traj")
    model, _, _ = _full_model(store)
    model.train(True)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    for b in _traj_batches(store):
        train_step(model, opt, b, store["lam"].tolist())
    with pytest.raises(AssertionError):
        _check_trajectory_end(store, model, float(store["traj/lr"].sum()))


def test_literal_train_body_with_torch_adam_vs_reference_golden(golden):
    """The loop body of the reference's train() (:515-547), line for line — two ``model(...)`` calls, the model's own
    ``loss_probability`` / ``consist_loss`` / ``OrthogonalConstraint``, ``loss.backward()``, ``torch.optim.Adam.step()``
    — on the HIP model: loss, terms, and the post-step parameters of ``full_b32``."""
    import torch.nn.functional as F
    from torch import nn
    from torch.optim import Adam
    from igcn_amd.data import Batch
    from igcn_amd.train import HP as hp
    store = golden("full_b32")
    model, graphs, _ = _full_model(store)
    device = torch.device("cuda")
    lambda_loss = store["lam"].tolist()
    optimizer = Adam(model.parameters(), lr=1e-3, weight_decay=0)
    criterion_recon = nn.MSELoss(reduction='none')
    temperature = torch.tensor(0.1, device=device)
    loader = [Batch.from_data_list(graphs)]
    isSoftSimilarity = True
    # ---- verbatim from here -------------------------------------------------------------------------------
    model.train()
    total_loss = 0
    for data in loader:
        optimizer.zero_grad()
        data = data.to(device)
        for param in model.parameters():
            param.requires_grad = True
        out, snps_hat, out_feat, out_lin, _, our_reg = model(data, temperature, device)
        loss_ce = lambda_loss[0] * F.nll_loss(out, data.y.view(-1))
        out_prob, snps_hat_prob, out_feat_prob, out_lin_prob, _, our_reg_prob = model(data, temperature, device,
                                                                                      isExplain=True)
        loss_mi = lambda_loss[0] * F.nll_loss(out_prob, data.y.view(-1))
        loss_reg = lambda_loss[1] * (F.mse_loss(our_reg.view(-1), data.clini_score.view(-1))
                                     + F.mse_loss(our_reg_prob.view(-1), data.clini_score.view(-1))) / 2
        loss_prob = lambda_loss[2] * model.loss_probability(data.x, data.edge_index, data.edge_attr, hp)
        recon_loss = lambda_loss[3] * (torch.sum(criterion_recon(snps_hat, data.snps_feat))
                                       + torch.sum(criterion_recon(snps_hat_prob, data.snps_feat))) / 2
        cluster_loss = 0
        if isSoftSimilarity:
            cluster_loss += lambda_loss[4] * (model.consist_loss(out_feat, data.tsne_fdim)
                                              + model.consist_loss(out_feat_prob, data.tsne_fdim)) / 2
        orthogonal_loss = lambda_loss[5] * model.OrthogonalConstraint(out_feat)
        if lambda_loss[0] == 0:
            loss_ce = 0.0
            loss_mi = 0.0
        loss = hp.lamda_ce * loss_ce + hp.lamda_mi * loss_mi + loss_reg + loss_prob + recon_loss + cluster_loss \
            + orthogonal_loss
        loss.backward()
        total_loss += loss.detach().cpu().item() * data.num_graphs
        optimizer.step()
    # ---- end of the verbatim body -------------------------------------------------------------------------
    ref_orth = float(store["step/term/orth"])
    slack = 2e-3 * abs(ref_orth)
    ref_loss = float(store["step/loss"])
    assert abs(total_loss / len(graphs) - ref_loss) <= 2e-4 * max(1.0, abs(ref_loss)) + slack
    got = {"ce": loss_ce, "mi": loss_mi, "reg": loss_reg, "prob": loss_prob, "recon": recon_loss,
           "cluster": cluster_loss, "orth": orthogonal_loss}
    for k, v in got.items():
        ref = float(store[f"step/term/{k}"])
        assert abs(float(v) - ref) <= 2e-4 * max(1.0, abs(ref)) + (slack if k == "orth" else 0.0), (k, float(v), ref)
    params = dict(model.named_parameters())
    wg = golden_group(store, "step/grad")
    assert_matches(data.x.grad, wg.pop("data.x"), 1e-3, "grad data.x")
    lr = 1e-3
    for k, w in golden_group(store, "step/param_after").items():
        p = params[k].detach().cpu()
        g = wg.get(k)
        if isinstance(w, tuple) or g is None or isinstance(g, tuple):
            assert_matches(p, w, 2.5 * lr, "param " + k, floor=1.0)
            continue
        g = torch.from_numpy(g)
        diff = (p - torch.from_numpy(w)).abs()
        solid = g.abs() > 5e-2 * g.abs().max() if g.abs().max() > 0 else torch.zeros_like(g, dtype=torch.bool)
        sib = wg.get(k[:-5] + ".weight") if k.endswith(".bias") else None
        if sib is not None and not isinstance(sib, tuple) and float(g.abs().max()) < 1e-2 * float(np.abs(sib).max()):
            solid = torch.zeros_like(solid)
        assert float(diff[solid].max() if solid.any() else 0.0) <= 5e-5, "param " + k
        assert float(diff.max()) <= 2.01 * lr, "param (noise-level grads) " + k


def test_loss_probability_is_recomputed_from_its_arguments():
    """the mask regulariser cached by a forward is handed out ONCE and only for that forward's inputs; a
    second call, a call on another batch, or a call after the parameters moved recomputes from the arguments, as
    the reference does (kernel/sgcn_img_snp.py:153-181)."""
    from igcn_amd import synth
    from igcn_amd.data import Batch
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    from igcn_amd.train import HP, losses
    pool = (40, 20, 10, 4, 1)
    go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=3)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
    torch.manual_seed(0)
    for rois, dense in ((90, False), (64, True)):
        model = SGCN_GCN_IMGSNP(2, 16, a_g, a, pool_dim, 32, "cuda", rois=rois, H_0=3, num_classes=3,
                                isSoftSimilarity=True, rbf_gamma=0.01, isCrossAtten=True, num_regr=3,
                                isuseProb4Regr=True, isImageOnly=False, isSNPsOnly=False).cuda().train()
        for m in (model, model.go_network):
            m._dropout_enabled = False
        b1 = Batch.from_data_list(synth.brain_graph_list(4, seed=5, rois=rois, tsne_dim=16, dense=dense)).to("cuda")
        b2 = Batch.from_data_list(synth.brain_graph_list(4, seed=6, rois=rois, tsne_dim=16, dense=dense)).to("cuda")
        _, terms, _ = losses(model, b1, LAM)                      # the batched sweep reduces the regulariser itself
        want1 = float(terms["prob"]) / LAM[2]
        fresh1 = float(model.loss_probability(b1.x, b1.edge_index, b1.edge_attr, HP))      # 2nd call: recomputed
        assert fresh1 == pytest.approx(want1, rel=2e-5)
        model(b1, None, "cuda", isExplain=True)
        other = float(model.loss_probability(b2.x, b2.edge_index, b2.edge_attr, HP))       # another batch
        _, terms2, _ = losses(model, b2, LAM)
        assert other == pytest.approx(float(terms2["prob"]) / LAM[2], rel=2e-5)
        assert abs(other - want1) > 1e-6
        model(b1, None, "cuda", isExplain=True)
        with torch.no_grad():
            model.prob.add_(0.25)                                   # the parameters moved after the forward
        moved = float(model.loss_probability(b1.x, b1.edge_index, b1.edge_attr, HP))
        model(b1, None, "cuda", isExplain=True)
        again = float(model.loss_probability(b1.x, b1.edge_index, b1.edge_attr, HP))
        assert moved == pytest.approx(again, rel=2e-5) and abs(moved - want1) > 1e-5


def _small_model(seed=3, rois=90):
    from igcn_amd import synth
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    pool = (60, 30, 20, 9, 1)
    go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=2)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, "cuda")
    torch.manual_seed(seed)
    m = SGCN_GCN_IMGSNP(2, 16, a_g, a, pool_dim, 32, "cuda", rois=rois, H_0=3, num_classes=3, isSoftSimilarity=True,
                        rbf_gamma=0.01, isCrossAtten=True, num_regr=3, isuseProb4Regr=True, isImageOnly=False,
                        isSNPsOnly=False).cuda().train()
    for mod in (m, m.go_network):
        mod._dropout_enabled = False
    return m


def test_fit_epoch_with_ragged_tail_equals_the_eager_loop():
    """An epoch of 3 x 32 + 20 graphs (``DataLoader(dataset, 32)`` without drop_last, :96-97), three epochs with the
    rate halved after the second (:169-171), through ``fit_epoch``: epoch 1 runs [eager, capture+replay, replay, eager
    tail], epoch 2 captures the tail shape too, epoch 3 is replays only — against the same 12 steps through the eager
    ``train_step`` on a twin model; returned epoch losses = sum(loss_b * graphs_b) / len(dataset) (:546,548)."""
    from igcn_amd import synth
    from igcn_amd.data import DataLoader
    from igcn_amd.train import FlatAdam, fit_epoch, train_step
    m1 = _small_model()
    m2 = copy.deepcopy(m1)
    graphs = synth.brain_graph_list(3 * 32 + 20, seed=60, rois=90, tsne_dim=16)
    loader = DataLoader(graphs, 32, shuffle=False)
    o1, o2 = FlatAdam(m1.parameters(), lr=1e-3), FlatAdam(m2.parameters(), lr=1e-3)
    for epoch in range(1, 4):
        got = fit_epoch(m1, o1, loader, None, LAM, device="cuda")
        total = 0.0
        for data in loader:
            data = data.to("cuda")
            total += float(train_step(m2, o2, data, LAM)) * data.num_graphs
        want = total / len(graphs)
        assert got == pytest.approx(want, rel=1e-4), (epoch, got, want)
        if epoch == 2:
            for o in (o1, o2):
                for group in o.param_groups:
                    group['lr'] = 0.5 * group['lr']
    tr = next(iter(o1._igcn_epoch_trainers.values()))
    assert tr.counts == {"eager": 2, "captured": 2, "replayed": 10}, tr.counts
    assert int(o1.step_count.item()) == int(o2.step_count.item()) == 12
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        d = (p1.detach() - p2.detach()).abs()
        tol = torch.full_like(d, 4e-4) if p2.grad is None else torch.where(p2.grad.abs() > 1e-6, 4e-4, 1.1e-2)
        assert bool((d <= tol).all()), (k, float(d.max()))
    for (k, b1), (_, b2) in zip(m1.named_buffers(), m2.named_buffers()):
        assert_matches(b1.float(), b2.float().cpu().numpy(), 1e-4, "buffer " + k, floor=1e-2)


def test_flat_adam_state_dict_interchanges_with_torch_adam():
    """``state_dict()`` has torch.optim.Adam's layout: a FlatAdam checkpoint loads into torch.optim.Adam over the same
    parameter list and vice versa, and both continue identically (same gradients -> same parameters)."""
    torch.manual_seed(1)
    shapes = [(7, 5), (64,), (3, 4, 2), (1,)]
    base = [torch.randn(s, device="cuda") for s in shapes]
    grads = [[torch.randn(s, device="cuda") for s in shapes] for _ in range(4)]
    from igcn_amd.train import FlatAdam

    def make(kind, lr=2e-3):
        ps = [torch.nn.Parameter(b.clone()) for b in base]
        ps.append(torch.nn.Parameter(torch.ones(5, device="cuda")))            # never receives a gradient
        return ps, (FlatAdam(ps, lr=lr) if kind == "flat" else torch.optim.Adam(ps, lr=lr))

    def run(ps, opt, steps):
        for g in steps:
            opt.zero_grad()
            for p, gi in zip(ps, g):
                p.grad = gi.clone()
            opt.step()

    for src, dst in (("flat", "torch"), ("torch", "flat"), ("flat", "flat")):
        ps_a, opt_a = make(src)
        run(ps_a, opt_a, grads[:2])
        for group in opt_a.param_groups:
            group['lr'] = 0.5 * group['lr']
        sd = copy.deepcopy(opt_a.state_dict())
        assert sorted(sd["state"].keys()) == [0, 1, 2, 3] and sd["param_groups"][0]["lr"] == pytest.approx(1e-3)
        assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 2.0
        ps_b, opt_b = make(dst, lr=123.0)
        with torch.no_grad():
            for pb, pa in zip(ps_b, ps_a):
                pb.copy_(pa)
        opt_b.load_state_dict(sd)
        assert opt_b.param_groups[0]["lr"] == pytest.approx(1e-3)
        run(ps_a, opt_a, grads[2:])
        run(ps_b, opt_b, grads[2:])
        for pa, pb in zip(ps_a, ps_b):
            assert float((pa - pb).abs().max()) <= 2e-6, (src, dst)
        assert torch.equal(ps_b[-1].detach(), torch.ones(5, device="cuda"))
    # the schedule reaches a parameter through the device scalar
    ps, opt = make("flat", lr=1e-2)
    assert float(opt.lr_dev.item()) == pytest.approx(1e-2)
    opt.param_groups[0]["lr"] *= 0.1
    assert float(opt.lr_dev.item()) == pytest.approx(1e-3) and opt.lr == pytest.approx(1e-3)
    with pytest.raises(ValueError):
        opt.param_groups[0]["weight_decay"] = 1e-4


def test_copy_multi_moves_the_same_bytes_as_copy():
    """igcn_copy_multi (the one-launch hand-over of GraphedTrainStep.load): mixed dtypes and sizes, unaligned views
    (byte path), empty tensors, more pairs than one launch takes — bit-identical to Tensor.copy_."""
    from igcn_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(5)
    shapes = [((23040, 3), torch.float32), ((2, 69120), torch.int64), ((69120,), torch.float32), ((256, 54), torch.float32),
              ((256,), torch.int64), ((768,), torch.float32), ((257,), torch.int64), ((0,), torch.float32),
              ((13,), torch.uint8), ((1000003,), torch.uint8)]
    shapes = shapes + shapes                                   # 20 pairs: two launches
    pairs = []
    for shp, dt in shapes:
        n = int(np.prod(shp))
        raw = torch.randint(0, 255, (n * torch.empty(0, dtype=dt).element_size() + 32,), dtype=torch.uint8, device="cuda",
                            generator=g)
        src = raw[16:16 + n * torch.empty(0, dtype=dt).element_size()].view(dt).view(shp)
        dst = torch.zeros(shp, dtype=dt, device="cuda")
        pairs.append((dst, src))
    # an unaligned source / destination (byte path): 1-byte offset views
    a = torch.randint(0, 255, (4099,), dtype=torch.uint8, device="cuda", generator=g)
    b = torch.zeros(4099, dtype=torch.uint8, device="cuda")
    pairs.append((b[1:4098], a[2:4099]))
    # a host source and a dtype change fall back to copy_
    host = torch.arange(77, dtype=torch.float32)
    dev = torch.zeros(77, device="cuda")
    as_int = torch.zeros(5, dtype=torch.int32, device="cuda")
    pairs += [(dev, host), (as_int, torch.arange(5, device="cuda", dtype=torch.int64))]
    _lib.copy_multi(pairs)
    torch.cuda.synchronize()
    def raw(t):                                            # bytes, not values: random bytes make NaNs, and NaN != NaN
        return t.cpu().contiguous().reshape(-1).view(torch.uint8)
    for dst, src in pairs:
        if dst.dtype == src.dtype:
            assert torch.equal(raw(dst), raw(src)), (dst.shape, dst.dtype)
        else:
            assert torch.equal(dst.cpu(), src.cpu().to(dst.dtype)), (dst.shape, dst.dtype)
    assert int(b[0]) == 0 and int(b[4098]) == 0            # nothing written outside the ranges


def test_captured_step_draws_its_dropout_masks_in_the_plan_builds_launch(monkeypatch):
    """Dropout ON in the captured step: the masks of a replay are drawn by extra workgroups of its per-graph plan build
    (ops.dropout_masks ``ride``; model.predraw_dropout) — the same masks, hence the same losses, as with the mask launch
    of its own (IGCN_NO_DROPOUT_RIDER=1); fresh masks on every replay; num_batches_tracked advances by the two passes."""
    from igcn_amd import synth
    from igcn_amd.data import Batch
    from igcn_amd.train import FlatAdam, GraphedTrainStep

    def run():
        torch.manual_seed(77)                                       # (seeds the device-side mask counter too)
        m = _small_model()
        for mod in (m, m.go_network):
            mod._dropout_enabled = True
        opt = FlatAdam(m.parameters(), lr=1e-3)
        data = Batch.from_data_list(synth.brain_graph_list(16, seed=61, rois=90, tsne_dim=16)).to("cuda")
        data.x.requires_grad_(True)
        step = GraphedTrainStep(m, opt, data, LAM, warmup=1)
        losses = [float(step()) for _ in range(4)]
        return losses, int(m.go_network.latent[1].num_batches_tracked), m

    want, nb_want, _ = run()
    assert len(set(round(v, 6) for v in want)) > 1                  # the masks differ from replay to replay
    monkeypatch.setenv("IGCN_NO_DROPOUT_RIDER", "1")
    got, nb_got, m = run()
    assert nb_got == nb_want and nb_want >= 8                       # two passes per step, warm-up rolled back or not
    assert all(abs(a - b) <= 1e-6 * max(1.0, abs(b)) for a, b in zip(got, want)), (got, want)
    assert getattr(m.go_network, "_predrawn", None) is None
