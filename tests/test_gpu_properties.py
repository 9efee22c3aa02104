"""Size-independent properties at the FULL benchmark sizes (BASELINE configs[2]: B=256, R=90, GO N=3000), where the
CPU oracle is too slow to be the checker: sample-permutation equivariance, linearity of the scatter-aggregate,
idempotence / consistency of the graph plan, equality of the batched sweep with two separate passes, and the
per-column normalisation of the GDC transform."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bench
    from igcn_amd import synth
    from igcn_amd.data import Batch
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    model.eval()
    graphs = synth.brain_graph_list(256, seed=4242, rois=90, tsne_dim=90)
    return model, graphs, Batch, dev


def test_sample_permutation_equivariance(full):
    """Eval mode: permuting the graphs of a batch permutes every output row and changes nothing else."""
    model, graphs, Batch, dev = full
    perm = np.random.default_rng(0).permutation(len(graphs))
    with torch.no_grad():
        a = model(Batch.from_data_list(graphs).to(dev), None, dev, isExplain=True)
        b = model(Batch.from_data_list([graphs[i] for i in perm]).to(dev), None, dev, isExplain=True)
    p = torch.from_numpy(perm).to(dev)
    for name, x, y in zip(("logp", "x_hat", "out_z", "out_lin", "lin_f", "reg"), a, b):
        err = float((x[p] - y).abs().max())
        assert err <= 1e-5 * max(1.0, float(x.abs().max())), (name, err)


def test_batched_sweep_equals_two_passes(full):
    """forward_pair (one 2B-sample sweep) == forward(plain), forward(isExplain) at the full batch size."""
    model, graphs, Batch, dev = full
    with torch.no_grad():
        pair = model.forward_pair(Batch.from_data_list(graphs).to(dev), None, dev)
        one = [model(Batch.from_data_list(graphs).to(dev), None, dev, isExplain=e) for e in (False, True)]
    for k in range(2):
        for x, y in zip(pair[k], one[k]):
            assert float((x - y).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max()))


def test_scatter_aggregate_is_linear_and_plan_is_idempotent(full):
    from igcn_amd import ops
    model, graphs, Batch, dev = full
    data = Batch.from_data_list(graphs).to(dev)
    plan = ops.plan_for(data)
    plan.check()
    again = ops.GraphPlan(data.edge_index, data.x.shape[0])              # radix-sort build of the same batch
    for name in ("tgt_ptr", "tgt_perm", "src_ptr", "src_perm", "loop_edge", "src32", "dst32"):
        assert torch.equal(getattr(plan, name), getattr(again, name)), name
    before = {n: getattr(plan, n).clone() for n in ("tgt_ptr", "tgt_perm", "src_ptr", "src_perm", "loop_edge")}
    plan.rebuild(data.edge_index)                                           # in place, same input: same plan
    for n, t in before.items():
        assert torch.equal(getattr(plan, n), t), n
    # every edge appears exactly once in each grouping
    for perm in (plan.tgt_perm, plan.src_perm):
        assert torch.equal(torch.sort(perm.long()).values, torch.arange(plan.n_edges, device=dev))
    coef = ops.GcnNorm.apply(data.edge_attr, plan)
    g = torch.Generator(device="cpu").manual_seed(1)
    h1 = torch.randn(data.x.shape[0], 16, generator=g).to(dev)
    h2 = torch.randn(data.x.shape[0], 16, generator=g).to(dev)
    zero = torch.zeros(16, device=dev)
    agg = lambda h: ops.GcnPropagate.apply(h, coef[0], coef[1], zero, plan, False, coef[2], coef[3])   # noqa: E731
    lhs, rhs = agg(2.0 * h1 - 3.0 * h2), 2.0 * agg(h1) - 3.0 * agg(h2)
    assert float((lhs - rhs).abs().max()) <= 1e-5 * float(rhs.abs().max())
    # GCN normalisation of column-stochastic GDC graphs: the aggregate of a constant feature is bounded by it
    ones = agg(torch.ones(data.x.shape[0], 16, device=dev))
    assert bool(torch.isfinite(ones).all()) and float(ones.min()) > 0.0


def test_gdc_columns_sum_to_one_at_full_batch():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from igcn_amd.gdc import diffusion_topk
    rng = np.random.default_rng(3)
    s = rng.random((256, 90, 90)).astype(np.float32)
    s = (s + s.transpose(0, 2, 1)) / 2
    s[s < 0.8] = 0.0
    s[:, np.arange(90), np.arange(90)] = 0.0
    s[:, np.arange(89), np.arange(1, 90)] = 1.0                             # a path keeps every graph connected
    s[:, np.arange(1, 90), np.arange(89)] = 1.0
    ei, ew, ptr = diffusion_topk(torch.from_numpy(s).cuda(), 3)
    assert int(ptr[-1]) == 256 * 270 and ptr.tolist() == list(range(0, 256 * 270 + 1, 270))
    col_sum = torch.zeros(256 * 90, dtype=torch.float64, device="cuda").index_add_(0, ei[1], ew.double())
    assert float((col_sum - 1.0).abs().max()) <= 1e-6
    assert bool(((ei[0] // 90) == (ei[1] // 90)).all())                     # block diagonal
    assert int((ei[0] == ei[1]).sum()) == 256 * 90                          # every diagonal entry survives top-3


def test_train_steps_reduce_the_loss_and_stay_finite(full):
    """Twenty graphed steps on one full-size batch: finite throughout, loss lower at the end than at the start."""
    import copy
    from igcn_amd.train import FlatAdam, GraphedTrainStep
    model, graphs, Batch, dev = full
    m = copy.deepcopy(model).train()
    data = Batch.from_data_list(graphs).to(dev)
    data.x.requires_grad_(True)
    step = GraphedTrainStep(m, FlatAdam(m.parameters(), lr=1e-3), data, warmup=1)
    losses = [float(step()) for _ in range(20)]
    assert all(np.isfinite(losses)), losses
    assert np.mean(losses[-5:]) < np.mean(losses[:5]), losses
