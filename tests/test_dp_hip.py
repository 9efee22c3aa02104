"""The HIP data-parallel step under world_size 2 (SURVEY §4 tier 4, §8e): two fresh child processes share cuda:0,
each runs ``train_step(world_size=2)`` — and the graphed step — on its shard; the post-Adam parameters must be
bit-identical across ranks and equal the oracle's mean-of-shard-gradients step (DDP convention: local BatchNorm
statistics and batch-level losses, gradients averaged)."""
import os
import socket
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_matches

pytestmark = pytest.mark.gpu
WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def ranks(tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    out = str(tmp_path_factory.mktemp("dp") / "res")
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), str(WORLD), port,
                               out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(WORLD)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("data-parallel workers timed out")
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    return [torch.load(f"{out}.rank{r}.pt") for r in range(WORLD)]


def _oracle_step():
    """Mean of the shard gradients (each shard's own train loss, dropout off), then one Adam step."""
    import dp_worker as W
    from _weights import seeded_state
    from igcn_amd import synth
    from igcn_amd.data import Batch
    from igcn_amd.train import shard_batch
    from oracle import go_network as OG, sgcn_img_snp as OS
    go_snps, adj, _ = synth.go_hierarchy(W.POOL, seed=2)
    a_g, a = synth.go_sparse_inputs(go_snps, adj)
    idx = OG.go_index_sets(a_g, a, list(W.POOL), 2)
    shapes = dict(OS.sgcn_param_shapes(W.LAYERS, W.HIDDEN, rois=W.ROIS))
    shapes.update({"go_network." + k: v for k, v in OG.go_param_shapes(idx, l_dim=32,
                                                                        d_att=W.LAYERS * W.HIDDEN).items()})
    for nm, c in (("batch_norm_1d", W.ROIS * W.LAYERS * W.HIDDEN + 32), ("batch_norm", W.LAYERS * W.HIDDEN)):
        shapes.update({f"{nm}.weight": (c,), f"{nm}.bias": (c,), f"{nm}.running_mean": (c,),
                       f"{nm}.running_var": (c,), f"{nm}.num_batches_tracked": ()})
    sd = seeded_state(shapes, W.SEED)
    cfg = SimpleNamespace(num_layers=W.LAYERS, rois=W.ROIS, image_only=False, rbf_gamma=0.01)
    graphs = W.all_graphs()
    grads, losses = [], []
    for r in range(WORLD):
        # fp64: at R * D = 2880 the reference-style fp32 evaluation of OrthogonalConstraint (an (R D)^2 matrix of
        # squares) is itself ~1e-3 off the exact value; the HIP path's Gram form is not
        st = OS.make_leaf_state(sd, dtype=torch.float64)
        data = Batch.from_data_list(shard_batch(graphs, r, WORLD))
        data.x = data.x.double().requires_grad_(True)
        data.edge_attr, data.snps_feat = data.edge_attr.double(), data.snps_feat.double()
        data.tsne_fdim, data.clini_score = data.tsne_fdim.double(), data.clini_score.double()
        loss, _, _ = OS.train_losses(st, cfg, idx, data, W.LAM, dropout=False)
        loss.backward()
        grads.append({k: st[k].grad for k in OS.trainable_keys(st)})
        losses.append(float(loss))
    mean = {k: (None if grads[0][k] is None else sum(g[k] for g in grads) / WORLD) for k in grads[0]}
    mean = {k: (None if g is None else g.float()) for k, g in mean.items()}
    params = {k: torch.nn.Parameter(sd[k].clone()) for k in mean}
    opt = torch.optim.Adam(list(params.values()), lr=1e-3)
    for k, p in params.items():
        p.grad = mean[k]
    opt.step()
    return mean, {k: p.detach() for k, p in params.items()}, losses


def test_ranks_end_bit_identical_and_match_the_oracle_step(ranks):
    r0, r1 = ranks
    for k in r0["param_after"]:
        assert torch.equal(r0["param_after"][k], r1["param_after"][k]), "replicas diverged: " + k
        assert torch.equal(r0["grad_sum"][k], r1["grad_sum"][k]), "all-reduced bucket differs: " + k
    mean, after, losses = _oracle_step()
    for r, res in enumerate(ranks):
        assert abs(res["loss"] - losses[r]) <= 2e-4 * max(1.0, abs(losses[r])), (r, res["loss"], losses[r])
    lr = 1e-3
    for k, g in mean.items():
        got = r0["grad_sum"][k] / WORLD
        if g is None:                                    # parameters the model never touches: zero-filled slots
            assert not bool(got.abs().max() > 0), k
            assert torch.equal(r0["param_after"][k], after[k]), k
            continue
        sib = mean.get(k[:-5] + ".weight") if k.endswith(".bias") else None
        floor = max(1e-5, 0.5 * float(sib.abs().max())) if sib is not None else 1e-5
        assert_matches(got, g.numpy(), 5e-3, "mean grad " + k, floor=floor)
        diff = (r0["param_after"][k] - after[k]).abs()
        solid = g.abs() > 5e-2 * g.abs().max() if float(g.abs().max()) > 0 else torch.zeros_like(g, dtype=torch.bool)
        if sib is not None and float(g.abs().max()) < 1e-2 * float(sib.abs().max()):
            solid = torch.zeros_like(solid)
        assert float(diff[solid].max() if solid.any() else 0.0) <= 5e-5, "param " + k
        assert float(diff.max()) <= 2.01 * lr, "param (noise-level grads) " + k


def test_batchnorm_buffers_stay_local(ranks):
    """DDP convention: every rank normalises with (and tracks) its own shard's statistics."""
    r0, r1 = ranks
    differ = [k for k in r0["buffers"] if "running_mean" in k and not torch.equal(r0["buffers"][k], r1["buffers"][k])]
    assert differ, "running means of different shards should differ"


def test_graphed_distributed_step_equals_the_eager_one(ranks):
    for res in ranks:
        assert res["graphed_step_count"] == 1            # warm-up steps are rolled back
        assert abs(res["graphed_loss"] - res["loss"]) <= 1e-6 * max(1.0, abs(res["loss"]))
        for k, p in res["param_after"].items():
            d = float((res["graphed_param_after"][k] - p).abs().max()) if p.numel() else 0.0
            assert d <= 1e-6, (k, d)
    for k in ranks[0]["graphed_param_after"]:
        assert torch.equal(ranks[0]["graphed_param_after"][k], ranks[1]["graphed_param_after"][k]), k


def test_two_bucket_exchange_gives_the_default_step_bit_for_bit(ranks):
    """``two_buckets=True``: the backward in two sweeps, the heads' gradients — most of the bucket —
    all-reduced on a side stream while the second sweep runs, the remainder behind it; eager and as three captured graphs.
    Same kernels, same arithmetic: the reduced gradients and the parameters after one (eager) / two (graphed) steps equal
    the default form's bit for bit, on every rank."""
    for res in ranks:
        assert res["two_used"] and res["two_graphed_used"]
        assert res["two_early_share"] > 0.7                       # lin1 + lin1_regr + lin2 + lin2_regr
        assert res["two_loss"] == res["loss"]
        for k, g in res["grad_sum"].items():
            assert torch.equal(res["two_grad_sum"][k], g), k
        for k, p in res["param_after"].items():
            assert torch.equal(res["two_param_after"][k], p), k
        assert abs(res["two_graphed_loss"] - res["graphed_loss"]) == 0.0
        bad1 = {k: float((res["two_graphed_param_after"][k] - p).abs().max()) for k, p in res["graphed_param_after"].items()
                if not torch.equal(res["two_graphed_param_after"][k], p)}
        bad2 = {k: float((res["two_graphed_param_after_2_steps"][k] - p).abs().max())
                for k, p in res["graphed_param_after_2_steps"].items()
                if not torch.equal(res["two_graphed_param_after_2_steps"][k], p)}
        assert not bad1 and not bad2, (bad1, bad2)
    for k in ranks[0]["two_param_after"]:
        assert torch.equal(ranks[0]["two_param_after"][k], ranks[1]["two_param_after"][k]), k


def test_epoch_loop_in_data_parallel_keeps_the_ranks_identical(ranks):
    """``fit_epoch(world_size=2)`` on every rank's shard (batches of 6 + 6 + 4 graphs, three epochs, the rate halved after
    the first): the ranks take the same route per batch (eager -> capture -> replay), end bit-identical, and land where
    the same nine steps through the eager data-parallel ``train_step`` land (kernel/train_eval_sgcn_img_snps.py:96-97,
    169-171,511-548 under SURVEY §8e's partitioning)."""
    r0, r1 = ranks
    assert r0["epoch_counts"] == r1["epoch_counts"] == {"eager": 2, "captured": 2, "replayed": 7}, r0["epoch_counts"]
    assert r0["epoch_step_count"] == r1["epoch_step_count"] == (9, 9)
    for k, p in r0["epoch_param_after"].items():
        assert torch.equal(p, r1["epoch_param_after"][k]), "replicas diverged in the epoch loop: " + k
        ref = r0["epoch_ref_param_after"][k]
        assert float((p - ref).abs().max()) <= 2e-4 if p.numel() else True, k
    assert all(np.isfinite(v) for v in r0["epoch_losses"] + r1["epoch_losses"])


def test_comm_abi_single_rank_allreduce_and_graph_capture():
    """igcn_comm_* (RCCL through the C ABI) on the one GPU this box has: a 1-rank communicator, the all-reduce on
    the launch stream, eagerly and captured into a hipGraph between two kernels."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from igcn_amd.comm import Comm
    comm = Comm(rank=0, world_size=1)
    x = torch.arange(1000, dtype=torch.float32, device="cuda")
    want = x.clone()
    comm.all_reduce_(x)
    torch.cuda.synchronize()
    assert torch.equal(x, want)
    y = torch.ones(4096, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        comm.all_reduce_(y)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            y.mul_(2.0)
            comm.all_reduce_(y)
            y.add_(1.0)
    except Exception as exc:                             # noqa: BLE001
        comm.close()
        pytest.skip(f"this RCCL build refuses stream capture ({type(exc).__name__}): the step uses two graphs")
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert bool((y == 7.0).all())                        # ((1*2)+1)*2+1
    comm.close()
