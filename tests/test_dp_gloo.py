"""Data-parallel path on CPU (gloo, world_size 2): graph-batch sharding, the flat gradient bucket and the
single all-reduce of train.py, checked against the oracle run independently on both shards (DDP convention:
local BatchNorm statistics / local batch-level losses, gradients averaged — SURVEY §8e)."""
import os
import socket
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import assert_matches
from igcn_amd import synth
from igcn_amd.data import Batch
from igcn_amd.train import FlatAdam, allreduce_mean_, shard_batch


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup():
    from oracle import go_network as OG, sgcn_img_snp as OS
    from _weights import seeded_state
    pool = (12, 6, 4, 2, 1)
    go_snps, adj, _ = synth.go_hierarchy(pool, seed=2)
    a_g, a = synth.go_sparse_inputs(go_snps, adj)
    idx = OG.go_index_sets(a_g, a, list(pool), 2)
    shapes = dict(OS.sgcn_param_shapes(2, 4, rois=10))
    shapes.update({"go_network." + k: v for k, v in OG.go_param_shapes(idx, l_dim=32, d_att=8).items()})
    sd = seeded_state(shapes, 3)
    cfg = SimpleNamespace(num_layers=2, rois=10, image_only=False, rbf_gamma=0.01)
    graphs = synth.brain_graph_list(8, seed=4, rois=10, tsne_dim=6)
    return OS, idx, sd, cfg, graphs


def _shard_grads(OS, idx, sd, cfg, graphs):
    st = OS.make_leaf_state(sd)
    data = Batch.from_data_list(graphs)
    data.x.requires_grad_(True)
    loss, _, _ = OS.train_losses(st, cfg, idx, data, dropout=False)
    loss.backward()
    keys = OS.trainable_keys(st)
    return keys, [st[k].grad if st[k].grad is not None else torch.zeros_like(st[k]) for k in keys]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        OS, idx, sd, cfg, graphs = _setup()
        keys, grads = _shard_grads(OS, idx, sd, cfg, shard_batch(graphs, rank, world))
        params = [torch.nn.Parameter(sd[k].clone()) for k in keys]
        opt = FlatAdam(params, lr=1e-3)                 # flat views exist on any device; step() needs the GPU
        for p, g in zip(params, grads):
            p.grad.add_(g)                              # what autograd's AccumulateGrad does in place
        allreduce_mean_(opt.grad, world)
        if rank == 0:
            torch.save({k: p.grad.clone() for k, p in zip(keys, params)}, out)
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_exchange_equals_mean_of_shard_gradients(tmp_path):
    port, out = _free_port(), str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    OS, idx, sd, cfg, graphs = _setup()
    k0, g0 = _shard_grads(OS, idx, sd, cfg, shard_batch(graphs, 0, 2))
    _, g1 = _shard_grads(OS, idx, sd, cfg, shard_batch(graphs, 1, 2))
    for k, a, b in zip(k0, g0, g1):
        # 4-sample training BatchNorm amplifies the thread-count dependent fp32 summation order
        assert_matches(got[k], ((a + b) / 2).numpy(), 2e-3, k, floor=1e-6)


def test_shard_batch_is_a_contiguous_partition():
    graphs = list(range(12))
    parts = [shard_batch(graphs, r, 4) for r in range(4)]
    assert sum(parts, []) == graphs and all(len(p) == 3 for p in parts)
    with pytest.raises(ValueError):
        shard_batch(graphs, 0, 5)


def test_flat_adam_views_and_loud_failure_without_gpu():
    from igcn_amd._lib import IgcnError
    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5))]
    before = [p.detach().clone() for p in ps]
    opt = FlatAdam(ps)
    assert all(torch.equal(p.detach(), b) for p, b in zip(ps, before))
    (ps[0].sum() * 2 + ps[1].sum() * 3).backward()
    assert bool((ps[0].grad == 2.0).all()) and bool((ps[1].grad == 3.0).all())
    assert float(opt.grad.sum()) == 12 * 2.0 + 5 * 3.0            # gradients live in the flat bucket (64-B aligned slots)
    opt.zero_grad()
    assert not opt.grad.any() and ps[0].grad.data_ptr() == opt.grad.data_ptr()
    assert (ps[1].grad.data_ptr() - opt.grad.data_ptr()) % 64 == 0
    with pytest.raises(IgcnError):
        opt.step()                                      # CPU tensors: the HIP path refuses, no fallback
