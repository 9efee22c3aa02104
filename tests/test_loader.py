"""The vectorised collation of uniform brain graphs (igcn_amd/loader.py) moves the same bytes as the reference-pinned
``Batch.from_data_list`` (tests/test_collate_golden.py pins THAT against the reference's batch.py:24-123)."""
import numpy as np
import pytest
import torch

from igcn_amd import synth
from igcn_amd.data import Batch, Data
from igcn_amd.loader import UniformGraphStore, collate_uniform
from make_golden import ragged_graph_list


def _same(a, b):
    assert sorted(a.keys) == sorted(b.keys)
    for k in a.keys:
        x, y = a[k], b[k]
        assert torch.is_tensor(x) and torch.is_tensor(y), k
        assert x.dtype == y.dtype and x.shape == y.shape, (k, x.dtype, y.dtype, x.shape, y.shape)
        assert torch.equal(x, y), k
    assert torch.equal(a.batch, b.batch)
    assert a.num_graphs == b.num_graphs and a._max_nodes == b._max_nodes and a._max_edges == b._max_edges


def test_collate_uniform_equals_from_data_list(golden):
    store = golden("batch_collate")
    n, seed, rois, top_k, tsne = [int(v) for v in store["brain/cfg"]]
    graphs = synth.brain_graph_list(n, seed=seed, rois=rois, top_k=top_k, tsne_dim=tsne)
    got = collate_uniform(graphs)
    _same(got, Batch.from_data_list(graphs))
    for k in [str(k) for k in store["brain/keys"]]:                 # and therefore the reference's own collation
        assert np.array_equal(got[k].numpy(), store[f"brain/{k}"]), k


def test_collate_uniform_falls_back_on_ragged_lists(golden):
    store = golden("batch_collate")
    graphs = ragged_graph_list(int(store["ragged/seed"]))
    got, want = collate_uniform(graphs), Batch.from_data_list(graphs)
    for k in want.keys:
        if torch.is_tensor(want[k]):
            assert torch.equal(got[k], want[k]), k
        else:
            assert got[k] == want[k], k


def test_store_batches_equal_from_data_list_and_reuse_their_slot():
    graphs = synth.brain_graph_list(24, seed=5, rois=30, top_k=3, tsne_dim=7)
    graphs.append(Data(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in graphs[3].__dict__.items()}))
    store = UniformGraphStore(graphs)
    idx = torch.tensor([5, 0, 24, 17, 5, 9])
    want = Batch.from_data_list([graphs[i] for i in idx.tolist()])
    first = store.batch(idx)
    _same(first, want)
    ptrs = {k: first[k].data_ptr() for k in first.keys}
    idx2 = torch.tensor([1, 2, 3, 23, 22, 21])
    again = store.batch(idx2, out=first)
    assert again is first and all(first[k].data_ptr() == p for k, p in ptrs.items())
    _same(first, Batch.from_data_list([graphs[i] for i in idx2.tolist()]))
    with pytest.raises(ValueError):
        UniformGraphStore(ragged_graph_list(3))


def test_epoch_index_visits_every_subject_once_per_epoch():
    """The feeders' index order is DataLoader(shuffle=True)'s (kernel/train_eval_sgcn_img_snps.py:96-97): a permutation per
    epoch cut into batches, reproducible from the seed; the ragged tail is dropped or handed out as a short slice."""
    from igcn_amd.loader import EpochIndex
    ix = EpochIndex(10, 4, seed=3, shuffle=True, drop_last=True)
    assert ix.per_epoch() == 2
    e1 = torch.cat([ix.next(), ix.next()])
    e2 = torch.cat([ix.next(), ix.next()])
    assert ix.epoch == 2 and len(set(e1.tolist())) == 8 and len(set(e2.tolist())) == 8
    assert e1.tolist() != e2.tolist()
    again = EpochIndex(10, 4, seed=3, shuffle=True, drop_last=True)
    assert torch.equal(torch.cat([again.next(), again.next()]), e1)
    full = EpochIndex(10, 4, seed=3, shuffle=True, drop_last=False)
    parts = [full.next() for _ in range(3)]
    assert [p.numel() for p in parts] == [4, 4, 2] and sorted(torch.cat(parts).tolist()) == list(range(10))
    assert full.next().numel() == 4 and full.epoch == 2
    seq = EpochIndex(6, 3, shuffle=False)
    assert [seq.next().tolist() for _ in range(3)] == [[0, 1, 2], [3, 4, 5], [0, 1, 2]]
    with pytest.raises(ValueError):
        EpochIndex(3, 4)
