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
