"""Collation parity: ``igcn_amd.data.Batch.from_data_list`` against the reference's own
``batch.py:24-123`` (+ ``num_graphs`` :188-191), captured by tests/golden/make_golden.py::capture_collate on the same
seeded graph lists — uniform 90-ROI brain graphs with the attribute set of sgcn_data.py:262-282, and a ragged list
with an extra ``*_index`` key, a bool tensor and python scalars (the ``__cat_dim__`` / ``__inc__`` contract).
Integer / index tensors are compared bit for bit, floating tensors too (collation only moves bytes)."""
import numpy as np
import pytest
import torch

from igcn_amd import synth
from igcn_amd.data import Batch, DataLoader
from make_golden import ragged_graph_list


def _lists(store):
    n, seed, rois, top_k, tsne = [int(v) for v in store["brain/cfg"]]
    return {"brain": synth.brain_graph_list(n, seed=seed, rois=rois, top_k=top_k, tsne_dim=tsne),
            "ragged": ragged_graph_list(int(store["ragged/seed"]))}


@pytest.mark.parametrize("tag", ["brain", "ragged"])
def test_from_data_list_equals_reference_batch(golden, tag):
    store = golden("batch_collate")
    b = Batch.from_data_list(_lists(store)[tag])
    ref_keys = [str(k) for k in store[f"{tag}/keys"]]
    # the reference's key set, plus the host-known offsets this build adds (ptr / edge_ptr: no reference counterpart)
    assert sorted(set(b.keys) - {"ptr", "edge_ptr"}) == sorted(ref_keys)
    for k in ref_keys:
        want = store[f"{tag}/{k}"]
        got = b[k]
        got = got.numpy() if torch.is_tensor(got) else np.asarray(got)
        assert got.shape == want.shape, (k, got.shape, want.shape)
        assert got.dtype == want.dtype, (k, got.dtype, want.dtype)
        assert np.array_equal(got, want), k
    assert b.num_graphs == int(store[f"{tag}/num_graphs"])
    # the offsets agree with the reference's batch vector / edge ranges
    bvec = store[f"{tag}/batch"]
    assert np.array_equal(np.searchsorted(bvec, np.arange(b.num_graphs + 1)), b.ptr.numpy())
    ei = store[f"{tag}/edge_index"]
    owner = bvec[ei[0]]
    assert np.all(np.diff(owner) >= 0)
    assert np.array_equal(np.searchsorted(owner, np.arange(b.num_graphs + 1)), b.edge_ptr.numpy())


def test_num_graphs_fallback_reads_the_batch_vector(golden):
    store = golden("batch_collate")
    b = Batch(batch=torch.from_numpy(store["brain/batch"]))
    assert b.num_graphs == int(store["brain/num_graphs"])          # batch.py:188-191


def test_dataloader_collates_like_the_reference(golden):
    store = golden("batch_collate")
    graphs = _lists(store)["brain"]
    batches = list(DataLoader(graphs, batch_size=len(graphs), shuffle=False))
    assert len(batches) == 1
    assert np.array_equal(batches[0].edge_index.numpy(), store["brain/edge_index"])
    assert np.array_equal(batches[0].x.numpy(), store["brain/x"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["brain", "ragged"])
def test_graph_plan_from_reference_collation_is_bit_exact(golden, tag):
    """The graph plan built on the device from the REFERENCE-collated edge_index equals a stable numpy sort."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from igcn_amd import ops
    store = golden("batch_collate")
    b = Batch.from_data_list(_lists(store)[tag]).to("cuda")
    ei = store[f"{tag}/edge_index"]
    assert np.array_equal(b.edge_index.cpu().numpy(), ei)
    plan = ops.plan_for(b)
    plan.check()
    n = int(store[f"{tag}/x"].shape[0])
    for key, ptr_t, perm_t in ((ei[1], plan.tgt_ptr, plan.tgt_perm), (ei[0], plan.src_ptr, plan.src_perm)):
        perm = np.argsort(key, kind="stable")
        assert np.array_equal(perm_t.cpu().numpy()[:ei.shape[1]], perm.astype(np.int32))
        assert np.array_equal(ptr_t.cpu().numpy(), np.searchsorted(key[perm], np.arange(n + 1)).astype(np.int32))
    assert np.array_equal(plan.src32.cpu().numpy()[:ei.shape[1]].astype(np.int64), ei[0])
    assert np.array_equal(plan.dst32.cpu().numpy()[:ei.shape[1]].astype(np.int64), ei[1])
