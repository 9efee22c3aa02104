"""GDC pre-transform + on-device collation: oracle vs the reference's own util_gdc.py outputs (CPU),
igcn_gdc_topk vs oracle / goldens (GPU; edge indices bit-exact, weights to fp32 rounding)."""
import numpy as np
import pytest
import torch

from oracle import gdc as OG


def _cases(store):
    c = 0
    while f"case{c}/A" in store:
        rois, k, _ = [int(v) for v in store[f"case{c}/cfg"]]
        yield c, rois, k, float(store[f"case{c}/alpha"]), store[f"case{c}/A"], store[f"case{c}/edge_index"], \
            store[f"case{c}/edge_attr"]
        c += 1


def test_oracle_matches_reference(golden):
    store = golden("gdc")
    n = 0
    for c, rois, k, alpha, a, ei, ew in _cases(store):
        got_ei, got_ew = OG.diffusion_topk(a, k, alpha)
        assert np.array_equal(got_ei, ei), c
        assert np.array_equal(got_ew, ew), c                      # same float64 arithmetic -> identical float32
        n += 1
    assert n == 5


def test_reference_edge_statistics(golden):
    """exactly k stored entries per column, every column sums to 1, all diagonal entries survive top-3."""
    store = golden("gdc")
    _, rois, k, alpha, a, ei, ew = next(_cases(store))
    assert ei.shape[1] == rois * k
    col_sum = np.zeros(rois)
    np.add.at(col_sum, ei[1], ew.astype(np.float64))
    assert np.allclose(col_sum, 1.0, atol=1e-6)
    assert int((ei[0] == ei[1]).sum()) == rois


@pytest.mark.gpu
def test_gdc_kernel_vs_reference_golden(golden):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from igcn_amd.gdc import diffusion_topk
    store = golden("gdc")
    for c, rois, k, alpha, a, ei, ew in _cases(store):
        adj = torch.from_numpy(a).cuda()[None]
        got_ei, got_ew, edge_ptr = diffusion_topk(adj, k, alpha)
        assert edge_ptr.tolist() == [0, ei.shape[1]], c
        assert np.array_equal(got_ei.cpu().numpy(), ei), c                                   # bit-exact indexing
        np.testing.assert_allclose(got_ew.cpu().numpy(), ew, rtol=3e-7, atol=0, err_msg=str(c))


@pytest.mark.gpu
@pytest.mark.parametrize("bsz,rois,k", [(64, 90, 3), (7, 40, 5), (256, 90, 3)])
def test_gdc_batch_vs_oracle(bsz, rois, k):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from igcn_amd import ops
    from igcn_amd.gdc import batch_from_dense
    rng = np.random.default_rng(bsz)
    adjs = []
    for _ in range(bsz):
        s = rng.random((rois, rois))
        s = (s + s.T) / 2
        np.fill_diagonal(s, 0.0)
        s[s < np.sort(s, axis=1)[:, -6][:, None]] = 0.0
        adjs.append(np.maximum(s, s.T).astype(np.float32))
    want_ei, want_ew, want_ptr = OG.diffusion_topk_batch(adjs, k)
    adj = torch.from_numpy(np.stack(adjs)).cuda()
    x = torch.rand(bsz * rois, 3, device="cuda")
    batch = batch_from_dense(adj, x, top_k=k, snps_feat=torch.rand(bsz, 54, device="cuda"),
                             clini_score=torch.rand(bsz, 3, device="cuda"))
    assert np.array_equal(batch.edge_index.cpu().numpy(), want_ei)
    np.testing.assert_allclose(batch.edge_attr.cpu().numpy(), want_ew, rtol=3e-7, atol=0)
    assert np.array_equal(batch.edge_ptr.cpu().numpy(), want_ptr)
    assert batch.num_graphs == bsz and batch.clini_score.shape == (bsz * 3,)
    assert batch.batch.tolist() == np.repeat(np.arange(bsz), rois).tolist()
    plan = ops.plan_for(batch)                      # the collated batch feeds the one-launch segmented plan build
    assert plan.segmented
    plan.check()
    assert np.array_equal(plan.tgt_perm.cpu().numpy(), np.argsort(want_ei[1], kind="stable").astype(np.int32))


@pytest.mark.gpu
def test_gdc_isolated_entries_are_squeezed_out():
    """A kept entry that is exactly 0 is not an edge for scipy's coo_matrix: fewer than R*k edges, no padding left."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from igcn_amd.gdc import diffusion_topk
    a = np.zeros((2, 6, 6), dtype=np.float32)
    a[0, :3, :3] = 1 - np.eye(3)                  # two disconnected triangles: PPR is block diagonal, so a
    a[0, 3:, 3:] = 1 - np.eye(3)                  # column has only 3 non-zeros and top-4 keeps an exact zero
    r = np.random.default_rng(0).random((6, 6))
    a[1] = ((r + r.T) / 2 * (1 - np.eye(6))).astype(np.float32)          # no exact ties (tie order is unspecified)
    want_ei, want_ew, want_ptr = OG.diffusion_topk_batch(list(a), 4)
    ei, ew, ptr = diffusion_topk(torch.from_numpy(a).cuda(), 4)
    assert ptr.tolist() == want_ptr.tolist() and ptr[-1] < 2 * 6 * 4
    assert np.array_equal(ei.cpu().numpy(), want_ei)
    np.testing.assert_allclose(ew.cpu().numpy(), want_ew, rtol=3e-7)


@pytest.mark.gpu
def test_gdc_rejects_too_many_rois():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from igcn_amd import _lib
    from igcn_amd.gdc import diffusion_topk
    big = _lib.load().igcn_gdc_topk_max_rois() + 1
    with pytest.raises(_lib.IgcnError):
        diffusion_topk(torch.rand(1, big, big, device="cuda"), 3)
