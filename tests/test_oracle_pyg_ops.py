"""Known-answer tests for the restated PyG/torch-scatter operators (oracle/pyg_ops.py).

PyG 2.0.2 is not in /root/reference and not installable: these operators are *unpinned by the
reference*; the pin is the published dense formula  out = D^-1/2 (A_w^T + I') D^-1/2 X W^T + b  in fp64.
"""
import numpy as np
import pytest
import torch

from oracle import pyg_ops


def _random_graph(rng, n, e, loops=True, isolated=True):
    src = rng.integers(0, n, e)
    dst = rng.integers(0, n, e)
    if not loops:
        dst = np.where(dst == src, (dst + 1) % n, dst)
    if isolated and n > 2:
        src = np.where(src == n - 1, 0, src)
        dst = np.where(dst == n - 1, 1, dst)
    # unique (src,dst) pairs so the dense matrix is well defined edge by edge
    pairs = sorted(set(zip(src.tolist(), dst.tolist())))
    ei = torch.tensor(pairs, dtype=torch.long).t().contiguous()
    w = torch.from_numpy(rng.random(ei.shape[1]) + 0.1).float()
    return ei, w


def _dense_autograd(x, ei, w, weight, bias):
    """Differentiable dense fp64 GCN layer (self-loop weight = stored loop weight or 1)."""
    n = x.shape[0]
    src, dst = ei[0], ei[1]
    nl = src != dst
    a = torch.zeros(n, n, dtype=torch.float64).index_put((dst[nl], src[nl]), w[nl], accumulate=True)
    loop = torch.ones(n, dtype=torch.float64).index_put((src[~nl],), w[~nl])
    a = a + torch.diag(loop)
    deg = a.sum(1)
    dis = torch.where(deg == 0, torch.zeros_like(deg), deg.clamp_min(1e-300).pow(-0.5))
    return (dis[:, None] * a * dis[None, :]) @ (x @ weight.t()) + bias


@pytest.mark.parametrize("seed", range(6))
def test_gcn_conv_matches_dense_fp64(seed):
    rng = np.random.default_rng(seed)
    n, f_in, f_out = int(rng.integers(3, 40)), int(rng.integers(1, 8)), int(rng.integers(1, 9))
    ei, w = _random_graph(rng, n, int(rng.integers(1, 5 * n)), loops=seed % 2 == 0)
    x = torch.from_numpy(rng.standard_normal((n, f_in))).float()
    weight = torch.from_numpy(rng.standard_normal((f_out, f_in))).float()
    bias = torch.from_numpy(rng.standard_normal(f_out)).float()
    got = pyg_ops.gcn_conv(x, ei, w, weight, bias)
    want = pyg_ops.gcn_conv_dense_fp64(x, ei, w, weight, bias)
    assert torch.allclose(got.double(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("seed", range(4))
def test_gcn_conv_gradients_fp64(seed):
    rng = np.random.default_rng(100 + seed)
    n = 12
    ei, w = _random_graph(rng, n, 40, loops=True)
    x = torch.from_numpy(rng.standard_normal((n, 3))).double().requires_grad_(True)
    w = w.double().requires_grad_(True)
    weight = torch.from_numpy(rng.standard_normal((5, 3))).double().requires_grad_(True)
    bias = torch.from_numpy(rng.standard_normal(5)).double().requires_grad_(True)
    cot = torch.from_numpy(rng.standard_normal((n, 5))).double()
    g1 = torch.autograd.grad((pyg_ops.gcn_conv(x, ei, w, weight, bias) * cot).sum(), [x, w, weight, bias])
    g2 = torch.autograd.grad((_dense_autograd(x, ei, w, weight, bias) * cot).sum(), [x, w, weight, bias])
    for a, b in zip(g1, g2):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-10)


def test_gcn_norm_edge_order_and_loop_weights():
    ei = torch.tensor([[0, 1, 1, 2, 2], [1, 1, 2, 0, 2]])
    w = torch.tensor([0.5, 0.25, 2.0, 1.0, 4.0])
    ei2, w_hat = pyg_ops.gcn_norm(ei, w, 4)
    # non-loop edges in original order, then one loop per node in node order
    assert ei2.tolist() == [[0, 1, 2, 0, 1, 2, 3], [1, 2, 0, 0, 1, 2, 3]]
    deg = torch.tensor([1.0 + 1.0, 0.5 + 0.25, 2.0 + 4.0, 1.0])
    dis = deg.pow(-0.5)
    raw = torch.tensor([0.5, 2.0, 1.0, 1.0, 0.25, 4.0, 1.0])
    assert torch.allclose(w_hat, dis[ei2[0]] * raw * dis[ei2[1]])


def test_to_dense_batch_uniform_is_a_view_and_ragged_pads():
    x = torch.arange(12.0).view(6, 2)
    out, mask = pyg_ops.to_dense_batch(x, torch.tensor([0, 0, 0, 1, 1, 1]), -7.0)
    assert torch.equal(out, x.view(2, 3, 2)) and mask.all()
    out, mask = pyg_ops.to_dense_batch(x, torch.tensor([0, 1, 1, 1, 1, 2]), -7.0)
    assert out.shape == (3, 4, 2)
    assert torch.equal(out[0, 0], x[0]) and torch.all(out[0, 1:] == -7.0)
    assert torch.equal(out[1], x[1:5]) and torch.equal(out[2, 0], x[5])
    assert mask.sum() == 6


def test_scatter_sum_is_index_add():
    src = torch.arange(24.0).view(2, 4, 3)
    idx = torch.tensor([2, 0, 2, 1])
    out = pyg_ops.scatter_sum_dim1(src, idx, 3)
    assert torch.equal(out[:, 2], src[:, 0] + src[:, 2]) and torch.equal(out[:, 0], src[:, 1])
