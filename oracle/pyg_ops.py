"""Restatement of the third-party graph operators the reference calls (TEST INFRASTRUCTURE).

PyG 2.0.2 / torch-scatter 2.0.9 are pinned by the reference's environment.yml:183,211 but are
not vendored under /root/reference and cannot be installed here, so their *published* semantics
are restated.  **Parity unpinned by the reference**: these functions are
pinned only by the fp64 dense known-answer tests in tests/test_oracle_pyg_ops.py.

Call sites in the reference that these stand for:
  GCNConv            kernel/sgcn_img_snp.py:34,40,42,49 (ctor)  :218,221 (forward)
  to_dense_batch     kernel/sgcn_img_snp.py:226,265,294
  global_*_pool      kernel/sgcn_img_snp.py:231-233,248-250 (graph_pool=True)
  scatter(sum)       kernel/go_model.py:200
"""
import math

import torch


def gcn_norm(edge_index, edge_weight, num_nodes):
    """Symmetric GCN normalisation with ``add_remaining_self_loops`` (fill 1.0).

    Returns (edge_index' [2,E'], w_hat [E']): the non-loop edges in their original order followed
    by one loop per node in node order; an existing loop's weight replaces the 1.0 fill (when a node
    carries several stored loops the last one in edge order wins, as a sequential index_put does).
    """
    src, dst = edge_index[0], edge_index[1]
    if edge_weight is None:
        edge_weight = torch.ones(src.numel(), dtype=torch.float32, device=src.device)
    keep = src != dst
    loop_w = torch.ones(num_nodes, dtype=edge_weight.dtype, device=edge_weight.device)
    # differentiable "last stored loop wins": index_put_ is sequential on CPU
    loops = (~keep).nonzero().view(-1)
    if loops.numel() > 0:
        loop_w = loop_w.index_put((src[loops],), edge_weight[loops])
    ar = torch.arange(num_nodes, dtype=src.dtype, device=src.device)
    src2 = torch.cat([src[keep], ar])
    dst2 = torch.cat([dst[keep], ar])
    w2 = torch.cat([edge_weight[keep], loop_w])
    deg = torch.zeros(num_nodes, dtype=w2.dtype, device=w2.device).index_add(0, dst2, w2)
    dis = deg.pow(-0.5)
    dis = dis.masked_fill(dis == float("inf"), 0.0)
    w_hat = dis[src2] * w2 * dis[dst2]
    return torch.stack([src2, dst2]), w_hat


def gcn_conv(x, edge_index, edge_weight, weight, bias):
    """One GCNConv forward: normalise -> x @ W^T -> sum messages src->dst -> + bias.

    ``weight`` is ``<conv>.lin.weight`` [out,in]; ``bias`` is ``<conv>.bias`` [out].
    """
    n = x.shape[0]
    ei, w_hat = gcn_norm(edge_index, edge_weight, n)
    h = x @ weight.t()
    msg = w_hat.unsqueeze(1) * h[ei[0]]
    out = torch.zeros(n, h.shape[1], dtype=h.dtype, device=h.device).index_add(0, ei[1], msg)
    return out + bias


def gcn_conv_dense_fp64(x, edge_index, edge_weight, weight, bias):
    """Known-answer form: D^-1/2 (A_w^T + I') D^-1/2 X W^T + b, dense, fp64, numpy-free torch."""
    n = x.shape[0]
    a = torch.zeros(n, n, dtype=torch.float64)          # a[dst, src]
    loop = torch.ones(n, dtype=torch.float64)
    src, dst = edge_index[0].tolist(), edge_index[1].tolist()
    w = edge_weight.double().tolist()
    for s, d, v in zip(src, dst, w):
        if s == d:
            loop[s] = v
        else:
            a[d, s] += v
    a = a + torch.diag(loop)
    deg = a.sum(dim=1)
    dis = torch.where(deg == 0, torch.zeros_like(deg), deg.pow(-0.5))
    a_hat = dis.unsqueeze(1) * a * dis.unsqueeze(0)
    return a_hat @ (x.double() @ weight.double().t()) + bias.double()


def to_dense_batch(x, batch, fill_value=0.0):
    """[N,F] node rows -> ([B, N_max, F], mask [B, N_max]); padding rows hold ``fill_value``."""
    b = int(batch.max()) + 1 if batch.numel() else 0
    counts = torch.zeros(b, dtype=torch.long, device=x.device).index_add(
        0, batch, torch.ones_like(batch))
    starts = torch.cat([counts.new_zeros(1), counts.cumsum(0)[:-1]])
    n_max = int(counts.max()) if b else 0
    pos = torch.arange(batch.numel(), device=x.device) - starts[batch] + batch * n_max
    out = x.new_full((b * n_max, x.shape[1]), fill_value)
    out = out.index_put((pos,), x)
    mask = torch.zeros(b * n_max, dtype=torch.bool, device=x.device)
    mask[pos] = True
    return out.view(b, n_max, x.shape[1]), mask.view(b, n_max)


def global_pools(x, batch):
    """cat(global_mean_pool, global_max_pool, global_add_pool)(x, batch) of PyG 2.0.2
    (call sites kernel/sgcn_img_snp.py:231-235,248-252): per-graph reductions over the node rows.  The max routes
    its gradient to the FIRST maximal row (torch-scatter's CPU scatter_max updates on strictly-greater only)."""
    b = int(batch.max()) + 1 if batch.numel() else 0
    means, maxs, adds = [], [], []
    for g in range(b):
        rows = x[batch == g]
        adds.append(rows.sum(0))
        means.append(rows.mean(0))
        mx, am = rows.max(dim=0)
        first = (rows == mx.unsqueeze(0)).float().argmax(dim=0)      # first occurrence of the maximum
        maxs.append(rows.gather(0, first.unsqueeze(0)).squeeze(0))
    return torch.cat([torch.stack(means), torch.stack(maxs), torch.stack(adds)], dim=1)


def global_mean_pool(x, batch):
    d = x.shape[1]
    return global_pools(x, batch)[:, :d]


def global_max_pool(x, batch):
    d = x.shape[1]
    return global_pools(x, batch)[:, d:2 * d]


def global_add_pool(x, batch):
    d = x.shape[1]
    return global_pools(x, batch)[:, 2 * d:]


def scatter_sum_dim1(src, index, dim_size):
    """torch_scatter.scatter(src, index, dim=1, reduce='sum', out=zeros) == index_add along dim 1."""
    out = torch.zeros(src.shape[0], dim_size, *src.shape[2:], dtype=src.dtype, device=src.device)
    return out.index_add(1, index, src)


def glorot_(t):
    """PyG's glorot init for ``GCNConv.lin.weight`` (uniform +-sqrt(6/(fan_in+fan_out)))."""
    a = math.sqrt(6.0 / (t.shape[-2] + t.shape[-1]))
    with torch.no_grad():
        t.uniform_(-a, a)
    return t


class GCNConvModule(torch.nn.Module):
    """nn.Module wrapper with PyG-2.0.2 state_dict keys (``lin.weight`` [out,in], ``bias`` [out]).

    Used (a) by tests/golden/make_golden.py as the stand-in registered under
    ``torch_geometric.nn.GCNConv`` when the reference model file is executed, and (b) nowhere else.
    """

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = torch.nn.Linear(in_channels, out_channels, bias=False)
        self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.lin.weight)
        with torch.no_grad():
            self.bias.zero_()

    def forward(self, x, edge_index, edge_weight=None):
        return gcn_conv(x, edge_index, edge_weight, self.lin.weight, self.bias)
