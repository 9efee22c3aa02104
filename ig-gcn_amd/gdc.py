"""On-device batch construction: the reference's per-subject GDC pre-transform
``preprocess_diffusion_imgs_snps`` (util_gdc.py:71-101: personalised-PageRank diffusion, top-k per column,
column normalisation, COO) and the block-diagonal collation of ``Batch.from_data_list`` (batch.py:24-123), for a
whole batch of dense adjacencies in one kernel launch (igcn_gdc_topk, fp64 in LDS).

The reference runs the transform once per subject in numpy at dataset-build time and collates in a Python loop per
batch; at >= 100 k graphs/s of train-step throughput both are far too slow to feed the GPU.
"""
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr
from .data import Batch


def diffusion_topk(adj, top_k=3, alpha=0.05, check=True):
    """adj [B,R,R] f32 (device) -> (edge_index [2,E] int64 with graph g's nodes offset by g*R, edge_attr [E] f32,
    edge_ptr [B+1] int64).  Edges of a graph are in (row, col) order as scipy's coo_matrix emits them.
    ``check=False``: no host read — for inputs known to keep ``top_k`` entries in every column (a PPR matrix of a
    connected graph is dense), as a per-step producer in front of the graphed train step; a column that kept fewer
    leaves -1 slots behind, which the per-graph plan build reports in its status word."""
    if adj.dim() != 3 or adj.shape[1] != adj.shape[2]:
        raise _lib.IgcnError("adj must be [B,R,R]")
    adj = adj.to(torch.float32).contiguous()
    b, r, _ = adj.shape
    slots = b * r * top_k
    ei = torch.empty(2, slots, dtype=torch.int64, device=adj.device)
    ew = torch.empty(slots, dtype=torch.float32, device=adj.device)
    counts = torch.empty(b, dtype=torch.int32, device=adj.device)
    call("igcn_gdc_topk", b, r, int(top_k), float(alpha), ptr(adj), ptr(ei), ptr(ew), ptr(counts), stream_ptr())
    edge_ptr = torch.zeros(b + 1, dtype=torch.int64, device=adj.device)
    torch.cumsum(counts, 0, out=edge_ptr[1:])
    # pre-transform time, not the train step: one host read decides whether padding slots must be squeezed out
    if check and int(edge_ptr[-1]) != slots:
        valid = ei[0] >= 0
        ei, ew = ei[:, valid].contiguous(), ew[valid].contiguous()
    return ei, ew, edge_ptr


def batch_from_dense(adj, x, top_k=3, alpha=0.05, check=True, **per_graph):
    """A ``Batch`` (the attribute surface of data.Batch.from_data_list) straight from device tensors:
    adj [B,R,R], x [B*R,H0] or [B,R,H0]; ``per_graph`` tensors with leading dim B (snps_feat [B,54], y [B],
    clini_score [B,n] -> flattened like the reference's collation, tsne_fdim [B,F], clust_y [B], sbjID [B])."""
    b, r, _ = adj.shape
    ei, ew, edge_ptr = diffusion_topk(adj, top_k, alpha, check)
    out = Batch()
    out.x = x.reshape(b * r, -1).contiguous()
    out.edge_index, out.edge_attr = ei, ew
    out.A = adj.reshape(b * r, r)
    out.batch = torch.arange(b, device=adj.device).repeat_interleave(r)
    for key, val in per_graph.items():
        if key == "clini_score" or key == "demographics":
            val = val.reshape(-1)                       # 1-D per-graph attributes concatenate (batch.py:110)
        setattr(out, key, val)
    out._num_graphs = b
    out.ptr = torch.arange(b + 1, device=adj.device, dtype=torch.int64) * r
    out.edge_ptr = edge_ptr
    out._max_nodes = r
    out._max_edges = r * top_k
    return out
