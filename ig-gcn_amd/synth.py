"""Seeded synthetic ADNI-shaped inputs: 90-ROI brain graphs and a GO-SNP hierarchy.

Nothing here comes from the reference's (absent) data files.  The brain adjacency follows the
*statistics* the reference's pre-transform produces (util_gdc.py:7-31,71-86: PPR alpha=0.05, keep the
top-k entries per column, column-normalise, COO in row-major order with edge_index=[row;col]); the GO
hierarchy follows the output contract of snps_graph.py:251-293 (nodes sorted deepest level first,
``pool_dim=[[n4,n3,n2,n1,n0]]``, ``go_snps`` [N,54] with an all-ones root row, ``adj[parent,child]``).
"""
import numpy as np
import torch

from .data import Batch, Data

N_SNPS = 54


def diffusion_topk_graph(rng, rois=90, knn=5, alpha=0.05, top_k=3):
    """One brain graph: random symmetric kNN similarity -> PPR -> top-k per column -> column-normalise.

    Returns (edge_index [2,E] int64 row-major COO, edge_attr [E] f32, dense A [rois,rois] f32).
    """
    s = rng.random((rois, rois))
    s = (s + s.T) / 2
    np.fill_diagonal(s, 0.0)
    nbr = np.argsort(-s, axis=1)[:, :knn]
    a = np.zeros((rois, rois))
    rows = np.repeat(np.arange(rois), knn)
    a[rows, nbr.ravel()] = s[rows, nbr.ravel()]
    a = np.maximum(a, a.T)
    dinv = 1.0 / np.sqrt(a.sum(axis=1))
    h = dinv[:, None] * a * dinv[None, :]
    ppr = alpha * np.linalg.inv(np.eye(rois) - (1 - alpha) * h)
    drop = np.argsort(ppr, axis=0)[: rois - top_k]
    ppr[drop, np.arange(rois)] = 0.0
    norm = ppr.sum(axis=0)
    norm[norm <= 0] = 1
    ppr = ppr / norm
    r, c = np.nonzero(ppr)
    ei = torch.from_numpy(np.vstack([r, c])).long()
    ew = torch.from_numpy(ppr[r, c]).float()
    return ei, ew, torch.from_numpy(a).float()


def dense_graph(rng, rois):
    """Stress-config graph: dense adjacency, U(0,1] weights, column-normalised."""
    w = 1.0 - rng.random((rois, rois))
    w = w / w.sum(axis=0, keepdims=True)
    r, c = np.nonzero(w)
    return (torch.from_numpy(np.vstack([r, c])).long(), torch.from_numpy(w[r, c]).float(),
            torch.from_numpy(w).float())


def brain_graph_list(n_graphs, seed=1000, rois=90, h0=3, top_k=3, num_classes=3, num_regr=3,
                     tsne_dim=90, dense=False):
    """List of ``Data`` with the attribute set of sgcn_data.py:262-282."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_graphs):
        ei, ew, a = dense_graph(rng, rois) if dense else diffusion_topk_graph(rng, rois, top_k=top_k)
        out.append(Data(
            x=torch.from_numpy(rng.random((rois, h0))).float(),
            edge_index=ei, edge_attr=ew, A=a,
            y=torch.tensor([int(rng.integers(num_classes))]),
            clust_y=torch.tensor([int(rng.integers(2))]),
            snps_feat=torch.from_numpy(rng.random((1, N_SNPS))).float(),
            sbjID=torch.tensor([i]),
            tsne_fdim=torch.from_numpy(rng.random((1, tsne_dim))).float(),
            clini_score=torch.from_numpy(rng.random(num_regr)).float(),
            demographics=torch.from_numpy(rng.random(9)).float()))
    return out


def brain_batch(n_graphs, **kw):
    return Batch.from_data_list(brain_graph_list(n_graphs, **kw))


def go_hierarchy(pool=(1800, 800, 300, 99, 1), seed=0, p_snp=0.05, max_parents=2):
    """Synthetic GO DAG.  Returns (go_snps [N,54] f32 numpy, adj [N,N] f32 numpy with adj[parent,child]=1,
    pool_dim [[...]]) — the tuple shape of snps_graph.py:291-293 that the trainer consumes at
    train_eval_sgcn_img_snps.py:68-71."""
    rng = np.random.default_rng(seed)
    pool = [int(p) for p in pool]
    n = sum(pool)
    starts = np.cumsum([0] + pool)
    adj = np.zeros((n, n), dtype=np.float32)
    for lvl in range(len(pool) - 1):
        lo, hi = starts[lvl], starts[lvl + 1]
        plo, phi = starts[lvl + 1], starts[lvl + 2]
        for child in range(lo, hi):
            k = int(rng.integers(1, max_parents + 1))
            k = min(k, phi - plo)
            parents = rng.choice(np.arange(plo, phi), size=k, replace=False)
            adj[parents, child] = 1.0
    go_snps = (rng.random((n, N_SNPS)) < p_snp).astype(np.float32)
    go_snps[n - 1, :] = 1.0
    return go_snps, adj, [pool]


def go_sparse_inputs(go_snps, adj, device="cpu"):
    """(A_g, A) exactly as the trainer builds them (train_eval_sgcn_img_snps.py:69-70)."""
    a = torch.tensor(adj).float().t().to_sparse().coalesce().to(device)
    a_g = torch.tensor(go_snps).float().to_sparse().coalesce().to(device)
    return a_g, a
