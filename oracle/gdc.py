"""CPU restatement of the GDC pre-transform + block-diagonal collation (TEST INFRASTRUCTURE ONLY).

Follows /root/reference:
  util_gdc.py:7-15    get_ppr_matrix        -> ppr_matrix
  util_gdc.py:25-31   get_top_k_matrix      -> top_k_matrix
  util_gdc.py:71-86   preprocess_diffusion_imgs_snps (coo_matrix of the dense result: row-major non-zeros,
                      edge_index = [row; col] int64, edge_attr float32)               -> diffusion_topk
  batch.py:98-104     Batch.from_data_list index offsetting                           -> diffusion_topk_batch

numpy float64 like the reference.  Pinned by tests/golden/gdc.npz (the reference's own functions executed on
seeded adjacencies).
"""
import numpy as np


def ppr_matrix(adj, alpha=0.05):
    n = adj.shape[0]
    d = np.diag(1 / np.sqrt(adj.sum(axis=1)))
    h = d @ adj @ d
    return alpha * np.linalg.inv(np.eye(n) - (1 - alpha) * h)


def top_k_matrix(a, k=5):
    a = a.copy()
    n = a.shape[0]
    a[a.argsort(axis=0)[:n - k], np.arange(n)] = 0.0
    norm = a.sum(axis=0)
    norm[norm <= 0] = 1
    return a / norm


def diffusion_topk(adj, top_k=3, alpha=0.05):
    res = top_k_matrix(ppr_matrix(np.asarray(adj, dtype=np.float64), alpha), top_k)
    r, c = np.nonzero(res)
    return np.vstack([r, c]).astype(np.int64), res[r, c].astype(np.float32)


def diffusion_topk_batch(adjs, top_k=3, alpha=0.05):
    """-> (edge_index [2,E] with graph g offset by g*R, edge_attr [E], edge_ptr [B+1])."""
    eis, ews, ptr, off = [], [], [0], 0
    for a in adjs:
        ei, ew = diffusion_topk(a, top_k, alpha)
        eis.append(ei + off)
        ews.append(ew)
        off += a.shape[0]
        ptr.append(ptr[-1] + ei.shape[1])
    return np.concatenate(eis, axis=1), np.concatenate(ews), np.asarray(ptr, dtype=np.int64)
