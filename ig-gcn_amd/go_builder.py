"""GO-DAG builder: PANTHER over-representation JSON + root-connection paths + SNP->gene table -> the tuple the
trainer consumes.

Output contract of the reference's ``parse_go_json`` (snps_graph.py:12-174, ``build_graph`` :251-293,
``build_go_gene_snps`` :224-249; snps_get_root_go_by_html.py ``build_graph_after_loading`` :63-92) as used at
kernel/train_eval_sgcn_img_snps.py:68-71:

    go_snps [N,54]   1 where a GO term's genes include one of the SNP's genes; the root row is all ones
    adj [N,N]        adj[parent, child] = 1, nodes sorted DEEPEST LEVEL FIRST (root last)
    pool_dim         [[n_level4, n_level3, n_level2, n_level1, n_level0]]
    n_l, go_level, go_ids, genes per node

Same algorithm, restated: two passes over the JSON (which terms form the sub-graph; then its edges: a term whose
``level`` exceeds that of an earlier term of the same group becomes that term's parent), the root-connection paths
(first three hops and the last), breadth-first levels from GO:0008150 (the reference's recursive walk computes the
same shortest distances), the reference's exact ``np.argsort(-level)`` ordering call, and the SNP table.  File paths
are arguments instead of the reference's hard-wired ``./data/...``.  Pure host code: runs once per data set.
"""
import json
from collections import deque

import numpy as np

ROOT_ID = "GO:0008150"


def _groups(data):
    for way in data["overrepresentation"]["group"]:
        res = way["result"]
        yield res if isinstance(res, list) else [res]


def _term(t):
    return t["term"]["id"], t["term"]["level"], list(t["input_list"]["mapped_id_list"]["mapped_id"])


def subgraph_ids(data):
    """Pass 1 (snps_graph.py:12-96): the first and the last term of every group and every term after which the
    level stops rising."""
    ids, keep = [], []
    for terms in _groups(data):
        levels, current, last = [], -1, -1
        for i, t in enumerate(terms):
            tid, level, _ = _term(t)
            if tid not in ids:
                ids.append(tid)
            last = ids.index(tid)
            if i == 0 and tid not in keep:
                keep.append(tid)
            if levels and not level > levels[-1]:
                keep.append(ids[current])
            current = last
            levels.append(level)
        if last >= 0:
            keep.append(ids[last])
    return keep


def subgraph_edges(data, keep):
    """Pass 2 (:98-160): ids, per-occurrence gene lists and (parent row, child col) edges of the kept terms."""
    keep = set(keep)
    ids, genes, rows, cols = [], [], [], []
    for terms in _groups(data):
        idx_way, lvl_way = [], []
        for t in terms:
            tid, level, g = _term(t)
            if tid not in keep:
                continue
            genes.append(g)                                   # one entry per kept OCCURRENCE, as the reference
            if tid not in ids:
                ids.append(tid)
            k = ids.index(tid)
            for j in range(len(lvl_way) - 1, -1, -1):         # nearest earlier term of lower level = its child
                if level > lvl_way[j]:
                    cols.append(idx_way[j])
                    rows.append(k)
                    break
            idx_way.append(k)
            lvl_way.append(level)
    return ids, genes, rows, cols


def add_root_connections(lines, ids, rows, cols):
    """snps_get_root_go_by_html.py:63-92: dotted root-to-term paths; hops 0,1,2 and the last term are chained."""
    ids, rows, cols = list(ids), list(rows), list(cols)
    for line in lines:
        parts = line.split(".")
        prev = -1
        for pos, term in enumerate(parts):
            if 2 < pos < len(parts) - 1:
                continue
            tid = "GO:" + term.replace("\n", "")
            if tid not in ids:
                ids.append(tid)
            k = ids.index(tid)
            if prev >= 0:
                cols.append(k)
                rows.append(prev)
            prev = k
    n = len(ids)
    adj = np.zeros((n, n), dtype=np.int64)
    if rows:
        adj[np.asarray(rows), np.asarray(cols)] = 1
    return ids, adj


def levels_from_root(adj, root):
    """Shortest distance from the root along adj[parent, child] (the recursive get_level :176-182 relaxes to the
    same values); unreachable terms keep +inf."""
    n = adj.shape[0]
    level = np.full(n, np.inf)
    level[root] = 0
    q = deque([root])
    children = [np.flatnonzero(adj[i] > 0) for i in range(n)]
    while q:
        i = q.popleft()
        for c in children[i]:
            if level[c] > level[i] + 1:
                level[c] = level[i] + 1
                q.append(int(c))
    return level


def go_snp_matrix(genes_per_node, snps_to_genes, root):
    """build_go_gene_snps :224-249."""
    n, s = len(genes_per_node), len(snps_to_genes)
    out = np.zeros((n, s))
    sets = [set(g) for g in snps_to_genes]
    for i, genes in enumerate(genes_per_node):
        for j in range(s):
            if any(g in sets[j] for g in genes):
                out[i, j] = 1
    out[root, :] = 1
    return out


def read_snps_to_genes(path):
    with open(path) as f:
        return [[g.replace("\n", "") for g in line.split(";")] for line in f]


def build_graph(ids_json, genes_json, ids_all, adj, snps_to_genes):
    """build_graph :251-293."""
    genes = {i: g for i, g in enumerate(genes_json)}          # index = occurrence number, as the reference's map
    for i in range(len(genes_json), len(ids_all)):
        genes[i] = []
    genes_list = [genes[i] for i in range(len(ids_all))]
    root = ids_all.index(ROOT_ID)
    level = levels_from_root(adj, root)
    order = np.argsort(-level)                                # the reference's call: ties in numpy's order
    level = level[order]
    ids_sorted = [ids_all[i] for i in order]
    genes_sorted = [genes_list[i] for i in order]
    adj = adj[order, :][:, order]
    root = ids_sorted.index(ROOT_ID)
    pool_dim = [[np.sum(level == i) for i in range(4, -1, -1)]]
    go_snps = go_snp_matrix(genes_sorted, snps_to_genes, root)
    return go_snps, adj, pool_dim, 4, level, ids_sorted, genes_sorted


def parse_go_json(json_path, connection_path="./data/go_root_connection.txt",
                  snps_to_gene_path="./data/snps_to_gene.txt"):
    """Drop-in for snps_graph.parse_go_json (same 7-tuple), with the two side files as arguments."""
    with open(json_path) as f:
        data = json.load(f)
    keep = subgraph_ids(data)
    ids, genes, rows, cols = subgraph_edges(data, keep)
    ids_json = list(ids)
    with open(connection_path) as f:
        ids_all, adj = add_root_connections(f.readlines(), ids, rows, cols)
    return build_graph(ids_json, genes, ids_all, adj, read_snps_to_genes(snps_to_gene_path))


def model_inputs(go_snps, adj, device="cpu"):
    """(A_g, A) sparse COO tensors exactly as kernel/train_eval_sgcn_img_snps.py:69-70 builds them."""
    import torch
    a_g = torch.tensor(np.asarray(go_snps)).float().to_sparse().coalesce().to(device)
    a = torch.tensor(np.asarray(adj)).float().t().to_sparse().coalesce().to(device)
    return a_g, a
