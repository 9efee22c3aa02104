"""GO-DAG builder against the reference's own parse_go_json run on synthetic PANTHER-format inputs
(tests/golden/go_builder.npz holds the three input files and the reference's 7-tuple)."""
import json

import numpy as np

from igcn_amd import go_builder


def _inputs(store, tmp_path):
    (tmp_path / "analysis.json").write_text(str(store["json"]))
    (tmp_path / "conn.txt").write_text(str(store["connection"]))
    (tmp_path / "s2g.txt").write_text(str(store["snps_to_gene"]))
    return str(tmp_path / "analysis.json"), str(tmp_path / "conn.txt"), str(tmp_path / "s2g.txt")


def test_builder_matches_reference(golden, tmp_path):
    store = golden("go_builder")
    go_snps, adj, pool_dim, n_l, level, ids, genes = go_builder.parse_go_json(*_inputs(store, tmp_path))
    assert ids == store["ids"].tolist()                                   # node order: deepest level first
    assert np.array_equal(np.asarray(adj), store["adj"])
    assert np.array_equal(np.asarray(pool_dim), store["pool_dim"]) and n_l == int(store["n_l"])
    assert np.array_equal(level, store["go_level"])
    assert np.array_equal(go_snps, store["go_snps"])
    assert genes == json.loads(str(store["genes"]))


def test_builder_output_contract(golden, tmp_path):
    """What the kernels rely on: levels sorted descending with the root last, pool_dim = level counts, an all-ones
    root row in go_snps, edges only from a level to the next deeper one or beyond."""
    store = golden("go_builder")
    go_snps, adj, pool_dim, n_l, level, ids, _ = go_builder.parse_go_json(*_inputs(store, tmp_path))
    finite = level[np.isfinite(level)]
    assert np.all(np.diff(finite) <= 0) and ids[-1] == go_builder.ROOT_ID and level[-1] == 0
    assert sum(pool_dim[0]) == int(np.sum(level <= 4)) and pool_dim[0][-1] == 1
    assert np.all(go_snps[-1] == 1) and go_snps.shape[1] == 54
    r, c = np.nonzero(adj)
    assert np.all(level[c] <= level[r] + 1)                               # a child is at most one level deeper
    a_g, a = go_builder.model_inputs(go_snps, adj)
    assert a_g.shape == (len(ids), 54) and a.shape == (len(ids), len(ids)) and a.is_coalesced()


def test_levels_equal_recursive_relaxation():
    """levels_from_root (breadth first) == the reference's recursive relaxation on a random DAG."""
    rng = np.random.default_rng(0)
    n = 40
    adj = np.triu((rng.random((n, n)) < 0.12).astype(np.int64), 1)
    want = np.full(n, np.inf)
    want[0] = 0

    def relax(i, lv):                                                     # snps_graph.py:176-182 in miniature
        for c in np.flatnonzero(adj[i] > 0):
            if want[c] > lv + 1:
                want[c] = lv + 1
            relax(c, lv + 1)
    relax(0, 0)
    assert np.array_equal(go_builder.levels_from_root(adj, 0), want)
