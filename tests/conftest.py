import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)

import igcn_amd  # noqa: E402,F401  (import shim: package dir is ``ig-gcn_amd/``)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    return torch.cuda.is_available()


def pytest_sessionstart(session):
    """On a GPU box make sure libigcn.so matches the sources (a digest check; compiles only when the library is
    missing or stale).  Test infrastructure only: the package itself never builds or falls back — it raises."""
    if has_gpu():
        import __graft_entry__
        __graft_entry__.build()


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_group(store, prefix):
    """{'name': array or ('summary', array)} for keys under ``prefix/``."""
    out = {}
    for k, v in store.items():
        if k.startswith(prefix + "/"):
            name = k[len(prefix) + 1:]
            if name.endswith("#summary"):
                out[name[:-8]] = ("summary", v)
            else:
                out[name] = v
    return out


def assert_matches(got, want, tol, what="", floor=0.0):
    """Scale-relative check: max|got-want| <= tol * max(max|want|, floor).

    got: tensor; want: ndarray or ('summary', ndarray) for tensors stored as signatures.
    """
    from _weights import summarise
    g = got.detach().cpu()
    if isinstance(want, tuple):
        s = summarise(g)
        w = want[1]
        scale = max(abs(w[1]), 1e-30)          # abs-sum sets the scale of all three signatures
        err = np.abs(s - w) / scale
        assert err.max() <= tol, f"{what}: summary mismatch {s} vs {w} (rel {err})"
        return
    w = torch.from_numpy(np.asarray(want))
    assert tuple(g.shape) == tuple(w.shape), f"{what}: shape {tuple(g.shape)} vs {tuple(w.shape)}"
    scale = max(float(w.abs().max()) if w.numel() else 0.0, floor, 1e-30)
    err = float((g.double() - w.double()).abs().max()) if w.numel() else 0.0
    assert np.isfinite(err) and err <= tol * scale, \
        f"{what}: max abs err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e} > tol {tol})"


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
