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


class relu_margins:
    """Context manager for ORACLE runs: records, per ``torch.relu`` call (``sites[n]`` in call order), a boolean tensor
    marking the pre-activations within ``band`` of zero relative to the tensor's largest magnitude — the ReLU decisions
    an fp32 evaluation may take the other way.  Nothing is modified.  A gradient that is a SHORT sum over the batch (the
    bias of a BatchNorm over GO nodes: 32 terms) moves by a whole summand when one of them flips; tests mask exactly those
    elements (``near_nodes``) and hold everything else to the stated tolerance."""

    def __init__(self, band=3e-6):
        self.band, self.sites = band, []

    def __enter__(self):
        self._orig = torch.relu

        def recorded(t):
            mag = t.detach().abs()
            self.sites.append(mag <= self.band * mag.max())
            return self._orig(t)
        torch.relu = recorded
        torch.nn.functional.relu = recorded
        return self

    def __exit__(self, *exc):
        torch.relu = self._orig
        torch.nn.functional.relu = self._orig
        return False

    def count(self):
        return sum(int(m.sum()) for m in self.sites), sum(m.numel() for m in self.sites)

    def near_nodes(self, site_ids, dim):
        """Boolean vector over axis ``dim`` of the sites ``site_ids``: True where ANY near-zero pre-activation sits."""
        out = None
        for sid in site_ids:
            m = self.sites[sid]
            other = [d for d in range(m.dim()) if d != dim]
            v = m.any(dim=other) if other else m
            out = v if out is None else (out | v)
        return out


class relu_forced:
    """Context manager for ORACLE runs: impose the ReLU decisions another evaluation took.  ``forced[site]`` (site =
    index of the ``torch.relu`` call, in call order) is a bool tensor of the pre-activation's shape — True = that
    evaluation let the value through — or None (site not observed: the oracle's own decisions stand).  Inside ``band``
    (relative to the tensor's largest magnitude) the forced decision replaces the oracle's: value passed through or
    zeroed, derivative 1 or 0.  OUTSIDE the band the two evaluations must agree — ``mismatch_outside`` counts the
    entries where they do not (a real discrepancy, not rounding).  ``flips`` counts the overridden decisions."""

    def __init__(self, forced, band=2e-5, ignore=None):
        self.forced, self.band, self.ignore = forced, band, ignore or {}
        self.flips, self.mismatch_outside, self._site = 0, 0, 0

    def __enter__(self):
        self._orig = torch.relu

        def decided(t):
            site, self._site = self._site, self._site + 1
            want = self.forced.get(site)
            if want is None:
                return self._orig(t)
            want = want.to(torch.bool)
            assert tuple(want.shape) == tuple(t.shape), (site, tuple(want.shape), tuple(t.shape))
            mag = t.detach().abs()
            near = mag <= self.band * mag.max()
            own = t.detach() > 0
            differ = own != want
            skip = self.ignore.get(site)
            if skip is not None:                       # entries nothing downstream reads (pooled-away nodes)
                differ = differ & ~skip
            self.mismatch_outside += int((differ & ~near).sum())
            flip = differ & near
            self.flips += int(flip.sum())
            return torch.where(flip, torch.where(want, t, torch.zeros_like(t)), self._orig(t))
        torch.relu = decided
        torch.nn.functional.relu = decided
        return self

    def __exit__(self, *exc):
        torch.relu = self._orig
        torch.nn.functional.relu = self._orig
        return False
