"""CPU checks of the drop-in boundary: libigcn.so loads, exports every symbol include/igcn.h declares, and the
ctypes signature table (igcn_amd/_lib.py) agrees with the header prototype by prototype.  No compute calls."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from igcn_amd import _lib

HEADER = os.path.join(ROOT, "include", "igcn.h")

CTYPE = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t, "float": ctypes.c_float,
         "double": ctypes.c_double, "unsigned": ctypes.c_uint}


def _prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|const char\s*\*)\s+(igcn_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append(ctypes.c_void_p)
                else:
                    types.append(CTYPE[a.replace("const ", "").split()[0]])
        protos[name] = (ret.replace(" ", ""), types)
    return protos


def test_header_declares_what_python_binds():
    protos = _prototypes()
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, (ret, types) in protos.items():
        res, args = _lib.SIGNATURES[name]
        assert list(args) == types, f"{name}: header {types} vs ctypes {args}"
        want = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "constchar*": ctypes.c_char_p}[ret]
        assert res is want, name


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in _prototypes():
        assert hasattr(lib, name), name
    declared = int(re.search(r"#define\s+IGCN_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert lib.igcn_version() == declared == _lib.ABI_VERSION
    # host-only helpers may be called without a GPU
    assert lib.igcn_gcn_propagate_bwd_scratch_floats(23040, 16) >= 1440 * 16
    assert lib.igcn_go_attn_bwd_scratch_floats(256, 3000, 5, 5) > 4 * 256 * 3000


def test_go_attn_walk_order_is_a_balanced_permutation():
    """igcn_go_attn_walk_order (host code): every node exactly once, idle slots -1, the lanes of a wave-pass get
    nodes of (nearly) equal column degree, and no wave is left with much more list work than the average."""
    import numpy as np
    import torch
    import igcn_amd  # noqa: F401
    from igcn_amd import synth
    lib = _lib.load()
    _, adj, _ = synth.go_hierarchy()
    deg = (np.asarray(adj) != 0).sum(1).astype(np.int64)      # readers of each node = its children (adj[parent, child])
    n = deg.size
    t_ptr = torch.zeros(n + 1, dtype=torch.int32)
    t_ptr[1:] = torch.from_numpy(np.cumsum(deg)).to(torch.int32)
    assert lib.igcn_go_attn_bwd_threads(n, 2, 5) == 1024 and lib.igcn_go_attn_bwd_threads(1200, 5, 5) == 512
    slots = lib.igcn_go_attn_walk_slots(n, 2, 5)
    assert slots == 3072 and lib.igcn_go_attn_walk_slots(1200, 5, 5) == 1536
    order = torch.full((slots,), -7, dtype=torch.int32)
    assert lib.igcn_go_attn_walk_order(n, 2, 5, t_ptr.data_ptr(), order.data_ptr()) == 0
    o = order.numpy()
    assert sorted(o[o >= 0].tolist()) == list(range(n)) and set(o[o < 0].tolist()) <= {-1}
    steps = np.zeros(16)
    spread = []
    for it in range(3):
        for w in range(16):
            grp = o[it * 1024 + w * 64: it * 1024 + w * 64 + 64]
            d = deg[grp[grp >= 0]]
            d = d[d <= 12]                                    # hubs are walked by the whole wave
            if d.size:
                steps[w] += (d.max() + 1) // 2
                spread.append(d.max() - d.min())
    spread.sort()                                             # sorted by degree: no lane waits on a long neighbour
    assert spread[-2] <= 2 and spread[-1] <= 6                # (the group holding the tail of the distribution aside)
    # in node order the slowest wave walked 10 steps against a mean of 4; now one wave holds the indivisible group with
    # the longest ordinary lists (6 steps) and everybody else is within 2 of the mean
    assert steps.max() <= 6 and np.sort(steps)[-2] <= steps.mean() + 2, steps


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.IgcnError):
        _lib.load()


def test_cpu_tensors_are_rejected():
    import torch
    with pytest.raises(_lib.IgcnError):
        _lib.ptr(torch.zeros(3))
