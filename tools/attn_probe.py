#!/usr/bin/env python3
"""Phase stamps of k_attn_mfma_fwd at the bench shape (library built with IGCN_HIPCC_EXTRA=-DAM_PROBE_ON): waves 0-1 of
workgroups 0, 128, ..., 896 — 0 start, 1 K | V staged, 2 barrier passed, 3 tile loop done — in ns after the first start."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import _lib  # noqa: E402
from igcn_amd._lib import call, stream_ptr  # noqa: E402

b, lq, lk, h, d = 512, 90, 400, 2, 32
q = torch.randn(b, lq, d, device="cuda")
kv = torch.randn(b, lk, 2 * d, device="cuda")
o = torch.empty_like(q)
lse = torch.empty(b, h, lq, device="cuda")
for _ in range(3):
    call("igcn_attn_core_fwd", b, d, h, lq, lk, q.data_ptr(), kv.data_ptr(), o.data_ptr(), lse.data_ptr(), stream_ptr())
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 128)()
print("rc", raw.igcn_debug_attn_probe(buf))
t0 = min(buf[i * 8] for i in range(16))
for wg in range(8):
    for w in range(2):
        t = [buf[(wg * 2 + w) * 8 + i] for i in range(4)]
        print(f"wg {wg * 128:4d} wave {w}: start +{(t[0] - t0) * 10:6d} ns  staged +{(t[1] - t[0]) * 10:5d}  barrier +{(t[2] - t[0]) * 10:5d}"
              f"  loop done +{(t[3] - t[0]) * 10:5d}")
