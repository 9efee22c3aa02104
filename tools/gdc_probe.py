#!/usr/bin/env python3
"""Phase times inside k_gdc_topk (workgroup 0, thread 0): build with IGCN_HIPCC_EXTRA=-DGDC_PROBE_ON.

    IGCN_HIPCC_EXTRA=-DGDC_PROBE_ON python ig-gcn_amd/build.py --force && python tools/gdc_probe.py [B R]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import _lib  # noqa: E402
from igcn_amd.gdc import diffusion_topk  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r = int(sys.argv[2]) if len(sys.argv) > 2 else 90
rng = np.random.default_rng(0)
s = rng.random((b, r, r)).astype(np.float32)
s = (s + s.transpose(0, 2, 1)) / 2
s[s < 0.9] = 0.0
s[:, np.arange(r - 1), np.arange(1, r)] = 1.0
s[:, np.arange(1, r), np.arange(r - 1)] = 1.0
s[:, np.arange(r), np.arange(r)] = 0.0
adj = torch.from_numpy(s).cuda()
for _ in range(5):
    diffusion_topk(adj, 3, check=False)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 16)()
print("rc", raw.igcn_debug_gdc_probe(buf))
names = ["load A", "row sums", "registers + column 0", "elimination", "write-back", "top-k + weights", "row counts",
         "prefix", "emission + padding"]
for i, nm in enumerate(names[:8]):
    print(f"{nm:24s} {(buf[i + 1] - buf[i]) * 10 / 1e3:8.2f} us")
print(f"{'total':24s} {(buf[8] - buf[0]) * 10 / 1e3:8.2f} us   ({r} pivots: {(buf[4] - buf[3]) * 10 / r:.0f} ns each)")
laps = ["pivot search", "stage the pivot row", "barrier A", "update (fma per row)", "publish column p+1", "barrier B"]
print("inside the elimination, wave 0 (ns per pivot):")
for i, nm in enumerate(laps):
    print(f"  {nm:24s} {buf[9 + i] * 10 / r:8.0f}")
