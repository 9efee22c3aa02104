#!/usr/bin/env python3
"""k_multi_reduce durations by themselves, from a kernel trace: run under
  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/reduce_trace.py
Each listed entry is flushed ALONE 20 times (eager launches); `--summarise <dir>` prints the median duration per group."""
import csv
import ctypes
import glob
import os
import sys

SHAPES = [(4, 203648), (512, 6016), (512, 2400), (128, 16134), (16, 12800), (512, 1024), (800, 160), (8, 64)]
if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_multi_reduce" in r["Kernel_Name"] or "k_reduce_rows" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    assert len(d) == 40 * len(SHAPES), len(d)
    for i, (r, n) in enumerate(SHAPES):
        g = sorted(d[40 * i + 5:40 * i + 20])
        h = sorted(d[40 * i + 25:40 * i + 40])
        print(f"{r:5d} x {n:6d} ({r * n * 4 / 1e6:5.1f} MB): in k_multi_reduce median {g[len(g) // 2]:6.2f} us  min {g[0]:6.2f}   "
              f"stand-alone kernel ({rows[40 * i + 30]['Kernel_Name'][:28]}) median {h[len(h) // 2]:6.2f}  min {h[0]:6.2f}")
    sys.exit(0)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import _lib  # noqa: E402

lib = _lib.load()
fn = lib.igcn_debug_reduce_rows_final
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
for r, n in SHAPES:
    p, o = torch.randn(r, n, device="cuda"), torch.empty(n, device="cuda")
    torch.cuda.synchronize()
    for _ in range(20):
        lib.igcn_reduce_defer(1)
        assert fn(p.data_ptr(), r, n, n, o.data_ptr(), st) == 0
        assert lib.igcn_reduce_flush(st) == 0
        lib.igcn_reduce_defer(0)
        torch.cuda.synchronize()
    for _ in range(20):                                  # the same sum through its stand-alone kernel
        assert fn(p.data_ptr(), r, n, n, o.data_ptr(), st) == 0
        torch.cuda.synchronize()
