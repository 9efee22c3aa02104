#!/usr/bin/env python3
"""Phase stamps of k_spmm_long_lds (IGCN_HIPCC_EXTRA=-DSPMM_PROBE_ON) at the bench shapes: decode forward (sum_c = 0) and
encode backward dx (sum_c = 1): thread 0 of the first 8 workgroups at start / vector staged / first list done / end."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from igcn_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
model, _ = bench.build_model(dev)
go = model.go_network
B = 512
x = torch.rand(B, 54, device=dev, requires_grad=True)
val = torch.rand(2, go.gene_csr.nnz, device=dev, requires_grad=True)
xd = torch.rand(B, go.gene_t_csr.n_cols, device=dev, requires_grad=True)
vd = torch.rand(1, go.gene_t_csr.nnz, device=dev, requires_grad=True)
for _ in range(3):
    y = ops.SparseMap.apply(x, go.gene_csr, val)
    torch.autograd.grad(y.sum(), (x, val))
    yd = ops.SparseMap.apply(xd, go.gene_t_csr, vd)
    torch.autograd.grad(yd.sum(), (xd, vd))
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 128)()
print("rc", raw.igcn_debug_spmm_probe(buf))
for mode, name in ((0, "decode forward (C=1)"), (1, "encode backward dx (C=2, transposed lists)")):
    print(name)
    for wg in range(8):
        t = [buf[(mode * 8 + wg) * 8 + i] for i in range(4)]
        print(f"  wg {wg}: staging {(t[1] - t[0]) * 10:6d} ns  first list {(t[2] - t[1]) * 10:6d}  "
              f"other lists {(t[3] - t[2]) * 10:6d}  total {(t[3] - t[0]) * 10}")
