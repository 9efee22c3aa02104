#!/usr/bin/env python3
"""Phase timestamps of k_go_attn_bwd_lds (library built with -DGO_ABL_PROBE): wall_clock64 (100 MHz) of thread 0 of
the first 8 workgroups at the phase boundaries.  Build: IGCN_HIPCC_EXTRA=-DGO_ABL_PROBE python ig-gcn_amd/build.py --force"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from igcn_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda", 0)
model, _ = bench.build_model(dev)
go = model.go_network
B = 256
raw = ctypes.CDLL(_lib.LIB_PATH)
names = ["copy-in", "stats+row", "col walks", "barrier", "mfma", "final"]
for j, csr in enumerate(go.enc_csr):
    w = (go.w_inc[j].weight, go.w_s_loop[j].weight, go.w_att_in[j].weight.view(-1), go.w_att_s[j].weight.view(-1))
    fin = w[0].shape[1]
    x = torch.randn(B, fin, csr.n_rows, device=dev, requires_grad=True)
    y = ops.GoAttention.apply(x, *w, csr)
    dy = torch.randn_like(y)
    for _ in range(3):
        torch.autograd.grad(y, (x,) + w, dy, retain_graph=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 128)()
    rc = raw.igcn_debug_go_probe(buf)
    print(f"layer {j}: N={csr.n_rows} {fin}->{y.shape[1]} rc={rc}")
    def show():
        for wg in range(8):
            t = [buf[wg * 16 + i] for i in range(7)]
            print(f"  wg {wg}: start {(t[0] - buf[0]) * 10:6d} ns  " +
                  "  ".join(f"{n} {(t[i + 1] - t[i]) * 10:6d}" for i, n in enumerate(names)) +
                  f"  total {(t[6] - t[0]) * 10} ns")
    show()
    # the same layer with the LayerNorm block's backward formed in the copy-in (ops.GoAttentionLN)
    ln = go.G_B[j]
    keep = (torch.rand(B, csr.n_rows, device=dev) > 0.1).float() / 0.9
    wl = (go.w_inc[j].weight, go.w_s_loop[j].weight, go.w_att_in[j].weight, go.w_att_s[j].weight)
    z = ops.GoAttentionLN.apply(x, *wl, csr, ln.weight, ln.bias, keep, go.pool[j], ln.eps)
    dz = torch.randn_like(z)
    for _ in range(3):
        torch.autograd.grad(z, (x,) + wl + (ln.weight, ln.bias), dz, retain_graph=True)
    torch.cuda.synchronize()
    rc = raw.igcn_debug_go_probe(buf)
    print(f"layer {j} with the LayerNorm backward in the copy-in (fused_ok "
          f"{raw.igcn_go_attn_ln_fused_ok(csr.n_rows, fin, y.shape[1], go.pool[j])}):")
    show()
