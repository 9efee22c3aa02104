#!/usr/bin/env python3
"""Error of the attention cores against an fp64 evaluation of softmax(q k^T / sqrt(d)) v and its gradients, at the
bench shape (or: attn_error.py B Lq Lk): exact-fp32 MFMA (igcn_attn_core_*), split-bf16 (igcn_attn_core_split_*), bf16
operands (igcn_attn_core_bf16_*).  Scale-relative max error per tensor — the number the 1e-4 / 1e-3 test bounds are about."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import _lib  # noqa: E402
from igcn_amd._lib import call, stream_ptr  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lq = int(sys.argv[2]) if len(sys.argv) > 2 else 90
lk = int(sys.argv[3]) if len(sys.argv) > 3 else 400
h, d = 2, 32
hd = d // h
dev = "cuda"
torch.manual_seed(0)
amp = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0          # operand scale: larger = sharper softmax
q = torch.randn(b, lq, d, device=dev) * amp
kv = torch.randn(b, lk, 2 * d, device=dev) * amp
do = torch.randn(b, lq, d, device=dev)
qd, kvd, dod = (t.double().requires_grad_(t is not do) for t in (q, kv, do))
qh = qd.view(b, lq, h, hd).transpose(1, 2)
kh = kvd.view(b, lk, 2, h, hd)[:, :, 0].transpose(1, 2)
vh = kvd.view(b, lk, 2, h, hd)[:, :, 1].transpose(1, 2)
att = torch.softmax(qh @ kh.transpose(-1, -2) / hd ** 0.5, dim=-1)
ref = (att @ vh).transpose(1, 2).reshape(b, lq, d)
(ref * dod).sum().backward()
lib = _lib.load()
for core, fwd, bwd in (("fp32", "igcn_attn_core_fwd", "igcn_attn_core_bwd"),
                       ("split", "igcn_attn_core_split_fwd", "igcn_attn_core_split_bwd"),
                       ("bf16", "igcn_attn_core_bf16_fwd", "igcn_attn_core_bf16_bwd")):
    o = torch.empty_like(q)
    lse = torch.empty(b, h, lq, device=dev)
    call(fwd, b, d, h, lq, lk, q.data_ptr(), kv.data_ptr(), o.data_ptr(), lse.data_ptr(), stream_ptr())
    torch.cuda.synchronize()
    rel = lambda got, want: float((got.double() - want).abs().max() / want.abs().max())       # noqa: E731
    line = f"{core:6s} o {rel(o, ref.detach()):.2e}"
    if hasattr(lib, bwd):
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        scr = torch.empty(b * h * lq + 16, device=dev)
        call(bwd, b, d, h, lq, lk, q.data_ptr(), kv.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(), dq.data_ptr(),
             dkv.data_ptr(), scr.data_ptr(), stream_ptr())
        torch.cuda.synchronize()
        dk_ref, dv_ref = kvd.grad.view(b, lk, 2, d)[:, :, 0], kvd.grad.view(b, lk, 2, d)[:, :, 1]
        dk, dv = dkv.view(b, lk, 2, d)[:, :, 0], dkv.view(b, lk, 2, d)[:, :, 1]
        line += f"  dq {rel(dq, qd.grad):.2e}  dk {rel(dk, dk_ref):.2e}  dv {rel(dv, dv_ref):.2e}"
    print(line, flush=True)
