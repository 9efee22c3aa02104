#!/usr/bin/env python3
"""Attention core (igcn_attn_core_*) alone at the bench shape: event-timed forward / backward, for A/B runs and
rocprofv3 --pmc passes.  usage: attn_bench.py [B=512] [Lq=90] [Lk=400] [iters=20] [core=fp32|bf16]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd._lib import call, stream_ptr  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lq = int(sys.argv[2]) if len(sys.argv) > 2 else 90
lk = int(sys.argv[3]) if len(sys.argv) > 3 else 400
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
core = sys.argv[5] if len(sys.argv) > 5 else "fp32"
FWD, BWD = {"bf16": ("igcn_attn_core_bf16_fwd", "igcn_attn_core_bf16_bwd"),
            "split": ("igcn_attn_core_split_fwd", "igcn_attn_core_split_bwd")}.get(core, ("igcn_attn_core_fwd", "igcn_attn_core_bwd"))
ONLY = sys.argv[6] if len(sys.argv) > 6 else "fwd,bwd"
h, d = 2, 32
dev = "cuda"
q = torch.randn(b, lq, d, device=dev)
kv = torch.randn(b, lk, 2 * d, device=dev)
o = torch.empty_like(q)
lse = torch.empty(b, h, lq, device=dev)
do = torch.randn_like(q)
dq, dkv = torch.empty_like(q), torch.empty_like(kv)
scr = torch.empty(b * h * lq + 16, device=dev)


def fwd():
    call(FWD, b, d, h, lq, lk, q.data_ptr(), kv.data_ptr(), o.data_ptr(), lse.data_ptr(), stream_ptr())


def bwd():
    call(BWD, b, d, h, lq, lk, q.data_ptr(), kv.data_ptr(), o.data_ptr(), lse.data_ptr(),
         do.data_ptr(), dq.data_ptr(), dkv.data_ptr(), scr.data_ptr(), stream_ptr())


for name, fn in (("fwd", fwd), ("bwd", bwd)):
    if name not in ONLY.split(","):
        continue
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    flops = b * h * lq * lk * 16 * 2 * (2 if name == "fwd" else 5)
    print(f"{core} {name}: {us:8.1f} us   {flops / us / 1e6:6.1f} TFLOP/s (useful)", flush=True)
