#!/usr/bin/env python3
"""Phase stamps of k_attn_mfma_fwd (IGCN_HIPCC_EXTRA=-DAM_PROBE_ON) at the bench shape: workgroups 0, 128, ... 896,
waves 0 and 1: kernel-relative start, staging issued+stored, barrier passed, key loop done (ns)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa
from igcn_amd import _lib
from igcn_amd._lib import call, stream_ptr
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 128)()
b, lq, lk, h, d = 512, 90, 400, 2, 32
dev = "cuda"
q = torch.randn(b, lq, d, device=dev); kv = torch.randn(b, lk, 2 * d, device=dev)
o = torch.empty_like(q); lse = torch.empty(b, h, lq, device=dev)
for _ in range(200):
    call("igcn_attn_core_fwd", b, d, h, lq, lk, q.data_ptr(), kv.data_ptr(), o.data_ptr(), lse.data_ptr(), stream_ptr())
torch.cuda.synchronize()
raw.igcn_debug_attn_probe(buf)
t00 = min(buf[i * 8] for i in range(16))
for i in range(16):
    t = [buf[i * 8 + j] for j in range(4)]
    print(f"workgroup {i // 2 * 128:4d} wave {i % 2}: start {(t[0] - t00) * 10:6d}  staged +{(t[1] - t[0]) * 10:5d}  barrier +{(t[2] - t[1]) * 10:5d}  keys +{(t[3] - t[2]) * 10:6d} ns")
