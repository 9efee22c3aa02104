#!/usr/bin/env python3
"""Phase stamps of k_gemm_f32 (IGCN_HIPCC_EXTRA=-DG_PROBE_ON): workgroup (0, 0, z < 8): start, first tile staged, K loop
done — for a few shapes of the train step."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa
from igcn_amd import _lib, ops
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 64)()
dev = "cuda"
for what, m, n, k, form in (("Gram s s^T", 256, 256, 2880, "nt"), ("lin1", 512, 64, 2912, "nt"), ("Gram backward S s", 256, 2880, 256, "nn"),
                            ("kv projection", 204800, 64, 32, "nt"), ("lin1 dW", 64, 2912, 512, "tn")):
    if form == "nt":
        a, b = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev); fn = lambda: ops.gemm_nt(a, b)
    elif form == "nn":
        a, b = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev); fn = lambda: ops.gemm_nn(a, b)
    else:
        a, b = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev); fn = lambda: ops.gemm_tn(a, b)
    for _ in range(int(os.environ.get('WARM', '2000'))):
        fn()
    torch.cuda.synchronize()
    raw.igcn_debug_gemm_probe(buf)
    sk = ops._split_k(m, n, k)
    t = [buf[i] for i in range(3)]
    steps = -(-k // (32 * sk))
    print(f"{what:22s} {m}x{n}x{k} split {sk}: first tile {(t[1] - t[0]) * 10:5d} ns, K loop {(t[2] - t[1]) * 10:6d} ns "
          f"({steps} steps, {(t[2] - t[1]) * 10 / max(steps, 1):.0f} ns per step; "
          f"{(buf[6] - buf[5]) / max(steps, 1):.0f} s_memtime ticks per step, {(buf[6] - buf[4]) / max(t[2] - t[0], 1) * 100:.0f} MHz)")
    print("      shader cycles over the main loop (wave 0): multiply %d, LDS store %d, issue loads %d, barrier %d" % tuple(buf[16:20]))
