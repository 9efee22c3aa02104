#!/usr/bin/env python3
"""The train step's deferred final reductions (igcn_reduce_flush, one k_multi_reduce launch) by themselves: the queue of
the default workload rebuilt from its shapes (IGCN_DEBUG_REDUCE=1 lists them), flushed and event-timed — whole, and with
one entry left out at a time, to see which entries set the launch's duration (median of per-replay event times, graph
launch included).  usage: reduce_bench.py [iters=20] [cold]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import _lib  # noqa: E402

# (rows, n) of the default workload's queue, in queue order of one backward pass (bench.py --workload full)
SHAPES = [(8, 64), (32, 195), (512, 32), (512, 2400), (800, 160), (512, 64), (512, 50), (512, 336), (512, 2048),
          (512, 20), (512, 1024), (448, 5), (512, 6016), (3008, 2), (240, 1024), (2, 800), (2, 6000), (16, 800),
          (16, 6000), (16, 12800), (16, 1024), (128, 8067), (128, 16134), (4, 203648), (4, 186368), (4, 64), (4, 64),
          (512, 32), (512, 64), (8, 64), (32, 195), (512, 32), (512, 2400), (512, 6016)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
COLD = len(sys.argv) > 2 and sys.argv[2] == "cold"
junk = torch.zeros(256 << 20, device="cuda") if COLD else None
if os.environ.get("REDUCE_BENCH_LIB"):              # A/B of two builds inside ONE gpurun call (boxes differ by 40 %)
    _lib.LIB_PATH = os.environ["REDUCE_BENCH_LIB"]
lib = _lib.load()
fn = lib.igcn_debug_reduce_rows_final
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
bufs = [(torch.randn(r, n, device="cuda"), torch.empty(n, device="cuda")) for r, n in SHAPES]
st = torch.cuda.current_stream().cuda_stream


def flush(skip=None):
    lib.igcn_reduce_defer(st, 1)
    for i, ((p, o), (r, n)) in enumerate(zip(bufs, SHAPES)):
        if i != skip and (skip is None or not isinstance(skip, set) or i not in skip):
            assert fn(p.data_ptr(), r, n, n, o.data_ptr(), st) == 0
    assert lib.igcn_reduce_flush(st) == 0
    lib.igcn_reduce_defer(st, 0)


def timed(skip=None):
    """the flush captured into a graph (its host side — 27 queue calls — would otherwise set the pace)"""
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        global st
        keep, st = st, s.cuda_stream
        flush(skip)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            flush(skip)
        st = keep
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if COLD:
        junk.add_(1.0)
    e0.record()
    for _ in range(iters):                      # back to back: launch latency overlaps the previous replay
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


flush()
torch.cuda.synchronize()
for (p, o), (r, n) in zip(bufs, SHAPES):
    assert torch.allclose(o, p.sum(0), rtol=1e-4, atol=1e-3), (r, n)
whole = timed()
print(f"whole queue ({len(SHAPES)} entries, {sum(r * n for r, n in SHAPES) * 4 / 1e6:.1f} MB): {whole:7.1f} us", flush=True)
for keep in (1, 5, 10, 15, 20, 25, 30):
    t = timed(set(range(keep, len(SHAPES))))
    print(f"  first {keep:2d} entries only: {t:7.1f} us", flush=True)
def timed_immediate(i):
    """entry i through its stand-alone kernel (igcn_reduce_defer off), same graph-replay timing"""
    (p, o), (r, n) = bufs[i], SHAPES[i]
    g = torch.cuda.CUDAGraph()
    s_ = torch.cuda.Stream()
    with torch.cuda.stream(s_):
        assert fn(p.data_ptr(), r, n, n, o.data_ptr(), s_.cuda_stream) == 0
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s_):
            assert fn(p.data_ptr(), r, n, n, o.data_ptr(), s_.cuda_stream) == 0
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


seen = set()
for i, (r, n) in enumerate(SHAPES):                  # every distinct entry ALONE: its own latency chain above the floor
    if (r, n) in seen:
        continue
    seen.add((r, n))
    t = timed(set(range(len(SHAPES))) - {i})
    print(f"  alone {r:5d} x {n:6d} ({r * n * 4 / 1e6:5.1f} MB): {t:7.1f} us", flush=True)
for i, (r, n) in enumerate(SHAPES[:0]):
    t = timed(i)
    print(f"  without {r:5d} x {n:6d}: {t:7.1f} us  ({whole - t:+6.1f})", flush=True)
