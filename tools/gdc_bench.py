#!/usr/bin/env python3
"""Device time of the GDC pre-transform + collation (igcn_gdc_topk) for one batch of dense adjacencies."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
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
for _ in range(3):
    diffusion_topk(adj, 3)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    diffusion_topk(adj, 3)
e1.record()
torch.cuda.synchronize()
print(f"GDC + collation of {b} graphs x {r} ROIs: {e0.elapsed_time(e1) * 100:.1f} us per batch "
      f"(includes one host read of the edge count)")
e0.record()
for _ in range(10):
    diffusion_topk(adj, 3, check=False)
e1.record()
torch.cuda.synchronize()
print(f"  without the host read (check=False, what a per-step producer runs): {e0.elapsed_time(e1) * 100:.1f} us per batch")
