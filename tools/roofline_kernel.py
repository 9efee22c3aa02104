#!/usr/bin/env python3
"""Launch the GCN scatter-aggregate kernel (and a calibration copy) a fixed number of times, for rocprofv3:

  rocprofv3 --kernel-trace --stats --output-format csv -d out/trace -- python3 tools/roofline_kernel.py
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/fetch -- python3 tools/roofline_kernel.py
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/write -- python3 tools/roofline_kernel.py

Shapes: `bench` = the launch of the train step (512 x 90-ROI k=3 graphs, F=16); `stress` = 32 dense 512-ROI graphs.
The calibration kernel is a float4 device copy of a known byte count (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads
1/2 of the bytes of a wide coalesced stream; every other access width must be calibrated on a known pattern).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import ops, synth  # noqa: E402
from igcn_amd._lib import call, stream_ptr  # noqa: E402

LAUNCHES = 20
dev = torch.device("cuda", 0)


def run(plan, ew, n, f):
    coef = ops.GcnNorm.apply(ew, plan)
    h = torch.randn(n, f, device=dev)
    bias = torch.zeros(f, device=dev)
    out = torch.empty_like(h)
    torch.cuda.synchronize()
    for _ in range(LAUNCHES):
        call("igcn_gcn_propagate_fwd", n, plan.n_edges, f, 0, h.data_ptr(), f, coef[2].data_ptr(), coef[1].data_ptr(),
             bias.data_ptr(), plan.tgt_ptr.data_ptr(), out.data_ptr(), f, 1, stream_ptr())
    torch.cuda.synchronize()


# calibration: 256 MiB float4 copy (read 256 MiB, write 256 MiB), LAUNCHES times
src = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(LAUNCHES):
    dst.copy_(src)
torch.cuda.synchronize()

# bench shape
batch = synth.brain_batch(256, seed=1000, rois=90, tsne_dim=90).to(dev)
plan = ops.plan_for(batch).replicate(2)
run(plan, torch.cat([batch.edge_attr, batch.edge_attr]), 2 * batch.x.shape[0], 16)

# stress shape
rois, g = 512, 32
r = torch.arange(rois).repeat_interleave(rois)
c = torch.arange(rois).repeat(rois)
ei = torch.cat([torch.stack([r, c]) + k * rois for k in range(g)], dim=1).to(dev)
w = (torch.rand(ei.shape[1], device=dev) / rois)
run(ops.GraphPlan(ei, g * rois), w, g * rois, 16)
print("done")
