#!/usr/bin/env python3
"""Launch the scatter-aggregate kernels (and a calibration copy) a fixed number of times, for rocprofv3:

  rocprofv3 --kernel-trace --output-format csv -d out/trace -- python3 tools/roofline_kernel.py
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/fetch -- python3 tools/roofline_kernel.py
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/write -- python3 tools/roofline_kernel.py
  python3 tools/pmc_report.py out > profiles/rNN_pmc/traffic.json

Kernels: `k_sgcn_stack_fwd` = the LDS-resident SGCN stack at the bench shape (512 x 90-ROI k=3 graphs, F=16, L=2: the
kernel the default train step launches); `k_gcn_propagate_fwd_q` = the stand-alone scatter-aggregate at the same shape;
`k_gcn_propagate_fwd_lds` = the LDS-staged dense scatter-aggregate at the stress shape (64 dense 512-ROI graphs);
`k_ds_agg` = the dense-block aggregation of the same 32 graphs (both passes per launch: what configs[4] runs).
The calibration kernel is a float4 device copy of a known byte count (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads
1/2 of the bytes of a wide coalesced stream; every other access width must be calibrated on a known pattern).
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import ops, synth  # noqa: E402
from igcn_amd._lib import call, ptr, stream_ptr  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402

LAUNCHES = 20
dev = torch.device("cuda", 0)


def run_propagate(plan, ew, n, f, npg):
    coef = ops.GcnNorm.apply(ew, plan)
    h = torch.randn(n, f, device=dev)
    bias = torch.zeros(f, device=dev)
    out = torch.empty_like(h)
    torch.cuda.synchronize()
    for _ in range(LAUNCHES):
        call("igcn_gcn_propagate_fwd", n, plan.n_edges, f, npg, h.data_ptr(), f, coef[2].data_ptr(), coef[1].data_ptr(),
             bias.data_ptr(), plan.tgt_ptr.data_ptr(), out.data_ptr(), f, 1, stream_ptr())
    torch.cuda.synchronize()


# calibration: 256 MiB float4 copy (read 256 MiB, write 256 MiB), LAUNCHES times
src = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(LAUNCHES):
    dst.copy_(src)
torch.cuda.synchronize()

# bench shape: stand-alone kernel, then the LDS-resident stack
batch = synth.brain_batch(256, seed=1000, rois=90, tsne_dim=90).to(dev)
plan = ops.plan_for(batch).replicate(2)
ew2 = torch.cat([batch.edge_attr, batch.edge_attr])
n2 = 2 * batch.x.shape[0]
run_propagate(plan, ew2, n2, 16, 0)
h0, f, layers, rois = 3, 16, 2, 90
x2 = torch.cat([batch.x, batch.x])
ws = [torch.randn(f, h0 if l == 0 else f, device=dev) * 0.3 for l in range(layers)]
bs = [torch.randn(f, device=dev) * 0.1 for _ in range(layers)]
wp = (ctypes.c_void_p * layers)(*[w.data_ptr() for w in ws])
bp = (ctypes.c_void_p * layers)(*[b.data_ptr() for b in bs])
xcat = torch.empty(n2, layers * f, device=dev)
torch.cuda.synchronize()
for _ in range(LAUNCHES):
    call("igcn_sgcn_stack_fwd", n2 // rois, rois, plan._stack_dims[1], h0, f, layers, ptr(x2), ptr(ew2), ptr(plan.src32),
         ptr(plan.dst32), ptr(plan.tgt_ptr), ptr(plan.tgt_perm), ptr(plan.loop_edge), wp, bp, ptr(xcat), None, stream_ptr())
torch.cuda.synchronize()

# ... and the FRONT kernel of the default step's image branch (round 5: plan build + masks + stack of both passes in one
# launch; no dropout rider here, so the traffic is the kernel's own)
prob_f, pb_f, snps_f = torch.randn(rois, h0, device=dev), torch.randn(2 * h0, 1, device=dev), torch.randn(1, 54, device=dev)
wb_f = [t for pair in zip(ws, bs) for t in pair]
fplan = ops.plan_for(batch)
with torch.no_grad():
    for _ in range(LAUNCHES):
        ops.SgcnFront.apply(batch.x, prob_f, pb_f, batch.edge_attr, fplan, rois, snps_f, (0.1, 0.1, 0.1, 0.1, 1e-6),
                            batch.snps_feat, batch.edge_index, *wb_f)
torch.cuda.synchronize()

# stress shape: 64 dense 512-ROI graphs (both passes of a configs[4] step)
sb = Batch.from_data_list(synth.brain_graph_list(32, seed=1, rois=512, tsne_dim=8, dense=True)).to(dev)
splan = ops.plan_for(sb).replicate(2)
run_propagate(splan, torch.cat([sb.edge_attr, sb.edge_attr]), 2 * sb.x.shape[0], 16, 512)

# the same batch as COMPLETE graphs on the dense-block path (what the configs[4] step launches now): k_ds_agg — one
# launch aggregates BOTH passes of a layer, reading every 4-byte weight once
assert ops.plan_for(sb).dense_blocks
prob = torch.randn(512, 3, device=dev)
pb = torch.randn(6, 1, device=dev)
spr = torch.randn(1, 54, device=dev)
dw = [torch.randn(16, 3, device=dev) * 0.5, torch.zeros(16, device=dev), torch.randn(16, 16, device=dev) * 0.3,
      torch.zeros(16, device=dev)]
torch.cuda.synchronize()
for _ in range(LAUNCHES // 2):                      # two k_ds_agg launches (two layers) per forward
    ops.DenseSgcn.apply(sb.x, sb.edge_attr, prob, pb, spr, "both", 512, (0.1, 0.1, 0.1, 0.1, 1e-6), None, *dw)
torch.cuda.synchronize()
# ... and its backward: the transposed aggregation (two launches) and the mask-gradient edge pass
leaves = [t.requires_grad_(True) for t in (prob, pb, spr)]
cot = torch.randn(2 * sb.x.shape[0], 32, device=dev)
for _ in range(max(2, LAUNCHES // 4)):
    xcat, regp = ops.DenseSgcn.apply(sb.x, sb.edge_attr, prob, pb, spr, "both", 512, (0.1, 0.1, 0.1, 0.1, 1e-6), None, *dw)
    torch.autograd.grad((xcat * cot).sum() + regp.sum(), leaves)
torch.cuda.synchronize()
print("done")
