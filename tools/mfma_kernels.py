#!/usr/bin/env python3
"""The matrix-core kernels of the two benchmarked steps, launched a fixed number of times each, for the rocprofv3 passes
of tools/mfma_run.sh (kernel trace; then ONE --pmc pass of SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES
GRBM_GUI_ACTIVE; the program goes directly after `--`):

  k_ds_agg / k_ds_aggT / k_ds_mask_bwd   configs[4]: 32 complete 512-ROI graphs, both passes per launch
  k_attn_split_fwd / k_attn_split_bwd        configs[2]: 512 samples x 2 heads, 90 queries x 400 keys, head_dim 16 (the
                                             default: split bf16 operands); with IGCN_ATTN_EXACT_FP32=1 in the environment
                                             the same calls run k_attn_mfma_fwd / k_attn_mfma_bwd_shared (exact fp32)
  k_attn_bf16_*                              configs[4]: 64 samples x 2 heads, 512 x 1300, head_dim 16 (bf16 operands)
  k_gemm_f32                                 4096^3 (the calibration point: 65-69 % of the fp32 matrix peak by time) and
                                             lin1 of the default step (512 x 64 x 2912, split-K)
tools/mfma_report.py turns the two passes into profiles/rNN_pmc/mfma.json."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_amd  # noqa: E402,F401
from igcn_amd import ops, synth  # noqa: E402
from igcn_amd._lib import call, stream_ptr  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402

LAUNCHES = 10
dev = torch.device("cuda", 0)

# --- calibration + a step product ----------------------------------------------------------------------------------
a, b = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)
for _ in range(LAUNCHES):
    ops.gemm_nt(a, b)
torch.cuda.synchronize()
a1, b1 = torch.randn(512, 2912, device=dev), torch.randn(64, 2912, device=dev)
for _ in range(LAUNCHES):
    ops.gemm_nt(a1, b1)
torch.cuda.synchronize()


# --- attention cores -------------------------------------------------------------------------------------------------
def attention(bsz, lq, lk, core):
    fwd, bwd = ("igcn_attn_core_bf16_fwd", "igcn_attn_core_bf16_bwd") if core == "bf16" else \
        ("igcn_attn_core_fwd", "igcn_attn_core_bwd")
    h, d = 2, 32
    q = torch.randn(bsz, lq, d, device=dev)
    kv = torch.randn(bsz, lk, 2 * d, device=dev)
    o = torch.empty_like(q)
    lse = torch.empty(bsz, h, lq, device=dev)
    do = torch.randn_like(q)
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    scr = torch.empty(bsz * h * lq + 16, device=dev)
    for _ in range(LAUNCHES):
        call(fwd, bsz, d, h, lq, lk, q.data_ptr(), kv.data_ptr(), o.data_ptr(), lse.data_ptr(), stream_ptr())
    for _ in range(LAUNCHES):
        call(bwd, bsz, d, h, lq, lk, q.data_ptr(), kv.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(),
             dq.data_ptr(), dkv.data_ptr(), scr.data_ptr(), stream_ptr())
    torch.cuda.synchronize()


attention(512, 90, 400, "fp32")
attention(64, 512, 1300, "bf16")

# --- dense-block aggregation (configs[4]) ------------------------------------------------------------------------------
sb = Batch.from_data_list(synth.brain_graph_list(32, seed=1, rois=512, tsne_dim=8, dense=True)).to(dev)
assert ops.plan_for(sb).dense_blocks
prob = torch.randn(512, 3, device=dev)
pb = torch.randn(6, 1, device=dev)
spr = torch.randn(1, 54, device=dev)
dw = [torch.randn(16, 3, device=dev) * 0.5, torch.zeros(16, device=dev), torch.randn(16, 16, device=dev) * 0.3,
      torch.zeros(16, device=dev)]
leaves = [t.requires_grad_(True) for t in (prob, pb, spr)]
cot = torch.randn(2 * sb.x.shape[0], 32, device=dev)
for _ in range(LAUNCHES // 2):
    xcat, regp = ops.DenseSgcn.apply(sb.x, sb.edge_attr, prob, pb, spr, "both", 512, (0.1, 0.1, 0.1, 0.1, 1e-6), None, *dw)
    torch.autograd.grad((xcat * cot).sum() + regp.sum(), leaves)
torch.cuda.synchronize()
print("done")
