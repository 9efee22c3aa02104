"""Drop-in ``SGCN_GCN`` — the image-only sibling of the hot path (reference kernel/sgcn.py:272-388) on the
same kernels: masks (igcn_edge_mask_*), one gcn_norm per pass, MFMA feature transforms, scatter-aggregate,
the (R*D -> hidden_linear -> classes) head on the split-K GEMM, and the mask regulariser as one reduction.

Same constructor signature (``dataset`` is accepted and ignored exactly as in the reference), the same
``forward(data, isExplain=False) -> log_softmax [B, C]``, ``cal_probability`` (:321), ``loss_probability``
(:334; note the node-mask L1 term is divided by ``rois`` here, not by ``rois*H_0`` as in
kernel/sgcn_img_snp.py:160) and the same ``state_dict()`` keys.  ``lin1`` takes ``rois * num_layers * hidden``
inputs (the reference hard-codes 90 at :285, i.e. it only works for rois=90; identical there).

``forward_pair`` runs the plain and the masked pass of train() (kernel/train_eval_sgcn.py:303-306) as one
sweep over a 2-copy block-diagonal batch — there is no BatchNorm in this model, so the two passes do not
interact at all.
"""
import math

import os

import torch
import torch.nn.functional as F
from torch.nn import Linear, Parameter, init

from . import ops
from .sgcn_img_snp import GCNConv, sgcn_stack


class SGCN_GCN(torch.nn.Module):
    def __init__(self, dataset, num_layers, hidden, *args, hidden_linear=64, rois=90, H_0=3, num_features=3,
                 num_classes=2, **kwargs):
        super().__init__()
        self.input = None
        self.rois, self.prob_dim = rois, H_0
        self.conv1 = GCNConv(num_features, hidden)
        self.convs = torch.nn.ModuleList(GCNConv(hidden, hidden) for _ in range(num_layers - 1))
        self.lin1 = Linear(rois * num_layers * hidden, hidden_linear)
        self.lin2 = Linear(hidden_linear, num_classes)
        self.prob = Parameter(torch.empty(rois, H_0))
        self.prob_bias = Parameter(torch.empty(H_0 * 2, 1))
        self.edge_prob = Parameter(torch.empty(rois, rois))               # unused by forward (as in the reference)
        self._init_masks()
        self._dropout_enabled = True
        self.batched_passes = True

    def _init_masks(self):
        with torch.no_grad():
            for p in (self.prob_bias, self.prob, self.edge_prob):
                init.kaiming_uniform_(p, a=math.sqrt(5))

    def reset_parameters(self):
        self.conv1.reset_parameters()
        for conv in self.convs:
            conv.reset_parameters()
        self.lin1.reset_parameters()
        self.lin2.reset_parameters()
        self._init_masks()

    def cal_probability(self, x, edge_index, edge_weight, plan=None):
        """:321-332 -> (x*prob, edge_weight*e, prob, e)."""
        plan = plan if plan is not None else ops.GraphPlan(edge_index, x.shape[0])
        xm, ewm, e = ops.EdgeMask.apply(x, self.prob, self.prob_bias, edge_weight, plan, self.rois)
        return xm, ewm, self.prob, e

    def loss_probability(self, x, edge_index, edge_weight, hp, eps=1e-6, plan=None, edge_prob=None):
        """:334-358 (no SNP term; node-mask L1 = sum|sigmoid(prob)| / rois)."""
        if edge_prob is None:
            _, _, _, edge_prob = self.cal_probability(x, edge_index, edge_weight, plan=plan)
        none = self.prob.new_empty(0)
        return ops.MaskRegulariser.apply(self.prob, edge_prob, none, hp.lamda_x_l1 * self.prob_dim, hp.lamda_x_ent,
                                         hp.lamda_e_l1, hp.lamda_e_ent, eps)

    def forward(self, data, isExplain=False):
        """:360-388."""
        return self._forward_grouped(data, (bool(isExplain),))[0]

    def forward_pair(self, data):
        """(model(data), model(data, True)) of train() kernel/train_eval_sgcn.py:303,305 as one batched sweep."""
        return self._forward_grouped(data, (False, True))

    def _forward_grouped(self, data, explain_flags):
        x, edge_index, edge_weight = data.x, data.edge_index, data.edge_attr
        x.requires_grad = True                                         # :362 — populates data.x.grad
        self.input = x
        n = x.shape[0]
        if n % self.rois:
            raise ValueError(f"every graph must have exactly rois={self.rois} nodes (got {n} nodes)")
        bsz, g = n // self.rois, len(explain_flags)
        plan = ops.plan_for(data)
        plan.flush_pending_check()
        self.last_edge_prob = None
        if tuple(explain_flags) == (False, True) and x.is_cuda:
            # the train step's (plain | masked) pair: cal_probability writes both halves of the stacked batch itself
            x_in, ew_in, e = ops.EdgeMaskStacked.apply(x, self.prob, self.prob_bias, edge_weight, plan, self.rois)
            self.last_edge_prob = e
        else:
            x_m = ew_m = None
            if any(explain_flags):
                x_m, ew_m, _, e = self.cal_probability(x, edge_index, edge_weight, plan=plan)
                self.last_edge_prob = e
            xs = [x_m if f else x for f in explain_flags]
            ews = [ew_m if f else edge_weight for f in explain_flags]
            x_in = xs[0] if g == 1 else torch.cat(xs, dim=0)
            ew_in = ews[0] if g == 1 else torch.cat(ews, dim=0)
        plan_g = plan.replicate(g)
        xcat = sgcn_stack([self.conv1, *self.convs], x_in, ew_in, plan_g, self.rois,
                          os.environ.get("IGCN_NO_FUSED_SGCN", "0") != "1")
        z = xcat.view(g * bsz, -1)                                    # to_dense_batch == view (:378-381)
        f1 = ops.linear(z, self.lin1.weight, self.lin1.bias, relu=True)
        if self.training and self._dropout_enabled:
            f1 = F.dropout(f1, 0.5, True)
        logp = F.log_softmax(ops.linear(f1, self.lin2.weight, self.lin2.bias), dim=-1)
        return [logp[k * bsz:(k + 1) * bsz] for k in range(g)] if g > 1 else [logp]

    def __repr__(self):
        return self.__class__.__name__
