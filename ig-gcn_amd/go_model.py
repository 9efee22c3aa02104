"""Drop-in ``Gene_ontology_network`` on the HIP kernels.

Mirrors the interface of the reference's kernel/go_model.py: constructor (:24), ``forward(data, T,
device) -> (latent, x_D, [zeros(3)], atten_out)`` (:205,287) and an identical ``state_dict()`` key set,
so checkpoints interchange.  The arithmetic is restructured for the GPU:

* activations are channel-major [B, f, N]; one kernel launch covers every sample (the reference loops
  over samples in python and builds two sparse matrices per sample and layer, :236-244);
* the hierarchy's edge lists are built once on the host as CSR + transposed CSR (``ops.Csr``); the
  reference keeps COO index tensors and re-derives row sums with torch.sparse.sum per sample;
* LayerNorm-over-nodes, ReLU, Dropout2d and the level pooling of :246-251 are one fused kernel.

Dense read-outs (BatchNorm over nodes, the latent MLP) are small torch ops on the same stream.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

N_SNPS = 54


def _coo_rows_cols(sp):
    sp = sp.coalesce()
    idx = sp.indices().cpu()
    keep = (sp.values() != 0).cpu()          # the reference's to_dense().to_sparse() drops stored zeros
    return idx[0][keep], idx[1][keep]


class Gene_ontology_network(nn.Module):
    def __init__(self, A_g, A, in_f_dim, n_l, f_dim, pool_dim, l_dim, device, dim_snps_atten=5):
        super().__init__()
        self.device = device
        pool = [int(p) for p in pool_dim[0]]
        n = int(A.shape[0])
        self.pool, self.n_l, self.n_nodes = pool, n_l, n
        n_top = n - sum(pool[:n_l])
        self.n_top = n_top
        f_dim = [in_f_dim] + list(f_dim)
        self.f_dim = f_dim

        # ---- index structures (go_model.py:42-88), host-built once -------------------------------
        r, c = _coo_rows_cols(A)
        self.enc_csr = []
        for i in range(n_l):
            off = sum(pool[:i])
            m = (r >= off) & (c >= off)
            self.enc_csr.append(ops.Csr(r[m] - off, c[m] - off, n - off, n - off, device))
        rt, ct = _coo_rows_cols(A.t())
        self.dec_csr = []
        for i in range(n_l):
            ro, co = sum(pool[:n_l - i - 1]), sum(pool[:n_l - i])
            m = (rt >= ro) & (ct >= co)
            self.dec_csr.append(ops.Csr(rt[m] - ro, ct[m] - co, n - ro, n - co, device))
        ag = A_g.coalesce()
        gi = ag.indices().cpu()
        self.gene_csr = ops.Csr(gi[0], gi[1], n, N_SNPS, device)                       # rows = GO nodes
        agt = A_g.t().coalesce()
        gti = agt.indices().cpu()
        self.gene_t_csr = ops.Csr(gti[0], gti[1], N_SNPS, n, device)                   # rows = SNPs
        nnz_g = int(gi.shape[1])

        # ---- parameters: same names / shapes / init as the reference (:80-157) --------------------
        self.t = nn.ParameterList([nn.Parameter(torch.empty(nnz_g).normal_(1.0, 0.1)) for _ in range(in_f_dim)])
        self.t_D = nn.ParameterList([nn.Parameter(torch.empty(nnz_g).normal_(1.0, 0.1))])
        lin = lambda i, o: nn.Linear(i, o, bias=False)       # noqa: E731
        self.w_inc = nn.ModuleList([lin(f_dim[i], f_dim[i + 1]) for i in range(n_l)])
        self.w_s_loop = nn.ModuleList([lin(f_dim[i], f_dim[i + 1]) for i in range(n_l)])
        self.w_att_s = nn.ModuleList([lin(f_dim[i + 1], 1) for i in range(n_l)])
        self.G_B = nn.ModuleList([nn.LayerNorm(sum(pool[i:])) for i in range(n_l)])
        self.w_att_in = nn.ModuleList([lin(2 * f_dim[i + 1], 1) for i in range(n_l)])
        self.w_out = nn.ModuleList([lin(f_dim[i], f_dim[i - 1]) for i in range(n_l, 0, -1)])
        self.w_s_loop_out = nn.ModuleList([lin(f_dim[i], f_dim[i - 1]) for i in range(n_l, 0, -1)])
        self.G_B_D = nn.ModuleList([nn.LayerNorm(sum(pool[i:])) for i in range(n_l - 1, -1, -1)])
        self.conc_for_attention = nn.Sequential(lin(f_dim[-1], dim_snps_atten), nn.BatchNorm1d(n_top), nn.ReLU())
        self.conc = lin(f_dim[-1], 1)
        self.B = nn.Sequential(nn.BatchNorm1d(n_top), nn.ReLU(), nn.Dropout(0.5))
        self.conc_D = lin(f_dim[0], 1)
        self.B_D = nn.Sequential(nn.BatchNorm1d(n), nn.ReLU(), nn.Dropout(0.5))
        self.latent = nn.Sequential(lin(n_top, 32), nn.BatchNorm1d(32), nn.ReLU(), nn.Dropout(0.5),
                                    lin(32, l_dim), nn.BatchNorm1d(l_dim), nn.ReLU())
        # present in the reference's state_dict, never used by forward (:148-157)
        self.classification = nn.Sequential(nn.BatchNorm1d(l_dim + N_SNPS), nn.ReLU(), nn.Dropout(0.5),
                                            lin(l_dim + N_SNPS, 16), nn.ReLU(), nn.Dropout(0.3),
                                            nn.Linear(16, 1, bias=True), nn.Sigmoid())
        self.node_dropout_p = 0.4           # nn.Dropout2d(0.4) at :104,113
        self._dropout_enabled = True        # parity tests switch every dropout off

    # ---------------------------------------------------------------------------------------------
    def _batch_counters(self):
        """num_batches_tracked of the BatchNorms this forward runs in training mode (go_model.py:119-146)."""
        bns = (self.conc_for_attention[1], self.B[0], self.B_D[0], self.latent[1], self.latent[5])
        return [bn.num_batches_tracked for bn in bns if bn.track_running_stats and bn.num_batches_tracked is not None]

    def predraw_dropout(self, b, dev, extra=(), groups=1):
        """Draw the masks of the NEXT training forward now, as a rider of the launch that follows on this stream (the
        per-graph plan build of a train step: ops.dropout_masks ``ride``); the forward picks them up."""
        if not (self.training and self._dropout_enabled):
            return
        self._predrawn = None
        try:
            res = self._dropout_masks(b, dev, extra, groups, ride=True)
        except Exception:
            self._predrawn = None
            raise
        key = (int(b), str(dev), tuple((tuple(s), float(p)) for s, p in extra), int(groups))
        self._predrawn = (key, res, self._counters_done)

    def _dropout_masks(self, b, dev, extra=(), groups=1, ride=False):
        """Every dropout of this forward pass from ONE kernel launch (igcn_dropout_masks), as {0, 1/(1-p)} factors
        that the consumers multiply by inside their own kernels:
          * nn.Dropout2d(0.4) on [B,N,f] (:104,113) zeroes whole nodes per sample -> ``ln`` [B,n] per LayerNorm site;
          * nn.Dropout(0.5) on the read-outs inp_out [B,n_top] (:128), out_D [B,N] (:136) and the latent MLP's hidden
            layer [B,32] (:143);
          * ``extra`` [(shape, p), ...]: sites of the enclosing model (the two F.dropout of the heads).
        Returns (dict of GO masks, list of extra masks) — Nones when dropout is off."""
        ln_sizes = [c.n_rows for c in self.enc_csr] + [c.n_rows for c in self.dec_csr]
        if not (self.training and self._dropout_enabled):
            return {"ln": [None] * len(ln_sizes), "inp": None, "out_d": None, "h": None}, [None] * len(extra)
        pre = getattr(self, "_predrawn", None)
        if pre is not None and not ride:
            # drawn ahead of the forward (predraw_dropout: the job rode in the plan build's launch) for exactly this call
            self._predrawn = None
            key = (int(b), str(dev), tuple((tuple(s), float(p)) for s, p in extra), int(groups))
            if pre[0] != key:
                raise RuntimeError("dropout masks were drawn ahead for another forward call (predraw_dropout)")
            self._counters_done = pre[2]
            return pre[1]
        state = getattr(self, "_drop_state", None)
        if state is None or state.state.device != dev:
            state = self._drop_state = ops.DropoutState(dev)
        sites = [((b, n), self.node_dropout_p) for n in ln_sizes]
        sites += [((b, self.n_top), 0.5), ((b, self.n_nodes), 0.5), ((b, self.latent[0].weight.shape[0]), 0.5)]
        sites += list(extra)
        # the launch that draws the masks also advances the BatchNorm batch counters (one torch launch less per step)
        cnt = self._batch_counters()
        m = ops.dropout_masks(sites, state, cnt, groups, ride=ride)
        self._counters_done = bool(cnt)
        k = len(ln_sizes)
        return {"ln": m[:k], "inp": m[k], "out_d": m[k + 1], "h": m[k + 2]}, m[k + 3:]

    def _drop(self, x, p):
        return F.dropout(x, p, True) if (self.training and self._dropout_enabled) else x

    def _node_linear_bn(self, x, weight, bn, groups=1, keep=None):
        """dropout(relu(bn(linear(x)))) with bn = BatchNorm1d(#nodes): one fused op (igcn_node_linear_bn_*)."""
        if self.training and bn.track_running_stats:
            self._tracked.append(bn.num_batches_tracked)
        return ops.NodeLinearBN.apply(x, weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                      self.training, bn.momentum, bn.eps, groups, keep)

    def _bn_relu(self, x, bn, groups=1, keep=None):
        """dropout(relu(bn(x))) for the latent MLP's BatchNorm1d layers (igcn_bn1d_*)."""
        if self.training and bn.track_running_stats:
            self._tracked.append(bn.num_batches_tracked)
        return ops.BatchNorm1dGrouped.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training,
                                            bn.momentum, bn.eps, True, groups, keep)

    def _lin_bn_relu(self, x, lin, bn, groups=1, keep=None):
        """dropout(relu(bn(lin(x)))) of the latent MLP (:138-146); a split-K product's slabs are summed by the BatchNorm
        launch itself (ops.LinearBN1d: no slab-sum launch behind the wide layer)."""
        if lin.bias is None and ops.linear_bn1d_supported(x, lin.weight, groups):
            if self.training and bn.track_running_stats:
                self._tracked.append(bn.num_batches_tracked)
            return ops.LinearBN1d.apply(x, lin.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training,
                                        bn.momentum, bn.eps, groups, keep)
        return self._bn_relu(ops.linear(x, lin.weight, lin.bias), bn, groups, keep)

    def forward(self, data, T=None, device=None, groups=1, extra_dropout=()):
        """``groups`` > 1: ``data`` holds that many equally sized batches stacked along dim 0 that are treated as
        successive forward calls (own BatchNorm statistics, running statistics updated in order).
        ``extra_dropout`` [(shape, p), ...]: dropout sites of the caller drawn by the same launch; their factors are
        left in ``self.extra_masks``."""
        bsz, dev = data.shape[0], data.device
        self._tracked = []
        self._counters_done = False
        masks, self.extra_masks = self._dropout_masks(bsz, dev, extra_dropout, groups)
        keeps = masks["ln"]
        # gene encoding (:208-215)
        x = ops.SparseMap.apply(data, self.gene_csr, *self.t)                            # [B, in_f, N]
        # encoder (:219-251)
        for j in range(self.n_l):
            csr = self.enc_csr[j]
            # layer + LayerNorm block as ONE autograd node: its backward is one launch when the layer runs LDS-resident
            # (ops.GoAttentionLN).  Leaves (no .view(-1): a view's gradient would pass through another backward node, and
            # the op could not defer its final reductions)
            # the encoder OUTPUT has three consumers (two read-outs, the decoder): the last layer hands out three
            # aliases and its backward adds their gradients while it loads them — no sum launch, no autograd adds
            fan = 3 if (j == self.n_l - 1 and x.is_cuda and torch.is_grad_enabled()
                        and os.environ.get("IGCN_NO_GRAD_FAN", "0") != "1") else 1
            x = ops.GoAttentionLN.apply(x, self.w_inc[j].weight, self.w_s_loop[j].weight, self.w_att_in[j].weight,
                                        self.w_att_s[j].weight, csr, self.G_B[j].weight, self.G_B[j].bias, keeps[j],
                                        self.pool[j], self.G_B[j].eps, fan)
        x_fan = None
        if isinstance(x, tuple):
            x_fan, x = x[:2], x[2]
        # read-outs (:254-255): BatchNorm1d(n_top) normalises per NODE over (batch, feature); fused kernels
        bn_a, bn_i = self.conc_for_attention[1], self.B[0]
        if ops.node_linear_bn_pair_supported(x, self.conc_for_attention[0].weight, self.conc.weight, None) \
                and os.environ.get("IGCN_NO_READOUT_PAIR", "0") != "1":
            # both read-outs of the encoder output in paired launches.  The encoder output has three consumers (two
            # read-outs, the decoder): their gradients are summed in one launch (ops.GradFan), not by two adds
            x_alias = None
            if x_fan is not None:
                x_pair, x_alias = x_fan
            elif x.requires_grad and x.is_cuda and os.environ.get("IGCN_NO_GRAD_FAN", "0") != "1":
                x_pair, x_alias, x = ops.GradFan.apply(x, 3)
            else:
                x_pair = x
            if self.training:
                self._tracked += [bn.num_batches_tracked for bn in (bn_a, bn_i) if bn.track_running_stats]
            atten_out, inp_out = ops.NodeLinearBNPair.apply(
                x_pair, x_alias, self.conc_for_attention[0].weight, bn_a.weight, bn_a.bias, bn_a.running_mean, bn_a.running_var,
                bn_a.momentum, bn_a.eps, self.conc.weight, bn_i.weight, bn_i.bias, bn_i.running_mean, bn_i.running_var,
                bn_i.momentum, bn_i.eps, masks["inp"], self.training, groups)
            inp_out = inp_out.squeeze(2)
        else:
            xa, xi = x_fan if x_fan is not None else (x, x)
            atten_out = self._node_linear_bn(xa, self.conc_for_attention[0].weight, bn_a, groups)
            inp_out = self._node_linear_bn(xi, self.conc.weight, bn_i, groups, masks["inp"]).squeeze(2)
        # decoder (:258-275)
        for j in range(self.n_l):
            csr = self.dec_csr[j]
            x = ops.GoDecodeLN.apply(x, self.w_out[j].weight, self.w_s_loop_out[j].weight, csr, self.G_B_D[j].weight,
                                     self.G_B_D[j].bias, keeps[self.n_l + j], self.G_B_D[j].eps)
        # gene decoding (:278-282)
        out_d = self._node_linear_bn(x, self.conc_D.weight, self.B_D[0], groups, masks["out_d"]).squeeze(2)   # [B,N]
        x_d = ops.SparseMap.apply(out_d, self.gene_t_csr, self.t_D[0]).squeeze(1)               # [B, 54]
        # latent projection (:138-146,285)
        h = self._lin_bn_relu(inp_out.view(bsz, -1), self.latent[0], self.latent[1], groups, masks["h"])
        latent = self._lin_bn_relu(h, self.latent[4], self.latent[5], groups)
        if self._tracked and not self._counters_done:  # num_batches_tracked of the five BatchNorms: one launch (with
            torch._foreach_add_(self._tracked, groups)  # dropout on, the mask launch has advanced them already)
        self._tracked = []
        zeros3 = getattr(self, "_zeros3", None)          # placeholder of the reference's unused third output
        if zeros3 is None or zeros3.device != dev:
            zeros3 = self._zeros3 = torch.zeros(3, device=dev)
        return latent, x_d, [zeros3], atten_out
