"""One optimisation step of the IG-GCN hot path — mirror of ``train()`` in the reference's
kernel/train_eval_sgcn_img_snps.py:511-548 (two forwards, seven loss terms, backward, Adam), plus the
graph-batch data parallelism the reference does not have (one process per GPU, one RCCL all-reduce of a
flat fp32 gradient buffer per step).
"""
import os
import sys
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import call, ptr, stream_ptr

# sgcn_hyperparameters.py:18-23
HP = SimpleNamespace(lamda_x_l1=0.1, lamda_e_l1=0.1, lamda_x_ent=0.1, lamda_e_ent=0.1, lamda_mi=1, lamda_ce=1)
# main.py:73-78,204 : [disease, regr, prob, reco, simi, orth]
DEFAULT_LAMBDA = (0.0, 1.0, 0.5, 1.5e-6, 0.1, 0.0)


class _ParamGroup(dict):
    """The one ``param_groups`` entry of FlatAdam: a dict like torch.optim's, whose ``'lr'`` item is mirrored into the
    device scalar the Adam kernel reads — so the reference's schedule
    (``param_group['lr'] = lr_decay_factor * param_group['lr']``, kernel/train_eval_sgcn_img_snps.py:169-171) reaches
    a step that was captured into a hipGraph.  ``betas`` / ``eps`` are by-value launch arguments: changing them
    invalidates captured steps (GraphedTrainStep refuses to replay)."""

    def __init__(self, owner, **items):
        super().__init__(**items)
        self._owner = owner

    def __setitem__(self, key, value):
        if key == "lr":
            value = float(value)
            self._owner._write_lr(value)
        elif key in ("betas", "eps") and (key not in self or self[key] != value):
            self._owner._hyper_version += 1
        elif key in ("weight_decay", "amsgrad", "maximize") and value:
            raise ValueError(f"FlatAdam is Adam({key}={value!r}) of the reference only with weight_decay=0, "
                             "amsgrad=False, maximize=False (kernel/train_eval_sgcn_img_snps.py:108, main.py:92)")
        super().__setitem__(key, value)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def setdefault(self, key, default=None):
        if key not in self:
            self[key] = default
        return self[key]


class FlatAdam:
    """Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0) over ONE flat fp32 parameter buffer — the drop-in for
    ``Adam(model.parameters(), lr=lr, weight_decay=weight_decay)`` at kernel/train_eval_sgcn_img_snps.py:108
    (``weight_decay`` must be 0, the default of main.py).

    Parameters (and the Adam moments) are views into contiguous buffers.  Two gradient modes:

    * table mode (default on the GPU): ``zero_grad`` just drops the ``.grad`` references, autograd hands over
      freshly written gradient tensors, and ONE multi-tensor kernel (igcn_adam_step_multi) walks a device table
      of {param, grad, exp_avg, exp_avg_sq} pointers — no per-parameter AccumulateGrad add, no memset.  Under
      data parallelism igcn_pack_grads gathers the gradients into the flat bucket for the single all-reduce.
    * flat mode (``flat_grads=True``): ``.grad`` are views of one flat buffer that autograd accumulates into.

    Parameters without a gradient are left untouched, like torch.optim.Adam.  The step counter is ONE device int32
    for all parameters (torch keeps one per parameter: the two agree whenever the set of parameters that receive a
    gradient does not change from step to step, which holds for this model).

    The torch.optim surface the reference's epoch loop touches is here: ``param_groups`` (one group; assigning
    ``['lr']`` writes the device scalar the kernel reads — also inside a captured hipGraph), ``zero_grad``, ``step``,
    ``state_dict`` / ``load_state_dict`` in torch.optim.Adam's own format (checkpoints interchange both ways).
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, flat_grads=None):
        if weight_decay:
            raise ValueError("FlatAdam: weight_decay must be 0 (the reference trains with main.py's default 0)")
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        align = 16                                   # floats: every parameter starts on a 64-byte boundary, so
        offs, n = [], 0                              # the GEMM kernels can stage weights with 16-byte loads
        for p in self.params:
            offs.append(n)
            n += (p.numel() + align - 1) // align * align
        self.flat_grads = (dev.type != "cuda") if flat_grads is None else bool(flat_grads)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)    # read by k_adam / k_adam_multi at run time
        self._hyper_version = 0
        self.param_groups = [_ParamGroup(self, params=self.params, betas=tuple(betas), eps=float(eps),
                                         weight_decay=0, amsgrad=False, maximize=False)]
        self.param_groups[0]["lr"] = lr
        with torch.no_grad():
            for p, off in zip(self.params, offs):
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view_as(p)
                p.grad = self.grad[off:off + k].view_as(p) if self.flat_grads else None
        nt = len(self.params)
        self._offs = offs
        # parameters that have been through a step with a gradient (torch.optim.Adam creates their state then):
        # the entries state_dict() writes
        self._has_state = [False] * nt
        # pinned staging buffers for the pointer table, used round-robin: a buffer is rewritten only after the
        # upload that read it has completed (the host may run several eager steps ahead of the device)
        self._hosts = [torch.zeros(nt, 4, dtype=torch.int64, pin_memory=(dev.type == "cuda")) for _ in range(4)]
        self._uploaded = [None] * len(self._hosts)
        self._turn = 0
        for host in self._hosts:
            for t, (p, o) in enumerate(zip(self.params, offs)):
                host[t, 0] = self.flat.data_ptr() + 4 * o
                host[t, 2] = self.exp_avg.data_ptr() + 4 * o
                host[t, 3] = self.exp_avg_sq.data_ptr() + 4 * o
        self.table = torch.zeros(nt, 4, dtype=torch.int64, device=dev)
        self._table_owner = None                     # who wrote ``table`` last (a GraphedTrainStep restores its own)
        self._table_live = [False] * nt              # which parameters the current table holds a gradient for
        self.numel = torch.tensor([p.numel() for p in self.params], dtype=torch.int64, device=dev)
        self.offset = torch.tensor(offs, dtype=torch.int64, device=dev)
        self._off_host = [int(o) for o in offs]
        # block list of the multi-tensor step: one workgroup per chunk of a tensor (igcn_adam_step_blocks)
        self._blocks = None
        if dev.type == "cuda":
            chunk = int(_lib.load().igcn_adam_chunk())
            bt, bo = [], []
            for t, p in enumerate(self.params):
                for lo in range(0, p.numel(), chunk):
                    bt.append(t)
                    bo.append(lo)
            self._blocks = (torch.tensor(bt, dtype=torch.int32, device=dev), torch.tensor(bo, dtype=torch.int32, device=dev))

    # ---- hyper-parameters -------------------------------------------------------------------------
    def _write_lr(self, value):
        if self.lr_dev.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FlatAdam: the learning rate is written between steps, not inside a stream capture")
        self.lr_dev.fill_(value)                     # stream-ordered: the next step (eager or replayed) reads it

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value):
        self.param_groups[0]["lr"] = value

    @property
    def betas(self):
        return self.param_groups[0]["betas"]

    @property
    def eps(self):
        return self.param_groups[0]["eps"]

    def zero_grad(self, set_to_none=True):
        if self.flat_grads:
            self.grad.zero_()
        else:
            for p in self.params:
                p.grad = None

    def refresh_table(self, owner=None, table=None):
        """Upload the current gradient pointers (host-side, not capturable: under a hipGraph the gradient
        tensors keep their addresses, so this runs once after capture).  ``table``: a captured step's OWN pointer table
        (``GraphedTrainStep`` keeps one per graph, so captured steps can alternate without re-uploading anything)."""
        k = self._turn
        self._turn = (k + 1) % len(self._hosts)
        if self._uploaded[k] is not None:
            self._uploaded[k].synchronize()
        host = self._hosts[k]
        for t, p in enumerate(self.params):
            g = p.grad
            if g is not None and not g.is_contiguous():
                g = p.grad = g.contiguous()
            host[t, 1] = g.data_ptr() if g is not None else 0
            self._table_live[t] = g is not None
        (self.table if table is None else table).copy_(host, non_blocking=True)
        if table is None:
            self._table_owner = owner
        if self.table.is_cuda:
            self._uploaded[k] = torch.cuda.Event()
            self._uploaded[k].record()

    def pack_grads(self, refresh=True, table=None, lo=0, hi=None):
        """Gather the per-tensor gradients into the flat bucket ``self.grad`` (data-parallel exchange).  ``lo`` / ``hi``: only
        the parameters [lo, hi) of ``self.params`` (a bucket of the two-bucket exchange)."""
        if self.flat_grads:
            return self.grad
        if refresh:
            self.refresh_table(table=table)
        hi = len(self.params) if hi is None else hi
        if hi <= lo:
            return self.grad
        tab = self.table if table is None else table
        call("igcn_pack_grads", hi - lo, tab.data_ptr() + 32 * lo, self.numel.data_ptr() + 8 * lo,
             self.offset.data_ptr() + 8 * lo, ptr(self.grad), stream_ptr())
        return self.grad

    def bucket_of(self, params):
        """(lo, hi, flat slice) of a run of CONSECUTIVE parameters of ``self.params``: the slice of the flat bucket that
        holds exactly their gradients (every parameter starts on a 64-byte boundary; the padding in between belongs to the
        parameter in front and is zero)."""
        ids = [id(p) for p in self.params]
        idx = sorted(ids.index(id(p)) for p in params)
        lo, hi = idx[0], idx[-1] + 1
        if idx != list(range(lo, hi)):
            raise ValueError("bucket_of: the parameters are not consecutive in the optimiser")
        off = self._offsets_host()
        start = off[lo]
        end = off[hi] if hi < len(self.params) else int(self.grad.numel())
        return lo, hi, self.grad[start:end]

    def _offsets_host(self):
        if getattr(self, "_off_host", None) is None:
            self._off_host = [int(v) for v in self.offset.cpu().tolist()]
        return self._off_host

    def step(self, grad_scale=1.0, refresh=True, from_flat=None, table=None):
        """``from_flat``: read gradients from the flat bucket (after an all-reduce); default = flat mode only.
        When the backward's flush has advanced ``step_count`` already (``backward_to_grads(tick=True)`` with deferred
        reductions sets ``_ticked``), the one-thread counter launch in front is skipped."""
        hyper = (ptr(self.lr_dev), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(grad_scale))
        sfx = "_ticked" if getattr(self, "_ticked", False) else ""
        self._ticked = False
        if self.flat_grads or from_flat:
            call("igcn_adam_step" + sfx, self.flat.numel(), ptr(self.flat), ptr(self.grad), ptr(self.exp_avg),
                 ptr(self.exp_avg_sq), ptr(self.step_count), *hyper, stream_ptr())
            self._has_state = [True] * len(self.params)
            return
        if refresh:
            self.refresh_table(table=table)
        tab = self.table if table is None else table
        if self._blocks is not None:
            call("igcn_adam_step_blocks", int(self._blocks[0].numel()), ptr(tab), ptr(self.numel), ptr(self._blocks[0]),
                 ptr(self._blocks[1]), ptr(self.step_count), *hyper, 1 if sfx else 0, stream_ptr())
        else:
            call("igcn_adam_step_multi" + sfx, len(self.params), ptr(tab), ptr(self.numel), ptr(self.step_count), *hyper,
                 stream_ptr())
        self.mark_stepped()

    def mark_stepped(self):
        """Book-keeping of a table-mode step (also called per replay of a captured one): the parameters the table holds
        a gradient for now have Adam state."""
        live = self._table_live
        if live != self._has_state:
            self._has_state = [a or b for a, b in zip(self._has_state, live)]

    # ---- checkpointing: torch.optim.Adam's own layout ----------------------------------------------
    def state_dict(self):
        """``torch.optim.Adam(params).state_dict()`` of the same parameter list: ``state[i] = {step, exp_avg,
        exp_avg_sq}`` for every parameter that has been stepped, one param group.  Host-synchronising (a checkpoint)."""
        step = float(int(self.step_count.item()))
        state = {}
        for t, (p, o) in enumerate(zip(self.params, self._offs)):
            if not self._has_state[t] or step == 0:
                continue
            k = p.numel()
            state[t] = {"step": torch.tensor(step, dtype=torch.float32),
                        "exp_avg": self.exp_avg[o:o + k].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + k].view_as(p).clone()}
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts FlatAdam's and torch.optim.Adam's state_dict of the same parameter list."""
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("FlatAdam.load_state_dict: expected one param group over "
                             f"{len(self.params)} parameters")
        g = groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
            raise ValueError("FlatAdam.load_state_dict: weight_decay / amsgrad / maximize are not supported")
        index = {pid: t for t, pid in enumerate(g["params"])}
        steps = set()
        with torch.no_grad():
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            self._has_state = [False] * len(self.params)
            for pid, st in sd["state"].items():
                t = index[pid]
                p, o = self.params[t], self._offs[t]
                k = p.numel()
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"FlatAdam.load_state_dict: state {pid} has shape {tuple(st['exp_avg'].shape)}, "
                                     f"parameter {tuple(p.shape)}")
                self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
                self._has_state[t] = True
            if len(steps) > 1:
                raise ValueError(f"FlatAdam keeps one step counter; the state holds {sorted(steps)}")
            self.step_count.fill_(steps.pop() if steps else 0)
        self.param_groups[0]["betas"] = tuple(g["betas"])
        self.param_groups[0]["eps"] = float(g["eps"])
        self.param_groups[0]["lr"] = g["lr"]


def losses_sgcn(model, data, hp=HP):
    """Loss of the image-only sibling: train() kernel/train_eval_sgcn.py:303-308
    (``lamda_ce*ce + loss_probability + lamda_mi*mi``).  Returns (loss, terms dict, outputs)."""
    if getattr(model, "batched_passes", True):
        out, out_p = model.forward_pair(data)
    else:
        out, out_p = model(data), model(data, True)
    y = data.y.view(-1)
    t = {"ce": F.nll_loss(out, y), "mi": F.nll_loss(out_p, y),
         "prob": model.loss_probability(data.x, data.edge_index, data.edge_attr, hp,
                                        edge_prob=model.last_edge_prob)}
    return hp.lamda_ce * t["ce"] + t["prob"] + hp.lamda_mi * t["mi"], t, (out, out_p)


def _batched(model):
    """The model takes both passes of a train step in one batched sweep (``_losses_batched``)."""
    return (hasattr(model, "go_network") and getattr(model, "batched_passes", True)
            and hasattr(model, "_forward_grouped") and model.isSoftSimilarity)


def losses(model, data, lambda_loss=DEFAULT_LAMBDA, hp=HP, temperature=None, lazy_value=False):
    """train() :521-543.  Returns (loss, terms dict, outputs).  A model without a GO branch (``SGCN_GCN``)
    takes the three-term loss of kernel/train_eval_sgcn.py:303-308 (``lambda_loss`` is not used there).
    ``lazy_value`` (train_step / GraphedTrainStep): ``loss`` and the terms may hold their VALUES only once the backward
    that follows has finished — every gradient is complete without them, and the last step of the sum then rides in the
    backward's final launch (ops.HeadLoss) instead of sitting on the step's critical path."""
    if not hasattr(model, "go_network"):
        return losses_sgcn(model, data, hp)
    # (the fused loss launches prepare their backward for the cached unit upstream gradient when it exists: registered here,
    # so that a process's FIRST evaluation takes the same route — the same rounding — as every later one, whoever calls)
    if data.x.is_cuda and torch.is_grad_enabled():
        ensure_unit_grad(data.x.device)
    if hasattr(model, "_reg_hp"):                    # the dense-block SGCN path reduces loss_probability in its forward
        model._reg_hp = (float(hp.lamda_x_l1), float(hp.lamda_x_ent), float(hp.lamda_e_l1), float(hp.lamda_e_ent), 1e-6)
    if getattr(model, "batched_passes", True) and hasattr(model, "_forward_grouped") and model.isSoftSimilarity:
        return _losses_batched(model, data, lambda_loss, hp, temperature, lazy_value)
    lam = lambda_loss
    dev = data.x.device
    o1 = model(data, temperature, dev)
    o2 = model(data, temperature, dev, isExplain=True)
    out, snps_hat, out_feat, _, _, reg = o1
    out_p, snps_hat_p, out_feat_p, _, _, reg_p = o2
    y = data.y.view(-1)
    clin = data.clini_score.view(-1)
    t = {}
    t["ce"] = lam[0] * F.nll_loss(out, y)
    t["mi"] = lam[0] * F.nll_loss(out_p, y)
    t["reg"] = lam[1] * (F.mse_loss(reg.view(-1), clin) + F.mse_loss(reg_p.view(-1), clin)) / 2
    # the mask loss re-evaluates cal_probability on the same inputs in the reference (:528); the explain
    # pass has just produced that very edge mask, so it is reused (same value, gradients add up)
    t["prob"] = lam[2] * model.loss_probability(data.x, data.edge_index, data.edge_attr, hp,
                                                edge_prob=model.last_edge_prob)
    t["recon"] = lam[3] * (torch.sum((snps_hat - data.snps_feat) ** 2)
                           + torch.sum((snps_hat_p - data.snps_feat) ** 2)) / 2
    if model.isSoftSimilarity:
        # the RBF Laplacian depends on the batch only: built once, shared by both passes; consist_loss and
        # OrthogonalConstraint of the plain pass come from the same Gram matrix
        lap = model.laplacian(out_feat.shape[0], data.tsne_fdim)
        c1, orth = model.batch_losses(out_feat, lap)
        c2, _ = model.batch_losses(out_feat_p, lap)
        t["cluster"] = lam[4] * (c1 + c2) / 2
    else:
        orth = None
        t["cluster"] = 0.0
        for c in range(2):
            sel = data.clust_y.view(-1) == c
            t["cluster"] = t["cluster"] + lam[4] * (model.consist_loss(out_feat[sel])
                                                    + model.consist_loss(out_feat_p[sel])) / 2
    t["orth"] = lam[5] * (orth if orth is not None else model.OrthogonalConstraint(out_feat))
    if lam[0] == 0:
        t["ce"], t["mi"] = 0.0, 0.0
    loss = hp.lamda_ce * t["ce"] + hp.lamda_mi * t["mi"] + t["reg"] + t["prob"] + t["recon"] + t["cluster"] \
        + t["orth"]
    return loss, t, (o1, o2)


def _losses_batched(model, data, lam, hp, temperature, lazy_value=False):
    """Same seven terms on the outputs of ONE batched sweep over both passes (rows [0,B) = plain pass of :521,
    rows [B,2B) = isExplain pass of :523): the mask regulariser, one Gram matrix per pass, and ONE loss-head kernel
    per direction (igcn_loss_head_*) for the terms and their weighted sum — the per-pass means of equal-sized
    halves are taken on the stacked tensors, so nothing is sliced."""
    dev = data.x.device
    from . import ops
    # the Gram products of the batch losses (out_z out_z^T per pass) read what the heads' first layers read — the outputs
    # of the fusion — and depend on nothing else: queued from inside the forward, they ride in that grouped GEMM launch
    # (ops.gram_rider) instead of a launch of their own behind the forward
    pre = {}

    def queue_gram(z):
        # (a bf16 heads launch does not carry fp32 products: the batched launch behind the forward stays the better one)
        if z.is_cuda and z.dtype == torch.float32 and z.is_contiguous() and z.shape[0] % 2 == 0 \
                and not getattr(model, "bf16_transforms", False) \
                and os.environ.get("IGCN_NO_GRAM_RIDER", "0") != "1" and os.environ.get("IGCN_NO_GEMM_GROUPS", "0") != "1":
            pre["gram"], pre["hold"] = ops.gram_rider(z.detach(), 2)

    # (the fused output-heads + loss launch reads its weights from a cached device vector: uploaded outside captures only)
    heads = data.x.is_cuda and ops._loss_weights(lam, hp.lamda_ce, hp.lamda_mi, data.num_graphs,
                                                 model.lin2_regr.weight.shape[0], dev) is not None
    try:
        scores, x_hat, out_z, out_lin, lin_f, reg = model._forward_grouped(data, temperature, dev, (False, True),
                                                                           split=False, raw_scores=True,
                                                                           on_out_z=queue_gram, heads_to_loss=heads)
    except BaseException:
        if pre:
            call("igcn_rider_cancel", stream_ptr())      # the queued products must not outlive their buffers
        raise
    if pre:
        call("igcn_gemm_rider_flush", stream_ptr())      # (a forward whose heads took another route: launch them now)
    # the Gram terms and the mask regulariser arrive as un-reduced partial sums and the class scores raw: the loss
    # kernel adds the partials up and takes log_softmax itself (two reductions and two torch launches less); the RBF
    # Laplacian of consist_loss is built inside the Gram loss kernel (model.laplacian() is a launch of its own)
    soft = model.isSoftSimilarity and data.tsne_fdim is not None
    # (the loss head's gradient of these partials for a unit upstream is known here: the Gram loss forward prepares its
    # own backward for it — ops.GramLosses ``expect``)
    unit = ops.unit_dgram(lam) if (ops.UNIT_GRAD_PTRS and os.environ.get("IGCN_NO_LOSS_HEAD_FUSED", "0") != "1") else None
    # (with the output layers left to the loss launch, the Gram loss launch is left to it too: two roles of one grid)
    job = {} if (isinstance(scores, tuple) and os.environ.get("IGCN_NO_GRAM_LOSS_PAIRED", "0") != "1") else None
    gram = ops.GramLosses.apply(out_z, None, 2, "partials", (data.tsne_fdim if soft else None, model.rbf_gamma), unit,
                                pre.get("gram"), job)
    # (rows sum to [2,2] = (consist, orth) per pass)
    prob = model.loss_probability(data.x, data.edge_index, data.edge_attr, hp, edge_prob=model.last_edge_prob,
                                  partials=True)
    lam6 = [float(v) for v in lam]
    if isinstance(scores, tuple):
        # the output layers were left to the loss launch (ops.HeadLoss: lin2 | lin2_regr, log_softmax, three loss terms and
        # the backward of all of it in one multi-workgroup launch; ``lazy_value``: the loss value joins the backward's flush)
        _, hf, keep1, hr, keep2 = scores
        loss, terms, logp, reg = ops.HeadLoss.apply(hf, keep1, model.lin2.weight, model.lin2.bias, hr, keep2,
                                                    model.lin2_regr.weight, model.lin2_regr.bias, data.y.view(-1),
                                                    data.clini_score.view(-1), x_hat, data.snps_feat, gram, prob, lam6,
                                                    hp.lamda_ce, hp.lamda_mi, bool(lazy_value),
                                                    job.get("gram") if job is not None else None)
    else:
        loss, terms, logp = ops.LossHead.apply(scores, data.y.view(-1), reg, data.clini_score.view(-1), x_hat,
                                               data.snps_feat, gram, prob, lam6, hp.lamda_ce, hp.lamda_mi, True)
    t = dict(zip(("ce", "mi", "reg", "prob", "recon", "cluster", "orth"), terms.unbind(0)))
    return loss, t, (logp, x_hat, out_z, out_lin, lin_f, reg)


_UNIT = {}


def _unit_grad(loss):
    """d loss / d loss = 1 as a cached device scalar (autograd would otherwise launch a fill for it every step)."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    if key not in _UNIT:
        if loss.is_cuda and torch.cuda.is_current_stream_capturing():
            return None                  # never allocate the cached scalar from a capture's private pool
        _UNIT[key] = torch.ones(loss.shape, dtype=loss.dtype, device=loss.device)
        if loss.is_cuda and loss.dtype == torch.float32:
            from . import ops
            ops.UNIT_GRAD_PTRS.add(_UNIT[key].data_ptr())   # (the tensor lives as long as the process: its address is its identity)
    return _UNIT[key]


def ensure_unit_grad(device):
    """Register the cached d loss / d loss = 1 of ``device`` BEFORE a step's forward: the forward's fused loss launches
    (ops.LossHead / ops.HeadLoss) prepare their backward for exactly that upstream gradient when they know it exists — the
    very first step of a process would otherwise take the unfused route and differ from every later one in rounding."""
    device = torch.device(device)
    if device.type == "cuda" and (device, torch.float32, ()) not in _UNIT and not torch.cuda.is_current_stream_capturing():
        _unit_grad(torch.empty((), dtype=torch.float32, device=device))


def backward_to_grads(loss, optimizer, data=None, defer=False, tick=False):
    """``loss.backward()`` for the table-mode FlatAdam: the gradients are taken with ``torch.autograd.grad`` and
    assigned to ``.grad`` as they come.  ``backward()`` routes every leaf through AccumulateGrad, which CLONES a
    gradient it cannot steal — and the kernels here hand back several parameter gradients as slices of one flat
    buffer (dW_inc | dW_s | da_in | da_s ...), i.e. views: a dozen device copies per step that nothing needs, since
    the Adam kernel reads the gradients through a pointer table.  Other optimisers keep ``backward()``.

    ``tick`` (with ``defer``): the launch that ends the backward also advances ``optimizer.step_count`` — for callers
    that ALWAYS follow this backward with exactly one ``optimizer.step()`` (``train_step``, ``GraphedTrainStep``).  A
    loop that may skip the step (gradient accumulation, a non-finite-loss guard) leaves it off."""
    params = getattr(optimizer, "params", None)
    if params is None or getattr(optimizer, "flat_grads", True):
        loss.backward()
        return
    leaves = list(params)
    if data is not None and getattr(data, "x", None) is not None and data.x.requires_grad:
        leaves.append(data.x)
    if defer:
        from . import ops
        # the small "sum the block partials" launches that end ~30 backward kernels are queued and run as ONE launch
        # when the block exits: the gradients below are complete only after it.  Only for the batched sweep, where
        # every parameter enters the graph ONCE — a parameter used twice has its two gradients added by autograd
        # during the backward, i.e. before the flush.
        # ... and the flush launch advances the optimiser's step counter on its way (one launch less in front of Adam)
        counter = getattr(optimizer, "step_count", None) if tick else None
        if getattr(optimizer, "_ticked", False):
            raise RuntimeError("backward_to_grads(tick=True): the previous backward advanced the step counter and no "
                               "optimizer.step() followed it")
        with ops.deferred_reductions(tick=counter):
            grads = torch.autograd.grad(loss, leaves, grad_outputs=_unit_grad(loss), allow_unused=True)
        if counter is not None:
            optimizer._ticked = True
    else:
        grads = torch.autograd.grad(loss, leaves, grad_outputs=_unit_grad(loss), allow_unused=True)
    for t, g in zip(leaves, grads):
        t.grad = g


def backward_two_buckets(loss, optimizer, data, model, on_early, tick=False):
    """``backward_to_grads(defer=True)`` in TWO sweeps for the two-bucket gradient exchange: the forward ran with
    ``model._cut_heads`` (the heads took detached copies of their inputs, ``model._cut``).  Sweep 1 stops at those copies:
    the heads' parameter gradients are final after its flush, ``on_early()`` is called (pack + start their all-reduce),
    sweep 2 takes the rest of the model from the loss and from the cut tensors' gradients — autograd prunes the heads'
    first layers from it.  Same kernels, same arithmetic, same gradients bit for bit as the one-sweep backward; the price is
    a second flush launch and the loss head's (launch-free) backward node visited twice."""
    from . import ops
    cut = getattr(model, "_cut", None)
    if not cut:
        raise RuntimeError("backward_two_buckets: the forward did not cut the heads off (model._cut_heads)")
    early = list(model.head_parameters())
    early_ids = {id(p) for p in early}
    rest = [p for p in optimizer.params if id(p) not in early_ids]
    if data is not None and getattr(data, "x", None) is not None and data.x.requires_grad:
        rest.append(data.x)
    if getattr(optimizer, "_ticked", False):
        raise RuntimeError("backward_two_buckets(tick=True): the previous backward advanced the step counter and no "
                           "optimizer.step() followed it")
    unit = _unit_grad(loss)
    copies = [c for _, c in cut]
    with ops.deferred_reductions():
        g1 = torch.autograd.grad(loss, early + copies, grad_outputs=unit, retain_graph=True, allow_unused=True)
    for p, g in zip(early, g1[:len(early)]):
        p.grad = g
    on_early()
    counter = getattr(optimizer, "step_count", None) if tick else None
    outs = [loss] + [x for x, _ in cut]
    gouts = [unit] + list(g1[len(early):])
    with ops.deferred_reductions(tick=counter):
        g2 = torch.autograd.grad(outs, rest, grad_outputs=gouts, allow_unused=True)
    if counter is not None:
        optimizer._ticked = True
    for t, g in zip(rest, g2):
        t.grad = g
    model._cut = None


def _single_use_parameters(model):
    """True when the step runs both passes as ONE batched sweep (``losses`` picks ``_losses_batched`` /
    ``forward_pair``): every parameter then enters the autograd graph once, which is what deferring the final
    gradient reductions needs (``backward_to_grads``).  IGCN_NO_DEFER=1 switches the deferral off (A/B runs)."""
    if os.environ.get("IGCN_NO_DEFER", "0") == "1" or not getattr(model, "batched_passes", True):
        return False
    if hasattr(model, "go_network"):
        return hasattr(model, "_forward_grouped") and bool(model.isSoftSimilarity)
    return hasattr(model, "forward_pair")


def forget_riders(model=None):
    """Drop every rider still waiting on the current stream WITHOUT launching it (igcn_rider_cancel) together with the
    dropout masks a model drew ahead for a forward that never came (go_network._predrawn): the start of every step, and
    the error path of one that raised half way."""
    call("igcn_rider_cancel", stream_ptr())
    go = getattr(model, "go_network", None)
    if go is not None:
        go._predrawn = None


def stream_pending():
    """What is still queued for the current stream and has not been launched: the library's host-side queues
    (igcn_stream_pending: deferred reductions, dropout rider, product riders) plus the two parameter-gradient queues of
    ops (LayerNorm affine passes, SNP <-> GO value-gradient passes).  0 at the end of every step."""
    from . import ops
    return int(_lib.load().igcn_stream_pending(stream_ptr())) + len(ops._DEFER["ln_affine"]) + len(ops._DEFER["spmm_dval"])


def assert_nothing_pending(where):
    """Checked mode (IGCN_DEBUG_SYNC=1): a step must leave none of the library's queues behind."""
    if _lib._DEBUG_SYNC:
        n = stream_pending()
        if n:
            raise _lib.IgcnError(f"{where}: {n} queued launch(es) left on the stream at the end of the step")


def _two_buckets_possible(model, optimizer):
    """The two-bucket exchange needs the table-mode optimiser, the batched sweep with deferred reductions and a model whose
    heads can be cut off (SGCN_GCN_IMGSNP)."""
    return (hasattr(model, "head_parameters") and not getattr(optimizer, "flat_grads", True)
            and _single_use_parameters(model))


class _TwoBucketExchange:
    """The gradient exchange of a data-parallel step in TWO all-reduces (``two_buckets=True``; IGCN_DP_TWO_BUCKETS=1 in
    bench.py): the heads' gradients (``model.head_parameters()``: 4/5 of the bucket, complete after the first fifth of
    the backward) are reduced on a SIDE stream while the launch stream runs the rest of the backward; the remainder
    follows on the launch stream; Adam waits for both.  Collectives of one communicator are issued in the same order on
    every rank (early, then rest).  The reference has no multi-GPU code."""

    def __init__(self, model, optimizer, comm):
        self.opt, self.comm = optimizer, comm
        self.lo, self.hi, self.early = optimizer.bucket_of(model.head_parameters())
        flat = optimizer.grad
        start = self.early.data_ptr() - flat.data_ptr()
        n0 = start // 4
        n1 = n0 + self.early.numel()
        self.rest = [t for t in (flat[:n0], flat[n1:]) if t.numel()]
        self.side = torch.cuda.Stream(device=flat.device)
        self.ev_packed = torch.cuda.Event()
        self.ev_reduced = torch.cuda.Event()

    def _all_reduce(self, t):
        if self.comm is not None:
            self.comm.all_reduce_(t)
        else:
            torch.distributed.all_reduce(t)

    def pack_early(self, table=None, refresh=True):
        self.opt.pack_grads(refresh=refresh, table=table, lo=self.lo, hi=self.hi)

    def start_early(self):
        """Behind pack_early on the launch stream: the early bucket's all-reduce on the side stream."""
        main = torch.cuda.current_stream()
        self.ev_packed.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ev_packed)
            self._all_reduce(self.early)
            self.ev_reduced.record(self.side)

    def pack_rest(self, table=None, refresh=True):
        self.opt.pack_grads(refresh=refresh, table=table, lo=0, hi=self.lo)
        self.opt.pack_grads(refresh=False, table=table, lo=self.hi, hi=len(self.opt.params))

    def finish(self):
        """The remainder's all-reduce on the launch stream, then the join with the side stream."""
        for t in self.rest:
            self._all_reduce(t)
        torch.cuda.current_stream().wait_event(self.ev_reduced)


def train_step(model, optimizer, data, lambda_loss=DEFAULT_LAMBDA, hp=HP, temperature=None, world_size=1,
               comm=None, two_buckets=False):
    """One iteration of the loop body of train() :515-547.  Returns the (device) loss tensor.

    With ``world_size > 1`` every rank has run the step on its own shard of the graph batch; gradients are summed
    with ONE all-reduce over the flat buffer and averaged inside the Adam kernel (grad_scale = 1/W).  ``comm``
    (``igcn_amd.comm.Comm``): the all-reduce is libigcn's igcn_comm_allreduce (RCCL) on the launch stream;
    otherwise ``torch.distributed.all_reduce`` of the initialised process group (backend nccl == RCCL).
    BatchNorm running statistics stay LOCAL to each rank (DDP convention): call
    ``broadcast_buffers(model)`` before checkpointing if one rank's ``state_dict()`` should stand for all.
    """
    optimizer.zero_grad()
    if data.x.grad is not None:
        data.x.grad = None
    forget_riders(model)          # (a failed step before this one may have left riders / masks drawn ahead: ADVICE r4)
    ensure_unit_grad(data.x.device)
    if two_buckets and (world_size > 1 or comm is not None) and _two_buckets_possible(model, optimizer):
        ex = getattr(optimizer, "_two_bucket_exchange", None)
        if ex is None or ex.comm is not comm:
            ex = optimizer._two_bucket_exchange = _TwoBucketExchange(model, optimizer, comm)
        model._cut_heads = True
        try:
            loss, _, _ = losses(model, data, lambda_loss, hp, temperature, lazy_value=True)

            def on_early():
                ex.pack_early()
                ex.start_early()
            backward_two_buckets(loss, optimizer, data, model, on_early, tick=True)
        except BaseException:
            forget_riders(model)
            raise
        finally:
            model._cut_heads = False
        assert_nothing_pending("train_step")
        ex.pack_rest()
        ex.finish()
        optimizer.step(grad_scale=1.0 / world_size, from_flat=True)
        return loss.detach()
    try:
        loss, _, _ = losses(model, data, lambda_loss, hp, temperature, lazy_value=True)
        backward_to_grads(loss, optimizer, data, defer=_single_use_parameters(model), tick=True)
    except BaseException:
        forget_riders(model)
        raise
    assert_nothing_pending("train_step")
    if world_size > 1 or comm is not None:
        flat = optimizer.pack_grads()
        if comm is not None:
            comm.all_reduce_(flat)
        else:
            torch.distributed.all_reduce(flat)
        optimizer.step(grad_scale=1.0 / world_size, from_flat=True)
    else:
        optimizer.step()
    return loss.detach()


class _capture:
    """``torch.cuda.graph`` with the cyclic garbage collector held off while the capture is open.  torch collects once on
    entry; a collection that starts DURING the capture — any allocation of the step, on this thread or on the autograd
    engine's — may finalise an object whose destructor the runtime refuses inside a capture (another captured step's
    hipGraph: every GraphedTrainStep sits in a reference cycle with its optimiser's pointer table, so it dies whenever the
    collector gets to it) and the process aborts."""

    def __init__(self, graph, **kw):
        self.ctx = torch.cuda.graph(graph, **kw)

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        self.ctx.__enter__()            # (synchronises, collects once, empties the allocator cache, begins the capture)
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        try:
            return self.ctx.__exit__(*exc)
        finally:
            if self.was:
                gc.enable()


class GraphedTrainStep:
    """The whole optimisation step captured ONCE into a hipGraph and replayed per step.

    CDNA4 idiom for a launch-bound step (hundreds of microsecond-sized kernels): no tracing compiler,
    just stream capture of the eager step — graph-plan build, two forwards, losses, backward and Adam
    become one graph launch.  Inputs live in static device tensors (``self.data``); ``load(batch)`` copies
    a new batch of identical shape into them.  With ``world_size > 1`` the gradient all-reduce stays
    outside the graphs: [zero_grad .. backward] graph -> RCCL all-reduce -> [Adam] graph.

    The graph plan of the batch is rebuilt every step, inside the graph: all three builders (LDS per graph, tiled
    counting sort per graph, general LSD radix sort) are hand-written kernels with caller-owned workspace.

    Construction runs ``warmup`` eager steps (allocator / library warm-up) on the construction batch; the optimiser
    state (parameters, Adam moments, step count) and every module buffer (BatchNorm running statistics,
    ``num_batches_tracked``) are snapshotted before and restored after them, so N replays equal N reference steps.
    """

    def __init__(self, model, optimizer, data, lambda_loss=DEFAULT_LAMBDA, hp=HP, world_size=1, warmup=3,
                 distributed=None, comm=None, comm_in_graph=False, max_edges=None, two_buckets=False):
        """``comm`` (``igcn_amd.comm.Comm``): the gradient all-reduce is igcn_comm_allreduce on the launch stream.
        ``comm_in_graph`` (opt-in): capture it INTO the step graph (one graph: ... pack -> all-reduce -> Adam).  True =
        required (a refused capture raises); None = try — the choice is agreed across ranks (an all-reduce of a flag)
        and the cause of a refusal is printed — and fall back to [graph] -> all-reduce -> [Adam graph]; False
        (default) = the two-graph form: in-graph capture of a multi-rank collective is verified on single-rank
        communicators only.
        ``max_edges``: size the captured per-graph kernels for graphs of up to this many edges (default: the largest
        graph of the construction batch); ``load`` refuses batches beyond it.
        ``two_buckets`` (opt-in, distributed steps): THREE graphs — [forward .. the heads' backward + pack of their
        gradients], [the rest of the backward + pack], [Adam] — with the heads' all-reduce (4/5 of the bucket) on a side
        stream beside the second graph and the remainder's behind it (``_TwoBucketExchange``); collectives stay outside
        the graphs.  Same gradients, bit for bit, as the default form."""
        self.model, self.opt, self.data, self.world = model, optimizer, data, world_size
        # distributed=True with world_size 1 takes the multi-rank control flow (two graphs around a collective) on
        # a single-rank process group: the rehearsal of the N>1 path that a one-GPU box allows
        self.dist = dist = (world_size > 1 or comm is not None) if distributed is None else bool(distributed)
        self.comm = comm
        self.comm_in_graph = False
        self.two = None
        if two_buckets and dist and _two_buckets_possible(model, optimizer):
            self.two = _TwoBucketExchange(model, optimizer, comm)
            comm_in_graph = False
        self.lam, self.hp = lambda_loss, hp
        from . import ops
        self.plan = ops.plan_for(data)                  # static plan tensors: rebuilt in place every step
        if max_edges is not None and self.plan._stack_dims is not None:
            if int(max_edges) < self.plan._stack_dims[1]:
                raise ValueError("max_edges is smaller than the construction batch's largest graph")
            self.plan._stack_dims = (self.plan._stack_dims[0], int(max_edges))
            self.plan._copies = {}
        # every plan builder is hand-written and capturable: the build is part of the captured step
        self.plan_in_graph = True
        opt = self.opt
        saved = [t.clone() for t in (opt.flat, opt.exp_avg, opt.exp_avg_sq, opt.step_count)]
        saved_buf = [(b, b.clone()) for b in model.buffers()]
        saved_has = list(opt._has_state)
        # the dropout generator's stream counter is not a registered buffer: saved / restored like them, so that a shape
        # captured in the middle of an epoch does not shift the masks of every later step against an eager run
        drop = getattr(getattr(model, "go_network", None), "_drop_state", None)
        saved_drop = drop.state.clone() if drop is not None else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                     # eager steps: allocator + library warm-up
                if self.two is not None:
                    def between_eager():
                        self.two.pack_early()
                        self.two.start_early()
                    self._fwd_bwd(between=between_eager)
                    self.two.pack_rest()
                    self.two.finish()
                    self.opt.step(grad_scale=1.0 / world_size, from_flat=True)
                    continue
                self._fwd_bwd()
                if dist:
                    self.opt.pack_grads()
                    self._reduce()
                    self.opt.step(grad_scale=1.0 / world_size, from_flat=True)
                else:
                    self.opt.step()
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():                           # undo the warm-up steps (capture itself executes nothing)
            for dst, src in zip((opt.flat, opt.exp_avg, opt.exp_avg_sq, opt.step_count), saved):
                dst.copy_(src)
            for b, v in saved_buf:
                b.copy_(v)
            drop_now = getattr(getattr(model, "go_network", None), "_drop_state", None)
            if saved_drop is not None and drop_now is drop:
                drop.state.copy_(saved_drop)
        forget_riders(model)
        opt._has_state = saved_has
        torch.cuda.synchronize()
        self.g_main = torch.cuda.CUDAGraph()
        # this graph's OWN gradient-pointer table: the captured Adam / pack kernels read it, so captured steps of one
        # optimiser (the shapes of an epoch, two alternating input sets) never re-upload anything when they take turns
        self._table = torch.zeros_like(opt.table)
        if not self.plan_in_graph:
            self.plan.rebuild(self.data.edge_index)
        # with a process group alive, its watchdog / progress threads may touch the runtime while this thread
        # captures: flag only this thread's unsafe calls (the launches of the autograd thread still land in the
        # capturing stream and are captured)
        mode = "thread_local" if dist else "global"
        if dist and comm is not None and comm_in_graph is not False:
            # one graph for the whole distributed step: RCCL's all-reduce is captured between pack and Adam
            try:
                with _capture(self.g_main, capture_error_mode=mode):
                    self.loss = self._fwd_bwd(rebuild=self.plan_in_graph)
                    self.opt.pack_grads(refresh=False, table=self._table)
                    comm.all_reduce_(self.opt.grad)
                    self.opt.step(grad_scale=1.0 / world_size, from_flat=True)
                self.comm_in_graph = True
            except Exception as exc:                     # noqa: BLE001 — capture refused: two graphs instead
                if comm_in_graph:
                    raise
                import sys
                print(f"[igcn] all-reduce capture refused ({type(exc).__name__}: {exc}); using two graphs around it",
                      file=sys.stderr)
                torch.cuda.synchronize()
                self.g_main = torch.cuda.CUDAGraph()
            if comm_in_graph is None and world_size > 1 and torch.distributed.is_initialized():
                # every rank must replay the same form: one rank inside a captured collective and another outside
                # would hang.  A rank that captured but must step down re-captures below.
                flag = torch.tensor([1 if self.comm_in_graph else 0], device=self.opt.flat.device, dtype=torch.int32)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                if self.comm_in_graph and int(flag.item()) == 0:
                    self.comm_in_graph = False
                    self.g_main = torch.cuda.CUDAGraph()
        self.g_rest = None
        if self.two is not None:
            # two captures around ONE backward: the first ends (and the second begins) between its two sweeps, on the same
            # capture stream and without leaving it (torch.cuda.graph's own entry — synchronise, collect, empty the
            # allocator's cache — must not run in the middle of a backward whose tensors live in the first graph's pool)
            import gc
            self.g_rest = torch.cuda.CUDAGraph()
            cap_stream = torch.cuda.Stream()
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.empty_cache()
            was = gc.isenabled()
            gc.disable()
            cap_stream.wait_stream(torch.cuda.current_stream())
            open_graph = [None]
            try:
                with torch.cuda.stream(cap_stream):
                    self.g_main.capture_begin(capture_error_mode=mode)
                    open_graph[0] = self.g_main

                    def between():
                        self.two.pack_early(table=self._table, refresh=False)
                        self.g_main.capture_end()
                        open_graph[0] = None
                        self.g_rest.capture_begin(pool=self.g_main.pool(), capture_error_mode=mode)
                        open_graph[0] = self.g_rest
                    try:
                        self.loss = self._fwd_bwd(rebuild=self.plan_in_graph, between=between)
                        self.two.pack_rest(table=self._table, refresh=False)
                    finally:
                        if open_graph[0] is not None:
                            open_graph[0].capture_end()
                torch.cuda.current_stream().wait_stream(cap_stream)
            finally:
                if was:
                    gc.enable()
        elif not self.comm_in_graph:
            with _capture(self.g_main, capture_error_mode=mode):
                self.loss = self._fwd_bwd(rebuild=self.plan_in_graph)
                if not dist:
                    self.opt.step(refresh=False, table=self._table)   # reads the pointer table at replay time
                else:
                    self.opt.pack_grads(refresh=False, table=self._table)
        # the captured gradient tensors keep their addresses — which only helps if the table can point AT them: a
        # non-contiguous gradient would be copied by refresh_table, and the replays would never update the copy
        for p in self.opt.params:
            if p.grad is not None and not p.grad.is_contiguous():
                raise _lib.IgcnError("graphed step: a parameter gradient is not contiguous "
                                     f"(shape {tuple(p.shape)}, strides {p.grad.stride()})")
        self.opt.refresh_table(table=self._table)
        # the parameters' ``.grad`` belong to whoever stepped last: an eager ``train_step`` (the ragged last batch of an
        # epoch) or another captured step re-points them, and __call__ puts this graph's own back before it replays
        self._table_live = list(self.opt._table_live)
        self._grads = [p.grad for p in self.opt.params]
        self._x_grad = self.data.x.grad
        self._hyper_version = self.opt._hyper_version
        self.g_opt = None
        if dist and not self.comm_in_graph:
            self.g_opt = torch.cuda.CUDAGraph()
            with _capture(self.g_opt, pool=self.g_main.pool(), capture_error_mode=mode):
                self.opt.step(grad_scale=1.0 / world_size, from_flat=True)
        torch.cuda.synchronize()

    def _fwd_bwd(self, rebuild=True, between=None):
        """``between`` (two-bucket form): called between the two sweeps of the backward, when the heads' gradients are
        final (``backward_two_buckets``)."""
        self.opt.zero_grad()
        self.data._igcn_plan = self.plan
        forget_riders(self.model)                       # (leftovers of a step that failed half way must never launch)
        ensure_unit_grad(self.data.x.device)
        rider = (rebuild and _batched(self.model) and hasattr(self.model, "predraw_dropout")
                 and os.environ.get("IGCN_NO_DROPOUT_RIDER", "0") != "1")
        try:
            if rider:
                self.model.predraw_dropout(self.data)   # queued: the plan build below carries the mask generation
            if rebuild:
                # the plan is per batch: rebuilt (in place) every step — by the first consumer: the front kernel of the
                # image branch builds it while it reads the edges anyway (ops.SgcnFront), any other route launches the
                # build; either launch carries the dropout rider
                self.plan.rebuild(self.data.edge_index, lazy=True)
            else:
                self.plan._copies = {}                  # the replica of the batched sweep is derived in-graph
            if (rider and getattr(self.plan, "_pending_build", None) is None
                    and not getattr(self.plan, "dense_blocks", False)):
                # (a plan build that does not carry riders: a launch of its own.  A dense-block plan has no build at all: the
                # first edge pass of ops.DenseSgcn carries the rider, and the model flushes it where that path is not taken)
                call("igcn_rider_flush", stream_ptr())
            self.data.x.grad = None
            self.model._cut_heads = between is not None
            try:
                loss, _, _ = losses(self.model, self.data, self.lam, self.hp, lazy_value=True)
            finally:
                self.model._cut_heads = False
            if rider:
                call("igcn_rider_flush", stream_ptr())  # (nothing waiting unless no launch of the forward took the rider)
            if between is not None:
                backward_two_buckets(loss, self.opt, self.data, self.model, between, tick=True)
            else:
                backward_to_grads(loss, self.opt, self.data, defer=_single_use_parameters(self.model), tick=True)
        except BaseException:
            # a step that raised between queueing a rider and its carrier: the rider's buffers die with this frame, and
            # the masks drawn ahead for THIS forward must not be handed to a later one
            forget_riders(self.model)
            raise
        assert_nothing_pending("GraphedTrainStep._fwd_bwd")
        return loss.detach()

    def _reduce(self):
        if self.comm is not None:
            self.comm.all_reduce_(self.opt.grad)
        elif self.dist:
            torch.distributed.all_reduce(self.opt.grad)

    def load(self, batch):
        """Copy a new batch (same graph count / node count / edge count) into the static input tensors — including
        the per-graph node and edge offsets the segmented plan build reads (a batch with the same total edge count
        but different per-graph counts would otherwise be grouped on the wrong segments).  ``batch is self.data``: a
        producer wrote the static inputs in place (loader.DeviceFeeder(into=step.data)) — nothing to copy."""
        if batch is self.data:
            return
        if self.plan.segmented:
            node_ptr, edge_ptr, _, _ = self.plan._seg
            bp, be = getattr(batch, "ptr", None), getattr(batch, "edge_ptr", None)
            if bp is None or be is None or bp.shape != node_ptr.shape or be.shape != edge_ptr.shape:
                raise ValueError("graphed step (segmented plan): the new batch must carry ptr / edge_ptr of the "
                                 "same graph count")
            mn, me = getattr(batch, "_max_nodes", None), getattr(batch, "_max_edges", None)
            # LDS build: the kernel's own limits; tiled build: the tile count the workspace was sized for
            e_lim = self.plan._seg[3] if self.plan._tiled else self.plan.SEG_MAX_EDGES
            n_lim = self.plan._seg[2] if self.plan._tiled else self.plan.SEG_MAX_NODES
            if mn is None or me is None or mn > n_lim or me > e_lim:
                raise ValueError("graphed step (per-graph plan): the new batch exceeds the per-graph sizes the "
                                 f"captured build was sized for ({n_lim} nodes / {e_lim} edges)")
            if self.plan._stack_dims is not None and me > self.plan._stack_dims[1]:
                # the captured LDS-resident SGCN stack was launched (and its LDS sized) for the construction batch's
                # largest graph; a graph beyond that would be skipped by its workgroup (and flagged in plan.status)
                raise ValueError("graphed step: a graph of the new batch has more edges "
                                 f"({me}) than the captured SGCN stack was sized for ({self.plan._stack_dims[1]}); "
                                 "build the step on a batch that contains the largest graph, or pass max_edges")
            if self.plan.nodes_per_graph and int(mn) * (int(bp.numel()) - 1) != self.plan.n_nodes:
                raise ValueError("graphed step: the captured kernels assume uniform graphs of "
                                 f"{self.plan.nodes_per_graph} nodes")
            pairs = [(node_ptr, bp), (edge_ptr, be)]
        else:
            pairs = []
        for k in ("x", "edge_index", "edge_attr", "snps_feat", "y", "clini_score", "tsne_fdim", "clust_y"):
            dst, src = getattr(self.data, k, None), getattr(batch, k, None)
            if dst is not None and src is not None:
                if dst.shape != src.shape:
                    raise ValueError(f"graphed step needs fixed shapes; {k}: {tuple(src.shape)} vs {tuple(dst.shape)}")
                pairs.append((dst, src))
        with torch.no_grad():
            _lib.copy_multi(pairs)                      # one launch for the whole hand-over (igcn_copy_multi)

    def _own_the_table(self):
        opt = self.opt
        if opt._hyper_version != self._hyper_version:
            raise RuntimeError("graphed step: betas / eps of the optimiser changed after the capture (they are by-value "
                               "launch arguments; only the learning rate is read from device memory) — rebuild the step")
        if getattr(opt, "_ticked", False):
            raise RuntimeError("graphed step: an eager backward advanced the step counter and no optimizer.step() "
                               "followed it")
        if opt._table_owner is not self:                # (host book-keeping only: the device table is this graph's own)
            opt._table_owner = self
            opt._table_live = list(self._table_live)
            for p, g in zip(opt.params, self._grads):
                p.grad = g
            self.data.x.grad = self._x_grad

    def __call__(self):
        self._own_the_table()
        if not self.plan_in_graph:
            self.plan.rebuild(self.data.edge_index)
        self.g_main.replay()
        if self.two is not None:
            self.two.start_early()                      # the heads' all-reduce: side stream, beside the second graph
            self.g_rest.replay()
            self.two.finish()                           # the remainder's, then the join
            self.g_opt.replay()
            self.opt._has_state = [True] * len(self.opt.params)
        elif self.g_opt is not None:
            self._reduce()
            self.g_opt.replay()
            self.opt._has_state = [True] * len(self.opt.params)
        elif self.dist:
            self.opt._has_state = [True] * len(self.opt.params)
        else:
            self.opt.mark_stepped()
        if _lib._DEBUG_SYNC:                            # checked mode: surface the segmented build's status flag
            self.plan.check()
        return self.loss


def _batch_signature(data):
    """What a captured step is specialised on: graph count and the shapes of every tensor ``load`` copies."""
    sig = [int(data.num_graphs)]
    for k in ("x", "edge_index", "edge_attr", "snps_feat", "y", "clini_score", "tsne_fdim", "clust_y"):
        v = getattr(data, k, None)
        sig.append(tuple(v.shape) if torch.is_tensor(v) else None)
    return tuple(sig)


def _clone_batch(data):
    """A private copy of a batch: the static input tensors of a captured step (``GraphedTrainStep.data``)."""
    import copy
    out = copy.copy(data)
    for k, v in list(vars(data).items()):
        if torch.is_tensor(v):
            setattr(out, k, v.detach().clone())
        elif k == "_igcn_plan":
            setattr(out, k, None)
    return out


class EpochTrainer:
    """``train()`` of kernel/train_eval_sgcn_img_snps.py:511-548 over a whole loader, on the fast path.

    The reference's loader is ``DataLoader(train_dataset, batch_size, shuffle=True)`` (:96-97: no ``drop_last``), so an
    epoch is a run of full batches plus one ragged tail.  Every batch SHAPE (graph count + tensor shapes) gets its own
    captured step once it has been seen ``capture_after`` times (default: the second time — one-off shapes never pay for
    a capture); until then, and beyond ``max_graphs`` captured shapes, the batch runs through the eager ``train_step``.
    Both routes are the same kernels on the same optimiser state, so an epoch is the same sequence of Adam steps either
    way; with dropout ON the two routes also draw from the same mask stream — a capture's warm-up steps are rolled back
    including the generator's counter — but a captured step draws its masks in the launch of its plan build and the eager
    step in a launch of its own, so runs that differ in ``capture_after`` agree in distribution, not bit for bit.  All
    captured steps share the optimiser: each restores its own gradient-pointer table before it replays.

    The learning-rate schedule of the epoch loop (:169-171) is ``optimizer.param_groups[0]['lr'] *= factor`` exactly as
    with torch.optim.Adam: the rate lives in device memory and captured steps read it there.
    """

    def __init__(self, model, optimizer, lambda_loss=DEFAULT_LAMBDA, hp=HP, world_size=1, comm=None, capture_after=1,
                 max_graphs=4, warmup=2):
        self.model, self.opt, self.lam, self.hp = model, optimizer, lambda_loss, hp
        self.world, self.comm = world_size, comm
        self.capture_after, self.max_graphs, self.warmup = int(capture_after), int(max_graphs), int(warmup)
        self.steps = {}                              # signature -> GraphedTrainStep
        self.seen = {}                               # signature -> times met
        self.counts = {"replayed": 0, "eager": 0, "captured": 0}

    def step(self, data):
        """One optimisation step on ``data`` (a device batch).  Returns the loss as a device scalar that stays valid
        until the next step of the same shape."""
        sig = _batch_signature(data)
        g = self.steps.get(sig)
        if g is None:
            n = self.seen.get(sig, 0)
            self.seen[sig] = n + 1
            graphable = data.x.is_cuda and isinstance(self.opt, FlatAdam) and not self.opt.flat_grads
            if graphable and n >= self.capture_after and len(self.steps) < self.max_graphs:
                g = self.steps[sig] = GraphedTrainStep(self.model, self.opt, _clone_batch(data), self.lam, self.hp,
                                                       world_size=self.world, comm=self.comm, warmup=self.warmup)
                self.counts["captured"] += 1
        if g is None:
            self.counts["eager"] += 1
            return train_step(self.model, self.opt, data, self.lam, self.hp, world_size=self.world, comm=self.comm)
        g.load(data)
        self.counts["replayed"] += 1
        return g()

    def fit_epoch(self, loader, device=None):
        """The body of ``train()``: one pass over ``loader``; returns ``sum_b loss_b * num_graphs_b / len(dataset)``
        (:546,548).  The per-batch ``loss.item()`` of the reference is one device accumulation, read once at the end."""
        self.model.train()
        total, count = None, 0
        for data in _batches(loader, device):
            loss = self.step(data)
            part = loss.reshape(()) * float(data.num_graphs)          # (a new tensor: the replayed loss is overwritten)
            total = part if total is None else total + part
            count += int(data.num_graphs)
        if total is None:
            return 0.0
        size = len(loader.dataset) if hasattr(loader, "dataset") else count
        return float(total) / size


def fit_epoch(model, optimizer, loader, temperature=None, lambda_loss=DEFAULT_LAMBDA, hp=HP, device=None,
              world_size=1, comm=None):
    """``train(model, optimizer, loader, temperature, lambda_loss, ..., device)`` of the reference (:511-548) as a
    function: the EpochTrainer is kept on the optimiser, so calling this once per epoch — as the reference's epoch
    loop calls ``train`` — re-uses the captured steps."""
    # (``temperature`` is accepted for the reference's signature and, like the reference's GO network, never read.)
    # The captured steps bake the regulariser weights of ``hp`` in as launch arguments: they are part of the key; the
    # model and the communicator are held by weak reference, so an id() recycled after garbage collection cannot alias
    import weakref
    hp_key = tuple(float(getattr(hp, k)) for k in ("lamda_x_l1", "lamda_e_l1", "lamda_x_ent", "lamda_e_ent", "lamda_mi",
                                                    "lamda_ce"))
    key = (id(model), tuple(float(v) for v in lambda_loss), hp_key, world_size, id(comm))
    cache = optimizer.__dict__.setdefault("_igcn_epoch_trainers", {})
    tr = cache.get(key)
    if tr is not None and (tr._model_ref() is not model or (comm is not None and tr._comm_ref() is not comm)):
        tr = None                                             # the id belonged to an object that is gone
    if tr is None:
        tr = cache[key] = EpochTrainer(model, optimizer, lambda_loss, hp, world_size=world_size, comm=comm)
        tr._model_ref = weakref.ref(model)
        tr._comm_ref = weakref.ref(comm) if comm is not None else (lambda: None)
    return tr.fit_epoch(loader, device)


def _batches(loader, device):
    for data in loader:
        yield data.to(device) if device is not None else data


@torch.no_grad()
def eval_loss(model, loader, lambda_loss=DEFAULT_LAMBDA, hp=HP, temperature=None, device=None):
    """eval_loss() kernel/train_eval_sgcn_img_snps.py:564-600 (and kernel/train_eval_sgcn.py:328-347): the
    training loss in eval mode, graph-weighted mean over the loader.  The per-batch ``.item()`` of the
    reference becomes one device accumulation and a single read at the end."""
    model.eval()
    total, count = None, 0
    for data in _batches(loader, device):
        loss, _, _ = losses(model, data, lambda_loss, hp, temperature)
        total = loss * data.num_graphs if total is None else total + loss * data.num_graphs
        count += data.num_graphs
    return float(total) / count


@torch.no_grad()
def eval_acc(model, loader, temperature=None, device=None):
    """eval_acc() kernel/train_eval_sgcn_img_snps.py:551-561 / kernel/train_eval_sgcn.py:316-325."""
    model.eval()
    correct, count = None, 0
    for data in _batches(loader, device):
        out = model(data, temperature, data.x.device) if hasattr(model, "go_network") else model(data)
        logp = out[0] if isinstance(out, tuple) else out
        hit = logp.max(1)[1].eq(data.y.view(-1)).sum()
        correct = hit if correct is None else correct + hit
        count += data.num_graphs
    return int(correct) / count


@torch.no_grad()
def eval_outputs(model, loader, temperature=None, device=None):
    """The tensors eval_scores() (kernel/train_eval_sgcn_img_snps.py:602-631) collects before handing them to
    sklearn: class scores, predictions, regression outputs, hidden features — concatenated over the loader on
    the device; the metric arithmetic itself (roc/f1/pearson) is host-side sklearn code and out of scope."""
    model.eval()
    cols = {"logp": [], "pred": [], "reg": [], "out_lin": [], "linear_outf": []}
    for data in _batches(loader, device):
        if hasattr(model, "go_network"):
            logp, _, _, out_lin, lin_f, reg = model(data, temperature, data.x.device)
            cols["reg"].append(reg.reshape(-1, model.num_regr))
            cols["out_lin"].append(out_lin)
            cols["linear_outf"].append(lin_f)
        else:
            logp = model(data)
        cols["logp"].append(logp)
        cols["pred"].append(logp.max(1)[1])
    return {k: torch.cat(v) for k, v in cols.items() if v}


def output_importance(model):
    """The arrays util/output.py:20-32 writes per fold: node / SNP importance and the edge-mask weights."""
    out = {"node_importance": model.prob.detach().cpu().numpy(),
           "prob_bias": model.prob_bias.detach().cpu().numpy()}
    if hasattr(model, "snps_prob"):
        out["snps_importance"] = model.snps_prob.detach().cpu().numpy()
    return out


def broadcast_buffers(model, src=0):
    """Copy rank ``src``'s module buffers (BatchNorm running statistics, ``num_batches_tracked``) to every rank.
    The data-parallel step keeps them local (every rank normalises with its own shard's statistics, like DDP without
    SyncBatchNorm), so replicas' buffers drift apart; call this before a checkpoint or an evaluation that should not
    depend on the rank."""
    if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        for b in model.buffers():
            torch.distributed.broadcast(b, src)


def allreduce_mean_(flat_grad, world_size):
    """Sum the flat gradient bucket over ranks (one collective) and divide by the world size, in place."""
    if world_size > 1:
        torch.distributed.all_reduce(flat_grad)
        flat_grad.div_(world_size)
    return flat_grad


def shard_batch(graphs, rank, world_size):
    """Contiguous graph-range shard of a list of graphs for rank r."""
    n = len(graphs)
    per = n // world_size
    if per * world_size != n:
        raise ValueError(f"{n} graphs do not split evenly over {world_size} ranks")
    return graphs[rank * per:(rank + 1) * per]
