"""torch.autograd wrappers over the C ABI (include/igcn.h).  Every op here launches hand-written HIP
kernels from libigcn.so on the current torch stream; nothing falls back to PyTorch arithmetic.

Index structures (``GraphPlan`` for a batch of brain graphs, ``Csr``/``CsrPair`` for the GO hierarchy)
are plain int32 device tensors owned by Python.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import call, ptr, stream_ptr


def _f32(t):
    if t.dtype != torch.float32:
        raise _lib.IgcnError(f"libigcn computes in fp32; got {t.dtype}")
    return t.contiguous()


# =================================================================================================
# Deferred reductions of a backward pass (igcn_reduce_defer / igcn_reduce_flush)
# =================================================================================================
class _PerStream:
    """The deferral state of the CURRENT stream (a backward pass runs on the stream of its forward, whichever thread the
    autograd engine uses): trainers on different streams of one process keep their queues apart, like the library's."""

    def __init__(self):
        self.by_stream = {}

    def _state(self):
        key = stream_ptr() or 0
        st = self.by_stream.get(key)
        if st is None:
            st = self.by_stream[key] = {"on": False, "keep": [], "ln_affine": [], "spmm_dval": []}
        return st

    def __getitem__(self, k):
        return self._state()[k]

    def __setitem__(self, k, v):
        self._state()[k] = v


_DEFER = _PerStream()
# addresses of the cached "d loss / d loss = 1" scalars of train._unit_grad: ops.LossHead returns the gradients its
# forward wrote for that upstream instead of launching its backward kernel
UNIT_GRAD_PTRS = set()
# address -> values of the loss head's d loss / d (consist_1, orth_1, consist_2, orth_2) while a backward that took the
# unit path is running (set by LossHead.backward, taken by GramLosses.backward in the same pass)
UNIT_DGRAM = {}


def unit_dgram(lam):
    """d loss / d (consist, orth per pass) of train()'s weighted sum (:533-543) for d loss / d loss = 1 — what
    igcn_loss_head_fwd_grads writes into dgram: cluster = lam4 (c1 + c2) / 2, orth = lam5 o1."""
    return (float(lam[4]) * 0.5, float(lam[5]), float(lam[4]) * 0.5, 0.0)


class deferred_reductions:
    """``with ops.deferred_reductions(): backward`` — the ~30 "sum the block partials" launches whose output is a
    parameter gradient are queued by the library and performed by ONE launch at exit.  Gradients are therefore only
    valid after the block; the partial buffers are kept alive here until then.  Used by train.py around the backward
    of a step (the optimiser runs after the flush); plain ``loss.backward()`` elsewhere reduces immediately."""

    def __init__(self, tick=None):
        """``tick``: a device int32 counter (the optimiser's step) that the flush launch advances by one on its way —
        the optimiser then skips its own one-thread counter launch (train.FlatAdam.step(ticked=True))."""
        self.tick = tick

    def __enter__(self):
        if _DEFER["on"]:
            raise _lib.IgcnError("deferred_reductions does not nest")
        _DEFER["on"] = True
        call("igcn_reduce_defer", stream_ptr(), 1)
        return self

    def __exit__(self, *exc):
        ok = exc[0] is None
        try:
            if ok:                       # queued parameter-gradient passes: one launch per kind (their sums still deferred)
                _flush_ln_affine()
                _flush_spmm_dval()
        except BaseException:
            ok = False
            raise
        finally:
            # Whatever happened above, the library leaves defer mode and its queue is emptied HERE, while the partial
            # buffers the queued entries point at are still alive (``keep`` is cleared last): a failed block must not
            # leave entries behind that a later flush would run on freed memory.  On the error path the queued sums are
            # still launched — they only read buffers that exist and write gradients nobody will use.
            try:
                call("igcn_reduce_defer", stream_ptr(), 0)
                if self.tick is not None and ok:
                    call("igcn_reduce_flush_tick", stream_ptr(), ptr(self.tick))
                else:
                    call("igcn_reduce_flush", stream_ptr())
            finally:
                _DEFER["on"] = False
                _DEFER["ln_affine"].clear()
                _DEFER["spmm_dval"].clear()
                _DEFER["keep"].clear()
        if _lib._DEBUG_SYNC and ok:                  # checked mode: the flush has emptied the stream's queue
            n = int(_lib.load().igcn_stream_pending(stream_ptr()))
            if n:
                raise _lib.IgcnError(f"deferred_reductions: {n} queued launch(es) left on the stream after the flush")
        return False


def _flush_ln_affine():
    """The affine-gradient passes queued by NodesLayerNorm.backward, four layers per launch."""
    q = _DEFER["ln_affine"]
    for i in range(0, len(q), 4):
        part = q[i:i + 4]
        table = (ctypes.c_int64 * (13 * len(part)))()
        for j, (dims, tens) in enumerate(part):
            table[13 * j:13 * j + 13] = list(dims) + [ptr(t) or 0 for t in tens]
        call("igcn_nodes_ln_bwd_affine_multi", len(part), ctypes.addressof(table), stream_ptr())
    q.clear()


def _flush_spmm_dval():
    """The value-gradient passes queued by SparseMap.backward, two maps per launch."""
    q = _DEFER["spmm_dval"]
    for i in range(0, len(q), 2):
        part = q[i:i + 2]
        table = (ctypes.c_int64 * (12 * len(part)))()
        for j, (dims, tens) in enumerate(part):
            table[12 * j:12 * j + 11] = list(dims) + [ptr(t) or 0 for t in tens]
        call("igcn_spmm_bwd_dval_multi", len(part), ctypes.addressof(table), stream_ptr())
    q.clear()


def _keep(t):
    """Scratch whose block partials a queued reduction still has to read."""
    if _DEFER["on"] and t is not None:
        _DEFER["keep"].append(t)
    return t


def _leaves(*ts):
    """True when every given tensor is an autograd LEAF that takes a gradient (a trainable parameter): only then is its
    gradient final AND kept — a view of a parameter (``in_proj_weight.split``), or any tensor with a grad_fn, hands its
    gradient to another backward node, which would read it before the flush; and the gradient of a FROZEN parameter is
    dropped by autograd the moment the backward returns it, so a sum still queued for the flush would land in memory that a
    later tensor of the backward has inherited (the hazard train.backward_two_buckets ran into with the loss head)."""
    return all(t is None or (t.is_leaf and t.requires_grad) for t in ts)


class _immediate:
    """Reduce now, not at the flush, for one library call (the op's parameter-like inputs are not leaves)."""

    def __init__(self, final):
        self.off = _DEFER["on"] and not final

    def __enter__(self):
        if self.off:
            call("igcn_reduce_defer", stream_ptr(), 0)

    def __exit__(self, *exc):
        if self.off:
            call("igcn_reduce_defer", stream_ptr(), 1)
        return False


# =================================================================================================
# Graph plan (once per batch)
# =================================================================================================
class GraphPlan:
    """Stable by-target / by-source grouping of a batch's edges (igcn_graph_plan_build)."""

    SEG_MAX_NODES, SEG_MAX_EDGES = 1024, 4096

    def __init__(self, edge_index, n_nodes, node_ptr=None, edge_ptr=None, max_nodes=None, max_edges=None):
        """``node_ptr``/``edge_ptr`` (int64 device tensors [G+1]) + the host-known per-graph maxima select a per-graph
        build (graphs of at most 1024 nodes: in LDS up to 4096 edges per graph, tiled counting sort beyond);
        otherwise the general radix-sort build.  All three are hand-written and hipGraph-capturable."""
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise _lib.IgcnError("edge_index must be int64 [2,E]")
        ei = edge_index.contiguous()
        dev = ei.device
        self.n_nodes, self.n_edges = int(n_nodes), int(ei.shape[1])
        n, e = self.n_nodes, self.n_edges
        i32 = dict(dtype=torch.int32, device=dev)
        self.src32 = torch.empty(max(e, 1), **i32)
        self.dst32 = torch.empty(max(e, 1), **i32)
        self.tgt_ptr = torch.empty(n + 1, **i32)
        self.tgt_perm = torch.empty(max(e, 1), **i32)
        self.src_ptr = torch.empty(n + 1, **i32)
        self.src_perm = torch.empty(max(e, 1), **i32)
        self.loop_edge = torch.empty(n, **i32)
        self._copies = {}
        self.status = None
        self._seg = None
        self._tiled = False
        self._stack_dims = None
        # uniform graph size (every PyG batch of R-ROI brain graphs), passed to the aggregation as a hint
        self.nodes_per_graph = int(max_nodes) if (max_nodes and node_ptr is not None and
                                                  n == int(max_nodes) * (int(node_ptr.numel()) - 1)) else 0
        have_seg = (node_ptr is not None and edge_ptr is not None and max_nodes is not None and max_edges is not None
                    and 0 < max_nodes <= self.SEG_MAX_NODES and e > 0 and node_ptr.device == dev
                    and edge_ptr.device == dev)
        if have_seg:
            # per-graph builds (hand-written, hipGraph-capturable): all in LDS for small graphs, tiled counting sort
            # for graphs with more than SEG_MAX_EDGES edges (dense 512-ROI graphs)
            self.status = torch.zeros(2, **i32)          # [0] sticky flags (check()), [1] consumed by the dense kernels
            self._seg = (node_ptr.contiguous(), edge_ptr.contiguous(), int(max_nodes), int(max_edges))
            self._tiled = max_edges > self.SEG_MAX_EDGES
            if self.nodes_per_graph:                    # uniform graphs: (nodes, max edges) per graph
                self._stack_dims = (self.nodes_per_graph, int(max_edges))
            if self._tiled:
                nb = int(_lib.load().igcn_graph_plan_tiled_workspace_bytes(int(node_ptr.numel()) - 1, int(max_nodes),
                                                                           int(max_edges)))
                self._ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        else:
            self._ws = torch.empty(int(_lib.load().igcn_graph_plan_workspace_bytes(n, e)), dtype=torch.uint8,
                                   device=dev)
        self._build(ei)
        self.dense_blocks = self._detect_dense_blocks(ei)

    def _detect_dense_blocks(self, ei):
        """True when every graph of the batch is COMPLETE and stored in row-major order (a dense adjacency as COO:
        edge k of graph g = (g R + k / R, g R + k % R)): the SGCN kernels of csrc/sgcn_dense.hip then work on
        ``edge_attr`` as the dense matrix it is.  Decided once, at construction (one host read); ``rebuild`` re-verifies
        every new batch on the device (status bit 2).  IGCN_NO_DENSE_BLOCKS=1 keeps the general kernels."""
        r = self.nodes_per_graph
        if (not r or self._seg is None or os.environ.get("IGCN_NO_DENSE_BLOCKS", "0") == "1"
                or self._seg[3] != r * r or self.n_edges != (self.n_nodes // r) * r * r
                or not _lib.load().igcn_dense_sgcn_supported(r, 1, 16, 1) or ei.data_ptr() % 16
                or torch.cuda.is_current_stream_capturing()):
            return False
        flag = torch.zeros(2, dtype=torch.int32, device=ei.device)
        call("igcn_dense_blocks_check", self.n_nodes // r, r, ptr(ei), ptr(flag), stream_ptr())
        return int(flag[0].item()) == 0

    @property
    def segmented(self):
        return self._seg is not None

    def _build(self, ei):
        n, e = self.n_nodes, self.n_edges
        if self._seg is not None:
            node_ptr, edge_ptr, max_nodes, max_edges = self._seg
            head = (n, e, int(node_ptr.numel()) - 1, ptr(ei), ptr(node_ptr), ptr(edge_ptr), max_nodes, max_edges,
                    ptr(self.src32), ptr(self.dst32), ptr(self.tgt_ptr), ptr(self.tgt_perm), ptr(self.src_ptr),
                    ptr(self.src_perm), ptr(self.loop_edge), ptr(self.status))
            if self._tiled:
                call("igcn_graph_plan_build_tiled", *head, ptr(self._ws), self._ws.numel(), stream_ptr())
            elif self._copies:
                # a replica plan exists (the two passes of a train step as one block-diagonal problem): the build
                # fills it too, instead of a replicate launch behind it
                (copies, rep), = self._copies.items()
                call("igcn_graph_plan_build_segmented_rep", *head, copies, ptr(rep.src32), ptr(rep.dst32),
                     ptr(rep.tgt_ptr), ptr(rep.tgt_perm), ptr(rep.src_ptr), ptr(rep.src_perm), ptr(rep.loop_edge),
                     stream_ptr())
            else:
                call("igcn_graph_plan_build_segmented", *head, stream_ptr())
            return
        call("igcn_graph_plan_build", n, e, ptr(ei), ptr(self.src32), ptr(self.dst32), ptr(self.tgt_ptr),
             ptr(self.tgt_perm), ptr(self.src_ptr), ptr(self.src_perm), ptr(self.loop_edge), ptr(self._ws),
             self._ws.numel(), stream_ptr())

    def rebuild(self, edge_index, lazy=False):
        """Build again IN PLACE for a new ``edge_index`` of the same size (same graph segments): every plan tensor
        keeps its address, so kernels captured in a hipGraph keep reading the right memory.
        ``lazy``: a per-graph LDS build is only NOTED — the first consumer either builds the plan itself while it reads the
        edges anyway (ops.SgcnFront: ``take_pending_build``) or launches the build (``flush_pending_build``)."""
        if edge_index.shape != (2, self.n_edges) or edge_index.dtype != torch.int64:
            raise _lib.IgcnError("rebuild needs an int64 edge_index of the shape the plan was built for")
        self._pending_build = None
        if (lazy and self._seg is not None and not self._tiled and self._stack_dims is not None
                and not getattr(self, "dense_blocks", False)):
            self._pending_build = edge_index.contiguous()
            return
        if getattr(self, "dense_blocks", False):
            # complete row-major graphs: the new batch has the SAME structure or none the dense kernels can use — one
            # pass over edge_index verifies it (status bit 2); the sorted arrays of the first build stay valid.  The
            # pass is not launched here: it rides in the first edge pass of the dense SGCN forward that consumes this
            # plan (DenseSgcn takes it with ``take_pending_check``); any other consumer launches it (``flush_pending_check``).
            # [Forked onto a side branch of the captured step, joined at the loss head: +74 us per replay — a second
            # queue costs the replay far more than the 27 us it hides.]
            self._pending_check = edge_index.contiguous()
            return
        # the LDS per-graph build refills ONE cached replica in place (same addresses: capturable); any other build
        # drops the replicas, to be derived again on demand
        keep = (self._seg is not None and not self._tiled and len(self._copies) == 1
                and os.environ.get("IGCN_PLAN_REPLICATE_LAUNCH", "0") != "1")
        if not keep:
            self._copies = {}
        self._build(edge_index.contiguous())

    def take_pending_build(self):
        """The edge_index of a ``rebuild(lazy=True)`` nobody has performed yet (None: the plan is up to date)."""
        ei = getattr(self, "_pending_build", None)
        self._pending_build = None
        return ei

    def flush_pending_build(self):
        """Perform a deferred ``rebuild`` now (a consumer that reads the plan arrays instead of building them)."""
        ei = self.take_pending_build()
        if ei is not None:
            self.rebuild(ei)

    def take_pending_check(self):
        """The edge_index whose structure check is still to run (``rebuild`` of a dense-block plan), handed to the
        consumer that lets it ride in its own launch (igcn_dense_sgcn_fwd); None when nothing is pending."""
        ei = getattr(self, "_pending_check", None)
        self._pending_check = None
        return ei

    def flush_pending_check(self):
        """Launch a pending structure check now, as a launch of its own (a consumer other than the dense SGCN forward)."""
        ei = self.take_pending_check()
        if ei is not None:
            call("igcn_dense_blocks_check", self.n_nodes // self.nodes_per_graph, self.nodes_per_graph, ptr(ei),
                 ptr(self.status), stream_ptr())

    def check(self):
        """Host-synchronising validation of the segmented build (tests / debugging only)."""
        self.flush_pending_build()
        self.flush_pending_check()
        code = int(self.status[0].item()) if self.status is not None else 0
        if code & 4:
            raise _lib.IgcnError("dense-block plan: the batch is not made of complete graphs in row-major order any "
                                 "more (build a new plan for it)")
        if code & 2:
            raise _lib.IgcnError("fused SGCN stack: a graph has more edges than the launch was sized for "
                                 f"(max_edges = {self._stack_dims[1] if self._stack_dims else '?'}); its outputs "
                                 "were not computed")
        if code:
            raise _lib.IgcnError("graph plan: the batch is not block diagonal within the declared graph segments")

    def replicate(self, copies):
        """Plan of ``copies`` disjoint copies of this batch (igcn_graph_plan_replicate), cached."""
        if copies == 1:
            return self
        self.flush_pending_build()
        if copies not in self._copies:
            rep = object.__new__(GraphPlan)
            n, e = self.n_nodes, self.n_edges
            rep.n_nodes, rep.n_edges = n * copies, e * copies
            i32 = dict(dtype=torch.int32, device=self.src32.device)
            for name, size in (("src32", e * copies), ("dst32", e * copies), ("tgt_ptr", n * copies + 1),
                               ("tgt_perm", e * copies), ("src_ptr", n * copies + 1), ("src_perm", e * copies),
                               ("loop_edge", n * copies)):
                setattr(rep, name, torch.empty(max(size, 1), **i32))
            rep._copies, rep._seg, rep.status = {}, None, self.status     # one status word for the plan and its replicas
            rep.nodes_per_graph = self.nodes_per_graph
            rep._tiled, rep._stack_dims, rep.dense_blocks = False, self._stack_dims, False
            call("igcn_graph_plan_replicate", n, e, copies, ptr(self.src32), ptr(self.dst32), ptr(self.tgt_ptr),
                 ptr(self.tgt_perm), ptr(self.src_ptr), ptr(self.src_perm), ptr(self.loop_edge), ptr(rep.src32),
                 ptr(rep.dst32), ptr(rep.tgt_ptr), ptr(rep.tgt_perm), ptr(rep.src_ptr), ptr(rep.src_perm),
                 ptr(rep.loop_edge), stream_ptr())
            self._copies[copies] = rep
        return self._copies[copies]


def plan_for(data, keep_pending=False):
    """The GraphPlan of a batch object, built on first use and cached on it.  A build that ``rebuild(lazy=True)`` deferred
    is performed here unless the caller takes care of it itself (``keep_pending``: SGCN_GCN_IMGSNP._forward_grouped)."""
    plan = getattr(data, "_igcn_plan", None)
    if plan is None or plan.n_edges != data.edge_index.shape[1] or plan.src32.device != data.edge_index.device:
        plan = GraphPlan(data.edge_index, data.x.shape[0], getattr(data, "ptr", None),
                         getattr(data, "edge_ptr", None), getattr(data, "_max_nodes", None),
                         getattr(data, "_max_edges", None))
        try:
            data._igcn_plan = plan
        except AttributeError:
            pass
    if not keep_pending:
        plan.flush_pending_build()
    return plan


# =================================================================================================
# SGCN branch
# =================================================================================================
class EdgeMask(torch.autograd.Function):
    """cal_probability (kernel/sgcn_img_snp.py:133-151) -> (xm, ewm, e)."""

    @staticmethod
    def forward(ctx, x, prob, prob_bias, ew, plan, rois):
        x, prob, pb, ew = _f32(x), _f32(prob), _f32(prob_bias), _f32(ew)
        n, h0 = x.shape
        xm, e, ewm = torch.empty_like(x), torch.empty_like(ew), torch.empty_like(ew)
        call("igcn_edge_mask_fwd", n, plan.n_edges, rois, h0, ptr(x), ptr(prob), ptr(pb), ptr(ew),
             ptr(plan.src32), ptr(plan.dst32), ptr(xm), ptr(e), ptr(ewm), None, None, stream_ptr())
        ctx.save_for_backward(x, prob, pb, ew, e)
        ctx.plan, ctx.rois = plan, rois
        return xm, ewm, e

    @staticmethod
    def backward(ctx, d_xm, d_ewm, d_e):
        return _edge_mask_backward(ctx, d_xm, d_ewm, d_e, None)


def _edge_mask_backward(ctx, d_xm, d_ewm, d_e, d_x_plain):
    x, prob, pb, ew, e = ctx.saved_tensors[:5]
    plan, rois = ctx.plan, ctx.rois
    n, h0 = x.shape
    d_xm, d_ewm, d_e, d_x_plain = (_f32(t) if t is not None else None for t in (d_xm, d_ewm, d_e, d_x_plain))
    dx, dprob, dpb = torch.empty_like(x), torch.empty_like(prob), torch.empty_like(pb)
    scratch = torch.empty(n * h0 + 16 * ((n + 3) // 4) + 16, dtype=torch.float32, device=x.device)
    call("igcn_edge_mask_bwd", n, plan.n_edges, rois, h0, ptr(x), ptr(prob), ptr(pb), ptr(ew), ptr(e),
         ptr(d_xm), ptr(d_ewm), ptr(d_e), ptr(d_x_plain), ptr(plan.tgt_ptr), ptr(plan.tgt_perm), ptr(plan.src_ptr),
         ptr(plan.src_perm), ptr(dx), ptr(dprob), ptr(dpb), ptr(scratch), stream_ptr())
    return dx, dprob, dpb, None, None, None


class EdgeMaskStacked(torch.autograd.Function):
    """The (plain | masked) batch of the two passes of a train step in one launch: x_in = cat(x, xm) [2N,h0],
    ew_in = cat(ew, ewm) [2E], and the edge mask e — cal_probability (kernel/sgcn_img_snp.py:133-151) writing both
    halves itself instead of two concatenations behind it.

    With ``reg_hp`` = (l1_x, ent_x, l1_e, ent_e, eps) the launch also leaves loss_probability (:153-181) as a fourth
    output — workgroup partials whose SUM is the regulariser (ops.LossHead adds them up), over sigmoid(prob), e and
    sigmoid(``snps_logits``) — and the backward takes its gradient along: no regulariser launches in the step."""

    @staticmethod
    def forward(ctx, x, prob, prob_bias, ew, plan, rois, snps_logits=None, reg_hp=None, snps_feat=None):
        """``snps_feat`` [B, S] (with ``snps_logits`` and ``reg_hp``): a fifth output, the SNP mask of the stacked sweep
        cat(snps_feat, snps_feat * sigmoid(snps_logits)) [2B, S] (cal_probability :147-151), from the same launch."""
        x, prob, pb, ew = _f32(x), _f32(prob), _f32(prob_bias), _f32(ew)
        n, h0 = x.shape
        ne = ew.shape[0]
        x_in = torch.empty(2 * n, h0, dtype=torch.float32, device=x.device)
        ew_in = torch.empty(2 * ne, dtype=torch.float32, device=x.device)
        e = torch.empty_like(ew)
        ctx.reg = None
        snps = feat = full = None
        if reg_hp is not None:
            snps = _f32(snps_logits).reshape(-1) if snps_logits is not None else None
            ns = snps.numel() if snps is not None else 0
            rows = 0
            if snps_feat is not None:
                feat = _f32(snps_feat)
                rows = feat.shape[0]
                if snps is None or feat.dim() != 2 or feat.shape[1] != ns:
                    raise _lib.IgcnError("EdgeMaskStacked: snps_feat must be [B, S] with S = snps_logits.numel()")
                full = torch.empty(2 * rows, ns, dtype=torch.float32, device=x.device)
            hp = tuple(float(v) for v in reg_hp)
            regp = torch.empty(int(_lib.load().igcn_edge_mask_reg_blocks(n, plan.n_edges, h0, ns, rows)),
                               dtype=torch.float32, device=x.device)
            call("igcn_edge_mask_fwd_reg", n, plan.n_edges, rois, h0, ptr(x), ptr(prob), ptr(pb), ptr(ew), ptr(plan.src32),
                 ptr(plan.dst32), ptr(x_in[n:]), ptr(e), ptr(ew_in[ne:]), ptr(x_in[:n]), ptr(ew_in[:ne]), ptr(snps), ns, *hp,
                 ptr(regp), ptr(feat), rows, ptr(full), stream_ptr())
            ctx.reg = (hp, ns, snps_logits.shape if snps_logits is not None else None, rows)
        else:
            call("igcn_edge_mask_fwd", n, plan.n_edges, rois, h0, ptr(x), ptr(prob), ptr(pb), ptr(ew),
                 ptr(plan.src32), ptr(plan.dst32), ptr(x_in[n:]), ptr(e), ptr(ew_in[ne:]), ptr(x_in[:n]), ptr(ew_in[:ne]),
                 stream_ptr())
        ctx.save_for_backward(x, prob, pb, ew, e, snps, feat)
        ctx.plan, ctx.rois = plan, rois
        ctx.set_materialize_grads(False)
        if full is not None:
            return x_in, ew_in, e, regp, full
        if reg_hp is not None:
            return x_in, ew_in, e, regp
        return x_in, ew_in, e

    @staticmethod
    def backward(ctx, d_x_in, d_ew_in, d_e, d_regp=None, d_full=None):
        n, ne = ctx.saved_tensors[0].shape[0], ctx.saved_tensors[3].shape[0]
        d_xm = d_x_in[n:] if d_x_in is not None else None
        d_xp = d_x_in[:n] if d_x_in is not None else None
        d_ewm = d_ew_in[ne:] if d_ew_in is not None else None
        if ctx.reg is None:
            return _edge_mask_backward(ctx, d_xm, d_ewm, d_e, d_xp) + (None, None, None)
        x, prob, pb, ew, e, snps, feat = ctx.saved_tensors
        dx, dprob, dpb, dsnps = _edge_mask_reg_backward(x, prob, pb, ew, e, snps, feat, ctx.reg, ctx.plan, ctx.rois, d_xm,
                                                        d_ewm, d_e, d_xp, d_regp, d_full)
        return dx, dprob, dpb, None, None, None, dsnps, None, None


def _edge_mask_reg_backward(x, prob, pb, ew, e, snps, feat, reg, plan, rois, d_xm, d_ewm, d_e, d_xp, d_regp, d_full):
    """Backward of the stacked mask launch with the regulariser and the SNP mask riding in it (igcn_edge_mask_bwd_reg):
    (dx, dprob, dpb, dsnps)."""
    hp, ns, snps_shape, rows = reg
    n, h0 = x.shape
    d_xm, d_ewm, d_e, d_xp, d_full = (_f32(t) if t is not None else None for t in (d_xm, d_ewm, d_e, d_xp, d_full))
    # the partials' gradient is one scalar, repeated (ops.LossHead); none: the regulariser was not used
    greg = _f32(d_regp[:1]).reshape(1) if d_regp is not None else torch.zeros(1, dtype=torch.float32, device=x.device)
    dx, dprob, dpb = torch.empty_like(x), torch.empty_like(prob), torch.empty_like(pb)
    dsnps = torch.empty(snps_shape, dtype=torch.float32, device=x.device) if snps is not None else None
    scratch = torch.empty(n * h0 + 16 * ((n + 3) // 4) + 16, dtype=torch.float32, device=x.device)
    use_feat = feat is not None and d_full is not None
    call("igcn_edge_mask_bwd_reg", n, plan.n_edges, rois, h0, ptr(x), ptr(prob), ptr(pb), ptr(ew), ptr(e), ptr(d_xm),
         ptr(d_ewm), ptr(d_e), ptr(d_xp), ptr(plan.tgt_ptr), ptr(plan.tgt_perm), ptr(plan.src_ptr), ptr(plan.src_perm),
         ptr(greg), ptr(snps), ns, *hp, ptr(dx), ptr(dprob), ptr(dpb), ptr(dsnps), ptr(scratch),
         ptr(feat) if use_feat else None, rows if use_feat else 0, ptr(d_full) if use_feat else None, stream_ptr())
    return dx, dprob, dpb, dsnps


class GcnNorm(torch.autograd.Function):
    """PyG gcn_norm (add_remaining_self_loops + symmetric normalisation) -> (what [E], what_loop [N], streams).

    ``streams`` = (tstream, sstream): the same coefficients as 8-byte (neighbour, coefficient) records in by-target /
    by-source order, consumed by the aggregation kernels; not differentiable (gradients flow through ``what``)."""

    @staticmethod
    def forward(ctx, ew, plan):
        ew = _f32(ew)
        n, e = plan.n_nodes, plan.n_edges
        f = dict(dtype=torch.float32, device=ew.device)
        dis, wl = torch.empty(n, **f), torch.empty(n, **f)
        what, wloop = torch.empty(max(e, 1), **f), torch.empty(n, **f)
        tstream, sstream = torch.empty(max(e, 1), 2, **f), torch.empty(max(e, 1), 2, **f)
        call("igcn_gcn_norm_fwd", n, e, ptr(ew), ptr(plan.src32), ptr(plan.dst32), ptr(plan.tgt_ptr),
             ptr(plan.tgt_perm), ptr(plan.src_perm), ptr(plan.loop_edge), ptr(dis), ptr(wl), ptr(what), ptr(wloop),
             ptr(tstream), ptr(sstream), stream_ptr())
        ctx.save_for_backward(ew, dis, wl)
        ctx.plan = plan
        ctx.mark_non_differentiable(tstream, sstream)
        ctx.set_materialize_grads(False)          # no zero tensors for the two record streams in the backward
        return what, wloop, tstream, sstream

    @staticmethod
    def backward(ctx, dwhat, dwloop, _dt, _ds):
        ew, dis, wl = ctx.saved_tensors
        plan = ctx.plan
        n, e = plan.n_nodes, plan.n_edges
        dwhat = _f32(dwhat) if dwhat is not None else torch.zeros_like(ew)
        dwloop = _f32(dwloop) if dwloop is not None else torch.zeros_like(dis)
        dew = torch.empty_like(ew)
        scratch = torch.empty(n, dtype=torch.float32, device=ew.device)
        call("igcn_gcn_norm_bwd", n, e, ptr(ew), ptr(dis), ptr(wl), ptr(dwhat), ptr(dwloop), ptr(plan.src32),
             ptr(plan.dst32), ptr(plan.tgt_ptr), ptr(plan.tgt_perm), ptr(plan.src_ptr), ptr(plan.src_perm),
             ptr(plan.loop_edge), ptr(dew), ptr(scratch), stream_ptr())
        return dew, None


class GcnPropagate(torch.autograd.Function):
    """out = act(A_hat h + bias): the scatter-aggregate of GCNConv (+ F.relu of sgcn_img_snp.py:218,221).
    ``what`` carries the autograd dependency on the edge weights; the kernels read the sorted ``streams``."""

    @staticmethod
    def forward(ctx, h, what, wloop, bias, plan, relu, tstream, sstream):
        h, wloop = _f32(h), _f32(wloop)
        bias = _f32(bias) if bias is not None else None
        n, f = h.shape
        out = torch.empty_like(h)
        call("igcn_gcn_propagate_fwd", n, plan.n_edges, f, int(getattr(plan, "nodes_per_graph", 0) or 0), ptr(h), f,
             ptr(tstream), ptr(wloop), ptr(bias), ptr(plan.tgt_ptr), ptr(out), f, int(relu), stream_ptr())
        ctx.save_for_backward(h, what, wloop, out, sstream)
        ctx.plan, ctx.relu, ctx.has_bias = plan, int(relu), bias is not None
        ctx.final = _leaves(bias)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, what, wloop, out, sstream = ctx.saved_tensors
        plan = ctx.plan
        dout = _f32(dout)
        n, f = h.shape
        need_dw = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dh = torch.empty_like(h)
        dbias = torch.empty(f, dtype=torch.float32, device=h.device) if ctx.has_bias else None
        dwhat = torch.empty_like(what) if need_dw else None
        dwloop = torch.empty_like(wloop) if need_dw else None
        lib = _lib.load()
        scratch = _keep(torch.empty(int(lib.igcn_gcn_propagate_bwd_scratch_floats(n, f)), dtype=torch.float32,
                                    device=h.device))
        with _immediate(ctx.final):
            call("igcn_gcn_propagate_bwd", n, plan.n_edges, f, int(getattr(plan, "nodes_per_graph", 0) or 0), ptr(dout),
                 f, ptr(out), f, ctx.relu, ptr(h), f,
                 ptr(sstream), ptr(wloop), ptr(plan.src32), ptr(plan.dst32), ptr(plan.src_ptr), ptr(dh), f, ptr(dbias),
                 int(need_dw), ptr(dwhat), ptr(dwloop), ptr(scratch), stream_ptr())
        return dh, dwhat, dwloop, dbias, None, None, None, None


class GradFan(torch.autograd.Function):
    """``a, b[, c, d] = GradFan.apply(t, k)``: k aliases of ``t`` for k consumers.  The backward sums the k incoming
    gradients in ONE launch (igcn_sum_n) — autograd left to itself adds them pairwise, a library launch per add (seven
    per train step: the encoder output's three consumers, ``prob``'s three, ``data.x``, ``snps_prob``)."""

    @staticmethod
    def forward(ctx, t, k):
        ctx.set_materialize_grads(False)
        ctx.final = _leaves(t)          # a leaf's summed gradient is read by the optimiser only: a deferred reduction
        return tuple(t.view_as(t) for _ in range(k))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        ok = all(g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.data_ptr() % 16 == 0 for g in gs)
        if not ok or len(gs) > 4:
            out = gs[0]
            for g in gs[1:]:
                out = out + g
            return out, None
        out = torch.empty_like(gs[0])
        arr = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
        if ctx.final and _DEFER["on"]:
            for g in gs:
                _keep(g)                # the consumers' gradients live until the flush sums them
            call("igcn_sum_n_final", out.numel(), len(gs), arr, ptr(out), stream_ptr())
        else:
            call("igcn_sum_n", out.numel(), len(gs), arr, ptr(out), stream_ptr())
        return out, None


def sgcn_stack_supported(plan, rois, h0, f, layers):
    """The LDS-resident SGCN stack (igcn_sgcn_stack_*) covers this batch: per-graph plan of uniform graphs (block
    diagonal, verified by the builder), supported widths, and the graph fits LDS."""
    seg = getattr(plan, "_stack_dims", None)
    if seg is None or seg[0] != rois or f not in (4, 8, 16, 32) or not (1 <= layers <= 4) or not (1 <= h0 <= 8):
        return False
    lib = _lib.load()
    return int(lib.igcn_sgcn_stack_lds_bytes(rois, seg[1], h0, f, layers, 1)) <= 150 * 1024


class SgcnStack(torch.autograd.Function):
    """xcat = cat_l relu(GCNConv_l(...)) of kernel/sgcn_img_snp.py:218-224 for a batch of small uniform graphs: one
    LDS-resident kernel per direction (igcn_sgcn_stack_*) instead of gcn_norm + (GEMM, scatter-aggregate) per layer
    + concatenation.  ``wb`` = W_0, b_0, W_1, b_1, ..."""

    @staticmethod
    def forward(ctx, x_in, ew_in, plan, rois, *wb):
        """``rois`` negative: DUAL output — returns (xcat, xcat) as two autograd outputs sharing one buffer, for the two
        consumers of the model (attention query, head inputs); their gradients are added by the backward kernel while it
        stages the rows, instead of by an autograd add in front of it."""
        dual, rois = rois < 0, abs(rois)
        x_in, ew_in = _f32(x_in), _f32(ew_in)
        wb = [_f32(t) for t in wb]
        ws, bs = wb[0::2], wb[1::2]
        n, h0 = x_in.shape
        f, layers = ws[0].shape[0], len(ws)
        emax = plan._stack_dims[1]
        xcat = torch.empty(n, layers * f, dtype=torch.float32, device=x_in.device)
        wp = (ctypes.c_void_p * layers)(*[w.data_ptr() for w in ws])
        bp = (ctypes.c_void_p * layers)(*[b.data_ptr() for b in bs])
        call("igcn_sgcn_stack_fwd", n // rois, rois, emax, h0, f, layers, ptr(x_in), ptr(ew_in), ptr(plan.src32),
             ptr(plan.dst32), ptr(plan.tgt_ptr), ptr(plan.tgt_perm), ptr(plan.loop_edge), wp, bp, ptr(xcat),
             ptr(plan.status), stream_ptr())
        ctx.save_for_backward(x_in, ew_in, *wb)
        ctx.plan, ctx.rois = plan, rois
        ctx.final = _leaves(*wb)
        if dual:
            ctx.set_materialize_grads(False)
            return xcat, xcat.view(n, layers * f)
        return xcat

    @staticmethod
    def backward(ctx, dxcat, dxcat2=None):
        x_in, ew_in, *wb = ctx.saved_tensors
        dx, dew, grads = _sgcn_stack_backward(x_in, ew_in, wb, ctx.plan, ctx.rois, ctx.final, dxcat, dxcat2)
        return (dx, dew, None, None, *grads)


def _sgcn_stack_backward(x_in, ew_in, wb, plan, rois, final, dxcat, dxcat2):
    """igcn_sgcn_stack_bwd on the batched plan ``plan``: (dx_in, dew_in, [dW_0, db_0, dW_1, ...])."""
    ws, bs = wb[0::2], wb[1::2]
    if dxcat is None:
        dxcat, dxcat2 = dxcat2, None
    if dxcat is None:
        dxcat = torch.zeros(x_in.shape[0], len(ws) * ws[0].shape[0], dtype=torch.float32, device=x_in.device)
    dxcat = _f32(dxcat)
    dxcat2 = _f32(dxcat2) if dxcat2 is not None else None
    n, h0 = x_in.shape
    f, layers = ws[0].shape[0], len(ws)
    emax = plan._stack_dims[1]
    g = n // rois
    lib = _lib.load()
    npar = int(lib.igcn_sgcn_stack_param_floats(h0, f, layers))
    dx, dew = torch.empty_like(x_in), torch.empty_like(ew_in)
    dpar = torch.empty(npar, dtype=torch.float32, device=x_in.device)
    scratch = _keep(torch.empty(g * npar, dtype=torch.float32, device=x_in.device))
    wp = (ctypes.c_void_p * layers)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * layers)(*[b.data_ptr() for b in bs])
    with _immediate(final):
        call("igcn_sgcn_stack_bwd", g, rois, emax, h0, f, layers, ptr(x_in), ptr(ew_in), ptr(plan.src32),
             ptr(plan.dst32), ptr(plan.tgt_ptr), ptr(plan.tgt_perm), ptr(plan.src_ptr), ptr(plan.src_perm),
             ptr(plan.loop_edge), wp, bp, ptr(dxcat), ptr(dxcat2), ptr(dx), ptr(dew), ptr(dpar), ptr(scratch),
             ptr(plan.status), stream_ptr())
    grads, off = [], 0
    for l in range(layers):
        fin = h0 if l == 0 else f
        grads.append(dpar[off:off + f * fin].view(f, fin))
        grads.append(dpar[off + f * fin:off + f * fin + f])
        off += f * fin + f
    return dx, dew, grads


def sgcn_front_supported(plan, rois, h0, f, layers, snps_feat, snps_logits):
    """The one-launch front of a train step's image branch (igcn_sgcn_front_fwd) covers this batch: what the LDS-resident
    stack covers, on a plan that is built per graph in LDS (not tiled, not dense blocks), with the SNP mask riding along."""
    if (not sgcn_stack_supported(plan, rois, h0, f, layers) or plan._seg is None or plan._tiled
            or getattr(plan, "dense_blocks", False) or os.environ.get("IGCN_NO_FRONT_FUSED", "0") == "1"
            or os.environ.get("IGCN_NO_FUSED_SGCN", "0") == "1" or os.environ.get("IGCN_PLAN_REPLICATE_LAUNCH", "0") == "1"):
        return False
    if snps_feat is None or snps_feat.dim() != 2 or snps_logits is None or snps_feat.shape[1] != snps_logits.numel():
        return False
    return int(_lib.load().igcn_sgcn_front_lds_bytes(rois, plan._stack_dims[1], h0, f, layers)) <= 150 * 1024


class SgcnFront(torch.autograd.Function):
    """The front of the image branch of a train step — graph plan, cal_probability for the stacked (plain | masked) batch,
    loss_probability, the SNP mask and the GCNConv stack of both passes (kernel/sgcn_img_snp.py:133-181,218-224; train()
    :521-523) — as ONE launch (igcn_sgcn_front_fwd) instead of plan build -> EdgeMaskStacked -> SgcnStack.  Outputs
    (xcat, xcat alias, e, reg partials, stacked SNP mask); backward = the two existing backward launches (igcn_sgcn_stack_bwd
    on the 2-copy plan, igcn_edge_mask_bwd_reg), which walk the plan arrays this launch wrote."""

    @staticmethod
    def forward(ctx, x, prob, prob_bias, ew, plan, rois, snps_logits, reg_hp, snps_feat, edge_index, *wb):
        x, prob, pb, ew = _f32(x), _f32(prob), _f32(prob_bias), _f32(ew)
        wb = [_f32(t) for t in wb]
        ws, bs = wb[0::2], wb[1::2]
        n, h0 = x.shape
        ne = ew.shape[0]
        f, layers = ws[0].shape[0], len(ws)
        dev = x.device
        node_ptr, edge_ptr, _, _ = plan._seg
        g = int(node_ptr.numel()) - 1
        plan.take_pending_build()                       # a deferred build is not needed: this launch fills the plan ...
        plan_g = plan.replicate(2)                      # ... and its 2-copy replica (here: the arrays)
        snps = _f32(snps_logits).reshape(-1)
        feat = _f32(snps_feat)
        ns = snps.numel()
        f32 = dict(dtype=torch.float32, device=dev)
        x_in = torch.empty(2 * n, h0, **f32)
        ew_in = torch.empty(2 * ne, **f32)
        e = torch.empty(ne, **f32)
        regp = torch.empty(g, **f32)
        full = torch.empty(2 * feat.shape[0], ns, **f32)
        xcat = torch.empty(2 * n, layers * f, **f32)
        hp = tuple(float(v) for v in reg_hp)
        names = ("src32", "dst32", "tgt_ptr", "tgt_perm", "src_ptr", "src_perm", "loop_edge")
        p1 = (ctypes.c_void_p * 7)(*[getattr(plan, k).data_ptr() for k in names])
        p2 = (ctypes.c_void_p * 7)(*[getattr(plan_g, k).data_ptr() for k in names])
        wp = (ctypes.c_void_p * layers)(*[w.data_ptr() for w in ws])
        bp = (ctypes.c_void_p * layers)(*[b.data_ptr() for b in bs])
        call("igcn_sgcn_front_fwd", n, ne, g, rois, plan._stack_dims[1], h0, f, layers, ptr(edge_index), ptr(node_ptr),
             ptr(edge_ptr), p1, p2, ptr(plan.status), ptr(x), ptr(prob), ptr(pb), ptr(ew), ptr(x_in), ptr(ew_in), ptr(e),
             ptr(snps), ns, *hp, ptr(regp), ptr(feat), ptr(full), wp, bp, ptr(xcat), stream_ptr())
        ctx.save_for_backward(x, prob, pb, ew, e, snps, feat, x_in, ew_in, *wb)
        ctx.plan, ctx.plan_g, ctx.rois = plan, plan_g, rois
        ctx.reg = (hp, ns, snps_logits.shape, feat.shape[0])
        ctx.final = _leaves(*wb)
        ctx.set_materialize_grads(False)
        return xcat, xcat.view(2 * n, layers * f), e, regp, full

    @staticmethod
    def backward(ctx, dxcat, dxcat2, d_e, d_regp, d_full):
        x, prob, pb, ew, e, snps, feat, x_in, ew_in, *wb = ctx.saved_tensors
        n, ne = x.shape[0], ew.shape[0]
        dx_in, dew_in, grads = _sgcn_stack_backward(x_in, ew_in, wb, ctx.plan_g, ctx.rois, ctx.final, dxcat, dxcat2)
        dx, dprob, dpb, dsnps = _edge_mask_reg_backward(x, prob, pb, ew, e, snps, feat, ctx.reg, ctx.plan, ctx.rois,
                                                        dx_in[n:], dew_in[ne:], d_e, dx_in[:n], d_regp, d_full)
        return (dx, dprob, dpb, None, None, None, dsnps, None, None, None, *grads)


def dense_sgcn_supported(plan, rois, h0, f, layers):
    """The dense-block SGCN kernels (igcn_dense_sgcn_*) cover this batch: complete row-major graphs (verified by the
    plan) of a supported size / width."""
    return bool(getattr(plan, "dense_blocks", False) and plan.nodes_per_graph == rois
                and _lib.load().igcn_dense_sgcn_supported(rois, h0, f, layers))


class DenseSgcn(torch.autograd.Function):
    """cal_probability + gcn_norm + cat_l relu(GCNConv_l(.)) + the edge / node / SNP terms of loss_probability
    (kernel/sgcn_img_snp.py:133-181,218-224) for a batch of COMPLETE graphs (dense adjacency as COO), one or both
    passes of a train step: ``mode`` "plain" (isExplain=False), "masked" (isExplain=True) or "both" (rows [0, N) of
    ``xcat`` the plain pass, [N, 2N) the masked pass).  ``ew`` is read as the dense matrix [G, R, R] it is; no plan
    arrays, no per-edge intermediates (csrc/sgcn_dense.hip).  Returns (xcat, partials of loss_probability — their
    SUM is the loss; empty for "plain").  ``reg`` = (l1_x, ent_x, l1_e, ent_e, eps).  ``status``: the plan's device
    status words (``GraphPlan.status``, int32[2]) or None — when the structure check of the batch (``plan.rebuild``)
    has flagged it, the kernels turn the degrees into NaN, so loss and gradients are NaN instead of silently wrong.
    ``rois`` negative: DUAL output (xcat, regp, xcat again — two autograd handles of one buffer for the model's two
    consumers, attention query and head inputs; their gradients meet where the backward kernels read them)."""

    @staticmethod
    def forward(ctx, x, ew, prob, prob_bias, snps_prob, mode, rois, reg, status, *wb):
        """``status``: the plan's status words, a ``GraphPlan`` (its words + a pending structure check, which then rides
        in the first edge pass of this forward) or None."""
        check_ei = None
        if isinstance(status, GraphPlan):
            check_ei, status = status.take_pending_check(), status.status
        dual, rois = rois < 0, abs(rois)
        x, ew, prob, pb = _f32(x), _f32(ew), _f32(prob), _f32(prob_bias)
        sp = _f32(snps_prob) if snps_prob is not None else None
        wb = [_f32(t) for t in wb]
        ws_, bs_ = wb[0::2], wb[1::2]
        n, h0 = x.shape
        f, layers = ws_[0].shape[0], len(ws_)
        g = n // rois
        copies, first_masked = (2, 0) if mode == "both" else (1, int(mode == "masked"))
        lib = _lib.load()
        dev = x.device
        xcat = torch.empty(copies * n, layers * f, dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib.igcn_dense_sgcn_ws_floats(g, rois, layers, copies)), dtype=torch.float32, device=dev)
        anym = mode != "plain"
        regp = torch.empty(int(lib.igcn_dense_sgcn_reg_blocks(g, rois)) if anym else 0, dtype=torch.float32, device=dev)
        wp = (ctypes.c_void_p * layers)(*[w.data_ptr() for w in ws_])
        bp = (ctypes.c_void_p * layers)(*[b.data_ptr() for b in bs_])
        ctx.reg = tuple(float(v) for v in reg)
        call("igcn_dense_sgcn_fwd", g, rois, h0, f, layers, copies, first_masked, ptr(x), ptr(prob), ptr(pb), ptr(ew),
             wp, bp, ptr(sp), sp.numel() if sp is not None else 0, *ctx.reg, ptr(xcat), ptr(regp) if anym else None,
             ptr(ws), ptr(status) if status is not None and status.numel() >= 2 else None, ptr(check_ei), stream_ptr())
        ctx.save_for_backward(x, ew, prob, pb, sp, xcat, ws, *wb)
        ctx.cfg = (g, rois, h0, f, layers, copies, first_masked, anym)
        ctx.final = _leaves(pb, *wb)
        ctx.set_materialize_grads(False)
        if not anym:
            ctx.mark_non_differentiable(regp)
        if dual:
            return xcat, regp, xcat.view(copies * n, layers * f)
        return xcat, regp

    @staticmethod
    def backward(ctx, dxcat, dregp, dxcat2=None):
        x, ew, prob, pb, sp, xcat, ws, *wb = ctx.saved_tensors
        ws_ = wb[0::2]
        g, rois, h0, f, layers, copies, first_masked, anym = ctx.cfg
        lib = _lib.load()
        dev = x.device
        if dxcat is None:
            dxcat, dxcat2 = dxcat2, None
        dxcat = _f32(dxcat) if dxcat is not None else torch.zeros_like(xcat)
        dxcat2 = _f32(dxcat2) if dxcat2 is not None else None
        d_reg = dregp[:1] if (dregp is not None and anym) else None        # the same scalar in every partial
        if d_reg is not None and not d_reg.is_contiguous():
            d_reg = d_reg.contiguous()
        e32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)        # noqa: E731
        dx = e32(x.shape)
        dprob, dpb = (e32(prob.shape), e32(pb.shape)) if anym else (None, None)
        dsp = e32(sp.shape) if (anym and sp is not None) else None
        npar = int(lib.igcn_sgcn_stack_param_floats(h0, f, layers))
        dpar = e32(npar)
        bws = _keep(e32(int(lib.igcn_dense_sgcn_bwd_ws_floats(g, rois, h0, layers, copies))))
        wp = (ctypes.c_void_p * layers)(*[w.data_ptr() for w in ws_])
        with _immediate(ctx.final):
            call("igcn_dense_sgcn_bwd", g, rois, h0, f, layers, copies, first_masked, ptr(x), ptr(prob), ptr(pb), ptr(ew),
                 wp, ptr(sp), sp.numel() if sp is not None else 0, *ctx.reg, ptr(xcat), ptr(dxcat), ptr(dxcat2), ptr(d_reg),
                 ptr(ws), ptr(bws), ptr(dx), ptr(dprob), ptr(dpb), ptr(dsp), ptr(dpar), stream_ptr())
        grads, off = [], 0
        for l in range(layers):
            fin = h0 if l == 0 else f
            grads.append(dpar[off:off + f * fin].view(f, fin))
            grads.append(dpar[off + f * fin:off + f * fin + f])
            off += f * fin + f
        return (dx, None, dprob, dpb, dsp, None, None, None, None, *grads)


# =================================================================================================
# Dense transforms on MFMA
# =================================================================================================
def _split_k(m, n, k):
    """K-slices for a product with few output tiles: the library's own launch heuristic (gemm.hip)."""
    return int(_lib.load().igcn_gemm_f32_split_k(m, n, k))


def _gemm(bf16):
    return "igcn_gemm_bf16" if bf16 else "igcn_gemm_f32"


def gemm_nt(a, b, bias=None, act=0, out=None, bf16=False):
    """out[M,N] = act(a[M,K] @ b[N,K]^T + bias) on igcn_gemm_f32 (``bf16``: operands rounded to bf16, igcn_gemm_bf16)."""
    a, b = _f32(a), _f32(b)
    m, k = a.shape
    n = b.shape[0]
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    sk = _split_k(m, n, k)
    scratch = torch.empty(sk * m * n, dtype=torch.float32, device=a.device) if sk > 1 else None
    call(_gemm(bf16), m, n, k, ptr(a), k, 1, ptr(b), k, 1, ptr(bias), ptr(out), n, act, sk, ptr(scratch),
         stream_ptr())
    return out


def gemm_nn(a, b, out=None, bf16=False):
    """out[M,N] = a[M,K] @ b[K,N]."""
    a, b = _f32(a), _f32(b)
    m, k = a.shape
    n = b.shape[1]
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    sk = _split_k(m, n, k)
    scratch = torch.empty(sk * m * n, dtype=torch.float32, device=a.device) if sk > 1 else None
    call(_gemm(bf16), m, n, k, ptr(a), k, 1, ptr(b), 1, n, None, ptr(out), n, 0, sk, ptr(scratch), stream_ptr())
    return out


def gemm_tn(a, b, bf16=False, final_grad=False, out=None):
    """out[M,N] = a[K,M]^T @ b[K,N]  (weight gradient: reduction over the long row axis K).  ``final_grad``: the
    output is a parameter gradient — its split-K sum may be deferred (``deferred_reductions``)."""
    a, b = _f32(a), _f32(b)
    k, m = a.shape
    n = b.shape[1]
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    sk = _split_k(m, n, k)
    scratch = torch.empty(sk * m * n, dtype=torch.float32, device=a.device) if sk > 1 else None
    if final_grad:
        _keep(scratch)
    call(_gemm(bf16), m, n, k, ptr(a), 1, m, ptr(b), 1, n, None, ptr(out), n, 0x100 if final_grad else 0, sk,
         ptr(scratch), stream_ptr())
    return out


def gemm_group(specs, bf16=False, ride=False):
    """``ride``: do not launch — queue the products for the current stream (igcn_gemm_rider): the next grouped launch on it
    carries them as further members (``igcn_gemm_rider_flush`` launches them if none came).  Returns (outs, hold): the
    outputs are valid behind the carrying launch, ``hold`` keeps operands and slabs alive until then.

    Several independent products in ONE launch (igcn_gemm_f32_grouped; at most four).  ``specs``: tuples
    (form, a, b, out, bias, final_grad[, act]) with form "nt" (a [M,K], b [N,K]), "nn" (a [M,K], b [K,N]) or "tn"
    (a [K,M], b [K,N]); ``out`` None allocates; act 1 = ReLU ("nt" only).  Returns the outputs.  ``bf16``: operands
    rounded to bf16 on the way into LDS (igcn_gemm_bf16 semantics) for every member."""
    outs = []
    specs = [tuple(sp) + (0,) * (7 - len(sp)) for sp in specs]
    if ride and (len(specs) > 4 or os.environ.get("IGCN_NO_GEMM_GROUPS", "0") == "1"):
        raise _lib.IgcnError("gemm_group(ride=True): at most four products, grouped launches enabled")
    if len(specs) > 4 or os.environ.get("IGCN_NO_GEMM_GROUPS", "0") == "1":
        for form, a, b, out, bias, final, act in specs:
            if form == "nt":
                outs.append(gemm_nt(a, b, bias, act, out=out, bf16=bf16))
            elif form == "nn":
                outs.append(gemm_nn(a, b, out=out, bf16=bf16))
            else:
                outs.append(gemm_tn(a, b, bf16=bf16, final_grad=final, out=out))
        return outs
    table = (ctypes.c_int64 * (16 * len(specs)))()
    hold = []
    for i, (form, a, b, out, bias, final, act) in enumerate(specs):
        a, b = _f32(a), _f32(b)
        if form == "nt":
            (m, k), n = a.shape, b.shape[0]
            st = (k, 1, k, 1)
        elif form == "nn":
            (m, k), n = a.shape, b.shape[1]
            st = (k, 1, 1, n)
        else:
            (k, m), n = a.shape, b.shape[1]
            st = (1, m, 1, n)
        if out is None:
            out = torch.empty(m, n, dtype=torch.float32, device=a.device)
        sk = _split_k(m, n, k)
        scratch = torch.empty(sk * m * n, dtype=torch.float32, device=a.device) if sk > 1 else None
        if final:
            _keep(scratch)
        bias = _f32(bias) if bias is not None else None
        hold += [a, b, out, scratch, bias]
        table[16 * i:16 * i + 15] = [m, n, k, ptr(a) or 0, st[0], st[1], ptr(b) or 0, st[2], st[3], ptr(bias) or 0,
                                     ptr(out) or 0, n, act | (0x100 if final else 0), sk, ptr(scratch) or 0]
        table[16 * i + 15] = 1 if bf16 else 0
        outs.append(out)
    if ride:
        call("igcn_gemm_rider", stream_ptr(), len(specs), ctypes.addressof(table))
        return outs, hold
    call("igcn_gemm_f32_grouped", len(specs), ctypes.addressof(table), stream_ptr())
    return outs


class Linear(torch.autograd.Function):
    """y = act(x W^T + b) with W [out,in] (GCNConv.lin, lin1, lin1_regr, ...).  ``bf16``: the three products of the
    layer (forward, input gradient, weight gradient) take bf16 operands on the matrix cores, fp32 accumulation."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, bf16=False):
        y = gemm_nt(x, weight, _f32(bias) if bias is not None else None, 1 if relu else 0, bf16=bf16)
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.relu, ctx.has_bias, ctx.bf16 = relu, bias is not None, bf16
        ctx.w_final, ctx.b_final = _leaves(weight), _leaves(bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = _f32(dy)
        db = None
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        rows, cols = dy.shape
        if (ctx.relu or need_db) and cols <= 256:
            # ReLU mask and bias gradient in one pass over dy (igcn_bias_grad)
            lib = _lib.load()
            g = torch.empty_like(dy) if ctx.relu else None
            db = torch.empty(cols, dtype=torch.float32, device=dy.device)
            scratch = _keep(torch.empty(int(lib.igcn_bias_grad_scratch_floats(rows, cols)), dtype=torch.float32,
                                        device=dy.device))
            with _immediate(ctx.b_final):
                call("igcn_bias_grad", rows, cols, ptr(dy), ptr(y) if ctx.relu else None, ptr(g), ptr(db),
                     ptr(scratch), stream_ptr())
            if ctx.relu:
                dy = g
            if not need_db:
                db = None
        else:
            if ctx.relu:
                dy = dy * (y > 0)
            db = dy.sum(0) if need_db else None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            # dX = dy W and dW = dy^T x share dy and nothing else: one launch, their small grids side by side
            dx, dw = gemm_group([("nn", dy, weight, None, None, False), ("tn", dy, x, None, None, ctx.w_final)],
                                bf16=ctx.bf16)
        else:
            dx = gemm_nn(dy, weight, bf16=ctx.bf16) if ctx.needs_input_grad[0] else None
            dw = gemm_tn(dy, x, bf16=ctx.bf16, final_grad=ctx.w_final) if ctx.needs_input_grad[1] else None
        return dx, dw, db, None, None


class LinearReluOwed(torch.autograd.Function):
    """y = relu(x W^T + b) whose ReLU backward and bias gradient are taken by the CONSUMER of y: the incoming gradient
    is already that of the pre-activation, and b's gradient is returned by the consumer (HeadInputs with ``cross_bias``,
    which reads y anyway — kernel/sgcn_img_snp.py:241-242 then :284).  Only valid when y has that one consumer."""

    @staticmethod
    def forward(ctx, x, weight, bias, bf16=False):
        y = gemm_nt(x, weight, _f32(bias), 1, bf16=bf16)
        ctx.save_for_backward(x, weight)
        ctx.bf16 = bf16
        ctx.w_final = _leaves(weight)
        return y

    @staticmethod
    def backward(ctx, dz):
        x, weight = ctx.saved_tensors
        dz = _f32(dz)
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            dx, dw = gemm_group([("nn", dz, weight, None, None, False), ("tn", dz, x, None, None, ctx.w_final)],
                                bf16=ctx.bf16)
        else:
            dx = gemm_nn(dz, weight, bf16=ctx.bf16) if ctx.needs_input_grad[0] else None
            dw = gemm_tn(dz, x, bf16=ctx.bf16, final_grad=ctx.w_final) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None


def _bias_grad_into(dy, db, final):
    """db[cols] = column sums of dy (igcn_bias_grad without a ReLU mask), written into ``db`` (may be a slice)."""
    lib = _lib.load()
    rows, cols = dy.shape
    scratch = _keep(torch.empty(int(lib.igcn_bias_grad_scratch_floats(rows, cols)), dtype=torch.float32,
                                device=dy.device))
    with _immediate(final):
        call("igcn_bias_grad", rows, cols, ptr(dy), None, None, ptr(db), ptr(scratch), stream_ptr())


class LinearPair(torch.autograd.Function):
    """Two independent linear layers y_i = act(x_i W_i^T + b_i) as one op — the two heads' first layers (lin1 on
    [out_z | latent], lin1_regr on the regression features; kernel/sgcn_img_snp.py:299-304): their forward products
    share one grouped launch and their backward's four products (dX_1, dW_1, dX_2, dW_2) another, instead of running
    as six half-empty grids one after the other."""

    @staticmethod
    def forward(ctx, x1, w1, b1, x2, w2, b2, relu, bf16=False):
        x1, x2 = _f32(x1), _f32(x2)
        y1, y2 = gemm_group([("nt", x1, w1, None, b1, False, 1 if relu else 0),
                             ("nt", x2, w2, None, b2, False, 1 if relu else 0)], bf16=bf16)
        ctx.save_for_backward(x1, w1, x2, w2, y1 if relu else None, y2 if relu else None)
        ctx.relu, ctx.bf16 = relu, bf16
        ctx.final = (_leaves(w1), _leaves(b1), _leaves(w2), _leaves(b2))
        return y1, y2

    @staticmethod
    def backward(ctx, dy1, dy2):
        x1, w1, x2, w2, y1, y2 = ctx.saved_tensors
        lib = _lib.load()
        dy1, dy2 = _f32(dy1), _f32(dy2)
        rows, cols = dy1.shape
        if (not ctx.bf16 and dy1.is_cuda and dy2.shape == dy1.shape and os.environ.get("IGCN_NO_HEAD_FUSED", "0") != "1"
                and lib.igcn_head_bwd_supported(rows, cols, x1.shape[1]) and lib.igcn_head_bwd_supported(rows, cols, x2.shape[1])
                and all(t.is_contiguous() for t in (dy1, dy2, x1, x2, w1, w2))):
            # ReLU mask, bias gradients and the four products in ONE launch, the wide operands read once (igcn_head_bwd_pair)
            f32 = dict(dtype=torch.float32, device=dy1.device)
            dx1, dx2, dw1, dw2 = torch.empty_like(x1), torch.empty_like(x2), torch.empty_like(w1), torch.empty_like(w2)
            db1, db2 = torch.empty(cols, **f32), torch.empty(cols, **f32)
            s1 = _keep(torch.empty(int(lib.igcn_head_bwd_scratch_floats(rows, x1.shape[1])), **f32))
            s2 = _keep(torch.empty(int(lib.igcn_head_bwd_scratch_floats(rows, x2.shape[1])), **f32))
            with _immediate(all(ctx.final)):
                call("igcn_head_bwd_pair", rows, cols, x1.shape[1], ptr(dy1), ptr(y1) if ctx.relu else None, ptr(w1), ptr(x1),
                     ptr(dx1), ptr(dw1), ptr(db1), ptr(s1), x2.shape[1], ptr(dy2), ptr(y2) if ctx.relu else None, ptr(w2),
                     ptr(x2), ptr(dx2), ptr(dw2), ptr(db2), ptr(s2), stream_ptr())
            return dx1, dw1, db1, dx2, dw2, db2, None, None
        if dy2.shape == dy1.shape:                                          # both heads: ONE mask + bias-gradient launch
            gs = [torch.empty_like(dy1), torch.empty_like(dy2)] if ctx.relu else [dy1, dy2]
            dbs = [torch.empty(cols, dtype=torch.float32, device=dy1.device) for _ in range(2)]
            nscr = int(lib.igcn_bias_grad_scratch_floats(rows, cols))
            scr = [_keep(torch.empty(nscr, dtype=torch.float32, device=dy1.device)) for _ in range(2)]
            with _immediate(ctx.final[1] and ctx.final[3]):
                call("igcn_bias_grad_pair", rows, cols,
                     ptr(dy1), ptr(y1) if ctx.relu else None, ptr(gs[0]) if ctx.relu else None, ptr(dbs[0]), 0, ptr(scr[0]),
                     ptr(dy2), ptr(y2) if ctx.relu else None, ptr(gs[1]) if ctx.relu else None, ptr(dbs[1]), 0, ptr(scr[1]),
                     stream_ptr())
        else:
            gs, dbs = [], []
            for dy, y, b_final in ((dy1, y1, ctx.final[1]), (dy2, y2, ctx.final[3])):
                rows, cols = dy.shape
                g = torch.empty_like(dy) if ctx.relu else None
                db = torch.empty(cols, dtype=torch.float32, device=dy.device)
                scratch = _keep(torch.empty(int(lib.igcn_bias_grad_scratch_floats(rows, cols)), dtype=torch.float32,
                                            device=dy.device))
                with _immediate(b_final):                                   # ReLU mask + bias gradient in one pass
                    call("igcn_bias_grad", rows, cols, ptr(dy), ptr(y) if ctx.relu else None, ptr(g), ptr(db),
                         ptr(scratch), stream_ptr())
                gs.append(g if ctx.relu else dy)
                dbs.append(db)
        with _immediate(ctx.final[0] and ctx.final[2]):
            dx1, dw1, dx2, dw2 = gemm_group([("nn", gs[0], w1, None, None, False), ("tn", gs[0], x1, None, None, True),
                                             ("nn", gs[1], w2, None, None, False), ("tn", gs[1], x2, None, None, True)],
                                            bf16=ctx.bf16)
        return dx1, dw1, dbs[0], dx2, dw2, dbs[1], None, None


def linear_pair(x1, w1, b1, x2, w2, b2, relu=True, bf16=False):
    """(act(x1 W1^T + b1), act(x2 W2^T + b2)) with grouped launches; separate ops.linear calls when the shapes fall
    outside the grouped kernel (bias-less layers, wide outputs)."""
    if b1 is None or b2 is None or w1.shape[0] > 256 or w2.shape[0] > 256 or x1.dim() != 2 or x2.dim() != 2:
        return linear(x1, w1, b1, relu=relu, bf16=bf16), linear(x2, w2, b2, relu=relu, bf16=bf16)
    return LinearPair.apply(x1, w1, b1, x2, w2, b2, relu, bf16)


def _proj_forward(q2, m2, w, bias, d, bf16):
    """q = q2 W_q^T + b_q, k | v = m2 [W_k; W_v]^T + [b_k; b_v] of the packed input projection: one streaming launch
    (igcn_proj_fwd_pair: reduction depth 32, W resident in LDS) when the shape allows it, the grouped GEMM otherwise."""
    lib = _lib.load()
    if (not bf16 and q2.is_cuda and os.environ.get("IGCN_NO_PROJ_FUSED", "0") != "1" and q2.is_contiguous()
            and m2.is_contiguous() and w.is_contiguous() and bias.is_contiguous() and q2.shape[0] > 0 and m2.shape[0] > 0
            and lib.igcn_proj_bwd_supported(q2.shape[0], d, d) and lib.igcn_proj_bwd_supported(m2.shape[0], 2 * d, d)):
        q = torch.empty(q2.shape[0], d, dtype=torch.float32, device=q2.device)
        kv = torch.empty(m2.shape[0], 2 * d, dtype=torch.float32, device=q2.device)
        call("igcn_proj_fwd_pair", q2.shape[0], d, ptr(q2), ptr(w[:d]), ptr(bias[:d]), ptr(q), m2.shape[0], 2 * d, ptr(m2),
             ptr(w[d:]), ptr(bias[d:]), ptr(kv), d, stream_ptr())
        return q, kv
    return gemm_group([("nt", q2, w[:d], None, bias[:d], False), ("nt", m2, w[d:], None, bias[d:], False)], bf16=bf16)


def _proj_fused(ctx, dq, dkv, d):
    """The one-pass kernels (igcn_proj_bwd_pair*) cover this backward."""
    lib = _lib.load()
    return bool(ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not ctx.bf16 and dq.is_cuda
                and os.environ.get("IGCN_NO_PROJ_FUSED", "0") != "1"
                and lib.igcn_proj_bwd_supported(dq.shape[0], d, d) and lib.igcn_proj_bwd_supported(dkv.shape[0], 2 * d, d))


def _proj_backward(ctx, dq, dkv, q2, m2, w, dw, d, db=None):
    """The four products behind the packed input projection's backward — input gradients of query and memory, weight
    gradients of the two blocks (written into ``dw``) — in one launch when all four are wanted.  ``db`` [3d] (only with
    ``_proj_fused``): the bias gradients from the same pass — column sums of dq, exact zeros for the key bias (a
    softmax over keys cannot see it), column sums of the value gradient (= sum of the attention's incoming gradient:
    every query's weights sum to one; the reference's autograd takes this very sum)."""
    lib = _lib.load()
    if db is not None:
        dquery, dmem = torch.empty_like(q2), torch.empty_like(m2)
        f32 = dict(dtype=torch.float32, device=dq.device)
        s1 = _keep(torch.empty(int(lib.igcn_proj_bwd_scratch_floats(dq.shape[0], d)), **f32))
        s2 = _keep(torch.empty(int(lib.igcn_proj_bwd_scratch_floats(dkv.shape[0], 2 * d)), **f32))
        with _immediate(ctx.final):
            call("igcn_proj_bwd_pair_bias", dq.shape[0], d, ptr(dq), ptr(q2), ptr(w[:d]), ptr(dquery), ptr(dw[:d]), ptr(s1),
                 ptr(db[:d]), 0, dkv.shape[0], 2 * d, ptr(dkv), ptr(m2), ptr(w[d:]), ptr(dmem), ptr(dw[d:]), ptr(s2),
                 ptr(db[d:]), d, d, stream_ptr())
        return dquery.view(ctx.shapes[0]), dmem.view(ctx.shapes[1])
    if (ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not ctx.bf16 and dq.is_cuda
            and os.environ.get("IGCN_NO_PROJ_FUSED", "0") != "1"
            and lib.igcn_proj_bwd_supported(dq.shape[0], d, d) and lib.igcn_proj_bwd_supported(dkv.shape[0], 2 * d, d)):
        # input gradient AND weight gradient of a block from ONE pass over its incoming gradient (igcn_proj_bwd): the
        # key | value gradient (52 MB at the bench shape) is read once, not twice
        dquery, dmem = torch.empty_like(q2), torch.empty_like(m2)
        f32 = dict(dtype=torch.float32, device=dq.device)
        s1 = _keep(torch.empty(int(lib.igcn_proj_bwd_scratch_floats(dq.shape[0], d)), **f32))
        s2 = _keep(torch.empty(int(lib.igcn_proj_bwd_scratch_floats(dkv.shape[0], 2 * d)), **f32))
        with _immediate(ctx.final):                 # both blocks in one launch: the small one rides along
            call("igcn_proj_bwd_pair", dq.shape[0], d, ptr(dq), ptr(q2), ptr(w[:d]), ptr(dquery), ptr(dw[:d]), ptr(s1),
                 dkv.shape[0], 2 * d, ptr(dkv), ptr(m2), ptr(w[d:]), ptr(dmem), ptr(dw[d:]), ptr(s2), d, stream_ptr())
        return dquery.view(ctx.shapes[0]), dmem.view(ctx.shapes[1])
    if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
        with _immediate(ctx.final):
            dquery, dmem, _, _ = gemm_group([("nn", dq, w[:d], None, None, False), ("nn", dkv, w[d:], None, None, False),
                                             ("tn", dq, q2, dw[:d], None, True), ("tn", dkv, m2, dw[d:], None, True)],
                                            bf16=ctx.bf16)
        return dquery.view(ctx.shapes[0]), dmem.view(ctx.shapes[1])
    dquery = gemm_nn(dq, w[:d], bf16=ctx.bf16).view(ctx.shapes[0]) if ctx.needs_input_grad[0] else None
    dmem = gemm_nn(dkv, w[d:], bf16=ctx.bf16).view(ctx.shapes[1]) if ctx.needs_input_grad[1] else None
    with _immediate(ctx.final):
        gemm_tn(dq, q2, bf16=ctx.bf16, final_grad=True, out=dw[:d])
        gemm_tn(dkv, m2, bf16=ctx.bf16, final_grad=True, out=dw[d:])
    return dquery, dmem


class InProj(torch.autograd.Function):
    """The packed input projection of nn.MultiheadAttention for cross-attention (kernel/sgcn_img_snp.py:240-241:
    ``multihead_attn(query, memory, memory)``): q = query W_q^T + b_q, kv = memory [W_k; W_v]^T + [b_k; b_v] with
    W = in_proj_weight [3D, D] and bias = in_proj_bias [3D] taken WHOLE.  Two GEMMs forward; the backward writes the
    two weight-gradient blocks and the two bias-gradient blocks straight into ONE [3D, D] / [3D] gradient, so the
    parameters stay autograd leaves of this op: no concatenation of per-slice gradients (what ``in_proj_weight.split``
    costs in its backward), and their split-K / block-partial sums are final and can wait for the deferred flush."""

    @staticmethod
    def forward(ctx, query, memory, w, bias, bf16=False):
        d = w.shape[1]
        q2, m2 = _f32(query).reshape(-1, d), _f32(memory).reshape(-1, d)
        w, bias = _f32(w), _f32(bias)
        q, kv = _proj_forward(q2, m2, w, bias, d, bf16)
        ctx.save_for_backward(q2, m2, w)
        ctx.bf16, ctx.final = bf16, _leaves(w, bias)
        ctx.shapes = (query.shape, memory.shape)
        return q.view(*query.shape[:-1], d), kv.view(*memory.shape[:-1], 2 * d)

    @staticmethod
    def backward(ctx, dq, dkv):
        q2, m2, w = ctx.saved_tensors
        d = w.shape[1]
        dq, dkv = _f32(dq).reshape(-1, d), _f32(dkv).reshape(-1, 2 * d)
        dw = torch.empty_like(w)
        db = torch.empty(3 * d, dtype=torch.float32, device=w.device)
        _bias_grad_into(dq, db[:d], ctx.final)
        _bias_grad_into(dkv, db[d:], ctx.final)
        dquery, dmem = _proj_backward(ctx, dq, dkv, q2, m2, w, dw, d)
        return dquery, dmem, dw, db, None


class ConcatCols(torch.autograd.Function):
    """torch.cat(parts, dim=1) for up to four [N, F] tensors of equal width (F % 4 == 0): the jumping-knowledge
    concatenation of the GCN layer outputs, moved 16 bytes per lane (igcn_concat_cols)."""

    @staticmethod
    def forward(ctx, *parts):
        import ctypes
        parts = [_f32(p) for p in parts]
        n, f = parts[0].shape
        out = torch.empty(n, len(parts) * f, dtype=torch.float32, device=parts[0].device)
        arr = (ctypes.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
        call("igcn_concat_cols", n, f, len(parts), arr, ptr(out), stream_ptr())
        ctx.f, ctx.k = f, len(parts)
        return out

    @staticmethod
    def backward(ctx, d):
        return tuple(d[:, i * ctx.f:(i + 1) * ctx.f] for i in range(ctx.k))


def concat_cols(parts):
    parts = list(parts)
    ok = 1 <= len(parts) <= 4 and all(p.dim() == 2 and p.shape == parts[0].shape and p.is_cuda and
                                      p.dtype == torch.float32 for p in parts) and parts[0].shape[1] % 4 == 0
    if len(parts) == 1:
        return parts[0]
    return ConcatCols.apply(*parts) if ok else torch.cat(parts, dim=1)


class GraphPool(torch.autograd.Function):
    """global_mean_pool | global_max_pool | global_add_pool concatenated (kernel/sgcn_img_snp.py:230-235,246-252)
    for a batch of uniform graphs: x [G*R, D] -> [G, 3D] (igcn_graph_pool_*)."""

    @staticmethod
    def forward(ctx, x, nodes_per_graph):
        x = _f32(x)
        n, d = x.shape
        r = int(nodes_per_graph)
        if r <= 0 or n % r:
            raise _lib.IgcnError(f"graph pool: {n} nodes do not split into graphs of {r}")
        g = n // r
        out = torch.empty(g, 3 * d, dtype=torch.float32, device=x.device)
        arg = torch.empty(g, d, dtype=torch.int32, device=x.device)
        call("igcn_graph_pool_fwd", g, r, d, ptr(x), ptr(out), ptr(arg), stream_ptr())
        ctx.save_for_backward(arg)
        ctx.dims = (g, r, d)
        return out

    @staticmethod
    def backward(ctx, dout):
        arg, = ctx.saved_tensors
        g, r, d = ctx.dims
        dout = _f32(dout)
        dx = torch.empty(g * r, d, dtype=torch.float32, device=dout.device)
        call("igcn_graph_pool_bwd", g, r, d, ptr(dout), ptr(arg), ptr(dx), stream_ptr())
        return dx, None


class SnpsMask(torch.autograd.Function):
    """(snps * sigmoid(p), sigmoid(p)) of cal_probability (kernel/sgcn_img_snp.py:147-151); p [1,S] or [S]."""

    @staticmethod
    def forward(ctx, snps, p, stacked=False):
        """``stacked``: the first output is cat(snps, snps * sigmoid(p)) [2B,S] (plain | masked pass)."""
        snps, p = _f32(snps), _f32(p)
        b, s = snps.shape
        sp = torch.empty_like(p)
        if stacked:
            full = torch.empty(2 * b, s, dtype=torch.float32, device=snps.device)
            out = full[b:]
        else:
            full = out = torch.empty_like(snps)
        call("igcn_snps_mask_fwd", b, s, ptr(snps), ptr(p), ptr(out), ptr(sp), ptr(full[:b]) if stacked else None,
             stream_ptr())
        ctx.save_for_backward(snps, p)
        ctx.stacked = stacked
        ctx.set_materialize_grads(False)
        return full, sp

    @staticmethod
    def backward(ctx, dout, dsp):
        snps, p = ctx.saved_tensors
        if dout is None and dsp is None:
            return None, None, None
        b, s = snps.shape
        if dout is not None and ctx.stacked:
            dout = dout[b:]
        dp = torch.empty_like(p)
        call("igcn_snps_mask_bwd", b, s, ptr(snps), ptr(p), ptr(_f32(dout)) if dout is not None else None,
             ptr(_f32(dsp)) if dsp is not None else None, ptr(dp), stream_ptr())
        return None, dp, None


class HeadInputs(torch.autograd.Function):
    """(out_z, out_lin, feat) of kernel/sgcn_img_snp.py:284-297 from (img_out, out_cross, latent, x, prob) in one
    launch (igcn_head_inputs_*): out_z = (img + cross)/2, out_lin = out_z | latent, feat = out_lin | x * prob of the
    row's sample.  ``prob`` None: no regression features, ``feat`` is ``out_lin``."""

    @staticmethod
    def forward(ctx, img, cross, latent, x, prob, bsz, cross_bias=None):
        img, cross, latent = _f32(img), _f32(cross), _f32(latent)
        r, w = img.shape
        l = latent.shape[1]
        p = 0
        if prob is not None:
            x, prob = _f32(x), _f32(prob)
            p = prob.numel()
        dev = img.device
        out_z = torch.empty(r, w, dtype=torch.float32, device=dev)
        out_lin = torch.empty(r, w + l, dtype=torch.float32, device=dev)
        feat = torch.empty(r, w + l + p, dtype=torch.float32, device=dev) if p else None
        call("igcn_head_inputs_fwd", r, bsz, w, l, p, ptr(img), ptr(cross), ptr(latent), ptr(x) if p else None,
             ptr(prob) if p else None, ptr(out_z), ptr(out_lin), ptr(feat), stream_ptr())
        # cross_bias [D]: ``cross`` is the output of a LinearReluOwed layer with that bias — the backward also takes that
        # layer's ReLU mask and bias gradient (igcn_head_inputs_bwd_relu)
        ctx.owed = cross_bias is not None
        ctx.save_for_backward(x if p else None, prob if p else None, cross if ctx.owed else None)
        ctx.b_final = _leaves(cross_bias)
        ctx.feat_dim = cross_bias.numel() if ctx.owed else 0
        ctx.dims = (r, bsz, w, l, p)
        ctx.x_shape = x.shape if p else None
        ctx.set_materialize_grads(False)
        if p:
            return out_z, out_lin, feat
        none = out_lin.new_empty(0)
        ctx.mark_non_differentiable(none)
        return out_z, out_lin, none

    @staticmethod
    def backward(ctx, d_out_z, d_out_lin, d_feat):
        x, prob, cross = ctx.saved_tensors
        r, bsz, w, l, p = ctx.dims
        dev = d_out_z.device if d_out_z is not None else (d_out_lin.device if d_out_lin is not None else d_feat.device)
        gz = _f32(d_out_z) if d_out_z is not None else None
        gl = _f32(d_out_lin) if d_out_lin is not None else None
        gf = _f32(d_feat) if (d_feat is not None and p) else None
        d_mid = torch.empty(r, w, dtype=torch.float32, device=dev)
        d_latent = torch.empty(r, l, dtype=torch.float32, device=dev)
        dx = torch.empty(ctx.x_shape, dtype=torch.float32, device=dev) if p else None
        dprob = torch.empty(prob.shape, dtype=torch.float32, device=dev) if p else None
        if ctx.owed:
            d = ctx.feat_dim
            d_cross = torch.empty(r, w, dtype=torch.float32, device=dev)
            db = torch.empty(d, dtype=torch.float32, device=dev)
            part = _keep(torch.empty(int(_lib.load().igcn_head_inputs_bwd_blocks(r, w, l)), d, dtype=torch.float32,
                                     device=dev))
            with _immediate(ctx.b_final):
                call("igcn_head_inputs_bwd_relu", r, bsz, w, l, p, ptr(gz), ptr(gl), ptr(gf), ptr(x), ptr(prob),
                     ptr(d_mid), ptr(d_latent), ptr(dx), ptr(dprob), ptr(cross), ptr(d_cross), d, ptr(part), ptr(db),
                     stream_ptr())
            return d_mid, d_cross, d_latent, dx, dprob, None, db
        call("igcn_head_inputs_bwd", r, bsz, w, l, p, ptr(gz), ptr(gl), ptr(gf), ptr(x), ptr(prob), ptr(d_mid),
             ptr(d_latent), ptr(dx), ptr(dprob), stream_ptr())
        return d_mid, d_mid, d_latent, dx, dprob, None, None


class OutProjHeadInputs(torch.autograd.Function):
    """HeadInputs with the layer in front computed on the way (igcn_outproj_head_inputs_fwd): cross = relu(out_proj(o)) per
    graph node (kernel/sgcn_img_snp.py:241-242), then (out_z, out_lin, feat) of :284-297 — one launch instead of a GEMM and
    an elementwise pass.  Backward: igcn_head_inputs_bwd_relu (ReLU mask, bias gradient, the sums), then the layer's two
    products as one grouped launch."""

    @staticmethod
    def forward(ctx, o, weight, bias, img, latent, x, prob, bsz, bf16=False):
        o, weight, bias = _f32(o), _f32(weight), _f32(bias)
        img, latent = _f32(img), _f32(latent)
        r, w = img.shape
        l, d = latent.shape[1], weight.shape[0]
        p = 0
        if prob is not None:
            x, prob = _f32(x), _f32(prob)
            p = prob.numel()
        dev = img.device
        cross = torch.empty(r, w, dtype=torch.float32, device=dev)
        out_z = torch.empty(r, w, dtype=torch.float32, device=dev)
        out_lin = torch.empty(r, w + l, dtype=torch.float32, device=dev)
        feat = torch.empty(r, w + l + p, dtype=torch.float32, device=dev) if p else None
        call("igcn_outproj_head_inputs_fwd", r, bsz, w, l, p, d, ptr(o), ptr(weight), ptr(bias), ptr(img), ptr(latent),
             ptr(x) if p else None, ptr(prob) if p else None, ptr(cross), ptr(out_z), ptr(out_lin), ptr(feat), stream_ptr())
        ctx.save_for_backward(x if p else None, prob if p else None, cross, o, weight)
        ctx.dims = (r, bsz, w, l, p, d)
        ctx.x_shape = x.shape if p else None
        ctx.o_shape = o.shape
        ctx.bf16 = bf16
        ctx.w_final, ctx.b_final = _leaves(weight), _leaves(bias)
        ctx.set_materialize_grads(False)
        # cross is handed out as well, for inspection only (tests read the layer's ReLU decisions off it)
        if p:
            ctx.mark_non_differentiable(cross)
            return out_z, out_lin, feat, cross
        none = out_lin.new_empty(0)
        ctx.mark_non_differentiable(none, cross)
        return out_z, out_lin, none, cross

    @staticmethod
    def backward(ctx, d_out_z, d_out_lin, d_feat, _d_cross):
        x, prob, cross, o, weight = ctx.saved_tensors
        r, bsz, w, l, p, d = ctx.dims
        dev = cross.device
        gz = _f32(d_out_z) if d_out_z is not None else None
        gl = _f32(d_out_lin) if d_out_lin is not None else None
        gf = _f32(d_feat) if (d_feat is not None and p) else None
        d_mid = torch.empty(r, w, dtype=torch.float32, device=dev)
        d_latent = torch.empty(r, l, dtype=torch.float32, device=dev)
        dx = torch.empty(ctx.x_shape, dtype=torch.float32, device=dev) if p else None
        dprob = torch.empty(prob.shape, dtype=torch.float32, device=dev) if p else None
        d_cross = torch.empty(r, w, dtype=torch.float32, device=dev)
        db = torch.empty(d, dtype=torch.float32, device=dev)
        part = _keep(torch.empty(int(_lib.load().igcn_head_inputs_bwd_blocks(r, w, l)), d, dtype=torch.float32, device=dev))
        with _immediate(ctx.b_final):
            call("igcn_head_inputs_bwd_relu", r, bsz, w, l, p, ptr(gz), ptr(gl), ptr(gf), ptr(x), ptr(prob), ptr(d_mid),
                 ptr(d_latent), ptr(dx), ptr(dprob), ptr(cross), ptr(d_cross), d, ptr(part), ptr(db), stream_ptr())
        dz, o2 = d_cross.view(-1, d), o.reshape(-1, d)
        do, dw = gemm_group([("nn", dz, weight, None, None, False), ("tn", dz, o2, None, None, ctx.w_final)], bf16=ctx.bf16)
        return do.view(ctx.o_shape), dw, db, d_mid, d_latent, dx, dprob, None, None


def outproj_head_inputs_supported(d, width):
    return relu_owed_supported(d, width) and 512 % d == 0 and os.environ.get("IGCN_NO_OUTPROJ_FUSED", "0") != "1"


def head_inputs_supported(img, cross, latent, x, prob):
    ts = [img, cross, latent] + ([x, prob] if prob is not None else [])
    if not all(t.is_cuda and t.dtype == torch.float32 for t in ts):
        return False
    w, l = img.shape[1], latent.shape[1]
    p = prob.numel() if prob is not None else 0
    return img.shape == cross.shape and w % 2 == 0 and l % 2 == 0 and p % 2 == 0


def relu_owed_supported(d, width):
    """HeadInputs can take the ReLU backward and bias gradient of a D-feature layer whose output is its ``cross``
    operand (igcn_head_inputs_bwd_relu): D a power of two in [2, 64] dividing the row width."""
    return os.environ.get("IGCN_NO_RELU_OWED", "0") != "1" and 2 <= d <= 64 and (d & (d - 1)) == 0 and width % d == 0


class SmallLinear(torch.autograd.Function):
    """y = (x * keep) W^T + b for C <= 4 outputs (lin2 / lin2_regr: 64 -> 3) as one VALU kernel per direction
    (igcn_small_linear_*) instead of a GEMM forward and five launches backward.  ``keep`` (same shape as x, or None):
    dropout factors of the input, applied inside the kernels (F.dropout of kernel/sgcn_img_snp.py:289,299)."""

    @staticmethod
    def forward(ctx, x, weight, bias, keep=None):
        x, weight = _f32(x), _f32(weight)
        bias = _f32(bias) if bias is not None else None
        keep = _f32(keep) if keep is not None else None
        r, k = x.shape
        c = weight.shape[0]
        y = torch.empty(r, c, dtype=torch.float32, device=x.device)
        call("igcn_small_linear_fwd", r, k, c, ptr(x), ptr(keep), ptr(weight), ptr(bias), ptr(y), stream_ptr())
        ctx.save_for_backward(x, weight, keep)
        ctx.has_bias = bias is not None
        ctx.final = _leaves(weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, keep = ctx.saved_tensors
        dy = _f32(dy)
        r, k = x.shape
        c = weight.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dwb = torch.empty(c * k + c, dtype=torch.float32, device=x.device)
        scratch = _keep(torch.empty(int(_lib.load().igcn_small_linear_bwd_scratch_floats(r, k, c)),
                                    dtype=torch.float32, device=x.device))
        with _immediate(ctx.final):
            call("igcn_small_linear_bwd", r, k, c, ptr(x), ptr(keep), ptr(weight), ptr(dy), ptr(dx), ptr(dwb),
                 ptr(scratch), stream_ptr())
        return dx, dwb[:c * k].view(c, k), (dwb[c * k:] if ctx.has_bias else None), None


class SmallLinearPair(torch.autograd.Function):
    """Two SmallLinear layers over inputs of the same shape — lin2 on the classifier features and lin2_regr on the
    regression features — as one launch per direction (igcn_small_linear_pair_*): the two are independent and each is a
    few workgroups' worth of work."""

    @staticmethod
    def forward(ctx, x1, w1, b1, keep1, x2, w2, b2, keep2):
        f = lambda t: _f32(t) if t is not None else None               # noqa: E731
        x1, w1, b1, keep1, x2, w2, b2, keep2 = (f(t) for t in (x1, w1, b1, keep1, x2, w2, b2, keep2))
        r, k = x1.shape
        c1, c2 = w1.shape[0], w2.shape[0]
        y1 = torch.empty(r, c1, dtype=torch.float32, device=x1.device)
        y2 = torch.empty(r, c2, dtype=torch.float32, device=x1.device)
        call("igcn_small_linear_pair_fwd", r, k, c1, ptr(x1), ptr(keep1), ptr(w1), ptr(b1), ptr(y1),
             c2, ptr(x2), ptr(keep2), ptr(w2), ptr(b2), ptr(y2), stream_ptr())
        ctx.save_for_backward(x1, w1, keep1, x2, w2, keep2)
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.final = _leaves(w1, b1, w2, b2)
        return y1, y2

    @staticmethod
    def backward(ctx, dy1, dy2):
        x1, w1, keep1, x2, w2, keep2 = ctx.saved_tensors
        dy1, dy2 = _f32(dy1), _f32(dy2)
        r, k = x1.shape
        c1, c2 = w1.shape[0], w2.shape[0]
        lib = _lib.load()
        dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
        dwb1 = torch.empty(c1 * k + c1, dtype=torch.float32, device=x1.device)
        dwb2 = torch.empty(c2 * k + c2, dtype=torch.float32, device=x1.device)
        s1 = _keep(torch.empty(int(lib.igcn_small_linear_bwd_scratch_floats(r, k, c1)), dtype=torch.float32, device=x1.device))
        s2 = _keep(torch.empty(int(lib.igcn_small_linear_bwd_scratch_floats(r, k, c2)), dtype=torch.float32, device=x1.device))
        with _immediate(ctx.final):
            call("igcn_small_linear_pair_bwd", r, k, c1, ptr(x1), ptr(keep1), ptr(w1), ptr(dy1), ptr(dx1), ptr(dwb1),
                 ptr(s1), c2, ptr(x2), ptr(keep2), ptr(w2), ptr(dy2), ptr(dx2), ptr(dwb2), ptr(s2), stream_ptr())
        return (dx1, dwb1[:c1 * k].view(c1, k), dwb1[c1 * k:] if ctx.has_bias[0] else None, None,
                dx2, dwb2[:c2 * k].view(c2, k), dwb2[c2 * k:] if ctx.has_bias[1] else None, None)


def small_linear_pair(x1, w1, b1, keep1, x2, w2, b2, keep2):
    """(linear(x1, w1, b1, keep=keep1), linear(x2, w2, b2, keep=keep2)) for two narrow layers over inputs of the same
    2-D shape in one launch each way; two ops.linear calls otherwise."""
    if (x1.dim() == 2 and x1.shape == x2.shape and _small_linear_ok(x1, w1, False) and _small_linear_ok(x2, w2, False)):
        return SmallLinearPair.apply(x1, w1, b1, keep1, x2, w2, b2, keep2)
    return linear(x1, w1, b1, keep=keep1), linear(x2, w2, b2, keep=keep2)


def _small_linear_ok(x2, weight, relu):
    k, c = x2.shape[1], weight.shape[0]
    kq = k // 4
    return (not relu and x2.is_cuda and c <= 4 and k % 4 == 0 and 1 <= kq <= 64 and (kq & (kq - 1)) == 0
            and x2.shape[0] > 0)


def linear(x, weight, bias=None, relu=False, bf16=False, keep=None):
    """``keep``: dropout factors of the INPUT (same shape as x) — fused into the narrow-output kernel, a plain
    multiply in front of the GEMM otherwise."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if _small_linear_ok(x2, weight, relu) and not bf16:
        y = SmallLinear.apply(x2, weight, bias, keep.reshape(x2.shape) if keep is not None else None)
    else:
        if keep is not None:
            x2 = x2 * keep.reshape(x2.shape)
        y = Linear.apply(x2, weight, bias, relu, bf16)
    return y.view(*lead, weight.shape[0])


# =================================================================================================
# Dropout masks of a forward pass (one launch)
# =================================================================================================
class DropoutState:
    """Device-resident counter of the mask generator (igcn_dropout_masks): seeded from torch's generator, advanced by
    the kernel itself, so the launch is capturable and every replay draws fresh masks."""

    def __init__(self, device):
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        # data-parallel ranks usually share torch.manual_seed: mix the rank in, so that replicas draw different masks
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) \
            else int(os.environ.get("RANK", "0"))
        seed = (seed ^ (rank * 0x9E3779B97F4A7C15)) & (2 ** 62 - 1)
        words = int(_lib.load().igcn_dropout_state_words())
        self.state = torch.zeros(words, dtype=torch.int64, device=device)
        self.state[0] = seed


def dropout_masks(sites, state, counters=(), inc=0, ride=False):
    """``sites``: [(shape, p), ...] -> list of float tensors of {0, 1/(1-p)} factors, drawn by ONE kernel launch.
    ``counters`` (<= 8 int64 device scalars): bumped by ``inc`` by the same launch (BatchNorm's num_batches_tracked).
    ``ride``: do not launch — queue the job for the current stream (igcn_rider_dropout): the next per-graph plan build on
    it carries the mask generation in its own grid; ``igcn_rider_flush`` launches it if no build came.  The returned
    tensors are valid behind that launch.  (Only for site lists one launch takes.)"""
    import math
    sizes = [int(math.prod(shape)) for shape, _ in sites]
    # every site starts on a 16-byte boundary (its consumers read it with 16-byte loads)
    starts, total = [], 0
    for n in sizes:
        starts.append(total)
        total += (n + 3) // 4 * 4
    ends = [(starts[k + 1] if k + 1 < len(sites) else total) for k in range(len(sites))]
    out = torch.empty(total, dtype=torch.float32, device=state.state.device)
    per = int(_lib.load().igcn_dropout_max_segments())
    if ride and len(sites) <= per:
        seg_end = (ctypes.c_int64 * len(sites))(*ends)
        seg_p = (ctypes.c_float * len(sites))(*[float(p) for _, p in sites])
        cnt = list(counters)
        carr = (ctypes.c_void_p * max(len(cnt), 1))(*[c.data_ptr() for c in cnt])
        call("igcn_rider_dropout", stream_ptr(), total, len(sites), seg_end, seg_p, ptr(state.state), ptr(out), len(cnt),
             carr, int(inc))
        return [out[s0:s0 + n].view(*shape) for (shape, _), s0, n in zip(sites, starts, sizes)]
    for k0 in range(0, len(sites), per):                     # a deep GO hierarchy has more sites than one launch takes
        k1 = min(k0 + per, len(sites))
        base = starts[k0]
        seg_end = (ctypes.c_int64 * (k1 - k0))(*[e - base for e in ends[k0:k1]])
        seg_p = (ctypes.c_float * (k1 - k0))(*[float(p) for _, p in sites[k0:k1]])
        cnt = list(counters) if k0 == 0 else []
        carr = (ctypes.c_void_p * max(len(cnt), 1))(*[c.data_ptr() for c in cnt])
        call("igcn_dropout_masks", ends[k1 - 1] - base, k1 - k0, seg_end, seg_p, ptr(state.state), ptr(out[base:]),
             len(cnt), carr, int(inc), stream_ptr())
    return [out[s0:s0 + n].view(*shape) for (shape, _), s0, n in zip(sites, starts, sizes)]


# =================================================================================================
# GO hierarchy structures
# =================================================================================================
class Csr:
    """Row-grouped sparse structure + its transpose, as int32 device tensors.

    rows/cols: int64 CPU tensors of the non-zeros in ROW-MAJOR (coalesced) order.
    """

    def __init__(self, rows, cols, n_rows, n_cols, device):
        rows, cols = rows.long().cpu(), cols.long().cpu()
        self.n_rows, self.n_cols, self.nnz = int(n_rows), int(n_cols), int(rows.numel())
        key = rows * n_cols + cols
        assert self.nnz == 0 or bool((key[1:] > key[:-1]).all()), "non-zeros must be row-major and unique"
        row_ptr = torch.zeros(n_rows + 1, dtype=torch.long)
        row_ptr[1:] = torch.bincount(rows, minlength=n_rows).cumsum(0)
        order = torch.argsort(cols * n_rows + rows)          # column-major order; unique keys => deterministic
        t_ptr = torch.zeros(n_cols + 1, dtype=torch.long)
        t_ptr[1:] = torch.bincount(cols, minlength=n_cols).cumsum(0)
        i32 = lambda t: t.to(torch.int32).to(device).contiguous()     # noqa: E731
        pad = lambda t: t if t.numel() else torch.zeros(1, dtype=torch.long)   # noqa: E731
        self.row_ptr, self.col, self.row_of = i32(row_ptr), i32(pad(cols)), i32(pad(rows))
        self.t_ptr, self.t_row, self.t_k = i32(t_ptr), i32(pad(rows[order])), i32(pad(order))
        self.flat_pos = key.to(device)            # row * n_cols + col of every non-zero (dense formulation)
        self._dense = {}
        self._t_ptr_host = t_ptr.to(torch.int32).contiguous()
        self._walk_order = {}

    def walk_order(self, fin, fout):
        """Thread -> node map of the GO attention backward's column walks for a layer of this shape
        (igcn_go_attn_walk_order; square structures only), built on first use."""
        if self.n_rows != self.n_cols:
            return None
        key = (int(fin), int(fout))
        if key not in self._walk_order:
            lib = _lib.load()
            host = torch.empty(int(lib.igcn_go_attn_walk_slots(self.n_cols, *key)), dtype=torch.int32)
            call("igcn_go_attn_walk_order", self.n_cols, key[0], key[1], self._t_ptr_host.data_ptr(), host.data_ptr())
            self._walk_order[key] = host.to(self.flat_pos.device)
        return self._walk_order[key]

    def dense(self, channels):
        """Persistent zero-initialised dense image [channels, n_rows * n_cols]; only the non-zero positions are ever
        written (SparseMap scatters the current values into it), so the zeros stay zeros."""
        t = self._dense.get(channels)
        if t is None:
            t = torch.zeros(channels, self.n_rows * self.n_cols, dtype=torch.float32, device=self.flat_pos.device)
            self._dense[channels] = t
        return t


class SparseMap(torch.autograd.Function):
    """y[b,c,i] = sum_k val[c,k] x[b,col_k]  (gene encode go_model.py:208-215 / decode :281-282).

    Two formulations.  (a) The CSR kernels igcn_spmm_*: the structure is shared by all samples, so a workgroup reads it
    once and reuses it across a tile of samples whose operand rows sit in LDS — the default from SPARSE_MIN_BATCH samples
    (both passes of a 256-graph step: 512) up, where every launch fills the chip.  (b) For small batches (configs[4]:
    2 x 32 samples, 10 000 nodes) the values are scattered into a dense image and all three products run on the matrix
    cores (igcn_gemm_f32) — y = x T^T, dx = dy T, dT = dy^T x with the value gradients gathered back from dT; adding the
    structural zeros changes the fp32 summation order only.  IGCN_DENSE_MAPS=1 / IGCN_SPARSE_MAPS=1 force one of them;
    maps whose dense image would exceed DENSE_LIMIT floats always use (a)."""

    DENSE_LIMIT = 1 << 24
    SPARSE_MIN_BATCH = 256

    @staticmethod
    def _use_dense(b, c, csr):
        if c * csr.n_rows * csr.n_cols > SparseMap.DENSE_LIMIT or csr.nnz == 0:
            return False
        if os.environ.get("IGCN_DENSE_MAPS") == "1":
            return True
        if os.environ.get("IGCN_SPARSE_MAPS") == "1":
            return False
        return b < SparseMap.SPARSE_MIN_BATCH

    @staticmethod
    def forward(ctx, x, csr, *vals):
        """``vals``: the C per-channel value vectors [nnz] (the model's ParameterList entries, taken as they are: they
        stay autograd leaves of this op, so the value-gradient sums are final and deferrable) or ONE tensor [C, nnz]."""
        x = _f32(x)
        stacked = len(vals) == 1 and vals[0].dim() == 2
        ctx.vstride = None
        dense = SparseMap._use_dense(x.shape[0], vals[0].shape[0] if stacked else len(vals), csr)
        if stacked:
            val = _f32(vals[0])
        elif len(vals) == 1:
            val = _f32(vals[0]).reshape(1, -1)                   # one channel: a view, no launch
        else:
            vs = SparseMap._row_stride(vals, csr.nnz)
            if vs is not None:
                # the channels' vectors sit at a constant stride in one buffer (train.FlatAdam's flat parameters): the
                # kernels read them where they are — no torch.stack launch in front of every step
                ctx.vstride, val = vs, vals[0]
            else:
                val = torch.stack([_f32(v).reshape(-1) for v in vals])
        b, c = x.shape[0], (len(vals) if ctx.vstride is not None else val.shape[0])
        # ``x._igcn_grad_rows = (lo, hi)`` (set by the producer of x): only rows [lo, hi) of d x will ever be read — the
        # stacked (plain | masked) SNP batch of a train step, whose plain half is data — and the backward computes only
        # those (the rest of dx stays unwritten)
        ctx.grad_rows = getattr(x, "_igcn_grad_rows", None) if os.environ.get("IGCN_SNP_GRAD_ALL", "0") != "1" else None
        ctx.csr, ctx.stacked, ctx.nvals = csr, stacked, len(vals)
        ctx.final = _leaves(*vals)
        ctx.dense, ctx.channels = dense, c
        if ctx.dense:
            t = csr.dense(c)
            call("igcn_image_put", c, csr.nnz, ptr(csr.flat_pos), ptr(val),
                 ctx.vstride if ctx.vstride is not None else csr.nnz, ptr(t), csr.n_rows * csr.n_cols, stream_ptr())
            y = gemm_nt(x, t.view(c * csr.n_rows, csr.n_cols)).view(b, c, csr.n_rows)
            ctx.save_for_backward(x, val)
            return y
        y = torch.empty(b, c, csr.n_rows, dtype=torch.float32, device=x.device)
        if ctx.vstride is not None:
            call("igcn_spmm_fwd_strided", b, c, csr.n_rows, csr.n_cols, csr.nnz, ptr(csr.row_ptr), ptr(csr.col), ptr(val),
                 ctx.vstride, ptr(x), ptr(y), stream_ptr())
            ctx.save_for_backward(x, *vals)                        # (every channel's vector stays alive)
            return y
        call("igcn_spmm_fwd", b, c, csr.n_rows, csr.n_cols, csr.nnz, ptr(csr.row_ptr), ptr(csr.col), ptr(val),
             ptr(x), ptr(y), stream_ptr())
        ctx.save_for_backward(x, val)
        return y

    @staticmethod
    def _row_stride(vals, nnz):
        """Floats between consecutive channel vectors when they are fp32, contiguous, on one device and equally spaced
        in memory (stride >= nnz); None otherwise."""
        v0 = vals[0]
        if not all(v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() and v.numel() == nnz
                   and v.device == v0.device for v in vals):
            return None
        d = vals[1].data_ptr() - v0.data_ptr()
        if d <= 0 or d % 4 or d // 4 < nnz or any(vals[i].data_ptr() - v0.data_ptr() != i * d for i in range(2, len(vals))):
            return None
        return d // 4

    @staticmethod
    def backward(ctx, dy):
        x, val = ctx.saved_tensors[0], ctx.saved_tensors[1]
        csr = ctx.csr
        dy = _f32(dy)
        b, c = x.shape[0], ctx.channels
        need_val = any(ctx.needs_input_grad[2:])
        if ctx.dense:
            dy2 = dy.view(b, c * csr.n_rows)
            t = csr.dense(c)                     # still holds the values of the forward (same parameters)
            dx = gemm_nn(dy2, t.view(c * csr.n_rows, csr.n_cols)) if ctx.needs_input_grad[0] else None
            dval = None
            if need_val:
                dt = gemm_tn(dy2, x)                                  # [c * n_rows, n_cols]: the dense image of dT
                dval = torch.empty(c, csr.nnz, dtype=torch.float32, device=x.device)
                call("igcn_image_take", c, csr.nnz, ptr(csr.flat_pos), ptr(dt), csr.n_rows * csr.n_cols, ptr(dval),
                     stream_ptr())
        else:
            dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            dval = torch.empty(c, csr.nnz, dtype=torch.float32, device=x.device) if need_val else None
            scratch = _keep(torch.empty(int(_lib.load().igcn_spmm_bwd_scratch_floats(b, c, csr.n_rows, csr.n_cols,
                                                                                      csr.nnz)),
                                        dtype=torch.float32, device=x.device)) if dval is not None else None
            def bwd(dx_, dval_):
                bb, xx, dyy = b, x, dy
                if dval_ is None and dx_ is not None and ctx.grad_rows is not None and ctx.dense is False:
                    lo_, hi_ = ctx.grad_rows                      # input gradient of the rows somebody reads, only
                    if 0 <= lo_ < hi_ <= b:
                        bb, xx, dyy, dx_ = hi_ - lo_, x[lo_:hi_], dy[lo_:hi_], dx_[lo_:hi_]
                with _immediate(ctx.final):
                    if ctx.vstride is not None:
                        call("igcn_spmm_bwd_strided", bb, c, csr.n_rows, csr.n_cols, csr.nnz, ptr(csr.row_ptr), ptr(csr.col),
                             ptr(csr.row_of), ptr(csr.t_ptr), ptr(csr.t_row), ptr(csr.t_k), ptr(val), ctx.vstride, ptr(xx),
                             ptr(dyy), ptr(dx_), ptr(dval_), ptr(scratch), stream_ptr())
                    else:
                        call("igcn_spmm_bwd", bb, c, csr.n_rows, csr.n_cols, csr.nnz, ptr(csr.row_ptr), ptr(csr.col),
                             ptr(csr.row_of), ptr(csr.t_ptr), ptr(csr.t_row), ptr(csr.t_k), ptr(val), ptr(xx), ptr(dyy),
                             ptr(dx_), ptr(dval_), ptr(scratch), stream_ptr())
            if (dval is not None and _DEFER["on"] and ctx.final and csr.nnz > 0
                    and os.environ.get("IGCN_SPMM_DVAL_NOW", "0") != "1"):
                # the value gradients are parameter gradients: queued, and launched together with the other map's at
                # the end of the backward (each pass alone fills half the chip)
                if dx is not None:
                    bwd(dx, None)
                _DEFER["spmm_dval"].append(((b, c, csr.n_rows, csr.n_cols, csr.nnz),
                                            (csr.col, csr.row_of, x, dy, dval, scratch)))
            else:
                bwd(dx, dval)
        if dval is None:
            grads = (None,) * ctx.nvals
        elif ctx.stacked:
            grads = (dval,)
        else:
            grads = tuple(dval[i] for i in range(ctx.nvals))         # rows of one buffer: no copies
        return (dx, None) + grads


def _go_attn_backward(x, w_inc, w_s, a_in, a_s, csr, final, dy):
    """igcn_go_attn_bwd: (dx, dparams) of one encoder layer (GoAttention, GoAttentionLN's unfused path)."""
    b, fin, n = x.shape
    fout = w_inc.shape[0]
    lib = _lib.load()
    dx = torch.empty_like(x)
    dpar = torch.empty(2 * fout * fin + 3 * fout, dtype=torch.float32, device=x.device)
    scratch = _keep(torch.empty(int(lib.igcn_go_attn_bwd_scratch_floats(b, n, fin, fout)), dtype=torch.float32,
                                device=x.device))
    with _immediate(final):     # leaves: the parameter gradients join the deferred final reductions
        call("igcn_go_attn_bwd", b, n, fin, fout, ptr(csr.row_ptr), ptr(csr.col), ptr(csr.t_ptr), ptr(csr.t_row),
             ptr(csr.walk_order(fin, fout)), ptr(x), ptr(w_inc), ptr(w_s), ptr(a_in), ptr(a_s), ptr(dy), ptr(dx),
             ptr(dpar), ptr(scratch), stream_ptr())
    return dx, dpar


def _nodes_ln_backward(y, gamma, beta, keep, mean, rstd, pool, final, dz):
    """(dy, dgb [2, N]) of NodesLayerNorm: the dX pass now, the affine pass with the other layers' when deferred."""
    b, f, n = y.shape
    dy = torch.empty_like(y)
    dgb = torch.empty(2, n, dtype=torch.float32, device=y.device)
    lib = _lib.load()
    scratch = _keep(torch.empty(int(lib.igcn_nodes_ln_bwd_scratch_floats(b, f, n)), dtype=torch.float32,
                                device=y.device))
    if _DEFER["on"] and final and os.environ.get("IGCN_LN_AFFINE_NOW", "0") != "1":
        # d gamma / d beta are parameter gradients: their pass joins those of the other layers in ONE launch when
        # the backward ends (operands kept alive until then)
        call("igcn_nodes_ln_bwd_dy", b, f, n, pool, ptr(y), ptr(gamma), ptr(beta), ptr(keep), ptr(mean),
             ptr(rstd), ptr(dz), ptr(dy), stream_ptr())
        _DEFER["ln_affine"].append(((b, f, n, pool), (y, gamma, beta, keep, mean, rstd, dz, scratch, dgb)))
    else:
        with _immediate(final):
            call("igcn_nodes_ln_bwd", b, f, n, pool, ptr(y), ptr(gamma), ptr(beta), ptr(keep), ptr(mean),
                 ptr(rstd), ptr(dz), ptr(dy), ptr(dgb), ptr(scratch), stream_ptr())
    return dy, dgb


def _go_decode_backward(x, w_out, w_sout, csr, final, dy):
    """igcn_go_decode_bwd: (dx, dparams) of one decoder layer."""
    b, fin, nin = x.shape
    fout, nout = w_out.shape[0], csr.n_rows
    lib = _lib.load()
    dx = torch.empty_like(x)
    dpar = torch.empty(2 * fout * fin, dtype=torch.float32, device=x.device)
    scratch = _keep(torch.empty(int(lib.igcn_go_decode_bwd_scratch_floats(b, nin, fin, fout)),
                                dtype=torch.float32, device=x.device))
    with _immediate(final):
        call("igcn_go_decode_bwd", b, nin, nout, fin, fout, ptr(csr.row_ptr), ptr(csr.t_ptr), ptr(csr.t_row),
             ptr(x), ptr(w_out), ptr(w_sout), ptr(dy), ptr(dx), ptr(dpar), ptr(scratch), stream_ptr())
    return dx, dpar


class GoAttention(torch.autograd.Function):
    """One GO encoder layer for all samples (go_model.py:226-244).  x [B,fin,N] -> y [B,fout,N]."""

    @staticmethod
    def forward(ctx, x, w_inc, w_s, a_in, a_s, csr):
        x, w_inc, w_s, a_in, a_s = _f32(x), _f32(w_inc), _f32(w_s), _f32(a_in), _f32(a_s)
        b, fin, n = x.shape
        fout = w_inc.shape[0]
        y = torch.empty(b, fout, n, dtype=torch.float32, device=x.device)
        call("igcn_go_attn_fwd", b, n, fin, fout, ptr(csr.row_ptr), ptr(csr.col), ptr(x), ptr(w_inc), ptr(w_s),
             ptr(a_in), ptr(a_s), ptr(y), stream_ptr())
        ctx.save_for_backward(x, w_inc, w_s, a_in, a_s)
        ctx.csr, ctx.final = csr, _leaves(w_inc, w_s, a_in, a_s)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_inc, w_s, a_in, a_s = ctx.saved_tensors
        fout, fin = w_inc.shape[0], x.shape[1]
        dx, dpar = _go_attn_backward(x, w_inc, w_s, a_in, a_s, ctx.csr, ctx.final, _f32(dy))
        k = fout * fin
        return (dx, dpar[:k].view(fout, fin), dpar[k:2 * k].view(fout, fin),
                dpar[2 * k:2 * k + 2 * fout].view_as(a_in), dpar[2 * k + 2 * fout:].view_as(a_s), None)


class NodesLayerNorm(torch.autograd.Function):
    """LayerNorm over nodes + ReLU + node dropout + level pooling (go_model.py:246-251, :273-275)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, keep, pool, eps):
        y, gamma, beta = _f32(y), _f32(gamma), _f32(beta)
        keep = _f32(keep) if keep is not None else None
        b, f, n = y.shape
        z = torch.empty(b, f, n - pool, dtype=torch.float32, device=y.device)
        mean = torch.empty(b * f, dtype=torch.float32, device=y.device)
        rstd = torch.empty_like(mean)
        call("igcn_nodes_ln_fwd", b, f, n, pool, float(eps), ptr(y), ptr(gamma), ptr(beta), ptr(keep), ptr(z),
             ptr(mean), ptr(rstd), stream_ptr())
        ctx.save_for_backward(y, gamma, beta, keep, mean, rstd)
        ctx.pool = pool
        ctx.final = _leaves(gamma, beta)
        return z

    @staticmethod
    def backward(ctx, dz):
        y, gamma, beta, keep, mean, rstd = ctx.saved_tensors
        dy, dgb = _nodes_ln_backward(y, gamma, beta, keep, mean, rstd, ctx.pool, ctx.final, _f32(dz))
        return dy, dgb[0], dgb[1], None, None, None


def _al16(*ts):
    return all(t is None or t.data_ptr() % 16 == 0 for t in ts)


class GoAttentionLN(torch.autograd.Function):
    """One GO encoder layer AND the LayerNorm block behind it (go_model.py:226-251): z = dropout(relu(LN(attn(x))))[.., pool:].
    Forward: the two launches of GoAttention and NodesLayerNorm.  Backward: when the layer runs LDS-resident
    (igcn_go_attn_ln_fused_ok) ONE launch — the attention backward forms the LayerNorm's input gradient while it
    copies it in (igcn_go_attn_ln_bwd) and leaves per-sample d gamma | d beta rows to the final reductions; otherwise
    the backward passes of the two ops in sequence (same arithmetic per element)."""

    @staticmethod
    def forward(ctx, x, w_inc, w_s, a_in, a_s, csr, gamma, beta, keep, pool, eps, fan=1):
        """``fan`` > 1: returns that many aliases of z, one per consumer (the encoder output feeds two read-outs and the
        decoder); the backward adds their gradients while it loads them (``dz2`` / ``dz3`` of igcn_go_attn_ln_bwd)
        instead of a sum launch in front of it (ops.GradFan)."""
        x, w_inc, w_s, a_in, a_s = _f32(x), _f32(w_inc), _f32(w_s), _f32(a_in), _f32(a_s)
        gamma, beta = _f32(gamma), _f32(beta)
        keep = _f32(keep) if keep is not None else None
        b, fin, n = x.shape
        fout = w_inc.shape[0]
        y = torch.empty(b, fout, n, dtype=torch.float32, device=x.device)
        call("igcn_go_attn_fwd", b, n, fin, fout, ptr(csr.row_ptr), ptr(csr.col), ptr(x), ptr(w_inc), ptr(w_s),
             ptr(a_in), ptr(a_s), ptr(y), stream_ptr())
        z = torch.empty(b, fout, n - pool, dtype=torch.float32, device=x.device)
        mean = torch.empty(b * fout, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        call("igcn_nodes_ln_fwd", b, fout, n, pool, float(eps), ptr(y), ptr(gamma), ptr(beta), ptr(keep), ptr(z),
             ptr(mean), ptr(rstd), stream_ptr())
        ctx.save_for_backward(x, w_inc, w_s, a_in, a_s, y, gamma, beta, keep, mean, rstd)
        ctx.csr, ctx.pool, ctx.fan = csr, pool, fan
        ctx.final_attn, ctx.final_ln = _leaves(w_inc, w_s, a_in, a_s), _leaves(gamma, beta)
        if fan > 1:
            ctx.set_materialize_grads(False)
            return tuple(z.view_as(z) for _ in range(fan))
        return z

    @staticmethod
    def backward(ctx, *dzs):
        x, w_inc, w_s, a_in, a_s, y, gamma, beta, keep, mean, rstd = ctx.saved_tensors
        csr, pool = ctx.csr, ctx.pool
        b, fin, n = x.shape
        fout = w_inc.shape[0]
        lib = _lib.load()
        k = fout * fin
        gs = [_f32(g) for g in dzs if g is not None]
        none = (None,) * 12
        if not gs:
            return none
        fused = bool(lib.igcn_go_attn_ln_fused_ok(n, fin, fout, pool)) and _al16(x, y, gamma, beta, keep, *gs) \
            and os.environ.get("IGCN_NO_LN_FUSED", "0") != "1"
        if len(gs) > 1 and not (fused and len(gs) <= 3):
            # several consumers, no fused path to add them on load: one sum launch (what ops.GradFan does)
            total = torch.empty_like(gs[0])
            if len(gs) <= 4 and _al16(*gs):
                arr = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
                call("igcn_sum_n", total.numel(), len(gs), arr, ptr(total), stream_ptr())
            else:
                total = gs[0]
                for g in gs[1:]:
                    total = total + g
            gs = [total]
        dz = gs[0]
        dz2 = gs[1] if len(gs) > 1 else None
        dz3 = gs[2] if len(gs) > 2 else None
        if fused:
            dx = torch.empty_like(x)
            dpar = torch.empty(2 * k + 3 * fout, dtype=torch.float32, device=x.device)
            dgb = torch.empty(2, n, dtype=torch.float32, device=x.device)
            scratch = _keep(torch.empty(int(lib.igcn_go_attn_bwd_scratch_floats(b, n, fin, fout)),
                                        dtype=torch.float32, device=x.device))
            part = _keep(torch.empty(int(lib.igcn_go_ln_part_floats(b, n)), dtype=torch.float32, device=x.device))
            with _immediate(ctx.final_attn and ctx.final_ln):
                call("igcn_go_attn_ln_bwd", b, n, fin, fout, ptr(csr.row_ptr), ptr(csr.col), ptr(csr.t_ptr),
                     ptr(csr.t_row), ptr(csr.walk_order(fin, fout)), ptr(x), ptr(w_inc), ptr(w_s), ptr(a_in), ptr(a_s),
                     pool, ptr(y), ptr(gamma), ptr(beta), ptr(keep), ptr(mean), ptr(rstd), ptr(dz), ptr(dz2), ptr(dz3),
                     ptr(dx), ptr(dpar), ptr(dgb), ptr(scratch), ptr(part), stream_ptr())
        else:
            dy, dgb = _nodes_ln_backward(y, gamma, beta, keep, mean, rstd, pool, ctx.final_ln, dz)
            dx, dpar = _go_attn_backward(x, w_inc, w_s, a_in, a_s, csr, ctx.final_attn, dy)
        return (dx, dpar[:k].view(fout, fin), dpar[k:2 * k].view(fout, fin),
                dpar[2 * k:2 * k + 2 * fout].view_as(a_in), dpar[2 * k + 2 * fout:].view_as(a_s), None,
                dgb[0], dgb[1], None, None, None, None)


class GoDecodeLN(torch.autograd.Function):
    """One GO decoder layer AND the LayerNorm block behind it (go_model.py:262-275): z = dropout(relu(LN(decode(x)))).
    Backward in one launch when the layer runs LDS-resident (igcn_go_decode_ln_bwd), see GoAttentionLN."""

    @staticmethod
    def forward(ctx, x, w_out, w_sout, csr, gamma, beta, keep, eps):
        x, w_out, w_sout, gamma, beta = _f32(x), _f32(w_out), _f32(w_sout), _f32(gamma), _f32(beta)
        keep = _f32(keep) if keep is not None else None
        b, fin, nin = x.shape
        fout, nout = w_out.shape[0], csr.n_rows
        assert csr.n_cols == nin
        y = torch.empty(b, fout, nout, dtype=torch.float32, device=x.device)
        z = torch.empty_like(y)
        mean = torch.empty(b * fout, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        call("igcn_go_decode_fwd", b, nin, nout, fin, fout, ptr(csr.row_ptr), ptr(csr.col), ptr(x), ptr(w_out),
             ptr(w_sout), ptr(y), stream_ptr())
        call("igcn_nodes_ln_fwd", b, fout, nout, 0, float(eps), ptr(y), ptr(gamma), ptr(beta), ptr(keep), ptr(z),
             ptr(mean), ptr(rstd), stream_ptr())
        ctx.save_for_backward(x, w_out, w_sout, y, gamma, beta, keep, mean, rstd)
        ctx.csr = csr
        ctx.final_dec, ctx.final_ln = _leaves(w_out, w_sout), _leaves(gamma, beta)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w_out, w_sout, y, gamma, beta, keep, mean, rstd = ctx.saved_tensors
        csr = ctx.csr
        dz = _f32(dz)
        b, fin, nin = x.shape
        fout, nout = w_out.shape[0], csr.n_rows
        lib = _lib.load()
        k = fout * fin
        if lib.igcn_go_decode_ln_fused_ok(nin, nout, fin, fout) and _al16(y, dz, gamma, beta, keep) \
                and os.environ.get("IGCN_NO_LN_FUSED", "0") != "1":
            dx = torch.empty_like(x)
            dpar = torch.empty(2 * k, dtype=torch.float32, device=x.device)
            dgb = torch.empty(2, nout, dtype=torch.float32, device=x.device)
            scratch = _keep(torch.empty(int(lib.igcn_go_decode_bwd_scratch_floats(b, nin, fin, fout)),
                                        dtype=torch.float32, device=x.device))
            part = _keep(torch.empty(int(lib.igcn_go_ln_part_floats(b, nout)), dtype=torch.float32, device=x.device))
            with _immediate(ctx.final_dec and ctx.final_ln):
                call("igcn_go_decode_ln_bwd", b, nin, nout, fin, fout, ptr(csr.row_ptr), ptr(csr.t_ptr), ptr(csr.t_row),
                     ptr(x), ptr(w_out), ptr(w_sout), ptr(y), ptr(gamma), ptr(beta), ptr(keep), ptr(mean), ptr(rstd),
                     ptr(dz), ptr(dx), ptr(dpar), ptr(dgb), ptr(scratch), ptr(part), stream_ptr())
        else:
            dy, dgb = _nodes_ln_backward(y, gamma, beta, keep, mean, rstd, 0, ctx.final_ln, dz)
            dx, dpar = _go_decode_backward(x, w_out, w_sout, csr, ctx.final_dec, dy)
        return dx, dpar[:k].view(fout, fin), dpar[k:].view(fout, fin), None, dgb[0], dgb[1], None, None


class GoDecode(torch.autograd.Function):
    """One GO decoder layer (go_model.py:262-272).  x [B,fin,Nin] -> y [B,fout,Nout]."""

    @staticmethod
    def forward(ctx, x, w_out, w_sout, csr):
        x, w_out, w_sout = _f32(x), _f32(w_out), _f32(w_sout)
        b, fin, nin = x.shape
        fout, nout = w_out.shape[0], csr.n_rows
        assert csr.n_cols == nin
        y = torch.empty(b, fout, nout, dtype=torch.float32, device=x.device)
        call("igcn_go_decode_fwd", b, nin, nout, fin, fout, ptr(csr.row_ptr), ptr(csr.col), ptr(x), ptr(w_out),
             ptr(w_sout), ptr(y), stream_ptr())
        ctx.save_for_backward(x, w_out, w_sout)
        ctx.csr = csr
        ctx.final = _leaves(w_out, w_sout)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_out, w_sout = ctx.saved_tensors
        fout, fin = w_out.shape[0], x.shape[1]
        dx, dpar = _go_decode_backward(x, w_out, w_sout, ctx.csr, ctx.final, _f32(dy))
        k = fout * fin
        return dx, dpar[:k].view(fout, fin), dpar[k:].view(fout, fin), None


# =================================================================================================
# GO read-outs: node-wise linear + BatchNorm-over-nodes + ReLU
# =================================================================================================
class NodeLinearBN(torch.autograd.Function):
    """relu(BatchNorm1d_N(W x)) for x [B,F,N] channel-major -> [B,N,D]  (go_model.py:117-136,254-255,278).
    ``groups``: consecutive sample groups with independent batch statistics (passes batched into one launch)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, training, momentum, eps, groups=1, keep=None):
        """``keep`` [B,N] (D == 1 only): dropout factors of the output, applied inside the kernels."""
        x, weight, gamma, beta = _f32(x), _f32(weight), _f32(gamma), _f32(beta)
        keep = _f32(keep) if keep is not None else None
        b, f, n = x.shape
        d = weight.shape[0]
        lib = _lib.load()
        dev = x.device
        out = torch.empty(b, n, d, dtype=torch.float32, device=dev)
        mean = torch.empty(groups, n, dtype=torch.float32, device=dev)
        rstd = torch.empty(groups, n, dtype=torch.float32, device=dev)
        scratch = torch.empty(int(lib.igcn_node_linear_bn_scratch_floats(b, n, groups)), dtype=torch.float32,
                              device=dev)
        call("igcn_node_linear_bn_fwd", b, f, n, d, groups, ptr(x), ptr(weight), ptr(gamma), ptr(beta),
             ptr(running_mean), ptr(running_var), int(training), float(momentum), float(eps), ptr(keep), ptr(out),
             ptr(mean), ptr(rstd), ptr(scratch), stream_ptr())
        ctx.save_for_backward(x, weight, gamma, beta, mean, rstd, keep)
        ctx.training, ctx.groups = int(training), groups
        ctx.final = _leaves(weight, gamma, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, gamma, beta, mean, rstd, keep = ctx.saved_tensors
        dout = _f32(dout)
        b, f, n = x.shape
        d = weight.shape[0]
        lib = _lib.load()
        dev = x.device
        dx, dw = torch.empty_like(x), torch.empty_like(weight)
        dgb = torch.empty(2, n, dtype=torch.float32, device=dev)
        scratch = _keep(torch.empty(int(lib.igcn_node_linear_bn_bwd_scratch_floats(b, f, n, d, ctx.groups)),
                                    dtype=torch.float32, device=dev))
        with _immediate(ctx.final):
            call("igcn_node_linear_bn_bwd", b, f, n, d, ctx.groups, ctx.training, ptr(x), ptr(weight), ptr(gamma),
                 ptr(beta), ptr(mean), ptr(rstd), ptr(dout), ptr(keep), ptr(dx), ptr(dw), ptr(dgb), ptr(scratch),
                 stream_ptr())
        return dx, dw, dgb[0], dgb[1], None, None, None, None, None, None, None


class NodeLinearBNPair(torch.autograd.Function):
    """Two NodeLinearBN read-outs of the SAME input (go_model.py:254-255: conc_for_attention -> [B,N,D1] and conc ->
    [B,N,1] with its dropout) as one op: paired launches (igcn_node_linear_bn_pair_*), three forward and three
    backward for both, and ONE input gradient (the sum of the two) handed back to autograd."""

    @staticmethod
    def forward(ctx, x, x_alias, w1, g1, b1, rm1, rv1, mom1, eps1, w2, g2, b2, rm2, rv2, mom2, eps2, keep2, training,
                groups):
        """``x_alias``: a second autograd handle of the same tensor (ops.GradFan) or None.  With it the two read-outs'
        input gradients go back separately — the caller's GradFan sums them together with the decoder's in one launch
        — instead of being added here."""
        ctx.split = x_alias is not None
        x, w1, g1, b1, w2, g2, b2 = (_f32(t) for t in (x, w1, g1, b1, w2, g2, b2))
        keep2 = _f32(keep2) if keep2 is not None else None
        b, f, n = x.shape
        d1, d2 = w1.shape[0], w2.shape[0]
        lib = _lib.load()
        dev = x.device
        e = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)          # noqa: E731
        out1, out2 = e(b, n, d1), e(b, n, d2)
        mean1, rstd1, mean2, rstd2 = e(groups, n), e(groups, n), e(groups, n), e(groups, n)
        nscr = int(lib.igcn_node_linear_bn_scratch_floats(b, n, groups))
        s1, s2 = e(nscr), e(nscr)
        call("igcn_node_linear_bn_pair_fwd", b, f, n, groups, ptr(x), int(training),
             d1, ptr(w1), ptr(g1), ptr(b1), ptr(rm1), ptr(rv1), float(mom1), float(eps1), ptr(out1), ptr(mean1),
             ptr(rstd1), ptr(s1),
             d2, ptr(w2), ptr(g2), ptr(b2), ptr(rm2), ptr(rv2), float(mom2), float(eps2), ptr(keep2), ptr(out2),
             ptr(mean2), ptr(rstd2), ptr(s2), stream_ptr())
        ctx.save_for_backward(x, w1, g1, b1, mean1, rstd1, w2, g2, b2, mean2, rstd2, keep2)
        ctx.training, ctx.groups = int(training), groups
        ctx.final = _leaves(w1, g1, b1, w2, g2, b2)
        return out1, out2

    @staticmethod
    def backward(ctx, dout1, dout2):
        x, w1, g1, b1, mean1, rstd1, w2, g2, b2, mean2, rstd2, keep2 = ctx.saved_tensors
        dout1, dout2 = _f32(dout1), _f32(dout2)
        b, f, n = x.shape
        d1, d2 = w1.shape[0], w2.shape[0]
        lib = _lib.load()
        dev = x.device
        dx1, dx2 = torch.empty_like(x), torch.empty_like(x)
        dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
        dgb1 = torch.empty(2, n, dtype=torch.float32, device=dev)
        dgb2 = torch.empty(2, n, dtype=torch.float32, device=dev)
        s1 = _keep(torch.empty(int(lib.igcn_node_linear_bn_bwd_scratch_floats(b, f, n, d1, ctx.groups)),
                               dtype=torch.float32, device=dev))
        s2 = _keep(torch.empty(int(lib.igcn_node_linear_bn_bwd_scratch_floats(b, f, n, d2, ctx.groups)),
                               dtype=torch.float32, device=dev))
        with _immediate(ctx.final):
            call("igcn_node_linear_bn_pair_bwd", b, f, n, ctx.groups, ctx.training, ptr(x),
                 d1, ptr(w1), ptr(g1), ptr(b1), ptr(mean1), ptr(rstd1), ptr(dout1), ptr(dx1), ptr(dw1), ptr(dgb1), ptr(s1),
                 d2, ptr(w2), ptr(g2), ptr(b2), ptr(mean2), ptr(rstd2), ptr(dout2), ptr(keep2), ptr(dx2), ptr(dw2),
                 ptr(dgb2), ptr(s2), stream_ptr())
        if ctx.split:
            dxa, dxb = dx1, dx2
        else:
            dxa, dxb = dx1.add_(dx2), None
        return (dxa, dxb, dw1, dgb1[0], dgb1[1], None, None, None, None, dw2, dgb2[0], dgb2[1], None, None, None, None,
                None, None, None)


def node_linear_bn_pair_supported(x, w1, w2, keep1):
    return (keep1 is None and x.is_cuda and x.dim() == 3
            and bool(_lib.load().igcn_node_linear_bn_pair_supported(x.shape[1], w1.shape[0], w2.shape[0])))


class BatchNorm1dGrouped(torch.autograd.Function):
    """(ReLU of) BatchNorm1d(C) on [B,C] with grouped batch statistics (latent MLP, go_model.py:138-146)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, groups, keep=None):
        """``keep`` [B,C]: dropout factors of the output, applied inside the kernels."""
        x, gamma, beta = _f32(x), _f32(gamma), _f32(beta)
        keep = _f32(keep) if keep is not None else None
        b, c = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(groups, c, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        call("igcn_bn1d_fwd", b, c, groups, ptr(x), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
             int(training), float(momentum), float(eps), int(relu), ptr(keep), ptr(y), ptr(mean), ptr(rstd),
             stream_ptr())
        ctx.save_for_backward(x, gamma, beta, mean, rstd, keep)
        ctx.cfg = (int(training), int(relu), groups)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd, keep = ctx.saved_tensors
        training, relu, groups = ctx.cfg
        dy = _f32(dy)
        b, c = x.shape
        dx, dg, db = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(beta)
        call("igcn_bn1d_bwd", b, c, groups, training, relu, ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
             ptr(dy), ptr(keep), ptr(dx), ptr(dg), ptr(db), stream_ptr())
        return dx, dg, db, None, None, None, None, None, None, None, None


class LinearBN1d(torch.autograd.Function):
    """dropout(relu(BatchNorm1d(x W^T))) — a bias-less Linear and the BatchNorm1d block behind it (the wide layer of the
    latent MLP, kernel/go_model.py:138-146) with the product's split-K slabs summed by the BatchNorm launch while it
    loads its column: no slab-sum launch in between.  Backward: igcn_bn1d_bwd, then dX | dW as ops.Linear."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, training, momentum, eps, groups, keep=None):
        x, weight, gamma, beta = _f32(x), _f32(weight), _f32(gamma), _f32(beta)
        keep = _f32(keep) if keep is not None else None
        b, k = x.shape
        c = weight.shape[0]
        dev = x.device
        xl = torch.empty(b, c, dtype=torch.float32, device=dev)
        y = torch.empty_like(xl)
        mean = torch.empty(groups, c, dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        lib = _lib.load()
        tail = (ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), int(training), float(momentum), float(eps), 1,
                ptr(keep), ptr(y), ptr(mean), ptr(rstd), stream_ptr())
        sk = _split_k(b, c, k)
        scratch = torch.empty(sk * b * c, dtype=torch.float32, device=dev) if sk > 1 else None
        call("igcn_gemm_f32", b, c, k, ptr(x), k, 1, ptr(weight), k, 1, None, ptr(xl), c, 0x200, sk, ptr(scratch),
             stream_ptr())
        eff = int(lib.igcn_gemm_effective_split(k, sk))
        if eff > 1:
            call("igcn_bn1d_fwd_slabs", b, c, groups, ptr(scratch), eff, ptr(xl), *tail)
        else:
            call("igcn_bn1d_fwd", b, c, groups, ptr(xl), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                 int(training), float(momentum), float(eps), 1, ptr(keep), ptr(y), ptr(mean), ptr(rstd), stream_ptr())
        ctx.save_for_backward(x, weight, xl, gamma, beta, mean, rstd, keep)
        ctx.cfg = (int(training), groups)
        ctx.w_final = _leaves(weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, xl, gamma, beta, mean, rstd, keep = ctx.saved_tensors
        training, groups = ctx.cfg
        dy = _f32(dy)
        b, c = xl.shape
        dxl, dg, db = torch.empty_like(xl), torch.empty_like(gamma), torch.empty_like(beta)
        call("igcn_bn1d_bwd", b, c, groups, training, 1, ptr(xl), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dy),
             ptr(keep), ptr(dxl), ptr(dg), ptr(db), stream_ptr())
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            dx, dw = gemm_group([("nn", dxl, weight, None, None, False), ("tn", dxl, x, None, None, ctx.w_final)])
        else:
            dx = gemm_nn(dxl, weight) if ctx.needs_input_grad[0] else None
            dw = gemm_tn(dxl, x, final_grad=ctx.w_final) if ctx.needs_input_grad[1] else None
        return dx, dw, dg, db, None, None, None, None, None, None, None


def linear_bn1d_supported(x, weight, groups):
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous() and weight.is_contiguous()
            and os.environ.get("IGCN_NO_LINEAR_BN_FUSED", "0") != "1"
            and _split_k(x.shape[0], weight.shape[0], x.shape[1]) > 1      # (an unsplit product has no slab sum to save)
            and bool(_lib.load().igcn_bn1d_fwd_supported(x.shape[0], groups)))


# =================================================================================================
# Loss terms
# =================================================================================================
class MaskRegulariser(torch.autograd.Function):
    """loss_probability (kernel/sgcn_img_snp.py:153-181) as one reduction kernel + one elementwise backward."""

    @staticmethod
    def forward(ctx, prob, e, snps_prob, l1_x, ent_x, l1_e, ent_e, eps, partials=False):
        """``partials``: return the [blocks] workgroup partials whose SUM is the loss (LossHead adds them up inside its
        own kernel: one launch less); their gradient is the same scalar in every element."""
        prob, e, snps_prob = _f32(prob), _f32(e), _f32(snps_prob)
        dev = prob.device
        scratch = torch.empty(1024, dtype=torch.float32, device=dev)
        ctx.hp = (float(l1_x), float(ent_x), float(l1_e), float(ent_e), float(eps))
        loss = None if partials else torch.empty(1, dtype=torch.float32, device=dev)
        call("igcn_mask_reg_fwd", prob.numel(), e.numel(), snps_prob.numel(), ptr(prob), ptr(e), ptr(snps_prob),
             *ctx.hp, ptr(loss), ptr(scratch), stream_ptr())
        ctx.save_for_backward(prob, e, snps_prob)
        ctx.partials = partials
        if partials:
            return scratch[:int(_lib.load().igcn_mask_reg_blocks(prob.numel() + e.numel() + snps_prob.numel()))]
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        prob, e, snps_prob = ctx.saved_tensors
        gout = _f32(gout[:1] if ctx.partials else gout).reshape(1)     # partials: the same scalar in every element
        dprob, de, dsnps = torch.empty_like(prob), torch.empty_like(e), torch.empty_like(snps_prob)
        call("igcn_mask_reg_bwd", prob.numel(), e.numel(), snps_prob.numel(), ptr(prob), ptr(e), ptr(snps_prob),
             *ctx.hp, ptr(gout), ptr(dprob), ptr(de), ptr(dsnps), stream_ptr())
        return dprob, de, dsnps, None, None, None, None, None, None


def rbf_laplacian(tsne, n, gamma, device):
    """Lap = diag(W1) - W with W = exp(-gamma*cdist(t,t)^2) (util/image_cluster.py:15-31); tsne None -> W = 1."""
    lap = torch.empty(n, n, dtype=torch.float32, device=device)
    t = _f32(tsne) if tsne is not None else None
    call("igcn_rbf_laplacian", n, t.shape[1] if t is not None else 0, float(gamma), ptr(t), ptr(lap), stream_ptr())
    return lap


class GramLosses(torch.autograd.Function):
    """(consist_loss, OrthogonalConstraint) of s [G*B, R*D] per group of B rows, each from one B x B Gram matrix
    (sgcn_img_snp.py:183-205).  Returns two tensors of shape [G]."""

    @staticmethod
    def forward(ctx, s, lap, groups=1, packed=False, rbf=None, expect=None, pre=None, hold=None):
        """``expect`` (with ``rbf``): the upstream gradient [G*2] the caller expects in the backward, as host floats (a
        train step's d loss / d (consist, orth) are its loss weights): the forward kernel then writes the backward's S
        as it goes, and a backward whose upstream IS that (announced by LossHead through UNIT_DGRAM, recognised by
        address like the unit scalar itself) launches no igcn_gram_loss_bwd.
        ``pre``: the Gram matrices [G, B, B] of ``s``, already computed (``gram_rider``: the products rode in another
        grouped launch) — no GEMM here.
        ``packed``: return ONE tensor [G,2] = (consist, orth) per group (what igcn_loss_head_* consumes);
        ``packed == "partials"``: the un-reduced row partials [B, G*2] whose column sums are that tensor (LossHead adds
        them up inside its own kernel: one launch less); every row then receives the gradient of the sum.
        ``rbf`` = (tsne [B, T] or None, gamma) with ``lap`` None: the Laplacian D - W of consist_loss is built inside the
        loss kernel (igcn_gram_loss_fwd_rbf) instead of by its own launch in front."""
        s = _f32(s)
        gb, rd = s.shape
        b = gb // groups
        tsne = None
        if lap is None:
            tsne = _f32(rbf[0]) if rbf[0] is not None else None
            lap = torch.empty(b, b, dtype=torch.float32, device=s.device)
        else:
            lap, rbf = _f32(lap), None
        partials = packed == "partials"
        out = None if partials else torch.empty(groups, 2, dtype=torch.float32, device=s.device)
        scratch = torch.empty(2 * b * groups, dtype=torch.float32, device=s.device)
        if pre is not None:
            gram = pre
            if tuple(gram.shape) != (groups, b, b) or gram.dtype != torch.float32 or not gram.is_contiguous():
                raise _lib.IgcnError("GramLosses: the precomputed Gram matrices do not match the input")
        else:
            gram = torch.empty(groups, b, b, dtype=torch.float32, device=s.device)
            # the per-group Gram matrices s_g s_g^T in ONE launch (+ one slab sum)
            sk = _split_k(b * groups, b, rd)
            gscr = torch.empty(groups * sk * b * b, dtype=torch.float32, device=s.device) if sk > 1 else None
            call("igcn_gemm_f32_batched", b, b, rd, groups, ptr(s), rd, 1, b * rd, ptr(s), rd, 1, b * rd, ptr(gram), b * b,
                 b, sk, ptr(gscr), stream_ptr())
        ctx.sym = ctx.expect = None
        if (rbf is not None and expect is not None and groups <= 4 and len(expect) == 2 * groups
                and ctx.needs_input_grad[0]):
            ctx.expect = tuple(float(v) for v in expect)
            ctx.sym = torch.empty(groups, b, b, dtype=torch.float32, device=s.device)
            if hold is not None and partials:
                # ``hold`` (a dict of the caller): the launch is NOT issued here — its arguments are handed to the caller,
                # whose next launch runs it as a second role of its own grid (ops.HeadLoss: igcn_head_loss_gram_fwd).  The
                # returned partials hold their values behind THAT launch.
                hold["gram"] = (b, rd, groups, gram, tsne, tsne.shape[1] if tsne is not None else 0, float(rbf[1]), lap,
                                scratch, ctx.expect, ctx.sym)
            else:
                call("igcn_gram_loss_fwd_rbf_unit", b, rd, groups, ptr(gram), ptr(tsne),
                     tsne.shape[1] if tsne is not None else 0, float(rbf[1]), ptr(lap), ptr(out), ptr(scratch),
                     (ctypes.c_float * (2 * groups))(*ctx.expect), ptr(ctx.sym), stream_ptr())
        elif rbf is not None:
            call("igcn_gram_loss_fwd_rbf", b, rd, groups, ptr(gram), ptr(tsne), tsne.shape[1] if tsne is not None else 0,
                 float(rbf[1]), ptr(lap), ptr(out), ptr(scratch), stream_ptr())
        else:
            call("igcn_gram_loss_fwd", b, rd, groups, ptr(gram), ptr(lap), ptr(out), ptr(scratch), stream_ptr())
        ctx.save_for_backward(s, lap, gram)
        ctx.groups, ctx.packed = groups, packed
        if partials:
            return scratch.view(b, 2 * groups)
        if packed:
            return out
        return out[:, 0], out[:, 1]

    @staticmethod
    def backward(ctx, g_c, g_o=None):
        s, lap, gram = ctx.saved_tensors
        groups = ctx.groups
        b = s.shape[0] // groups
        if ctx.packed == "partials":
            gout = _f32(g_c[0])                                   # the same [G*2] row in every partial row
        elif ctx.packed:
            gout = _f32(g_c)
        else:
            zero = torch.zeros(groups, dtype=torch.float32, device=s.device)
            gout = torch.stack([g_c if g_c is not None else zero, g_o if g_o is not None else zero],
                               dim=1).contiguous()
        ds = torch.empty_like(s)
        if ctx.sym is not None and UNIT_DGRAM.pop(gout.data_ptr(), None) == ctx.expect:
            sym = ctx.sym                                         # written by the forward for exactly this upstream
        else:
            sym = torch.empty(groups, b, b, dtype=torch.float32, device=s.device)
            call("igcn_gram_loss_bwd", b, groups, ptr(gram), ptr(lap), ptr(gout), ptr(sym), stream_ptr())
        rd = s.shape[1]
        call("igcn_gemm_f32_batched", b, rd, b, groups, ptr(sym), b, 1, b * b, ptr(s), 1, rd, b * rd, ptr(ds), b * rd, rd,
             1, None, stream_ptr())                                # ds_g = S_g s_g, every group in one launch
        return ds, None, None, None, None, None, None, None


def gram_rider(s, groups):
    """Queue the per-group Gram products s_g s_g^T of ``s`` [G*B, RD] as RIDERS of the next grouped GEMM launch on this
    stream (the heads' first layers, which read the same HeadInputs outputs): returns (gram [G, B, B], hold) — pass
    ``gram`` to GramLosses(pre=...) after the carrying launch (or igcn_gemm_rider_flush)."""
    s = _f32(s)
    gb, rd = s.shape
    b = gb // groups
    gram = torch.empty(groups, b, b, dtype=torch.float32, device=s.device)
    specs = [("nt", s[g * b:(g + 1) * b], s[g * b:(g + 1) * b], gram[g], None, False) for g in range(groups)]
    _, hold = gemm_group(specs, ride=True)
    return gram, hold


class ProjectedAttention(torch.autograd.Function):
    """ops.InProj + ops.AttentionCore as ONE autograd node (the cross-attention of kernel/sgcn_img_snp.py:240-241 up to
    the output projection): q = query W_q^T + b_q, k | v = memory [W_k; W_v]^T + [b_k; b_v], o = softmax(q k^T/sqrt(hd)) v.
    Seeing the core's incoming gradient and the projections together lets the backward take the key / value bias
    gradients in closed form instead of summing the 52 MB gradient of k | v over its rows:
      d b_k = 0 exactly        (a key bias adds q . b_k to every score of a query: the softmax over keys cannot see it)
      d b_v = sum_{b,q} d o    (every query's attention weights sum to 1)
    — the reference's autograd evaluates the first as rounding noise around 0 and the second as the same sum, reached
    through the 400-key value gradient."""

    @staticmethod
    def forward(ctx, query, memory, w, bias, heads, bf16=False):
        d = w.shape[1]
        q2, m2 = _f32(query).reshape(-1, d), _f32(memory).reshape(-1, d)
        w, bias = _f32(w), _f32(bias)
        b, lq, lk = query.shape[0], query.shape[1], memory.shape[1]
        q, kv = _proj_forward(q2, m2, w, bias, d, bf16)
        q, kv = q.view(b, lq, d), kv.view(b, lk, 2 * d)
        o = torch.empty_like(q)
        lse = torch.empty(b, heads, lq, dtype=torch.float32, device=q.device)
        core16 = bool(bf16) and attn_core_bf16(d, heads, lq, lk)
        call("igcn_attn_core_bf16_fwd" if core16 else "igcn_attn_core_fwd", b, d, heads, lq, lk, ptr(q), ptr(kv), ptr(o),
             ptr(lse), stream_ptr())
        ctx.save_for_backward(q2, m2, w, q, kv, o, lse)
        ctx.heads, ctx.bf16, ctx.core16, ctx.final = heads, bf16, core16, _leaves(w, bias)
        ctx.shapes = (query.shape, memory.shape)
        return o

    @staticmethod
    def backward(ctx, dout):
        q2, m2, w, q, kv, o, lse = ctx.saved_tensors
        dout = _f32(dout)
        b, lq, d = q.shape
        lk = kv.shape[1]
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        nscr = int(_lib.load().igcn_attn_core_bwd_scratch_floats(b, ctx.heads, lq))
        scratch = torch.empty(nscr, dtype=torch.float32, device=q.device)
        call("igcn_attn_core_bf16_bwd" if ctx.core16 else "igcn_attn_core_bwd", b, d, ctx.heads, lq, lk, ptr(q), ptr(kv),
             ptr(o), ptr(lse), ptr(dout), ptr(dq), ptr(dkv), ptr(scratch), stream_ptr())
        dq2, dkv2 = dq.view(-1, d), dkv.view(-1, 2 * d)
        dw = torch.empty_like(w)
        db = torch.empty(3 * d, dtype=torch.float32, device=w.device)
        lib = _lib.load()
        rows = dq2.shape[0]
        scr = _keep(torch.empty(int(lib.igcn_bias_grad_scratch_floats(rows, 2 * d)), dtype=torch.float32, device=w.device))
        scr2 = _keep(torch.empty(int(lib.igcn_bias_grad_scratch_floats(rows, d)), dtype=torch.float32, device=w.device))
        if _proj_fused(ctx, dq2, dkv2, d) and os.environ.get("IGCN_NO_PROJ_BIAS_FUSED", "0") != "1":
            dquery, dmem = _proj_backward(ctx, dq2, dkv2, q2, m2, w, dw, d, db)      # bias gradients ride in the pass
            return dquery, dmem, dw, db, None, None
        with _immediate(ctx.final):                                 # d b_q (then d b_k = 0) and d b_v in one launch
            call("igcn_bias_grad_pair", rows, d, ptr(dq2), None, None, ptr(db), d, ptr(scr),
                 ptr(dout.reshape(-1, d)), None, None, ptr(db[2 * d:]), 0, ptr(scr2), stream_ptr())
        dquery, dmem = _proj_backward(ctx, dq2, dkv2, q2, m2, w, dw, d)
        return dquery, dmem, dw, db, None, None


class LossHead(torch.autograd.Function):
    """The seven loss terms of train() (kernel/train_eval_sgcn_img_snps.py:525-543) and their weighted sum on the
    stacked outputs of the batched sweep, one kernel per direction (igcn_loss_head_*).
    Returns (loss scalar, terms [7] — not differentiable: {ce, mi, reg, prob, recon, cluster, orth} lam-weighted)."""

    @staticmethod
    def forward(ctx, logp, y, reg, clin, x_hat, snps, gram, prob, lam, hp_ce, hp_mi, from_logits=False):
        """``from_logits``: ``logp`` holds the raw class scores; log_softmax is taken inside the kernel and returned as
        a third (non-differentiable) output.  ``gram`` [rows, 4] / ``prob`` [rows] may be un-reduced partials
        (GramLosses ``packed="partials"``, MaskRegulariser ``partials=True``): their rows are summed by the kernel."""
        logp, reg, clin, x_hat, snps, gram, prob = (_f32(t) for t in (logp, reg, clin, x_hat, snps, gram, prob))
        y = y.contiguous()
        b, c = logp.shape[0] // 2, logp.shape[1]
        nr, s = reg.numel() // (2 * b), snps.shape[1]
        if y.dtype != torch.int64 or y.numel() != b or clin.numel() != b * nr or x_hat.shape != (2 * b, s) \
                or gram.numel() % 4 != 0 or gram.numel() == 0 or prob.numel() == 0 or snps.shape[0] != b:
            raise _lib.IgcnError("loss head: inconsistent shapes")
        dev = logp.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        terms = torch.empty(7, dtype=torch.float32, device=dev)
        logp_out = torch.empty_like(logp) if from_logits else None
        lam6 = (ctypes.c_float * 6)(*[float(v) for v in lam])
        ctx.cfg = (b, c, nr, s, [float(v) for v in lam], float(hp_ce), float(hp_mi))
        ctx.gram_shape, ctx.prob_shape = tuple(gram.shape), tuple(prob.shape)
        # when a backward will follow, the forward writes the gradients for an upstream gradient of ONE as it goes — a
        # train step's d loss / d loss (train._unit_grad, recognised by its address): no backward launch for the loss head
        ctx.unit = None
        if any(ctx.needs_input_grad) and UNIT_GRAD_PTRS and os.environ.get("IGCN_NO_LOSS_HEAD_FUSED", "0") != "1":
            f32 = dict(dtype=torch.float32, device=dev)
            unit = (torch.empty(2 * b, c, **f32), torch.empty_like(reg), torch.empty_like(x_hat), torch.empty(4, **f32),
                    torch.empty(1, **f32))
            call("igcn_loss_head_fwd_grads", b, c, nr, s, ptr(logp), 1 if from_logits else 0, ptr(logp_out), ptr(y),
                 ptr(reg), ptr(clin), ptr(x_hat), ptr(snps), ptr(gram), gram.numel() // 4, ptr(prob), prob.numel(), lam6,
                 float(hp_ce), float(hp_mi), ptr(loss), ptr(terms), *[ptr(t) for t in unit], stream_ptr())
            ctx.unit = unit
        else:
            call("igcn_loss_head_fwd", b, c, nr, s, ptr(logp), 1 if from_logits else 0, ptr(logp_out), ptr(y), ptr(reg),
                 ptr(clin), ptr(x_hat), ptr(snps), ptr(gram), gram.numel() // 4, ptr(prob), prob.numel(), lam6,
                 float(hp_ce), float(hp_mi), ptr(loss), ptr(terms), stream_ptr())
        ctx.save_for_backward(y, reg, clin, x_hat, snps, logp_out)
        ctx.mark_non_differentiable(terms)
        ctx.set_materialize_grads(False)          # no zero tensor for `terms` in the backward
        if from_logits:
            ctx.mark_non_differentiable(logp_out)
            return loss.view(()), terms, logp_out
        return loss.view(()), terms

    @staticmethod
    def backward(ctx, gout, _gterms, _glogp=None):
        y, reg, clin, x_hat, snps, logp = ctx.saved_tensors
        b, c, nr, s, lam, hp_ce, hp_mi = ctx.cfg
        gout = _f32(gout).reshape(1)
        dev = reg.device
        UNIT_DGRAM.clear()
        if ctx.unit is not None and gout.data_ptr() in UNIT_GRAD_PTRS:
            dlogp, dreg, dxhat, dgram, dprob = ctx.unit              # written by the forward for exactly this upstream
            # the values behind dgram's address, for GramLosses (whose forward may have prepared its backward for them)
            UNIT_DGRAM[dgram.data_ptr()] = unit_dgram(lam)
        else:
            dlogp = torch.empty(2 * b, c, dtype=torch.float32, device=dev)
            dreg, dxhat = torch.empty_like(reg), torch.empty_like(x_hat)
            dgram = torch.empty(4, dtype=torch.float32, device=dev)
            dprob = torch.empty(1, dtype=torch.float32, device=dev)
            lam6 = (ctypes.c_float * 6)(*lam)
            call("igcn_loss_head_bwd", b, c, nr, s, ptr(y), ptr(reg), ptr(clin), ptr(x_hat), ptr(snps), ptr(logp), lam6,
                 hp_ce, hp_mi, ptr(gout), ptr(dlogp), ptr(dreg), ptr(dxhat), ptr(dgram), ptr(dprob), stream_ptr())
        # un-reduced partial inputs: every row gets the gradient of the sum (stride-0 views: no launch)
        gs, ps = ctx.gram_shape, ctx.prob_shape
        dgram = dgram.view(2, 2) if gs == (2, 2) else dgram.view(1, 4).expand(gs[0], 4)
        dprob = dprob.view(()) if ps == () else dprob.expand(ps)
        return dlogp, None, dreg, None, dxhat, None, dgram, dprob, None, None, None, None


_LOSS_WTS = {}


def _loss_weights(lam, hp_ce, hp_mi, b, nr, dev):
    """{lam[0..5], hp_ce, hp_mi, B, NR} as a cached DEVICE vector (igcn_loss_final reads its weights from memory: the
    job may run as an entry of the deferred flush, whose table has no room for ten floats)."""
    key = (tuple(float(v) for v in lam), float(hp_ce), float(hp_mi), int(b), int(nr), str(dev))
    t = _LOSS_WTS.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            return None                              # never upload from inside a capture: the caller takes the other route
        t = _LOSS_WTS[key] = torch.tensor(list(key[0]) + [key[1], key[2], float(b), float(nr)], dtype=torch.float32,
                                          device=dev)
    return t


def head_loss_supported(lin_f, w2, reg_f, w2r, keep1, keep2):
    """The fused output-heads + loss launch (igcn_head_loss_fwd) covers these layers, and a unit upstream gradient is what
    the backward will bring (train._unit_grad has registered its scalar)."""
    return (lin_f.is_cuda and lin_f.dim() == 2 and lin_f.shape == reg_f.shape and w2.shape[1] == w2r.shape[1] == lin_f.shape[1]
            and lin_f.dtype == torch.float32 and lin_f.is_contiguous() and reg_f.is_contiguous()
            and (keep1 is None) == (keep2 is None) and torch.is_grad_enabled() and bool(UNIT_GRAD_PTRS)
            and os.environ.get("IGCN_NO_LOSS_HEAD_FUSED", "0") != "1" and os.environ.get("IGCN_NO_HEAD_LOSS_FUSED", "0") != "1"
            and bool(_lib.load().igcn_head_loss_supported(lin_f.shape[1], w2.shape[0], w2r.shape[0])))


class HeadLoss(torch.autograd.Function):
    """lin2 | lin2_regr -> log_softmax -> the cross-entropy / regression / reconstruction terms of train() AND their
    backward for an upstream gradient of one, in ONE multi-workgroup launch (igcn_head_loss_fwd) — instead of
    SmallLinearPair.forward, LossHead.forward (one workgroup) and SmallLinearPair.backward on the step's critical path.
    Returns (loss, terms [7], log_softmax [2B, C], regression outputs [2B, NR]); the last three are not differentiable.

    The loss VALUE is the weighted sum of partial sums that three launches leave behind (this one, the Gram loss, the mask
    regulariser); nothing of the backward reads it.  ``lazy``: its last step (igcn_loss_final) is issued by the BACKWARD —
    inside ``deferred_reductions`` it joins the flush as one more workgroup — so ``loss`` / ``terms`` hold their values
    only after the step (train_step / GraphedTrainStep); otherwise a small launch right behind the forward."""

    @staticmethod
    def forward(ctx, lin_f, keep1, w2, b2, reg_f, keep2, w2r, b2r, y, clin, x_hat, snps, gram, prob, lam, hp_ce, hp_mi,
                lazy=False, gram_job=None):
        """``gram_job``: the arguments of a Gram-loss launch that GramLosses handed back instead of issuing
        (``hold``): it runs as a second role of this launch's grid (igcn_head_loss_gram_fwd)."""
        f = lambda t: _f32(t) if t is not None else None               # noqa: E731
        lin_f, keep1, w2, b2, reg_f, keep2, w2r, b2r, clin, x_hat, snps, gram, prob = (
            f(t) for t in (lin_f, keep1, w2, b2, reg_f, keep2, w2r, b2r, clin, x_hat, snps, gram, prob))
        y = y.contiguous()
        rows, k = lin_f.shape
        b, c, nr, s = rows // 2, w2.shape[0], w2r.shape[0], snps.shape[1]
        if y.dtype != torch.int64 or y.numel() != b or clin.numel() != b * nr or x_hat.shape != (2 * b, s) \
                or gram.numel() % 4 != 0 or gram.numel() == 0 or prob.numel() == 0 or snps.shape[0] != b:
            raise _lib.IgcnError("head loss: inconsistent shapes")
        dev = lin_f.device
        lib = _lib.load()
        f32 = dict(dtype=torch.float32, device=dev)
        nblk = int(lib.igcn_head_loss_blocks(b, k))
        out8 = torch.empty(8, **f32)
        logp, our_reg = torch.empty(2 * b, c, **f32), torch.empty(2 * b, nr, **f32)
        dx1, dx2, dxhat = torch.empty_like(lin_f), torch.empty_like(reg_f), torch.empty_like(x_hat)
        parts = torch.empty(nblk, 4, **f32)
        wcols = c * k + c + nr * k + nr
        wpart = torch.empty(nblk, wcols, **f32)
        dgram, dprob = torch.empty(4, **f32), torch.empty(1, **f32)
        lam6 = (ctypes.c_float * 6)(*[float(v) for v in lam])
        head = (b, k, c, nr, s, ptr(lin_f), ptr(keep1), ptr(w2), ptr(b2), ptr(reg_f), ptr(keep2), ptr(w2r), ptr(b2r), ptr(y),
                ptr(clin), ptr(x_hat), ptr(snps), lam6, float(hp_ce), float(hp_mi), ptr(logp), ptr(our_reg), ptr(dx1), ptr(dx2),
                ptr(dxhat), ptr(parts), ptr(wpart), ptr(dgram), ptr(dprob))
        if gram_job is not None:
            gb_, grd, ggr, gmat, gts, gt, ggam, glap, gscr, gexp, gsym = gram_job
            call("igcn_head_loss_gram_fwd", *head, gb_, grd, ggr, ptr(gmat), ptr(gts), gt, ggam, ptr(glap), ptr(gscr),
                 (ctypes.c_float * (2 * ggr))(*gexp), ptr(gsym), stream_ptr())
        else:
            call("igcn_head_loss_fwd", *head, stream_ptr())
        wts = _loss_weights(lam, hp_ce, hp_mi, b, nr, dev)
        ctx.final = (parts, gram, prob, wts, out8) if lazy else None
        if not lazy:
            call("igcn_loss_final", ptr(parts), nblk, ptr(gram), gram.numel() // 4, ptr(prob), prob.numel(), ptr(wts),
                 ptr(out8), stream_ptr())
        ctx.unit = (dx1, dx2, dxhat, dgram, dprob, wpart)
        ctx.cfg = (b, k, c, nr, nblk, [float(v) for v in lam], b2 is not None, b2r is not None)
        ctx.gram_shape, ctx.prob_shape = tuple(gram.shape), tuple(prob.shape)
        ctx.w_final = (_leaves(w2, b2), _leaves(w2r, b2r))
        loss, terms = out8[0], out8[1:]
        ctx.mark_non_differentiable(terms, logp, our_reg)
        ctx.set_materialize_grads(False)
        return loss, terms, logp, our_reg

    @staticmethod
    def backward(ctx, gout, _gt=None, _gl=None, _gr=None):
        b, k, c, nr, nblk, lam, has_b2, has_b2r = ctx.cfg
        dx1, dx2, dxhat, dgram, dprob, wpart = ctx.unit
        if ctx.final is not None:                    # the loss value: now, i.e. into the flush when the stream defers
            parts, gram, prob, wts, out8 = ctx.final
            call("igcn_loss_final", ptr(parts), nblk, ptr(gram), gram.numel() // 4, ptr(prob), prob.numel(), ptr(wts),
                 ptr(out8), stream_ptr())
            _keep(parts)
        dev = dx1.device
        wcols = c * k + c + nr * k + nr
        # kept until the flush: the sums below are QUEUED, and a backward sweep that does not ask for the heads' gradients
        # (the second sweep of train.backward_two_buckets visits this node again) drops dwb on return — its memory would be
        # handed to a later tensor of the sweep and the flush would write the sums into it
        dwb = _keep(torch.empty(wcols, dtype=torch.float32, device=dev))
        _keep(wpart)
        o1, o2 = c * k + c, nr * k + nr
        with _immediate(ctx.w_final[0]):
            call("igcn_reduce_rows_final", ptr(wpart), nblk, wcols, o1, ptr(dwb), stream_ptr())
        with _immediate(ctx.w_final[1]):
            call("igcn_reduce_rows_final", wpart.data_ptr() + 4 * o1, nblk, wcols, o2, ptr(dwb[o1:]), stream_ptr())
        gout = _f32(gout).reshape(1)
        UNIT_DGRAM.clear()
        if gout.data_ptr() in UNIT_GRAD_PTRS:
            UNIT_DGRAM[dgram.data_ptr()] = unit_dgram(lam)
        else:                                        # every gradient is linear in the upstream one
            dx1, dx2, dxhat, dgram, dprob, dwb = (t * gout for t in (dx1, dx2, dxhat, dgram, dprob, dwb))
        gs, ps = ctx.gram_shape, ctx.prob_shape
        dgram = dgram.view(2, 2) if gs == (2, 2) else dgram.view(1, 4).expand(gs[0], 4)
        dprob = dprob.view(()) if ps == () else dprob.expand(ps)
        dw2, db2 = dwb[:c * k].view(c, k), dwb[c * k:o1]
        dw2r, db2r = dwb[o1:o1 + nr * k].view(nr, k), dwb[o1 + nr * k:]
        return (dx1, None, dw2, db2 if has_b2 else None, dx2, None, dw2r, db2r if has_b2r else None, None, None, dxhat, None,
                dgram, dprob, None, None, None, None, None)


# =================================================================================================
# Cross-attention: projections + attention core
# =================================================================================================
def attn_core_supported(d, h, lq, lk):
    lib = _lib.load()
    return bool(lib.igcn_attn_core_lds_bytes(d, h, lq, lk, 0)) and bool(lib.igcn_attn_core_lds_bytes(d, h, lq, lk, 1))


def attn_core_bf16(d, h, lq, lk):
    """The bf16-operand core (v_mfma_f32_16x16x32_bf16) covers this shape (head_dim 16); IGCN_ATTN_FP32_CORE=1 (read
    once at load) keeps the fp32 core under bf16 feature transforms (A/B runs)."""
    return bool(_lib.load().igcn_attn_core_bf16_supported(d, h, lq, lk))


class AttentionCore(torch.autograd.Function):
    """softmax(q k^T/sqrt(hd)) v per head on the projection outputs in place: q [B,Lq,D], kv [B,Lk,2D] -> [B,Lq,D]."""

    @staticmethod
    def forward(ctx, q, kv, heads, bf16=False):
        q, kv = _f32(q), _f32(kv)
        b, lq, d = q.shape
        lk = kv.shape[1]
        o = torch.empty_like(q)
        lse = torch.empty(b, heads, lq, dtype=torch.float32, device=q.device)
        core16 = bool(bf16) and attn_core_bf16(d, heads, lq, lk)
        call("igcn_attn_core_bf16_fwd" if core16 else "igcn_attn_core_fwd", b, d, heads, lq, lk, ptr(q), ptr(kv), ptr(o),
             ptr(lse), stream_ptr())
        ctx.save_for_backward(q, kv, o, lse)
        ctx.heads, ctx.core16 = heads, core16
        return o

    @staticmethod
    def backward(ctx, dout):
        q, kv, o, lse = ctx.saved_tensors
        dout = _f32(dout)
        b, lq, d = q.shape
        lk = kv.shape[1]
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        nscr = int(_lib.load().igcn_attn_core_bwd_scratch_floats(b, ctx.heads, lq))
        scratch = torch.empty(nscr, dtype=torch.float32, device=q.device)
        call("igcn_attn_core_bf16_bwd" if ctx.core16 else "igcn_attn_core_bwd", b, d, ctx.heads, lq, lk, ptr(q), ptr(kv),
             ptr(o), ptr(lse), ptr(dout), ptr(dq), ptr(dkv), ptr(scratch), stream_ptr())
        return dq, dkv, None, None
