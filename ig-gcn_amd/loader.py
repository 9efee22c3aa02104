"""Feeding the train step: the reference's loop starts at ``for data in loader: data = data.to(device)``
(kernel/train_eval_sgcn_img_snps.py:515-517) over ``Batch.from_data_list`` (batch.py:24-123) — a Python loop per graph
per key, 13 ms per 256 graphs on an 8-core host, i.e. 13x the 1 ms step it feeds.

Brain-graph datasets are UNIFORM (every subject: R ROIs, the same attribute set, and after the GDC pre-transform R*k
edges), so the collation is a gather plus one offset add:

* ``UniformGraphStore``  the whole dataset as stacked tensors (host — pinned — or device).  ``batch(idx)`` yields the same
  bytes as ``Batch.from_data_list([dataset[i] for i in idx])`` (bit for bit: tests/test_loader.py) with a handful of
  ``index_select`` calls and ``edge_index += arange(B) * R``.
* ``collate_uniform``    the same for an ad-hoc list of ``Data`` (drop-in collate function; falls back to
  ``Batch.from_data_list`` when the list is not uniform).
* ``HostFeeder``         a producer thread: store on the host -> pinned staging -> non-blocking H2D on a copy stream
  into a ring of device staging batches; the consumer waits on the batch's event, copies it into the graphed step's
  static inputs (``GraphedTrainStep.load``) and replays.
* ``DeviceFeeder``       store on the device: collation is a few device gathers on the launch stream.
* ``DeviceGdcFeeder``    dense connectivity on the device: GDC pre-transform + collation of batch k + 1 on a second
  stream while step k replays.
* ``EpochIndex``         the index order all feeders draw from: ``randperm`` per epoch, every subject once (the
  semantics of ``DataLoader(shuffle=True)``), ragged tail dropped or handed out as a shorter slice.
"""
import queue
import threading

import torch

from .data import Batch, Data

# keys whose per-graph value is a 1-D vector that the reference's collation concatenates (batch.py:110): [n] -> [n*B]
_INDEX_KEY = ("index", "face")


def _is_index_key(key):
    return any(t in key for t in _INDEX_KEY)


def _uniform(data_list):
    d0 = data_list[0]
    keys = sorted(d0.keys)
    n, e = d0.num_nodes, d0.num_edges
    for d in data_list[1:]:
        if sorted(d.keys) != keys or d.num_nodes != n or d.num_edges != e:
            return None
        for k in keys:
            a, b = d0[k], d[k]
            if torch.is_tensor(a) != torch.is_tensor(b) or (torch.is_tensor(a) and (a.shape != b.shape or
                                                                                    a.dtype != b.dtype)):
                return None
    return keys, n, e


def collate_uniform(data_list):
    """``Batch.from_data_list`` for a list of graphs with identical shapes: one ``torch.stack`` per key and one offset
    add for the ``*index*`` keys instead of a Python loop per graph per key.  Identical output (bit for bit)."""
    info = _uniform(data_list) if len(data_list) else None
    if info is None or any(not torch.is_tensor(data_list[0][k]) for k in info[0]):
        return Batch.from_data_list(data_list)
    keys, n, e = info
    b = len(data_list)
    out = Batch()
    for k in keys:
        st = torch.stack([d[k] for d in data_list])                       # [B, ...]
        first = data_list[0][k]
        if _is_index_key(k):                                              # cat along the LAST dim, offset by nodes
            if st.dtype != torch.bool:
                st = st + (torch.arange(b, dtype=st.dtype) * n).view(b, *([1] * (st.dim() - 1)))
            st = st.movedim(0, -2).reshape(*first.shape[:-1], b * first.shape[-1])
        else:                                                             # cat along dim 0
            st = st.reshape(b * first.shape[0], *first.shape[1:]) if first.dim() else st
        out[k] = st.contiguous()
    out.batch = torch.arange(b, dtype=torch.long).repeat_interleave(n)
    _finish(out, b, n, e, "cpu")
    return out


def _finish(out, b, n, e, device):
    out._num_graphs = b
    out.ptr = torch.arange(b + 1, dtype=torch.long, device=device) * n
    out.edge_ptr = torch.arange(b + 1, dtype=torch.long, device=device) * e
    out._max_nodes, out._max_edges = n, e


class UniformGraphStore:
    """A dataset of uniform graphs as stacked tensors: ``cols[key]`` = [S, ...per-graph shape...]."""

    def __init__(self, data_list, device="cpu", pin=False):
        info = _uniform(data_list)
        if info is None:
            raise ValueError("UniformGraphStore needs graphs of identical shapes and keys")
        self.keys, self.nodes, self.edges = info
        self.size = len(data_list)
        self.device = torch.device(device)
        self.cols, self.shapes = {}, {}
        for k in self.keys:
            first = data_list[0][k]
            if not torch.is_tensor(first):
                raise ValueError(f"non-tensor attribute {k!r}")
            st = torch.stack([d[k] for d in data_list]).contiguous()
            self.shapes[k] = tuple(first.shape)
            st = st.to(self.device)
            if pin and self.device.type == "cpu":
                st = st.pin_memory()
            self.cols[k] = st

    def batch(self, idx, out=None):
        """Collate the graphs ``idx`` (int64 tensor on the store's device).  ``out``: a Batch made by an earlier call
        (same batch size) whose tensors are overwritten in place — stable addresses for pinned staging."""
        b = int(idx.numel())
        dev = self.device
        n, e = self.nodes, self.edges
        res = out if out is not None else Batch()
        for k in self.keys:
            shp = self.shapes[k]
            if out is not None and not _is_index_key(k):                  # gather straight into the destination
                torch.index_select(self.cols[k], 0, idx, out=getattr(res, k).view(b, *shp))
                continue
            sel = torch.index_select(self.cols[k], 0, idx)                # [B, *shp]
            if _is_index_key(k):
                if sel.dtype != torch.bool:
                    sel += (torch.arange(b, dtype=sel.dtype, device=dev) * n).view(b, *([1] * len(shp)))
                val = sel.movedim(0, -2).reshape(*shp[:-1], b * shp[-1])
            else:
                val = sel.reshape(b * shp[0], *shp[1:]) if len(shp) else sel
            if out is not None:
                getattr(res, k).copy_(val)
            else:
                res[k] = val.contiguous()
        if out is None:
            res.batch = torch.arange(b, dtype=torch.long, device=dev).repeat_interleave(n)
            _finish(res, b, n, e, dev)
        return res

    def one_launch(self):
        """Whether ``gather_into`` applies: a device store whose rows are whole 4-byte words and whose ``*index*`` keys
        are [2, E] int64 (every brain-graph dataset; a bool or int8 attribute takes ``batch(out=...)``)."""
        for k in self.keys:
            c = self.cols[k]
            if _is_index_key(k):
                if c.dtype != torch.int64 or len(self.shapes[k]) != 2 or self.shapes[k][0] != 2:
                    return False
            elif (c[0].numel() * c.element_size()) % 4:
                return False
        return self.device.type == "cuda"

    def gather_into(self, idx, out):
        """``batch(idx, out=out)`` as ONE kernel launch (igcn_gather_batch) — same bytes (tests/test_gpu_epoch.py)."""
        gather_rows(idx, self.nodes, [(getattr(out, k), self.cols[k], _is_index_key(k)) for k in self.keys])
        return out


def gather_rows(idx, nodes, items):
    """Every key of a batch in ONE launch (igcn_gather_batch) on the current stream.  ``items``: (dst, src, is_index)
    with src [S, ...per-graph...] and dst the collated tensor — rows ``src[idx[b]]`` back to back, or for an ``*index*``
    key ([S, 2, E] int64) the [2, B E] concatenation with graph b offset by ``b * nodes``."""
    import ctypes
    from ._lib import call, ptr, stream_ptr
    b = int(idx.numel())
    if not (idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous() and idx.dim() == 1):
        raise ValueError("gather_rows: the subjects must be a contiguous int64 vector on the device")
    for k in range(0, len(items), 16):
        chunk = items[k:k + 16]
        n = len(chunk)
        for dst, src, _ in chunk:
            if not (dst.is_cuda and src.is_cuda and dst.is_contiguous() and src.is_contiguous()
                    and dst.dtype == src.dtype and dst.numel() == b * src[0].numel() and src.shape[0] == items[0][1].shape[0]):
                raise ValueError("gather_rows: destination does not match B rows of the source")
        d = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _, _ in chunk])
        sp = (ctypes.c_void_p * n)(*[t.data_ptr() for _, t, _ in chunk])
        rb = (ctypes.c_int64 * n)(*[t[0].numel() * t.element_size() for _, t, _ in chunk])
        kind = (ctypes.c_int * n)(*[1 if ix else 0 for _, _, ix in chunk])
        call("igcn_gather_batch", n, b, int(nodes), int(items[0][1].shape[0]), ptr(idx), d, sp, rb, kind, stream_ptr())


def _like(batch, device, pin=False):
    """A copy of ``batch`` with every tensor re-allocated on ``device`` (a staging slot).  The contents are COPIED, not
    left empty: ``UniformGraphStore.batch(out=slot)`` rewrites the per-graph keys only, and the constant tensors of a
    uniform batch (``ptr``, ``edge_ptr``, ``batch``) must be valid in every slot — the plan build reads them."""
    out = Batch()
    for k, v in batch.__dict__.items():
        if torch.is_tensor(v):
            t = torch.empty(v.shape, dtype=v.dtype, device=device)
            if pin and torch.device(device).type == "cpu":
                t = t.pin_memory()
            t.copy_(v)
            setattr(out, k, t)
        else:
            setattr(out, k, v)
    return out


def _copy_into(dst, src, non_blocking=True):
    pairs = [(getattr(dst, k), v) for k, v in src.__dict__.items() if torch.is_tensor(v)]
    if pairs and all(d.is_cuda and v.is_cuda for d, v in pairs):
        from ._lib import copy_multi
        copy_multi(pairs)                               # device to device: one launch (igcn_copy_multi)
        return
    for d, v in pairs:
        d.copy_(v, non_blocking=non_blocking)


class EpochIndex:
    """The index stream of ``DataLoader(dataset, batch_size, shuffle=...)`` (kernel/train_eval_sgcn_img_snps.py:96-97)
    for fixed-shape consumers: per epoch one ``torch.randperm(size)`` (``shuffle=True``; drawn from the seeded generator,
    so the order of every epoch is reproducible) or ``arange(size)``, cut into consecutive slices of ``batch_size`` —
    every subject exactly once per epoch.  ``drop_last=True`` (default: the captured step needs one shape) drops the
    ``size % batch_size`` subjects of the ragged tail of each epoch, like ``DataLoader(drop_last=True)``;
    ``drop_last=False`` yields the tail as a shorter index tensor (route it through ``train.EpochTrainer``)."""

    def __init__(self, size, batch_size, seed=0, shuffle=True, drop_last=True, device="cpu"):
        if batch_size > size and drop_last:
            raise ValueError(f"batch_size {batch_size} exceeds the dataset ({size} subjects)")
        self.size, self.bsz, self.shuffle, self.drop_last = int(size), int(batch_size), shuffle, drop_last
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        self.epoch, self._order, self._pos = 0, None, 0

    def per_epoch(self):
        return self.size // self.bsz if self.drop_last else -(-self.size // self.bsz)

    def next(self):
        if self._order is None or self._pos >= (self.size - self.bsz + 1 if self.drop_last else self.size):
            self._order = torch.randperm(self.size, generator=self.gen, device=self.device) if self.shuffle \
                else torch.arange(self.size, device=self.device)
            self._pos = 0
            self.epoch += 1
        idx = self._order[self._pos:self._pos + self.bsz]
        self._pos += self.bsz
        return idx


class HostFeeder:
    """Iterator over device-resident batches produced by a host thread: collate (vectorised) into pinned memory,
    upload on a copy stream, hand over with an event.  ``depth`` staging slots in flight.

        for batch in feeder:                       # batch.ready: the upload's event
            torch.cuda.current_stream().wait_event(batch.ready)
            step.load(batch); batch.release(); step()
    """

    def __init__(self, store, batch_size, device, steps, depth=3, seed=0, shuffle=True):
        if store.device.type != "cpu":
            raise ValueError("HostFeeder reads a host-resident store")
        self.store, self.bsz, self.device, self.steps = store, int(batch_size), torch.device(device), int(steps)
        self.index = EpochIndex(store.size, batch_size, seed, shuffle)       # every subject once per epoch
        proto = store.batch(torch.arange(self.bsz))
        self.host = [_like(proto, "cpu", pin=True) for _ in range(depth)]
        self.dev = [_like(proto, self.device) for _ in range(depth)]
        self.uploaded = [torch.cuda.Event() for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.q = queue.Queue()                                   # filled slots, in order
        self.free = queue.Queue()                                # (slot, event after the consumer's copy-out | None)
        for k in range(depth):
            self.free.put((k, None))
        self.thread = threading.Thread(target=self._produce, daemon=True)
        self.error = None

    def _indices(self):
        return self.index.next()

    def _produce(self):
        try:
            torch.cuda.set_device(self.device)
            # gathers of a few hundred KB: one thread.  (A fresh thread would otherwise spin up its own OpenMP team of
            # os.cpu_count() workers beside the main thread's — on a box whose CPU share is smaller than its CPU count
            # the two teams' spinning workers cost 10 ms per batch.)
            torch.set_num_threads(1)
            for _ in range(self.steps):
                k, done = self.free.get()                       # a slot the consumer has released ...
                if done is not None:
                    done.synchronize()                          # ... and whose copy-out has finished on the device
                self.store.batch(self._indices(), out=self.host[k])
                with torch.cuda.stream(self.copy_stream):
                    _copy_into(self.dev[k], self.host[k], non_blocking=True)
                    self.uploaded[k].record(self.copy_stream)
                self.q.put(k)
        except Exception as exc:                                # noqa: BLE001 — surface it in the consumer
            self.error = exc
            self.q.put(None)

    def __iter__(self):
        self.thread.start()
        for _ in range(self.steps):
            k = self.q.get()
            if k is None:
                raise self.error
            b = self.dev[k]
            b.ready = self.uploaded[k]
            b.release = lambda k=k: self._release(k)
            yield b
        self.thread.join()

    def _release(self, k):
        ev = torch.cuda.Event()
        ev.record()                                             # behind the consumer's copies out of slot k
        self.free.put((k, ev))


class _AheadOnSideStream:
    """Batches built on the device ONE AHEAD of the train step and on a stream of their own: batch k + 1 has no
    dependency on step k, so its gathers (and, for ``DeviceGdcFeeder``, the graph-diffusion transform) run beside the
    replay of step k instead of in front of step k + 1 on the launch stream.  Hand-over by events:

        for batch in feeder:
            torch.cuda.current_stream().wait_event(batch.ready)      # the side stream has finished this slot
            step.load(batch); batch.release(); step()                # release: the slot may be refilled

    ``depth`` output slots (>= 2) are allocated once; temporaries live on the side stream.  Subclasses provide
    ``self.index`` (an ``EpochIndex`` on the device), ``_build(idx) -> Batch`` (the prototype of the slots) and may
    override ``_fill(slot, idx)`` to write a slot in place."""

    def _init_slots(self, device, steps, depth):
        self.steps, self.depth = int(steps), max(2, int(depth))
        self.side = torch.cuda.Stream(device=device)
        proto = self._build(torch.arange(self.bsz, device=device), check=True)
        self.slots = [_like(proto, device) for _ in range(self.depth)]
        self.ready = [torch.cuda.Event() for _ in range(self.depth)]
        self.consumed = [None] * self.depth
        torch.cuda.synchronize(device)

    def _produce(self, k):
        slot = k % self.depth
        if self.consumed[slot] is not None:
            self.side.wait_event(self.consumed[slot])   # the consumer has copied the slot's previous batch out
        with torch.cuda.stream(self.side):              # everything below is side-stream work: no wait on the step
            self._fill(self.slots[slot], self.index.next())
            self.ready[slot].record(self.side)

    def _fill(self, slot, idx):
        _copy_into(slot, self._build(idx), non_blocking=True)

    def __iter__(self):
        self._produce(0)
        for k in range(self.steps):
            if k + 1 < self.steps:
                self._produce(k + 1)                    # enqueued before step k is: overlaps its replay
            slot = k % self.depth
            b = self.slots[slot]
            b.ready = self.ready[slot]
            b.release = lambda slot=slot: self._release(slot)
            yield b

    def _release(self, slot):
        ev = torch.cuda.Event()
        ev.record()                                     # behind the consumer's copies out of the slot
        self.consumed[slot] = ev


class DeviceFeeder(_AheadOnSideStream):
    """Store resident in HBM: every batch is ONE gather launch (igcn_gather_batch: all keys, index offsets included) —
    no host data path.

    ``into=step.data`` (the static inputs of a ``GraphedTrainStep``): the gather runs on the CURRENT stream straight into
    the step's inputs, in front of the replay — no staging slot, no copy, no second queue (a 5 us kernel on a side
    stream costs the replay more than it hides: DESIGN §6).  Without ``into`` batches are made one ahead on a side
    stream into ``depth`` slots (see ``_AheadOnSideStream`` for the hand-over protocol).  Either way the consumer is

        for batch in feeder:
            torch.cuda.current_stream().wait_event(batch.ready); step.load(batch); batch.release(); step()
    """

    def __init__(self, store, batch_size, steps, seed=0, shuffle=True, depth=2, into=None):
        if store.device.type != "cuda":
            raise ValueError("DeviceFeeder reads a device-resident store")
        self.store, self.bsz, self.into = store, int(batch_size), into
        self.index = EpochIndex(store.size, batch_size, seed, shuffle, device=store.device)
        if into is None:
            self._init_slots(store.device, steps, depth)
        else:
            self.steps = int(steps)
            if int(into.num_graphs) != self.bsz:
                raise ValueError(f"DeviceFeeder: the target holds {into.num_graphs} graphs, batch_size is {self.bsz}")

    def _build(self, idx, check=False):
        return self.store.batch(idx)

    def _fill(self, slot, idx):
        if self.store.one_launch():
            self.store.gather_into(idx, slot)
        else:
            self.store.batch(idx, out=slot)

    def __iter__(self):
        if self.into is None:
            yield from super().__iter__()
            return
        done = torch.cuda.Event()
        for _ in range(self.steps):
            with torch.no_grad():
                self._fill(self.into, self.index.next())
            done.record()                               # same stream as the consumer: the wait is free
            self.into.ready, self.into.release = done, _nothing
            yield self.into


def _nothing():
    return None


class DeviceGdcFeeder(_AheadOnSideStream):
    """Dense connectivity matrices resident in HBM -> graph-diffusion pre-transform (util_gdc.py:7-31,71-86) + collation
    (batch.py:24-123) of every batch ON THE DEVICE, one batch ahead of the train step on a side stream: the fp64
    Gauss-Jordan of batch k + 1 (igcn_gdc_topk, one workgroup per graph) runs beside the replay of step k.

    ``cols``: per-subject tensors [S, ...] on the device (x [S, R, H0], snps_feat, y, clini_score, tsne_fdim, clust_y)."""

    def __init__(self, adj, cols, batch_size, steps, top_k=3, alpha=0.05, seed=0, shuffle=True, depth=2):
        from .gdc import batch_from_dense
        if adj.device.type != "cuda":
            raise ValueError("DeviceGdcFeeder reads device-resident matrices")
        self.adj, self.cols, self.bsz = adj, cols, int(batch_size)
        self.top_k, self.alpha = int(top_k), float(alpha)
        self.index = EpochIndex(adj.shape[0], batch_size, seed, shuffle, device=adj.device)
        self._make = batch_from_dense
        self._counts = torch.empty(self.bsz, dtype=torch.int32, device=adj.device)
        if adj.dtype != torch.float32 or not adj.is_contiguous():
            raise ValueError("DeviceGdcFeeder: adj must be a contiguous float32 [S, R, R]")
        self._init_slots(adj.device, steps, depth)

    def _build(self, idx, check=False):
        b = int(idx.numel())
        sel = {k: torch.index_select(v, 0, idx) for k, v in self.cols.items()}
        x = sel.pop("x")
        per = {}
        for k, v in sel.items():
            per[k] = v.reshape(b, -1) if k in ("snps_feat", "tsne_fdim", "clini_score") else v.reshape(-1)
        out = self._make(torch.index_select(self.adj, 0, idx), x, top_k=self.top_k, alpha=self.alpha, check=check,
                         **per)
        out.A = None                                    # the dense matrices are not an input of the step
        return out

    def _fill(self, slot, idx):
        """Three launches on the side stream: the per-subject attributes (igcn_gather_batch), the diffusion of the
        drawn matrices read in place (igcn_gdc_topk_of) straight into the slot's edge list, the edge offsets."""
        from ._lib import call, ptr, stream_ptr
        b, r = int(idx.numel()), int(self.adj.shape[1])
        gather_rows(idx, r, [(getattr(slot, k), v, False) for k, v in self.cols.items()])
        call("igcn_gdc_topk_of", b, r, self.top_k, self.alpha, ptr(self.adj), int(self.adj.shape[0]), ptr(idx), ptr(slot.edge_index),
             ptr(slot.edge_attr), ptr(self._counts), stream_ptr())
        torch.cumsum(self._counts, 0, out=slot.edge_ptr[1:])


def as_data_list(batch_size, **kw):
    """Convenience for tests: ``synth.brain_graph_list`` without the dense [R,R] adjacency (not on the hot path)."""
    from . import synth
    graphs = synth.brain_graph_list(batch_size, **kw)
    return [Data(**{k: v for k, v in g.__dict__.items() if k != "A"}) for g in graphs]
