"""Minimal ``Data`` / ``Batch`` / ``DataLoader`` with the attribute surface the IG-GCN hot path uses.

Mirrors the collation contract of the reference's ``batch.py:24-123`` (``Batch.from_data_list``) and
``dataloader.py:11-48`` for the keys of the brain-graph samples built at ``sgcn_data.py:262-282``:
tensors are concatenated along dim 0, except keys containing ``index`` which are concatenated along
the last dim and offset by the cumulative node count (PyG ``__cat_dim__`` / ``__inc__``); ``batch[i]``
is the graph id of node ``i``.  The NGNN sub-graph keys of ``batch.py:47-87`` are out of scope.

A ``Batch`` additionally carries the host-known sizes (``num_graphs``, ``ptr``, ``edge_ptr``) so the hot
path never needs the ``batch[-1].item()`` device sync of ``batch.py:188-191``.
"""
import re

import torch


class Data:
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    # --- PyG-style dict surface -------------------------------------------------------------
    @property
    def keys(self):
        return [k for k, v in self.__dict__.items() if not k.startswith("_") and v is not None]

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.keys

    def __iter__(self):
        for k in sorted(self.keys):
            yield k, getattr(self, k)

    @property
    def num_nodes(self):
        x = getattr(self, "x", None)
        if x is not None:
            return x.size(0)
        ei = getattr(self, "edge_index", None)
        return int(ei.max()) + 1 if ei is not None and ei.numel() else None

    @property
    def num_edges(self):
        ei = getattr(self, "edge_index", None)
        return ei.size(1) if ei is not None else 0

    def __cat_dim__(self, key, value):
        return -1 if bool(re.search("(index|face)", key)) else 0

    def __inc__(self, key, value):
        if "batch" in key:                              # PyG 2.0.2: per-graph assignment vectors count graphs
            return int(value.max()) + 1
        return self.num_nodes if bool(re.search("(index|face)", key)) else 0

    def to(self, device, non_blocking=False):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device, non_blocking=non_blocking))
        return self

    def contiguous(self):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.contiguous())
        return self

    def __repr__(self):
        parts = [f"{k}={list(v.shape) if torch.is_tensor(v) else v}" for k, v in self]
        return f"{type(self).__name__}({', '.join(parts)})"


class Batch(Data):
    """Block-diagonal mini-batch of graphs."""

    def __init__(self, batch=None, **kwargs):
        super().__init__(**kwargs)
        self.batch = batch
        self._num_graphs = None

    @staticmethod
    def from_data_list(data_list, follow_batch=()):
        keys = sorted(set().union(*[set(d.keys) for d in data_list]))
        assert "batch" not in keys
        out = Batch()
        cols = {k: [] for k in keys}
        cumsum = {k: 0 for k in keys}                   # batch.py:41,89: running increment per key
        node_off, node_ptr, edge_ptr, bvec = 0, [0], [0], []
        for i, d in enumerate(data_list):
            n = d.num_nodes
            for k in d.keys:
                item = d[k]
                if torch.is_tensor(item) and item.dtype != torch.bool and cumsum[k]:
                    item = item + cumsum[k]
                cumsum[k] = cumsum[k] + d.__inc__(k, item)
                cols[k].append(item)
            bvec.append(torch.full((n,), i, dtype=torch.long))
            node_off += n
            node_ptr.append(node_off)
            edge_ptr.append(edge_ptr[-1] + d.num_edges)
        for k in keys:
            first = cols[k][0]
            if torch.is_tensor(first):
                out[k] = torch.cat(cols[k], dim=data_list[0].__cat_dim__(k, first))
            elif isinstance(first, (int, float)):
                out[k] = torch.tensor(cols[k])
            else:
                out[k] = cols[k]
        out.batch = torch.cat(bvec)
        out._num_graphs = len(data_list)
        out.ptr = torch.tensor(node_ptr, dtype=torch.long)
        out.edge_ptr = torch.tensor(edge_ptr, dtype=torch.long)
        out._max_nodes = max(b - a for a, b in zip(node_ptr[:-1], node_ptr[1:]))
        out._max_edges = max(b - a for a, b in zip(edge_ptr[:-1], edge_ptr[1:]))
        return out.contiguous()

    @property
    def num_graphs(self):
        if self._num_graphs is not None:
            return self._num_graphs
        # foreign construction: fall back to the reference's device read (batch.py:188-191)
        self._num_graphs = int(self.batch[-1]) + 1
        return self._num_graphs

    @property
    def keys(self):
        return [k for k in super().keys]


class DataLoader(torch.utils.data.DataLoader):
    """torch DataLoader whose collate is ``Batch.from_data_list`` (reference dataloader.py:24-48)."""

    def __init__(self, dataset, batch_size=1, shuffle=False, **kwargs):
        super().__init__(dataset, batch_size, shuffle,
                         collate_fn=lambda items: Batch.from_data_list(items), **kwargs)
