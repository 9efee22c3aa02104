"""RCCL gradient exchange behind the C ABI (igcn_comm_*, include/igcn.h) for the data-parallel train step.

One process per GPU.  ``Comm`` creates the RCCL communicator through libigcn (the unique id is broadcast with the
already-initialised ``torch.distributed`` group — any backend; it is only the bootstrap channel), and
``all_reduce_(flat)`` enqueues ONE ncclAllReduce(sum) over the flat fp32 gradient bucket on the CURRENT torch stream:
between the kernel that packs the gradients and the Adam kernel, with no hop to another stream, and capturable into
the step's hipGraph.  The reference has no multi-GPU code; semantics are DDP's: local BatchNorm
statistics and batch-level losses, gradients averaged (the 1/W lives in the Adam kernel's ``grad_scale``).
"""
import ctypes

import torch

from . import _lib
from ._lib import call, ptr, stream_ptr


class Comm:
    def __init__(self, rank=None, world_size=None, group=None):
        dist = torch.distributed
        if rank is None or world_size is None:
            if not dist.is_initialized():
                raise _lib.IgcnError("Comm: give (rank, world_size) or initialise torch.distributed first")
            rank, world_size = dist.get_rank(group), dist.get_world_size(group)
        self.rank, self.world_size = int(rank), int(world_size)
        lib = _lib.load()
        nbytes = int(lib.igcn_comm_unique_id_bytes())
        buf = (ctypes.c_char * nbytes)()
        if self.rank == 0:
            call("igcn_comm_get_unique_id", ctypes.addressof(buf))
        if self.world_size > 1:
            box = [bytes(buf)]
            dist.broadcast_object_list(box, src=0, group=group)          # host-side bootstrap of 128 bytes
            buf = (ctypes.c_char * nbytes).from_buffer_copy(box[0])
        handle = ctypes.c_void_p()
        call("igcn_comm_init", self.world_size, self.rank, ctypes.addressof(buf), ctypes.addressof(handle))
        self._handle = handle

    def all_reduce_(self, flat):
        """In-place sum over ranks of a contiguous fp32 device tensor, on the current stream."""
        if flat.dtype != torch.float32:
            raise _lib.IgcnError("Comm.all_reduce_: fp32 bucket expected")
        call("igcn_comm_allreduce", self._handle, ptr(flat), flat.numel(), stream_ptr())
        return flat

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            torch.cuda.synchronize()
            call("igcn_comm_destroy", self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 — interpreter shutdown
            pass
