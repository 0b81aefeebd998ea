"""The C-ABI collectives (include/moviigen_hip.h: mg_comm_*, mg_sp_all_to_all*, mg_sp_all_gather, mg_shard_all_gather)
as a transport for the sequence-parallel exchange: an RCCL communicator owned by libmoviigen_hip.so, every collective
enqueued directly on the caller's HIP stream.

`torch.distributed` (backend "nccl" = RCCL) remains the default transport; this one is selected with
MOVIIGEN_SP_TRANSPORT=rccl_direct (or `DirectComm(group)` by hand).  torch.distributed is still what bootstraps it: the
128-byte RCCL unique id travels from the group's rank 0 to the other ranks by object broadcast."""
import ctypes
import os

import torch
import torch.distributed as dist

from ..backend import lib


def enabled():
    return os.environ.get('MOVIIGEN_SP_TRANSPORT', '') == 'rccl_direct'


class DirectComm:
    """one RCCL communicator over the ranks of `group` (default WORLD), created collectively."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised (it carries the RCCL unique id)')
        self.group = group if group is not None else dist.group.WORLD
        self.size, self.rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            lib.call('mg_comm_unique_id', uid)
        box = [bytes(uid.raw) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0), group=self.group)
        uid = ctypes.create_string_buffer(box[0], 128)
        handle = ctypes.c_void_p()
        lib.call('mg_comm_create', uid, self.size, self.rank, ctypes.byref(handle))
        self.handle = handle

    def _st(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def all_to_all(self, recv, send):
        """equal chunks along dim 0 (all_to_all_single semantics), contiguous device tensors."""
        assert recv.is_contiguous() and send.is_contiguous() and recv.numel() == send.numel()
        per_peer = send.numel() * send.element_size() // self.size
        lib.call('mg_sp_all_to_all', self.handle, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()),
                 per_peer, self._st())

    def all_to_all_4d(self, x, out, heads, head_dim, seq_to_head, workspace):
        lib.call('mg_sp_all_to_all_4d_bf16', self.handle, ctypes.c_void_p(x.data_ptr()), x.stride(0), x.shape[0], int(heads),
                 int(head_dim), int(bool(seq_to_head)), ctypes.c_void_p(out.data_ptr()), out.stride(0),
                 ctypes.c_void_p(workspace.data_ptr()), self._st())
        return out

    def all_gather(self, out, x):
        assert out.is_contiguous() and x.is_contiguous() and out.numel() == x.numel() * self.size
        lib.call('mg_sp_all_gather', self.handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                 x.numel() * x.element_size(), self._st())
        return out

    def shard_all_gather(self, full, shard):
        """per-block parameter gather of the block-sharded weights (mg_shard_all_gather), on the CURRENT stream."""
        assert full.is_contiguous() and shard.is_contiguous() and full.numel() == shard.numel() * self.size
        lib.call('mg_shard_all_gather', self.handle, ctypes.c_void_p(shard.data_ptr()), ctypes.c_void_p(full.data_ptr()),
                 shard.numel() * shard.element_size(), self._st())
        return full

    def destroy(self):
        if self.handle:
            lib.call('mg_comm_destroy', self.handle)
            self.handle = None


_COMMS = {}     # ProcessGroup object -> DirectComm (the dict holds the group, so the key cannot be recycled)


def comm_for(group):
    """the DirectComm of a process group, created on first use (collectively: every rank of the group gets here at
    the same point of the forward).  None and dist.group.WORLD name the same ranks and share one communicator."""
    group = group if group is not None else dist.group.WORLD
    comm = _COMMS.get(group)
    if comm is None:
        if not _COMMS:
            import atexit
            atexit.register(destroy_all)
        comm = _COMMS[group] = DirectComm(group)
    return comm


def destroy_all():
    """ncclCommDestroy for every communicator created through comm_for (registered with atexit on first use)."""
    for comm in list(_COMMS.values()):
        try:
            comm.destroy()
        except Exception:       # the HIP runtime may already be gone at interpreter exit
            pass
    _COMMS.clear()
