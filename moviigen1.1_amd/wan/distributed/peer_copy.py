"""Peer-copy transport for the Ulysses exchange (MOVIIGEN_SP_TRANSPORT=peer_copy): the all-to-all as ONE-SIDED device
copies into the peers' receive buffers, `hipMemcpyAsync` device-to-device on the communication stream — the copy
engines (SDMA over the xGMI links) move the bytes, no compute unit is taken from the attention kernel that runs
meanwhile.  (RCCL's all-to-all is a kernel: with the persistent attention kernel holding one 256-register workgroup on
every CU, its copy workgroups have to wait for — or steal — CUs.  Which of the two transports wins on a real 8-GPU node
is a measurement this repository could not take yet; tools/scale_sweep.sh runs both.)

How it works.  The receive buffers of a `HeadExchange` are persistent, so they are registered ONCE: every rank exports
an IPC handle per buffer (torch's CUDA-IPC reduction = hipIpcGetMemHandle; dmabuf-based, `HSA_ENABLE_IPC_MODE_LEGACY=0`),
the handles travel by `all_gather_object`, and every rank maps its peers' buffers.  An exchange is then

    flag all-reduce   (every rank has reached this exchange on its comm stream => its previous consumer of the
                       receive buffer — ordered before the pack kernel this exchange waited for — is done: the buffer
                       may be overwritten)
    P copies          chunk p of my send buffer -> slot [my rank] of rank p's receive buffer (p == me: local copy)
    flag all-reduce   (stream-ordered after the copies on every rank => all slots of MY buffer have landed)

The two flag all-reduces are 4-byte RCCL collectives (or, for the gloo tests on one GPU, a stream sync + host barrier).
torch.distributed stays the control plane; this file moves no tensor through it.
"""
import os

import torch
import torch.distributed as dist

from . import collectives


def enabled():
    return os.environ.get('MOVIIGEN_SP_TRANSPORT', '') == 'peer_copy'


def open_windows(group, buffers):
    """PeerWindows over `buffers`, or None — with a warning, the exchange then stays on the collective transport — when the
    mapping cannot be set up on EVERY rank of the group (IPC export refused, e.g. under expandable_segments; a peer's
    device not visible to this process: the mapped tensor lives on the SENDER's device index, so all GPUs of the group
    must be visible to every rank — per-rank HIP_VISIBLE_DEVICES masks are not supported by this transport)."""
    import logging
    try:
        win = PeerWindows(group, buffers)
        ok = 1
    except Exception as e:      # noqa: BLE001 — any failure to export / map means: do not use this transport
        logging.warning(f'MOVIIGEN_SP_TRANSPORT=peer_copy: mapping the peers\' receive buffers failed ({type(e).__name__}: {e}); '
                        'falling back to the all-to-all collective')
        win, ok = None, 0
    flag = torch.tensor([ok], dtype=torch.int32, device=buffers[0].device)
    g = group if group is not None else dist.group.WORLD
    if dist.get_backend(g) == 'gloo':
        flag = flag.cpu()
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=g)           # all ranks or none
    return win if int(flag.item()) == 1 else None


class PeerWindows:
    """the peers' views of a list of persistent receive buffers (same shapes on every rank of `group`)."""

    def __init__(self, group, buffers):
        from torch.multiprocessing.reductions import reduce_tensor
        self.group = group if group is not None else dist.group.WORLD
        self.P, self.rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        self.local = list(buffers)
        dev = self.local[0].device
        try:
            mine = [reduce_tensor(b) for b in self.local]      # (rebuild_fn, args): picklable IPC description
        except Exception as e:      # noqa: BLE001 — export refused here: tell the peers instead of leaving them in the gather
            mine = f'{type(e).__name__}: {e}'
        everyone = [None] * self.P
        dist.all_gather_object(everyone, mine, group=self.group)
        failed = [f'rank {p}: {m}' for p, m in enumerate(everyone) if isinstance(m, str)]
        if failed:
            raise RuntimeError('IPC export of the receive buffers failed on ' + '; '.join(failed))
        self.views = []                                          # views[i][p] = buffer i of rank p, mapped here
        for i, b in enumerate(self.local):
            row = []
            for p in range(self.P):
                if p == self.rank:
                    row.append(b)
                else:
                    fn, args = everyone[p][i]
                    row.append(fn(*args))                        # hipIpcOpenMemHandle; stays mapped while referenced
                    assert row[-1].shape == b.shape and row[-1].dtype == b.dtype
            self.views.append(row)
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self._keep = everyone

    def _rendezvous(self):
        collectives.rendezvous(self.flag, self.group)           # 4-byte all-reduce enqueued on the current (communication) stream

    def all_to_all(self, i, send):
        """buffer i of every rank <- the P chunks of `send` ([P, ...], chunk p goes to rank p), all_to_all_single layout:
        slot [source rank] of the destination's buffer."""
        P, r = self.P, self.rank
        assert send.shape[0] == P and send.is_contiguous()
        self._rendezvous()
        for k in range(P):
            p = (r + k) % P                                      # start with the local copy, then walk the ring: no hot peer
            dst = self.views[i][p].view(P, *send.shape[1:])[r]
            dst.copy_(send[p], non_blocking=True)                # contiguous, same dtype: one hipMemcpyAsync D2D (peer)
        self._rendezvous()
        return self.local[i]
