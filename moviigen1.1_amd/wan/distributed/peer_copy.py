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


def enabled():
    return os.environ.get('MOVIIGEN_SP_TRANSPORT', '') == 'peer_copy'


class PeerWindows:
    """the peers' views of a list of persistent receive buffers (same shapes on every rank of `group`)."""

    def __init__(self, group, buffers):
        from torch.multiprocessing.reductions import reduce_tensor
        self.group = group if group is not None else dist.group.WORLD
        self.P, self.rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        self.gloo = dist.get_backend(self.group) == 'gloo'
        self.local = list(buffers)
        dev = self.local[0].device
        mine = [reduce_tensor(b) for b in self.local]          # (rebuild_fn, args): picklable IPC description
        everyone = [None] * self.P
        dist.all_gather_object(everyone, mine, group=self.group)
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
        if self.gloo:           # test plumbing on a shared GPU: host-synchronous
            torch.cuda.current_stream().synchronize()
            dist.barrier(group=self.group)
        else:
            dist.all_reduce(self.flag, group=self.group)        # enqueued on the current (communication) stream

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
