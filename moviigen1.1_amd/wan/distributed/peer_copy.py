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


def mode():
    """MOVIIGEN_SP_TRANSPORT: 'auto' (default) | 'peer_copy' | 'torch' | 'rccl_direct'"""
    return os.environ.get('MOVIIGEN_SP_TRANSPORT', '') or 'auto'


def enabled():
    return mode() == 'peer_copy'


def wanted(group, device):
    """does a HeadExchange on `group` try the copy-engine transport?  Always when it is asked for by name; in `auto` mode when the group
    runs on RCCL with more than one rank on device buffers — i.e. on a real multi-GPU launch: the copy engines move the bytes there and the
    persistent attention grid keeps every CU (a kernel transport only runs once an attention launch has ended).  A window that
    cannot be opened on every rank, or that fails its self-check, leaves the exchange on the collective (logged, `transport.used` says so)."""
    if torch.device(device).type != 'cuda' or group is None:
        return False
    if mode() == 'peer_copy':
        return True
    if mode() != 'auto':
        return False
    try:
        return dist.get_backend(group) == 'nccl' and dist.get_world_size(group) > 1
    except Exception:      # noqa: BLE001 — not a torch.distributed group (the rank emulation): the collective path
        return False


def self_check(win, group, buffer_index=0):
    """ONE exchange of a known pattern through the windows, before the first real one: chunk p of rank r's send image carries the value
    1000 r + p, so slot s of MY buffer must read 1000 s + my rank.  All ranks or none: a mismatch anywhere (a mapping that opened but does
    not reach the right memory) sends the whole group back to the collective.  -> win or None"""
    import logging
    buf = win.local[buffer_index]
    P, r = win.P, win.rank
    keep = buf.clone()
    send = torch.empty((P,) + tuple(buf.view(P, -1).shape[1:]), dtype=buf.dtype, device=buf.device)
    for p in range(P):
        send[p].fill_(float((7 * r + p) % 251))          # exact in bf16
    res = win.all_to_all(buffer_index, send, probe=True)
    got = buf.view(P, -1)
    want = torch.tensor([float((7 * s_ + r) % 251) for s_ in range(P)], dtype=torch.float32, device=buf.device)
    if isinstance(res, Exception):
        logging.warning(f'peer-copy transport: a copy of the self-check was refused on rank {r} ({type(res).__name__}: {res})')
        ok = 0
    else:
        ok = int(torch.equal(got.float(), want[:, None].expand_as(got)))
    buf.copy_(keep)
    flag = torch.tensor([ok], dtype=torch.int32, device=buf.device)
    g = win.group
    if dist.get_backend(g) == 'gloo':
        flag = flag.cpu()
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=g)
    if int(flag.item()) != 1:
        logging.warning('peer-copy transport: the self-check exchange did not arrive intact on every rank; falling back to the all-to-all collective')
        return None
    return win


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
    if int(flag.item()) != 1:
        return None
    return self_check(win, g)


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

    def all_to_all(self, i, send, probe=False):
        """buffer i of every rank <- the P chunks of `send` ([P, ...], chunk p goes to rank p), all_to_all_single layout:
        slot [source rank] of the destination's buffer.  probe (the self-check): a copy the runtime refuses on THIS rank must not leave the
        peers waiting in the second rendezvous — it is caught, the rendezvous still happens, and the error is returned instead of the buffer."""
        P, r = self.P, self.rank
        assert send.shape[0] == P and send.is_contiguous()
        self._rendezvous()
        error = None
        try:
            for k in range(P):
                p = (r + k) % P                                  # start with the local copy, then walk the ring: no hot peer
                dst = self.views[i][p].view(P, *send.shape[1:])[r]
                dst.copy_(send[p], non_blocking=True)            # contiguous, same dtype: one hipMemcpyAsync D2D (peer)
        except Exception as e:      # noqa: BLE001
            if not probe:
                raise
            error = e
        self._rendezvous()
        return error if probe and error is not None else self.local[i]
