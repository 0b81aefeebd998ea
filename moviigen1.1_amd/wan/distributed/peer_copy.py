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
    """MOVIIGEN_SP_TRANSPORT: 'torch' (default: the all-to-all collective) | 'auto' | 'peer_copy' | 'rccl_direct'.
    The copy-engine transport is OPT-IN: it has run on gloo ranks sharing one GPU and on RCCL at world size 1 only — until a
    multi-GPU run has shown it correct and faster (`bench.py --preflight` measures both transports before the warm-up, tools/scale_sweep.sh
    runs both), a real multi-GPU launch stays on the collective unless `auto` / `peer_copy` is asked for by name."""
    return os.environ.get('MOVIIGEN_SP_TRANSPORT', '') or 'torch'


def enabled():
    return mode() == 'peer_copy'


def wanted(group, device):
    """does a HeadExchange on `group` try the copy-engine transport?  When it is asked for by name; in `auto` mode when the group runs on
    RCCL with more than one rank on device buffers — i.e. on a real multi-GPU launch: the copy engines move the bytes there and the
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


PROBE_ELEMS = 4096      # elements of every slot the self-check writes (and restores)


def _vote(ok, group, device):
    """MIN over the group of a 0/1 flag: all ranks or none.  Every rank reaches this, whatever happened before."""
    return int(collectives.control_reduce(int(ok), 'min', group, device, dtype=torch.int32)) == 1


def self_check(win, group, buffer_indices=None):
    """ONE small pattern exchange through the windows before the first real one, on the first buffer of each half of the list (a q|k|v
    receive buffer and a return buffer): rank r writes the value (7 r + p) % 251 (exact in bf16) into the first PROBE_ELEMS elements of
    slot [r] of rank p's buffer, so slot s of MY buffer must read (7 s + my rank) % 251.  All ranks or none: a mismatch or ANY exception
    anywhere (a mapping that opened but does not reach the right memory, a copy the runtime refuses, an allocation that fails) sends the
    whole group back to the collective — nothing here can raise past the vote, so no rank is left waiting in it.  The probe touches
    P x PROBE_ELEMS elements per buffer and restores them; no full-size temporaries.  -> win or None"""
    import logging
    ok = 1
    P, r = win.P, win.rank
    if buffer_indices is None:
        buffer_indices = sorted({0, len(win.local) // 2})

    def guarded(what, fn):
        """one local step; an exception makes this rank vote no but NEVER skips a rendezvous: every rank performs the same sequence of
        collectives (two rendezvous per probed buffer, then the vote), whatever fails in between"""
        nonlocal ok
        try:
            return fn()
        except Exception as e:      # noqa: BLE001
            logging.warning(f'peer-copy transport: self-check step "{what}" failed on rank {r} ({type(e).__name__}: {e})')
            ok = 0
            return None
    for bi in buffer_indices:
        st = {}

        def save():
            st['mine'] = win.local[bi].view(P, -1)
            st['n'] = min(PROBE_ELEMS, st['mine'].shape[1])
            st['keep'] = st['mine'][:, :st['n']].clone()

        def write():
            mine, n = st['mine'], st['n']
            for k in range(P):
                p = (r + k) % P
                src = torch.full((n,), float((7 * r + p) % 251), dtype=mine.dtype, device=mine.device)
                win.views[bi][p].view(P, -1)[r, :n].copy_(src, non_blocking=True)

        def compare_and_restore():
            nonlocal ok
            mine, n = st['mine'], st['n']
            for s_ in range(P):                                  # per slot, in the buffer's own dtype
                val = torch.tensor(float((7 * s_ + r) % 251), dtype=mine.dtype, device=mine.device)
                if not bool((mine[s_, :n] == val).all()):
                    ok = 0
            mine[:, :n].copy_(st['keep'])
        guarded('save the probed elements', save)
        win._rendezvous()                                        # every rank saved its slots
        if 'keep' in st:
            guarded('write the pattern into the peers', write)
        win._rendezvous()                                        # all writes have landed everywhere
        if 'keep' in st:
            guarded('compare and restore', compare_and_restore)
    # the restores above are asynchronous copies on THIS stream; the exchanges that follow run on another one (HeadExchange.comm) and a peer's
    # first real copy may land as soon as the vote lets it go: the restore must be complete before this rank votes (it was the vote's own
    # device-to-host copy that ordered this in round 5; found as a once-in-a-dozen-runs wrong answer of the 8-rank one-GPU test)
    guarded('wait for the restores', lambda: torch.cuda.current_stream(win.local[0].device).synchronize())
    try:
        agreed = _vote(ok, win.group, win.local[0].device)
    except Exception as e:      # noqa: BLE001 — the control plane itself failed: nothing to fall back with, but say what happened
        logging.warning(f'peer-copy transport: the fallback vote failed ({type(e).__name__}: {e})')
        raise
    if not agreed:
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
    g = group if group is not None else dist.group.WORLD
    if not _vote(ok, g, buffers[0].device):                         # all ranks or none
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

    def all_to_all(self, i, send):
        """buffer i of every rank <- the P chunks of `send` ([P, ...], chunk p goes to rank p), all_to_all_single layout:
        slot [source rank] of the destination's buffer.  A copy the runtime refuses on THIS rank must not leave the peers waiting in the
        second rendezvous: it is caught, 1 is added to this rank's flag word — the rendezvous all-reduces (sums) that word, so from then on
        it is non-zero on EVERY rank — and the rendezvous still happens.  `failed()` reads the word; the caller (HeadExchange.peer_failed,
        once per forward) then takes the whole group back to the collective and repeats the forward."""
        P, r = self.P, self.rank
        assert send.shape[0] == P and send.is_contiguous()
        self._rendezvous()
        try:
            for k in range(P):
                p = (r + k) % P                                  # start with the local copy, then walk the ring: no hot peer
                dst = self.views[i][p].view(P, *send.shape[1:])[r]
                dst.copy_(send[p], non_blocking=True)            # contiguous, same dtype: one hipMemcpyAsync D2D (peer)
        except Exception as e:      # noqa: BLE001
            import logging
            logging.warning(f'peer-copy transport: a copy was refused on rank {r} ({type(e).__name__}: {e}); the group will fall back to the collective')
            self.flag.add_(1)
        self._rendezvous()
        return self.local[i]

    def failed(self):
        """has any rank's copy failed since the windows were opened?  (one 4-byte read: a host sync — call it once per forward, not per
        exchange.)  The word only ever changes through the rendezvous all-reduce, so every rank reads the same answer after the same exchange."""
        word = int(self.flag.item())
        if collectives._test_transport.staged(self.flag, self.group):     # the one-GPU test transport's rendezvous is a host barrier: the word is local
            word = int(collectives.control_reduce(word, 'max', self.group, self.flag.device, dtype=torch.int32))
        return word != 0
