"""The collective primitives of the distributed hot path, in ONE place: device buffers on RCCL (torch.distributed
backend "nccl" IS RCCL on ROCm; xGMI links between the GPUs of a node), optionally on the library's own communicator
(MOVIIGEN_SP_TRANSPORT=rccl_direct: the C-ABI collectives of csrc/sp_rccl.hip).

Each function is the production call, preceded by one guard: `_test_transport.staged(...)` — true only for device
tensors on a `gloo` group, i.e. in the N-ranks-on-one-GPU tests — hands the call to wan/distributed/_test_transport.py.
CPU tensors on gloo (the CPU tests) take the same torch.distributed lines as RCCL does."""
import torch
import torch.distributed as dist

from . import _test_transport, rccl_direct


def trace_summary(trace):
    """[(kind, event), ...] in begin / end pairs, kind 'comm' (a collective on its comm stream) or 'wait' (the compute
    stream waiting for one) -> dict(comm_ms, exposed_ms, hidden_frac, collectives).  Call after a device sync."""
    tot = {'comm': 0.0, 'wait': 0.0}
    n = {'comm': 0, 'wait': 0}
    for i in range(0, len(trace) - 1, 2):
        (k0, a), (k1, b) = trace[i], trace[i + 1]
        assert k0 == k1
        tot[k0] += a.elapsed_time(b)
        n[k0] += 1
    hidden = 1.0 - tot['wait'] / tot['comm'] if tot['comm'] > 0 else None
    return {'comm_ms': tot['comm'], 'exposed_ms': tot['wait'], 'hidden_frac': hidden, 'collectives': n['comm']}


def _direct(t):
    return t.is_cuda and rccl_direct.enabled()


def all_to_all(recv, send, group):
    """recv[p] <- rank p's send[my rank]  (all_to_all_single layout: dim 0 = peer)."""
    if _test_transport.staged(send, group):
        return _test_transport.all_to_all(recv, send, group)
    if _direct(send):
        rccl_direct.comm_for(group).all_to_all(recv, send)      # mg_sp_all_to_all: grouped ncclSend / ncclRecv
    else:
        dist.all_to_all_single(recv, send, group=group)


def all_gather(out, x, group, shard=False):
    """out [P * x.numel()] (any shape) <- rank-order concatenation of x; shard=True: the block-shard gather
    (mg_shard_all_gather on the direct transport)."""
    if _test_transport.staged(x, group):
        return _test_transport.all_gather(out, x, group)
    if _direct(x):
        comm = rccl_direct.comm_for(group)
        (comm.shard_all_gather if shard else comm.all_gather)(out, x)
    else:
        dist.all_gather_into_tensor(out.view(-1), x.reshape(-1), group=group)     # flat views: any (out, x) shapes with out = P x


def broadcast(t, src, group):
    if _test_transport.staged(t, group):
        return _test_transport.broadcast(t, src, group)
    dist.broadcast(t, src=src, group=group)


def send(x, dst, group=None):
    """the fp32 activation of a pipeline cut -> rank dst, enqueued behind the current stream's work.  The receiver knows
    the shape (WanVAE_.stage_out_shape): no header, no host round trip."""
    if _test_transport.staged(x, group):
        return _test_transport.send(x, dst, group)
    dist.send(x.contiguous(), dst, group=group)


def recv(shape, src, device, group=None):
    """-> a new fp32 tensor of `shape` on `device`, filled by rank src (stream-ordered on the current stream)."""
    x = torch.empty(*shape, dtype=torch.float32, device=device)
    if _test_transport.staged(x, group):
        return _test_transport.recv(x, src, group)
    dist.recv(x, src, group=group)
    return x


def neighbor_exchange(sends, recvs, group=None):
    """one batched point-to-point exchange: sends = [(contiguous tensor, group rank)], recvs = [(tensor to fill, group rank)] — the halo columns
    of the W-band VAE decode go to / come from the left and right neighbour in ONE group of isend / irecv (no ordering between the two directions:
    a chain of blocking pairs would unroll rank by rank).  Stream-ordered on return."""
    if not sends and not recvs:
        return
    probe = (sends or recvs)[0][0]
    world = group is None or group is dist.group.WORLD
    gr = (lambda r: r) if world else (lambda r: dist.get_global_rank(group, r))
    if _test_transport.staged(probe, group):
        return _test_transport.neighbor_exchange([(t, gr(r)) for t, r in sends], [(t, gr(r)) for t, r in recvs], group)
    ops = [dist.P2POp(dist.isend, t, gr(r), group) for t, r in sends] + [dist.P2POp(dist.irecv, t, gr(r), group) for t, r in recvs]
    for w in dist.batch_isend_irecv(ops):
        w.wait()


class _Works:
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


def ring_hop(send_bufs, recv_bufs, group, P, rank):
    """post one hop of a ring: send_bufs -> rank+1, recv_bufs <- rank-1 (batched isend / irecv); -> object with wait()."""
    world = group is None or group is dist.group.WORLD
    nxt = (rank + 1) % P if world else dist.get_global_rank(group, (rank + 1) % P)
    prv = (rank - 1) % P if world else dist.get_global_rank(group, (rank - 1) % P)
    if _test_transport.staged(send_bufs[0], group):
        return _test_transport.RingHop(send_bufs, recv_bufs, nxt, prv, group)
    ops = [dist.P2POp(dist.isend, t, nxt, group) for t in send_bufs] + [dist.P2POp(dist.irecv, t, prv, group) for t in recv_bufs]
    return _Works(dist.batch_isend_irecv(ops))


def rendezvous(flag, group):
    """every rank of the group has reached this point of its current stream (4-byte all-reduce, stream-ordered)."""
    if _test_transport.staged(flag, group):
        return _test_transport.rendezvous(group)
    dist.all_reduce(flag, group=group)


# ---- control plane: scalars, not tensors of the path ---------------------------------------------------------------------------------
def _on_host(group):
    """gloo moves host memory only (the CPU tests and the ranks-sharing-one-GPU tests); RCCL moves device memory."""
    return dist.get_backend(group) == 'gloo'


def control_reduce(value, op, group, device, dtype=torch.float64):
    """one python number reduced over the group ('min' / 'max' / 'sum') -> python number, the same on every rank.  Host round trip: for votes,
    clocks and flags around the path (fallback decisions, preflight timings) — never inside a layer."""
    group = group if group is not None else dist.group.WORLD
    t = torch.tensor([value], dtype=dtype, device='cpu' if _on_host(group) else device)
    dist.all_reduce(t, op={'min': dist.ReduceOp.MIN, 'max': dist.ReduceOp.MAX, 'sum': dist.ReduceOp.SUM}[op], group=group)
    return t.item()


def control_broadcast(value, src, group, device, dtype=torch.float64):
    """group rank `src`'s python number -> every rank."""
    group = group if group is not None else dist.group.WORLD
    t = torch.tensor([value], dtype=dtype, device='cpu' if _on_host(group) else device)
    dist.broadcast(t, src=src if group is dist.group.WORLD else dist.get_global_rank(group, src), group=group)
    return t.item()
