"""TEST TRANSPORT — not a product path.  The multi-process tests of this repository run N ranks on ONE GPU (the only
kind of box the build has) with the `gloo` backend, and gloo cannot move device tensors: every collective below stages
them through host memory, synchronously.  `collectives.py` enters this module on exactly one condition —
`staged(tensor, group)`: the group's backend is gloo AND the tensor lives on a device — so under `nccl` (RCCL; every
real multi-GPU run) none of this code is reachable.  Nothing here does arithmetic."""
import torch
import torch.distributed as dist


class EmulatedGroup:
    """MEASUREMENT ONLY (tools/emulate_rank.py, `bench.py --emulate-rank`): stands for a process group of `size` ranks of which only THIS
    process exists — a single-GPU box predicting what one rank of a multi-GPU run does.  Every collective on it is a LOOP-BACK: device
    copies of the real message sizes on the calling stream (the peers' data is this rank's own, repeated), and the bytes that would have
    crossed a link are counted.  The numbers such a run prints are marked `invalid: emulation`; nothing in a real launch creates one."""

    def __init__(self, size, rank=0, name='emulated'):
        self.size, self.rank, self.name = int(size), int(rank), name
        self.link_bytes = {'all_to_all': 0, 'all_gather': 0, 'p2p': 0}      # bytes this rank would send over ONE of its P - 1 links (p2p: to one neighbour)
        self.calls = {'all_to_all': 0, 'all_gather': 0, 'p2p': 0}

    def _count(self, kind, total_bytes):
        self.calls[kind] += 1
        self.link_bytes[kind] += total_bytes // self.size           # 1 / P of the buffer goes to each peer, each over its own link


def staged(t, group):
    if isinstance(group, EmulatedGroup):
        return True
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def all_to_all(recv, send, group):
    if isinstance(group, EmulatedGroup):
        recv.view(-1).copy_(send.reshape(-1))                      # chunk p <- "rank p's" chunk = my own
        return group._count('all_to_all', send.numel() * send.element_size())
    s, r = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
    dist.all_to_all_single(r, s, group=group)
    recv.copy_(r)


def all_gather(out, x, group):
    if isinstance(group, EmulatedGroup):
        out.view(group.size, -1).copy_(x.reshape(1, -1).expand(group.size, -1))
        return group._count('all_gather', x.numel() * x.element_size() * group.size)
    P = dist.get_world_size(group)
    parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(P)]
    dist.all_gather(parts, x.cpu().contiguous(), group=group)
    out.copy_(torch.cat([p.reshape(-1) for p in parts]).view(out.shape))


def broadcast(t, src, group):
    if isinstance(group, EmulatedGroup):
        return
    h = t.cpu()
    dist.broadcast(h, src=src, group=group)
    t.copy_(h)


def send(x, dst, group):
    if isinstance(group, EmulatedGroup):
        group.calls['p2p'] += 1
        group.link_bytes['p2p'] += x.numel() * x.element_size()
        return
    dist.send(x.detach().cpu().contiguous(), dst, group=group)


def recv(x, src, group):
    if isinstance(group, EmulatedGroup):
        return x                                                   # (contents unspecified: an emulated run is a timing run)
    h = torch.empty(x.shape, dtype=x.dtype)
    dist.recv(h, src, group=group)
    x.copy_(h)
    return x


class RingHop:
    """send_bufs -> nxt, recv_bufs <- prv through host copies; wait() lands them in the device buffers."""

    def __init__(self, send_bufs, recv_bufs, nxt, prv, group):
        self.host_r = [torch.empty(b.shape, dtype=b.dtype) for b in recv_bufs]
        self.recv_bufs = recv_bufs
        ops = [dist.P2POp(dist.isend, b.cpu(), nxt, group) for b in send_bufs] + \
              [dist.P2POp(dist.irecv, t, prv, group) for t in self.host_r]
        self.works = dist.batch_isend_irecv(ops)

    def wait(self):
        for w in self.works:
            w.wait()
        for h, d in zip(self.host_r, self.recv_bufs):
            d.copy_(h)


def neighbor_exchange(sends, recvs, group):
    if isinstance(group, EmulatedGroup):
        for (t, _), (src, _) in zip(recvs, sends):                  # "the neighbour's" border = my own
            t.copy_(src)
        group.calls['p2p'] += 1
        group.link_bytes['p2p'] += max((t.numel() * t.element_size() for t, _ in sends), default=0)      # the two directions use two links
        return
    host_r = [torch.empty(t.shape, dtype=t.dtype) for t, _ in recvs]
    ops = [dist.P2POp(dist.isend, t.detach().cpu().contiguous(), r, group) for t, r in sends] + \
          [dist.P2POp(dist.irecv, h, r, group) for h, (_, r) in zip(host_r, recvs)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    for h, (t, _) in zip(host_r, recvs):
        t.copy_(h)


def rendezvous(group):
    if isinstance(group, EmulatedGroup):
        return
    torch.cuda.current_stream().synchronize()
    dist.barrier(group=group)
