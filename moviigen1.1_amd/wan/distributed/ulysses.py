"""Ulysses sequence-parallel data movement on RCCL (torch.distributed backend "nccl" == RCCL on
ROCm; xGMI point-to-point links between the 8 GPUs of a node).

The reference delegates this to un-vendored libraries: xfuser's xFuserLongContextAttention
(wan/distributed/xdit_context_parallel.py:185-190) and FastVideo's all_to_all_4D
(scripts/train/model/model_seq.py:232-234,256).  Semantics:

  seq -> head : [L/P tokens, N heads]   -> [L tokens, N/P heads]   (q, k, v before attention)
  head -> seq : [L tokens, N/P heads]   -> [L/P tokens, N heads]   (attention output)
  all_gather_seq : rank-order concatenation along tokens            (head output, once/forward)

`HeadExchange` is what WanModel runs per layer: the rank's heads are cut into G head groups and
the layer becomes a software pipeline over them —

    compute stream :  pack(0..G-1) | attn(0)      | attn(1)      | ... | unpack(0..G-1)
    comm stream    :    qkv a2a(0) qkv a2a(1) ... | o a2a(0)     | o a2a(1) ...

  * ONE packed exchange per group carries q, k and v (mg_sp_pack_qkv_bf16 writes the
    [dest][token][q|k|v] send layout straight from the fused qkv activations; the receive buffer is
    then directly the row-major [L][3w] operand matrix — no transposes, no unpack on that side);
  * all buffers are persistent (allocated once per shape), nothing is allocated inside the layer loop;
  * collectives are issued on a dedicated HIP stream and ordered against the kernels with events, so
    the exchange of group g+1 and the return of group g-1 run under the attention of group g.
    On the fully connected xGMI mesh an all-to-all sends each peer its 1/P slice over its own direct
    link, all 7 links concurrently.  Tokens are sharded contiguously: rank r owns [r*L/P, (r+1)*L/P).

Transports of the exchange (MOVIIGEN_SP_TRANSPORT): `auto` (default) = one-sided peer copies on the copy engines (peer_copy.py) when the
group runs on RCCL with more than one rank AND the peers' buffers can be mapped and pass a pattern exchange on every rank, else
torch.distributed backend "nccl" (= RCCL) — which `torch` selects by name; `rccl_direct` = the library's own RCCL communicator (C-ABI
mg_sp_all_to_all); `peer_copy` = the copy engines or a logged fall-back — the calls themselves live in collectives.py (device to device; the
host-staged gloo transport of the one-GPU multi-process tests is behind its single guard there).  `seq_to_head` /
`head_to_seq` keep the one-tensor call shape of the reference libraries (used by the training-side SP forward and tests).
"""
import os

import torch

from ..backend import ops
from . import collectives, peer_copy

_a2a = collectives.all_to_all


def split_heads(n_loc, max_groups):
    """[(first local head, heads)] of the pipeline groups: sizes differ by at most one."""
    G = max(1, min(int(n_loc), int(max_groups)))
    base, extra = divmod(n_loc, G)
    out, h0 = [], 0
    for g in range(G):
        n = base + (1 if g < extra else 0)
        out.append((h0, n))
        h0 += n
    return out


def attention_rounds(n_heads, L, n_cu=256, qblock=256):
    """rounds ONE launch of the persistent head-dim-128 kernel takes for `n_heads` heads over L queries: heads x ceil(L / 256) items of
    one 256-query block each, dealt to the 8 XCDs in contiguous ranges and walked by the n_cu / 8 workgroups of an XCD in rounds
    (csrc/attn_hd128_m16.hip: the work loop).  A round lasts as long as one item whatever the number of busy workgroups."""
    items = n_heads * ((L + qblock - 1) // qblock)
    per_xcd_wg = max(1, (n_cu & ~7) // 8)
    if items <= (n_cu & ~7):
        return 1
    return max(((x + 1) * items // 8 - x * items // 8 + per_xcd_wg - 1) // per_xcd_wg for x in range(8))


def choose_groups(n_loc, L, P, dim=None, max_groups=5, n_cu=256, tflops_per_cu=5.8, link_gbps=45.0):
    """How many pipeline groups the rank's n_loc heads are cut into (a fixed 5 ignored the round quantisation of
    the persistent attention grid — five 2-head groups of configs[3] take 6 rounds per launch for 5.08 rounds of work, -15 %).
    Cost of a layer with G groups = sum over the groups of attention_rounds(n_g) x (time of one 256-query item over L keys)
                                   + the part of the exchange no attention can hide: the first group's q|k|v exchange and the last group's
                                     return = 1 / G of the layer's exchange (4 * (L / P) * dim * 2 bytes / P per link);
    the smallest cost wins, ties go to MORE groups.  With the exchange term small this is "fewest rounds, then most groups".
    -> (G, rounds, cost_ms): e.g. configs[2] (L = 131 040, 5 local heads): 5 groups, 10 rounds; configs[3] (L = 166 320, cfg2 x SP4, 10
    local heads): 2 groups of 5 heads, 26 rounds (5 x 2 heads: 30); 720p x SP8 (5 heads, 296 query blocks): ONE group, 6 rounds (5 x 1: 10)."""
    dim = dim if dim is not None else 128 * n_loc * P
    t_item = 4.0 * 256 * L * 128 / (tflops_per_cu * 1e12) * 1e3                      # ms
    t_exch = 4.0 * (L / P) * dim * 2 / P / (link_gbps * 1e9) * 1e3 if P > 1 else 0.0   # ms per layer on one link (all links concurrent)
    best = None
    for G in range(1, max(1, min(int(n_loc), int(max_groups))) + 1):
        rounds = sum(attention_rounds(n, L, n_cu) for _, n in split_heads(n_loc, G))
        cost = rounds * t_item + t_exch / G
        if best is None or cost < best[2] - 1e-9 or (abs(cost - best[2]) <= 1e-9 and G > best[0]):
            best = (G, rounds, cost)
    return best


class HeadExchange:
    """persistent buffers, events and the communication stream of the pipelined exchange for one
    (group, Lloc) shape.  `run(q, k, v, out, attend)` executes one layer's exchange + attention.

    Measurement hook (bench.py): while `HeadExchange.trace` is a list, every collective is bracketed by timing events
    on the comm stream and every wait of the compute stream on the comm stream by timing events on the compute
    stream; `overlap_summary()` turns them into (exchange time, time the compute stream stood waiting for it)."""

    trace = None        # None = off; a list = collect (kind, start event, end event)

    @classmethod
    def _mark(cls, kind, stream):
        if cls.trace is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        cls.trace.append((kind, ev))
        return ev

    @classmethod
    def overlap_summary(cls):
        """-> dict(exchange_ms, exposed_ms, hidden_frac) over everything traced so far (call after a device sync)."""
        s = collectives.trace_summary(cls.trace or [])
        return {'exchange_ms': s['comm_ms'], 'exposed_ms': s['exposed_ms'], 'hidden_frac': s['hidden_frac'], 'collectives': s['collectives']}

    def __init__(self, group, P, heads, head_dim, Lloc, device, max_groups=None):
        if heads % P:
            raise ValueError(f'`num_heads` {heads} cannot be divided evenly by the sequence-parallel size {P}')
        self.group, self.P, self.hd, self.Lloc = group, P, head_dim, Lloc
        self.n_loc = heads // P
        self.cols = self.n_loc * head_dim                      # columns of one destination's head slice
        # the CUs an attention launch of this exchange really gets: the device's, minus what MOVIIGEN_SP_RESERVE_CUS leaves to a collective
        # kernel (only honoured on a kernel transport with >= 2 groups, see reserve_cus below) — the group choice and the reported round
        # count are made for THAT grid
        n_cu = torch.cuda.get_device_properties(device).multi_processor_count if torch.device(device).type == 'cuda' else 256
        want_reserve = 0 if peer_copy.wanted(group, device) else (int(os.environ.get('MOVIIGEN_SP_RESERVE_CUS', '0') or 0) + 7) & ~7
        n_cu_attn = max(8, (n_cu & ~7) - want_reserve)
        if max_groups is None:
            env = os.environ.get('MOVIIGEN_SP_GROUPS', '') or 'auto'      # an empty value means auto
            if env == 'auto':                                  # by shape: rounds of the persistent attention grid vs exposed exchange
                max_groups = choose_groups(self.n_loc, P * Lloc, P, dim=heads * head_dim, n_cu=n_cu_attn)[0] if head_dim == 128 else min(5, self.n_loc)
            else:
                max_groups = int(env)
        self.groups = split_heads(self.n_loc, max_groups)
        bf = torch.bfloat16
        e = lambda *s: torch.empty(*s, dtype=bf, device=device)  # noqa: E731
        self.send, self.recv, self.ag, self.orecv = [], [], [], []
        for _, n in self.groups:
            w = n * head_dim
            self.send.append(e(P, Lloc, 3 * w))
            self.recv.append(e(P * Lloc, 3 * w))
            self.ag.append(e(P * Lloc, w))
            self.orecv.append(e(P, Lloc, w))
        # The copy-engine transport (opt-in: MOVIIGEN_SP_TRANSPORT=auto on an RCCL group of more than one rank, or =peer_copy): the
        # receive buffers are mapped into the peers once, checked with one pattern exchange, and an exchange is then P one-sided device
        # copies on the comm stream (copy engines, no CUs) between two 4-byte rendezvous
        self.peer = None
        if peer_copy.wanted(group, device):
            self.peer = peer_copy.open_windows(group, self.recv + self.orecv)      # None (logged) when IPC mapping is unavailable
        self.comm = torch.cuda.Stream(device=device)
        # CUs the attention launches of this layer leave free while exchanges are in flight (ops.attention_hd128
        # reserve_cus): the collective transports are kernels and the attention grid is persistent, one workgroup per CU
        # with the whole register file, so an RCCL kernel only runs once an attention launch has ended.  Measured at the
        # configs[2] group size (profiles/r04b_sp_overlap.txt): leaving 8 CUs free costs MORE than the exposed
        # exchange it could hide (512 query blocks of a one-head group on 248 instead of 256 workgroups = three rounds
        # instead of two: +28 % attention time) — the default is 0; the copy-engine transport needs none either way.
        self.reserve_cus = 0 if self.peer is not None or len(self.groups) < 2 else int(os.environ.get('MOVIIGEN_SP_RESERVE_CUS', '0') or 0)
        self.rounds = sum(attention_rounds(n, P * Lloc, (n_cu & ~7) - ((self.reserve_cus + 7) & ~7)) for _, n in self.groups)
        ev = lambda: [torch.cuda.Event() for _ in self.groups]  # noqa: E731
        self.ev_pack, self.ev_recv, self.ev_attn, self.ev_o = ev(), ev(), ev(), ev()

    def peer_failed(self):
        return self.peer is not None and self.peer.failed()

    def drop_peer(self):
        """back to the all-to-all collective for good (a copy of the copy-engine transport failed somewhere in the group)."""
        import logging
        if self.peer is not None:
            logging.warning('Ulysses exchange: the peer-copy transport reported a failed copy; this exchange now uses the all-to-all collective')
        self.peer = None
        self.reserve_cus = 0 if len(self.groups) < 2 else int(os.environ.get('MOVIIGEN_SP_RESERVE_CUS', '0') or 0)

    def run(self, q, k, v, out, attend):
        """q, k, v: [Lloc, heads*hd] bf16 (column slices of the fused qkv buffer are fine); out [Lloc, heads*hd].
        attend(qg, kg, vg, ag, n_heads): attention of n_heads heads over ALL tokens of the group, operands are
        column views (row stride 3w) of the receive buffer, ag a contiguous [P*Lloc, w] result buffer."""
        P, hd, cur = self.P, self.hd, torch.cuda.current_stream()
        for g, (h0, n) in enumerate(self.groups):
            ops.sp_pack_qkv(q, k, v, P, self.cols, h0 * hd, n * hd, self.send[g])
            self.ev_pack[g].record(cur)
        with torch.cuda.stream(self.comm):
            for g in range(len(self.groups)):
                self.comm.wait_event(self.ev_pack[g])
                self._mark('comm', self.comm)
                if self.peer is not None:
                    self.peer.all_to_all(g, self.send[g])
                else:
                    _a2a(self.recv[g].view(P, self.Lloc, -1), self.send[g], self.group)
                self._mark('comm', self.comm)
                self.ev_recv[g].record(self.comm)
        for g, (h0, n) in enumerate(self.groups):
            w = n * hd
            self._mark('wait', cur)
            cur.wait_event(self.ev_recv[g])
            self._mark('wait', cur)
            r = self.recv[g]
            attend(r[:, :w], r[:, w:2 * w], r[:, 2 * w:], self.ag[g], n)
            self.ev_attn[g].record(cur)
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(self.ev_attn[g])
                self._mark('comm', self.comm)
                if self.peer is not None:
                    self.peer.all_to_all(len(self.groups) + g, self.ag[g].view(P, self.Lloc, w))
                else:
                    _a2a(self.orecv[g], self.ag[g].view(P, self.Lloc, w), self.group)
                self._mark('comm', self.comm)
                self.ev_o[g].record(self.comm)
        for g, (h0, n) in enumerate(self.groups):
            self._mark('wait', cur)
            cur.wait_event(self.ev_o[g])
            self._mark('wait', cur)
            ops.sp_unpack_o(self.orecv[g], P, self.cols, h0 * hd, n * hd, out)
        return out


# ---- one-tensor exchanges in the call shape of the reference libraries ----------------------------
def seq_to_head(x, out, group, P, heads, head_dim):
    """x [Lloc, heads*hd] (row stride free) -> out [P*Lloc, (heads/P)*hd], tokens in rank order."""
    Lloc = x.shape[0]
    nl = (heads // P) * head_dim
    if x.is_cuda:
        send = torch.empty(P, Lloc, nl, dtype=x.dtype, device=x.device)
        ops.sp_copy_blocks(x, nl, x.stride(0), send, Lloc * nl, nl, P, Lloc, nl)
    else:        # CPU tensors: host-side index plumbing of the gloo tests (no arithmetic)
        send = x.reshape(Lloc, P, nl).transpose(0, 1).contiguous()
    _a2a(out.view(P, Lloc, nl), send, group)
    return out


def head_to_seq(x, out, group, P, heads, head_dim):
    """x [P*Lloc, (heads/P)*hd] -> out [Lloc, heads*hd]."""
    nl = (heads // P) * head_dim
    Lloc = x.shape[0] // P
    recv = torch.empty(P, Lloc, nl, dtype=x.dtype, device=x.device)  # [source rank = head group][token][..]
    _a2a(recv, x.view(P, Lloc, nl), group)
    if x.is_cuda:
        ops.sp_copy_blocks(recv, Lloc * nl, nl, out, nl, out.stride(0), P, Lloc, nl)
    else:
        out.view(Lloc, P, nl).copy_(recv.transpose(0, 1))
    return out


def all_gather_seq(x, group, P):
    """x [Lloc, C] -> [P*Lloc, C] (rank-order concatenation; get_sp_group().all_gather(dim=1))."""
    out = torch.empty(P * x.shape[0], x.shape[1], dtype=x.dtype, device=x.device)
    collectives.all_gather(out, x.contiguous(), group)
    return out

