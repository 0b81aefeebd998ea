"""shard_model — block-sharded DiT weights (the `--dit_fsdp` mode of the reference).

The reference wraps each `model.blocks[i]` in torch FSDP (FULL_SHARD, bf16 params,
wan/distributed/fsdp.py:10-32): before a block runs, NCCL all-gathers its flat parameter, after
it the full copy is dropped.  Inference only — no reduce-scatter.

MI355X-native restatement, same memory behaviour, explicit instead of hook-driven:

  * each block's bf16 GEMM weights (q,k,v,o, cross q,k,v,o, ffn.0, ffn.2 — 351 M params = 703 MB
    at 14B) are flattened in that order into ONE flat tensor, padded to a multiple of P, and each
    rank keeps its 1/P slice.  q|k|v and cross k|v are adjacent in the flat layout, so the fused
    `[3d,d]` / `[2d,d]` GEMM operands are plain views of the gathered buffer.
  * two full-size gather buffers; block i+1 is all-gathered (`all_gather_into_tensor`, RCCL:
    1-hop on the xGMI mesh) on a dedicated HIP stream while block i computes; events order
    producer/consumer in both directions.  Small fp32 tensors (biases, norm weights, modulation)
    stay replicated.
  * `sync_module_states=True` broadcasts rank 0's weights first, as FSDP does.

Works on CPU tensors with gloo as well (no streams) — that is how the CPU test drives it.
"""
import torch
import torch.distributed as dist

from . import collectives

ORDER = ('self_attn.q', 'self_attn.k', 'self_attn.v', 'self_attn.o', 'cross_attn.q', 'cross_attn.k', 'cross_attn.v',
         'cross_attn.o', 'ffn.0', 'ffn.2')


def _get(block, dotted):
    m = block
    for part in dotted.split('.'):
        m = m[part] if isinstance(m, torch.nn.ModuleDict) else getattr(m, part)
    return m


class BlockShards:
    # Measurement hook (bench.py --dit-fsdp), same form as HeadExchange.trace: while `trace` is a list every gather is
    # bracketed by timing events on the comm stream ('comm') and every wait of the compute stream for a gathered block by
    # timing events on the compute stream ('wait'); collectives.trace_summary() turns them into gather / exposed time.
    trace = None

    @classmethod
    def _mark(cls, kind, stream):
        if cls.trace is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            cls.trace.append((kind, ev))

    def __init__(self, model, group=None, sync_module_states=True):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised (launch with torchrun, one process per GPU)')
        # an OWN communicator over the ranks (a second RCCL comm = its own stream): the prefetch gathers then overlap
        # the attention all-to-alls instead of queueing behind them on the WORLD communicator
        self.group = group if group is not None else dist.new_group(list(range(dist.get_world_size())))
        self.P = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        self.shapes, self.offsets, self.shards = [], [], []
        dev = model.patch_embedding.weight.device
        self.dev = dev
        n_pad_max = 0
        for blk in model.blocks:
            ws = [_get(blk, n).weight for n in ORDER]
            flat = torch.cat([w.data.reshape(-1) for w in ws])
            n = flat.numel()
            n_pad = (n + self.P * 8 - 1) // (self.P * 8) * (self.P * 8)
            if n_pad != n:
                flat = torch.cat([flat, flat.new_zeros(n_pad - n)])
            if sync_module_states:
                self._bcast(flat)
            per = n_pad // self.P
            self.shards.append(flat[self.rank * per:(self.rank + 1) * per].clone())
            shapes, offs, o = [], [], 0
            for w in ws:
                shapes.append(tuple(w.shape))
                offs.append(o)
                o += w.numel()
                w.data = w.data.new_empty(0)         # release the full copy
            self.shapes.append(shapes)
            self.offsets.append(offs)
            n_pad_max = max(n_pad_max, n_pad)
            del flat
        self.bufs = [torch.empty(n_pad_max, dtype=torch.bfloat16, device=dev) for _ in range(2)]
        self.in_buf = [None, None]                    # which block each buffer currently holds
        self.cuda = dev.type == 'cuda'
        if self.cuda:
            self.comm = torch.cuda.Stream(device=dev)
            self.ready = [torch.cuda.Event() for _ in range(2)]
            self.free = [torch.cuda.Event() for _ in range(2)]
            for e in self.free:
                e.record()
        model._shards = self
        model._invalidate()

    def _bcast(self, flat):
        collectives.broadcast(flat, dist.get_global_rank(self.group, 0), self.group)

    def _gather(self, i, slot):
        shard = self.shards[i]
        out = self.bufs[slot][:shard.numel() * self.P]
        # enqueued on the CURRENT stream: _issue() calls this under `with stream(self.comm)`
        collectives.all_gather(out, shard, self.group, shard=True)
        self.in_buf[slot] = i

    def _issue(self, i):
        slot = i % 2
        if self.in_buf[slot] == i:
            return
        if self.cuda:
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(self.free[slot])     # compute finished with the previous tenant
                self._mark('comm', self.comm)
                self._gather(i, slot)
                self._mark('comm', self.comm)
                self.ready[slot].record(self.comm)
        else:
            self._gather(i, slot)

    def fetch(self, i, prefetch=True):
        """full bf16 weights of block i as views of a gather buffer; starts the gather of i+1."""
        slot = i % 2
        self._issue(i)
        if self.cuda:
            cur = torch.cuda.current_stream()
            self._mark('wait', cur)
            cur.wait_event(self.ready[slot])
            self._mark('wait', cur)
        n_blocks = len(self.shards)
        if prefetch and n_blocks > 1:
            nxt = (i + 1) % n_blocks
            if self.cuda:
                # the other slot is free once everything enqueued so far on the compute stream is done
                self.free[nxt % 2].record(torch.cuda.current_stream())
            if nxt % 2 != slot:
                self._issue(nxt)
        buf = self.bufs[slot]
        views = {}
        for name, shape, off in zip(ORDER, self.shapes[i], self.offsets[i]):
            views[name] = buf[off:off + shape[0] * shape[1]].view(shape)
        d = self.shapes[i][0][0]
        o_q, o_ck = self.offsets[i][0], self.offsets[i][5]
        views['wqkv'] = buf[o_q:o_q + 3 * d * d].view(3 * d, d)
        views['wkv_c'] = buf[o_ck:o_ck + 2 * d * d].view(2 * d, d)
        return views

    def release(self, i):
        """compute is done with block i's buffer (recorded on the compute stream)."""
        if self.cuda:
            self.free[i % 2].record(torch.cuda.current_stream())


def shard_model(model, device_id=None, param_dtype=torch.bfloat16, reduce_dtype=torch.float32,
                buffer_dtype=torch.float32, process_group=None, sharding_strategy=None, sync_module_states=True):
    """reference signature (wan/distributed/fsdp.py:10-19).  param_dtype must be bf16 (what the
    engine stores); reduce/buffer dtypes have no role in inference."""
    if param_dtype != torch.bfloat16:
        raise NotImplementedError('the engine stores GEMM weights in bf16')
    BlockShards(model, process_group, sync_module_states)
    return model
