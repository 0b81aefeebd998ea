"""Ring attention over RCCL.  The reference gets it from yunchang inside
xFuserLongContextAttention when `--ring_size > 1` (scripts/inference/generate.py:102-106,225-229);
here it is the engine's second sequence-parallel layout next to Ulysses:

  tokens are sharded as for Ulysses (rank r owns [r*L/P, (r+1)*L/P)), but every rank keeps ALL heads.
  The packed K/V block of a rank travels round the ring (send to r+1, receive from r-1, P-1 hops,
  overlapped with the attention of the block at hand on a second buffer); each hop's result
  (normalised bf16 output + log-sum-exp, mg_attn_fwd_bf16_hd128_lse) is folded into an fp32 running
  result (mg_attn_merge_f32).  No head-count divisibility requirement (Ulysses needs heads % P == 0).

Not bit-identical to the single-GPU result (the merge rounds differently from one long softmax):
tested to the bf16 tolerance."""
import torch
import torch.distributed as dist

from ..backend import ops
from . import collectives


def enable_ring_attention(model, group=None):
    """shard the token axis of `model` over `group` with ring attention instead of Ulysses."""
    if not dist.is_initialized():
        raise RuntimeError('torch.distributed is not initialised (launch with torchrun, one process per GPU)')
    if model.dim // model.num_heads != 128:
        raise NotImplementedError('ring attention is built on the head_dim 128 kernels')
    model.sp_group = group if group is not None else dist.group.WORLD
    model.sp_size = dist.get_world_size(group)
    model.sp_rank = dist.get_rank(group)
    model.ring = True
    model.uly_group, model.uly_size, model.ring_group, model.ring_size = None, None, None, None
    model._ws = {}
    from .xdit_context_parallel import tag_attention_modules
    tag_attention_modules(model)        # the stand-alone self-attention operator then refuses the ring layout instead of
    return model                        # silently running Ulysses over WORLD


def enable_hybrid_sp(model, ulysses_size, ring_size, group=None):
    """the reference CLI's `--ulysses_size U --ring_size R` (U * R == world size, generate.py:209-229): tokens are
    sharded over all U*R ranks; ranks [g*U, (g+1)*U) form Ulysses group g (all-to-all: the group's tokens, heads/U),
    ranks {u, u+U, u+2U, ...} form the ring that rotates the K/V blocks of the groups.  Needs heads % U == 0."""
    if not dist.is_initialized():
        raise RuntimeError('torch.distributed is not initialised (launch with torchrun, one process per GPU)')
    P, rank = dist.get_world_size(group), dist.get_rank(group)
    U, R = int(ulysses_size), int(ring_size)
    if U * R != P:
        raise ValueError('The number of ulysses_size and ring_size should be equal to the world size.')
    if model.num_heads % U:
        raise ValueError(f'`num_heads` {model.num_heads} cannot be divided evenly by the ulysses size {U}')
    if R > 1 and model.dim // model.num_heads != 128:
        raise NotImplementedError('ring attention is built on the head_dim 128 kernels')
    glob = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    uly_groups = [dist.new_group([glob(g * U + u) for u in range(U)]) for g in range(R)]     # every rank creates every group
    ring_groups = [dist.new_group([glob(u + g * U) for g in range(R)]) for u in range(U)]
    model.sp_group = group if group is not None else dist.group.WORLD
    model.sp_size, model.sp_rank = P, rank
    model.uly_group, model.uly_size = uly_groups[rank // U], U
    model.ring_group, model.ring_size, model.ring_rank = ring_groups[rank % U], R, rank // U
    model.ring = R > 1
    model._ws = {}
    from .xdit_context_parallel import tag_attention_modules
    tag_attention_modules(model)
    return model


def ring_attention(q, k, v, out, ws, group, P, rank, heads, scale, prescaled=False):
    """q, k, v [Lloc, heads*128] bf16 (row strides free), out [Lloc, heads*128] bf16; prescaled: q already carries
    scale*log2(e) (ops.rmsnorm_rope out_scale).
    ws: dict with packed buffers kp0/vp0/kp1/vp1, part (bf16 [Lloc, heads*128]), acc (fp32), lse, lse_acc."""
    Lloc = q.shape[0]
    cur, nxt = (ws['kp0'], ws['vp0']), (ws['kp1'], ws['vp1'])
    ops.pack_kv(k, v, heads, cur[0], cur[1])
    for j in range(P):
        pending = None
        if j + 1 < P:
            pending = collectives.ring_hop(list(cur), list(nxt), group, P, rank)     # send to rank+1, receive from rank-1
        ops.attention_hd128_lse(q, cur[0], cur[1], ws['part'], ws['lse'], Lloc, heads, scale, prescaled=prescaled)
        ops.attention_merge(ws['acc'], ws['lse_acc'], ws['part'], ws['lse'], heads, first=(j == 0),
                            out=out if j == P - 1 else None)
        if pending is not None:
            pending.wait()
            cur, nxt = nxt, cur
    return out
