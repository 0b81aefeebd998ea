"""Sequence-parallel entry points with the reference's names (wan/distributed/
xdit_context_parallel.py:65-198).  The reference installs `usp_dit_forward` / `usp_attn_forward`
by method replacement around xfuser; here sequence parallelism is a property of the engine
(WanModel.sp_size/sp_rank/sp_group) and the collectives are in ulysses.py.  `usp_dit_forward` configures it and
runs the fused forward; `usp_attn_forward` is the stand-alone sequence-parallel attention operator with the
reference's call shape, so the reference's installation sequence (text2video.py:97-100)

    for block in model.blocks:
        block.self_attn.forward = types.MethodType(usp_attn_forward, block.self_attn)
    model.forward = types.MethodType(usp_dit_forward, model)

works unchanged."""
import torch.distributed as dist


def enable_sequence_parallel(model, group=None):
    """shard the token axis of `model` over `group` (default: WORLD).  heads % size must be 0."""
    if not dist.is_initialized():
        raise RuntimeError('torch.distributed is not initialised (launch with torchrun, one process per GPU)')
    size = dist.get_world_size(group)
    if model.num_heads % size:
        raise ValueError(f'`num_heads` {model.num_heads} cannot be divided evenly by the sequence-parallel '
                         f'size {size}')  # reference generate.py:238-239
    model.sp_group = group if group is not None else dist.group.WORLD
    model.sp_size = size
    model.sp_rank = dist.get_rank(group)
    model.ring, model.uly_group, model.uly_size, model.ring_group, model.ring_size = False, None, None, None, None
    model._ws = {}
    tag_attention_modules(model)
    return model


def tag_attention_modules(model):
    """record on every self-attention module WHICH ranks its stand-alone operator (usp_attn_forward) exchanges with:
    the model's Ulysses group, its size and this rank's index in it — the CFG-parallel halves, the training-side
    sub-groups and the hybrid layout's inner groups all differ from WORLD."""
    U, R = model._sp_layout()
    group = model.uly_group if model.uly_group is not None else model.sp_group
    for blk in model.blocks:
        if R > 1:       # ring / hybrid layouts: K/V blocks travel between the groups — only the fused forward does that
            blk.self_attn.sp = 'ring'
        else:           # (group, size, rank inside the group, rank that sets the RoPE position offset)
            blk.self_attn.sp = (group, U, model.sp_rank % U, model.sp_rank) if U > 1 else None
    return model


def usp_dit_forward(self, x, t, context, seq_len, clip_fea=None, y=None, guidance=None):
    if self.sp_size == 1 and dist.is_initialized() and dist.get_world_size() > 1:
        enable_sequence_parallel(self)
    return type(self).forward(self, x, t, context, seq_len, clip_fea=clip_fea, y=y)


def usp_attn_forward(self, x, seq_lens, grid_sizes, freqs, dtype=None):
    """reference xdit_context_parallel.py:155-198, same call shape: `self` is a block's self-attention module,
    x [B, L/P, C] the rank's token shard; RoPE with the rank's position offset, packed q|k|v all-to-all, attention over
    all tokens x heads/P, all-to-all back, output projection.  Installing it with
    `types.MethodType(usp_attn_forward, block.self_attn)` (reference text2video.py:97-100) is honoured: WanModel.forward
    recognises it and keeps its fused, pipelined implementation of exactly this operator."""
    sp = getattr(self, 'sp', None)      # set by enable_sequence_parallel / enable_cfg_parallel / enable_hybrid_sp
    if sp is None and not hasattr(self, 'sp') and dist.is_initialized() and dist.get_world_size() > 1:
        sp = (dist.group.WORLD, dist.get_world_size(), dist.get_rank(), dist.get_rank())   # never configured: the reference's default
    return type(self).forward(self, x, seq_lens, grid_sizes, freqs, sp=sp)
