"""Sequence-parallel entry points with the reference's names (wan/distributed/
xdit_context_parallel.py:65-198).  The reference installs `usp_dit_forward` / `usp_attn_forward`
by method replacement around xfuser; here sequence parallelism is a property of the engine
(WanModel.sp_size/sp_rank/sp_group) and the collectives are in ulysses.py — these functions
only configure it, so `types.MethodType(usp_dit_forward, model)` style callers keep working."""
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
    return model


def usp_dit_forward(self, x, t, context, seq_len, clip_fea=None, y=None, guidance=None):
    if self.sp_size == 1 and dist.is_initialized() and dist.get_world_size() > 1:
        enable_sequence_parallel(self)
    return type(self).forward(self, x, t, context, seq_len, clip_fea=clip_fea, y=y)


def usp_attn_forward(self, x, seq_lens, grid_sizes, freqs, dtype=None):
    raise RuntimeError('self-attention is fused into WanModel.forward on this engine; enable sequence '
                       'parallelism with enable_sequence_parallel(model) instead of patching self_attn')
