"""Training-side sequence-parallel WanModel forward on the MI355X engine — the counterpart of the reference's
scripts/train/model/model_seq.py: the DiT the training scripts instantiate for
sequence-parallel runs and call for validation during training.

Same class name and forward signature (reference :621-631):

    WanModel(...).forward(x, t, context, seq_len, batch_context=None, context_mask=None, clip_fea=None, y=None)

and the same parallel structure, executed by the HIP kernels of libmoviigen_hip.so:

  * the patch-embedded sequence is zero-padded to `seq_len` and chunked over the sequence-parallel ranks
    (:704-706, :757) — `seq_len % sp_size == 0`, a rank may hold padded rows;
  * `rope_apply_dist` (:37-76) = mg_rmsnorm_rope_bf16 with the rank's position offset (rows past the video's
    tokens pass through un-rotated, the reference's `pad_freqs` with ones);
  * self-attention (:197-256): packed q|k|v all-to-all ([L/P, N] -> [L, N/P]), attention with the padded keys
    masked (`k_lens=seq_lens`), all-to-all back — wan/distributed/ulysses.py;
  * cross-attention (:271-294): q through the all-to-all, K/V narrowed to the rank's heads (`shrink_head`),
    all-to-all back;
  * the head runs on the rank's rows and the [L/P, 64] fp32 results are all-gathered (the reference gathers the
    5120-wide hidden states first, :780 — same values, 80x fewer bytes).

Forward only (inference / validation): the engine has no backward pass.  The sequence-parallel state that the
reference keeps in FastVideo's `nccl_info` / `get_sequence_parallel_state()` is set with
`initialize_sequence_parallel_state(sp_size)` below (groups of `sp_size` consecutive ranks, as FastVideo builds them).
"""
import os
import sys

import torch
import torch.distributed as dist

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
_PKG = os.path.join(_ROOT, 'moviigen1.1_amd')
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from wan.distributed.xdit_context_parallel import enable_sequence_parallel  # noqa: E402
from wan.modules.model import WanModel as _EngineWanModel  # noqa: E402

__all__ = ['WanModel', 'initialize_sequence_parallel_state', 'get_sequence_parallel_state', 'nccl_info']


class _NcclInfo:
    """fastvideo.utils.parallel_states.nccl_info: the fields model_seq.py reads."""
    sp_size = 1
    rank_within_group = 0
    group = None
    group_id = 0


nccl_info = _NcclInfo()
_STATE = {'enabled': False}


def initialize_sequence_parallel_state(sequence_parallel_size):
    """groups of `sequence_parallel_size` consecutive ranks (every rank creates every group)."""
    sp = int(sequence_parallel_size)
    if sp <= 1:
        _STATE['enabled'] = False
        nccl_info.sp_size, nccl_info.rank_within_group, nccl_info.group = 1, 0, None
        return
    if not dist.is_initialized():
        raise RuntimeError('torch.distributed is not initialised (launch with torchrun, one process per GPU)')
    world, rank = dist.get_world_size(), dist.get_rank()
    assert world % sp == 0, 'world_size must be divisible by sequence_parallel_size'
    for g in range(world // sp):
        grp = dist.new_group(list(range(g * sp, (g + 1) * sp)))
        if rank // sp == g:
            nccl_info.group, nccl_info.group_id = grp, g
    nccl_info.sp_size, nccl_info.rank_within_group = sp, rank % sp
    _STATE['enabled'] = True


def get_sequence_parallel_state():
    return _STATE['enabled']


class WanModel(_EngineWanModel):
    """engine WanModel with the training-side forward signature and sequence-parallel structure."""

    def _configure(self):
        want = nccl_info.sp_size if get_sequence_parallel_state() else 1
        if want > 1 and (self.sp_size != want or self.sp_group is not nccl_info.group):
            enable_sequence_parallel(self, nccl_info.group)
        elif want == 1 and self.sp_size != 1:
            self.sp_size, self.sp_rank, self.sp_group = 1, 0, None
            self._ws = {}
            for blk in self.blocks:
                blk.self_attn.sp = None
        self.sp_mask_padded_keys = True        # pad to seq_len, chunk, mask the padded keys (:704-706, :757, :247-252)
        self.cross_attn_head_sharded = True    # :271-294

    def forward(self, x, t, context, seq_len, batch_context=None, context_mask=None, clip_fea=None, y=None):
        if clip_fea is not None or y is not None:
            raise NotImplementedError('image conditioning (i2v) is not part of MoviiGen1.1 T2V')
        self._configure()
        if context is None:
            assert batch_context is not None                                  # reference :748
            context = [batch_context[i] for i in range(len(x))]              # [text_len, text_dim] each, already padded
        if self.sp_size > 1:
            assert seq_len % self.sp_size == 0, 'seq_len must be divisible by the sequence-parallel size'
        return super().forward(x, t, context, seq_len)
