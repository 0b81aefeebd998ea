"""worker for test_sequence_parallel_two_ranks_one_gpu: 2 processes share cuda:0, collectives go
through gloo (staged via host memory — test plumbing; production is RCCL).  Each rank runs the
Ulysses-sharded forward, then the Ulysses + block-sharded-weights forward (BASELINE configs[3]
style hybrid); both must equal the unsharded single-GPU forward bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')]

import weights as W  # noqa: E402
import wan  # noqa: E402
from wan.distributed.fsdp import shard_model  # noqa: E402
from wan.distributed.xdit_context_parallel import enable_sequence_parallel  # noqa: E402

dist.init_process_group('gloo')
rank = dist.get_rank()
dev = torch.device('cuda:0')
cfg = dict(W.SMALL_DIT_HD128, num_layers=3)
m = wan.modules.WanModel(**cfg)
m.load_state_dict(W.make_dit_params(cfg, 0))
m.to(dev)
lat, ctx = W.randn((16, 2, 8, 16), 20).to(dev), W.randn((33, cfg['text_dim']), 30).to(dev)
t = torch.tensor([650], device=dev)
L = 2 * 4 * 8
single = m([lat], t=t, context=[ctx], seq_len=L)[0].clone()
enable_sequence_parallel(m)
assert m.sp_size == 2
out = m([lat], t=t, context=[ctx], seq_len=L)[0]
# every op is row-local except attention, whose key order is unchanged: results are bit-identical
assert torch.equal(out, single), (out - single).abs().max().item()
print(f'SP_OK rank{rank}', flush=True)
shard_model(m, device_id=0)
assert m.blocks[1].ffn['0'].weight.numel() == 0
for _ in range(2):
    ctx2 = ctx.clone()                      # new prompt tensor -> cross K/V recomputed through fetch()
    out = m([lat], t=t, context=[ctx2], seq_len=L)[0]
    assert torch.equal(out, single), (out - single).abs().max().item()
print(f'SP_FSDP_OK rank{rank}', flush=True)
dist.barrier()
dist.destroy_process_group()
