"""worker for test_cfg_parallel_one_gpu: WORLD_SIZE 2 (cond | uncond, no Ulysses) or 4 (two halves x
Ulysses 2) processes share cuda:0, collectives through gloo (staged via host memory — test plumbing;
production is RCCL).  Every rank must end up with the (cond, uncond) pair of the plain
single-process forwards, bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')]

import weights as W  # noqa: E402
import wan  # noqa: E402
from wan.distributed.cfg_parallel import enable_cfg_parallel  # noqa: E402

dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device('cuda:0')
cfg = dict(W.SMALL_DIT_HD128, num_layers=2)
m = wan.modules.WanModel(**cfg)
m.load_state_dict(W.make_dit_params(cfg, 0))
m.to(dev)
lat = W.randn((16, 2, 8, 16), 20).to(dev)
ctx, ctx_null = W.randn((33, cfg['text_dim']), 30).to(dev), W.randn((7, cfg['text_dim']), 31).to(dev)
t = torch.tensor([650], device=dev)
L = 2 * 4 * 8
ref_c = m([lat], t=t, context=[ctx], seq_len=L)[0].clone()
ref_u = m([lat], t=t, context=[ctx_null], seq_len=L)[0].clone()
cp = enable_cfg_parallel(m)
assert cp is not None and cp.sp_size == world // 2 and m.sp_size == max(1, world // 2)
assert cp.branch == rank // (world // 2)
mine = m([lat], t=t, context=[ctx_null if cp.branch else ctx], seq_len=L)[0]
c, u = cp.exchange(mine)
assert torch.equal(c, ref_c) and torch.equal(u, ref_u), ((c - ref_c).abs().max().item(), (u - ref_u).abs().max().item())
print(f'CFGP_OK rank{rank}/{world}', flush=True)
dist.barrier()
dist.destroy_process_group()
