"""worker for test_vae_pipelined_one_gpu: WORLD_SIZE processes share cuda:0 (gloo, staged via host
memory — test plumbing; production is RCCL send/recv).  The layer-pipelined decode must give rank 0
the single-GPU video bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')]

import weights as W  # noqa: E402
import wan  # noqa: E402
from wan.modules.vae import partition_costs  # noqa: E402

dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
vae = wan.modules.WanVAE(state_dict=W.make_vae_params(8, 1), device='cuda:0')
z = W.randn((16, 4, 6, 10), 41)
ref = vae.decode([z])[0]
out = vae.decode_pipelined([z])[0]
if rank == 0:
    assert out is not None and torch.equal(out, ref), (out - ref).abs().max().item()
    costs = vae.model.stage_costs(6, 10)
    cuts = partition_costs(costs, world)
    assert cuts[0] == 0 and cuts[-1] == len(costs) and all(b > a for a, b in zip(cuts, cuts[1:]))
else:
    assert out is None
print(f'VAEPIPE_OK rank{rank}/{world}', flush=True)
dist.barrier()
dist.destroy_process_group()
