"""worker for test_ulysses_gloo_world2 (launched by torch.distributed.run, backend gloo)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')]

from oracle import dit  # noqa: E402
from wan.distributed import ulysses  # noqa: E402

dist.init_process_group('gloo')
rank, P = dist.get_rank(), dist.get_world_size()
N, hd, L = 4, 8, 12
g = torch.Generator().manual_seed(5)
full = [torch.randn(L // P, N * hd, generator=g) for _ in range(P)]       # every rank builds all shards
sim = dit.all_to_all_seq_to_head([f.view(L // P, N, hd) for f in full])  # oracle simulation
out = torch.empty(L, (N // P) * hd)
ulysses.seq_to_head(full[rank], out, dist.group.WORLD, P, N, hd)
assert torch.equal(out.view(L, N // P, hd), sim[rank]), 'seq_to_head'
# strided input (the v slice of a fused qkv buffer)
wide = torch.zeros(L // P, 3 * N * hd)
wide[:, 2 * N * hd:] = full[rank]
out2 = torch.empty(L, (N // P) * hd)
ulysses.seq_to_head(wide[:, 2 * N * hd:], out2, dist.group.WORLD, P, N, hd)
assert torch.equal(out2, out)
back = torch.empty(L // P, N * hd)
ulysses.head_to_seq(out, back, dist.group.WORLD, P, N, hd)
assert torch.equal(back, full[rank]), 'head_to_seq inverse'
sim_back = dit.all_to_all_head_to_seq(sim)
assert torch.equal(back.view(L // P, N, hd), sim_back[rank])
gath = ulysses.all_gather_seq(full[rank], dist.group.WORLD, P)
assert torch.equal(gath, torch.cat(full, 0)), 'all_gather_seq'
print(f'ULYSSES_OK rank{rank}', flush=True)
dist.destroy_process_group()
