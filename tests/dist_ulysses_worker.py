"""worker for the CPU multi-process tests (launched by torch.distributed.run, backend gloo,
world_size 2): Ulysses data movement and block-sharded weights."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')]

import weights as W  # noqa: E402
from oracle import dit  # noqa: E402
import wan  # noqa: E402
from wan.distributed import fsdp, ulysses  # noqa: E402

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

# ---- CFG-parallel groups: 2 ranks = cond | uncond, no Ulysses inside the halves -----------------------
from wan.distributed.cfg_parallel import enable_cfg_parallel  # noqa: E402


class _NoModel:
    sp_size = 1


cp = enable_cfg_parallel(_NoModel())
assert cp is not None and cp.branch == rank and cp.sp_size == 1
c, u = cp.exchange(torch.full((2, 3), float(rank + 1)))
assert torch.equal(c, torch.full((2, 3), 1.0)) and torch.equal(u, torch.full((2, 3), 2.0))
print(f'CFGP_HOST_OK rank{rank}', flush=True)

# ---- training-side sequence-parallel state (scripts/train/model/model_seq.py: FastVideo's nccl_info) ----------------
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location('model_seq', os.path.join(ROOT, 'scripts', 'train', 'model', 'model_seq.py'))
ms = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ms)
assert not ms.get_sequence_parallel_state() and ms.nccl_info.sp_size == 1
ms.initialize_sequence_parallel_state(P)
assert ms.get_sequence_parallel_state() and ms.nccl_info.sp_size == P and ms.nccl_info.rank_within_group == rank
assert dist.get_world_size(ms.nccl_info.group) == P and ms.nccl_info.group_id == 0
msm = ms.WanModel(**dict(W.TINY_DIT, num_layers=1))
msm._configure()
assert msm.sp_size == P and msm.sp_rank == rank and msm.sp_mask_padded_keys and msm.cross_attn_head_sharded
ms.initialize_sequence_parallel_state(1)
msm._configure()
assert msm.sp_size == 1 and not ms.get_sequence_parallel_state()
print(f'TRAIN_SP_STATE_OK rank{rank}', flush=True)

# ---- pipelined exchange: head groups ---------------------------------------------------------------------------
assert ulysses.split_heads(5, 5) == [(0, 1), (1, 1), (2, 1), (3, 1), (4, 1)]
assert ulysses.split_heads(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)] and ulysses.split_heads(3, 8) == [(0, 1), (1, 1), (2, 1)]
assert ulysses.split_heads(20, 1) == [(0, 20)]

# ---- block-sharded weights: every rank keeps 1/P, fetch(i) reassembles block i exactly -------------
cfg = dict(W.TINY_DIT, num_layers=3)
Pm = W.make_dit_params(cfg, 0)
m = wan.modules.WanModel(**cfg)
m.load_state_dict(Pm)
if rank == 1:  # sync_module_states must overwrite rank 1's (deliberately different) weights
    for p in m.parameters():
        if p.dtype == torch.bfloat16:
            p.data.add_(1.0)
fsdp.shard_model(m, device_id=None)
sh = m._shards
per_rank = sum(s.numel() for s in sh.shards)
total = sum(Pm[f'blocks.{i}.{n}.weight'].numel() for i in range(3) for n in fsdp.ORDER)
assert per_rank * P >= total and per_rank * P < total + 3 * P * 8
assert m.blocks[0].ffn['0'].weight.numel() == 0          # full copies released
for rep in range(2):
    for i in range(3):
        v = sh.fetch(i)
        for n in fsdp.ORDER:
            ref = Pm[f'blocks.{i}.{n}.weight'].to(torch.bfloat16)
            assert torch.equal(v[n], ref), (i, n)
        d = cfg['dim']
        assert torch.equal(v['wqkv'][d:2 * d], Pm[f'blocks.{i}.self_attn.k.weight'].to(torch.bfloat16))
        assert torch.equal(v['wkv_c'][d:], Pm[f'blocks.{i}.cross_attn.v.weight'].to(torch.bfloat16))
print(f'SHARDS_OK rank{rank}', flush=True)
dist.destroy_process_group()
