"""worker for test_preflight_gloo_world2 (CPU, gloo, 2 ranks): wan/distributed/preflight.py end to end on host tensors — the stages, the
time box decided by rank 0 for everyone, the report's keys — and the fallback vote / control plane of the copy-engine transport."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'moviigen1.1_amd')]
from wan.distributed import collectives, peer_copy, preflight  # noqa: E402

dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device('cpu')
rep = preflight.run(None, dev, probe_peer_copy=False, budget_s=60, probe_bytes=1 << 20)
assert rep['backend'] == 'gloo' and rep['world'] == world and rep['rccl_ranks'] == 0 and not rep['errors'], rep
assert rep['link_gbps_measured']['all_to_all'] > 0 and rep['ipc_open'] is None and rep['recommended'] == 'torch'
assert [d['rank'] for d in rep['rank_devices']] == list(range(world)) and len(rep['peer_access']) == world
p = preflight.parse(rep)
assert p['transport_recommended'] == 'torch' and p['rccl_ranks'] == 0 and p['preflight_errors'] == [] and p['link_gbps_measured']['all_to_all'] > 0
# an exhausted time box: rank 0's clock skips the remaining stages for EVERY rank (rank 1 alone could not decide that)
rep0 = preflight.run(None, dev, probe_peer_copy=True, budget_s=0.0 if rank == 0 else 1e9, probe_bytes=1 << 20)
assert rep0['stages_skipped'] == ['all_to_all', 'peer_copy'] and not rep0['link_gbps_measured'] and not rep0['errors'], rep0
# the control plane: one number, the same on every rank
assert collectives.control_reduce(rank + 1, 'max', None, dev) == world and collectives.control_reduce(rank + 1, 'min', None, dev) == 1
assert collectives.control_broadcast(40 + rank, 1, None, dev) == 41
# the transport's all-or-none vote: ONE rank saying no means no for everyone
assert peer_copy._vote(True, dist.group.WORLD, dev) is True
assert peer_copy._vote(rank != 1, dist.group.WORLD, dev) is False
print(f'PREFLIGHT_OK rank{rank}/{world}', flush=True)
dist.barrier()
dist.destroy_process_group()
