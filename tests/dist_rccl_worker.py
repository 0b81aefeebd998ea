"""worker for the RCCL smoke test (-m gpu): ONE process, backend "nccl" (= RCCL), world_size 1.
Exercises the exact torch.distributed calls of the production path (all_to_all_single /
all_gather_into_tensor / broadcast / barrier on device tensors) and a forward with the sequence-
parallel branch forced on a size-1 group, which must equal the plain forward bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')]

import weights as W  # noqa: E402
import wan  # noqa: E402
from wan.distributed import fsdp, ulysses  # noqa: E402
from wan.distributed.xdit_context_parallel import enable_sequence_parallel  # noqa: E402

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29577')
torch.cuda.set_device(0)
dev = torch.device('cuda:0')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
G = dist.group.WORLD
N, hd, L = 4, 128, 96
x = torch.randn(L, 3 * N * hd, device=dev).bfloat16()
out = torch.empty(L, N * hd, dtype=torch.bfloat16, device=dev)
ulysses.seq_to_head(x[:, N * hd:2 * N * hd], out, G, 1, N, hd)
assert torch.equal(out, x[:, N * hd:2 * N * hd])
back = torch.empty_like(out)
ulysses.head_to_seq(out, back, G, 1, N, hd)
assert torch.equal(back, out)
assert torch.equal(ulysses.all_gather_seq(out, G, 1), out)
dist.barrier()
print('RCCL_COLLECTIVES_OK', flush=True)

cfg = W.SMALL_DIT_HD128
m = wan.modules.WanModel(**cfg)
m.load_state_dict(W.make_dit_params(cfg, 0))
m = m.to(dev).eval()
lat = W.randn((16, 2, 8, 8), 3).to(dev)
ctx = W.randn((20, cfg['text_dim']), 4).to(dev)
t = torch.tensor([500.0], device=dev)
ref = m([lat], t=t, context=[ctx], seq_len=32)[0].clone()
enable_sequence_parallel(m)
assert m.sp_size == 1
m.sp_force = True                       # take the Ulysses branch on the size-1 RCCL group
got = m([lat], t=t, context=[ctx], seq_len=32)[0]
assert torch.equal(got, ref), (got - ref).abs().max().item()
print('RCCL_SP_BRANCH_OK', flush=True)
# ---- the C-ABI collectives on the library's own RCCL communicator (include/moviigen_hip.h: mg_comm_*, mg_sp_*) --------
from wan.distributed import rccl_direct  # noqa: E402
dc = rccl_direct.DirectComm()
assert dc.size == 1 and dc.rank == 0 and dc.handle
xs = torch.randn(L, N * hd, device=dev).bfloat16()
r1 = torch.empty_like(xs)
dc.all_to_all(r1, xs)
wsb = torch.empty_like(xs)
o4 = torch.empty(L, N * hd, dtype=torch.bfloat16, device=dev)
dc.all_to_all_4d(x[:, N * hd:2 * N * hd], o4, N, hd, True, wsb)          # strided source (k slice of a fused qkv buffer)
b4 = torch.empty(L, 3 * N * hd, dtype=torch.bfloat16, device=dev)
dc.all_to_all_4d(o4, b4[:, :N * hd], N, hd, False, wsb)                     # strided destination
g1 = torch.empty_like(xs)
dc.all_gather(g1, xs)
torch.cuda.synchronize()
assert torch.equal(r1, xs) and torch.equal(o4, x[:, N * hd:2 * N * hd]) and torch.equal(b4[:, :N * hd], o4) and torch.equal(g1, xs)
os.environ['MOVIIGEN_SP_TRANSPORT'] = 'rccl_direct'
m._ws = {}
got = m([lat], t=t, context=[ctx], seq_len=32)[0]                          # sp_force is still on: exchange via mg_sp_all_to_all
assert torch.equal(got, ref), (got - ref).abs().max().item()
print('RCCL_DIRECT_OK', flush=True)
os.environ['MOVIIGEN_SP_TRANSPORT'] = 'peer_copy'                          # one-sided copies between two RCCL flag all-reduces
m._ws = {}
got = m([lat], t=t, context=[ctx], seq_len=32)[0]
xch = m._ws[next(iter(m._ws))]['xchg']
assert xch.peer is not None and torch.equal(got, ref), (got - ref).abs().max().item()
print('PEER_COPY_OK', flush=True)
# ---- round 6: the control plane and the preflight on the RCCL backend (device tensors, not gloo's host tensors), the W-band VAE decode's
# collectives at world size 1, and the peer-copy failure word after a forward ----------------------------------------------------------------
from wan.distributed import collectives, preflight  # noqa: E402
assert collectives.control_reduce(3, 'max', G, dev) == 3 and collectives.control_reduce(0, 'min', G, dev, dtype=torch.int32) == 0
assert collectives.control_broadcast(7.5, 0, G, dev) == 7.5
assert not xch.peer_failed() and not m._peer_transport_failed()
rep = preflight.run(None, dev, probe_peer_copy=True, budget_s=30)
assert rep['backend'] == 'nccl' and rep['rccl_ranks'] == 1 and rep['peer_access'] == [[True]] and not rep['errors'], rep
assert preflight.parse(rep)['transport_recommended'] == 'torch' and rep['rank_devices'][0]['device'] == 0
collectives.neighbor_exchange([], [], G)                                   # an edge rank with no neighbours: nothing to post
vae = wan.modules.WanVAE(state_dict=W.make_vae_params(8, 1), device=dev)
zv = W.randn((16, 3, 4, 6), 44)
assert torch.equal(vae.decode_spatial([zv], G)[0], vae.decode([zv])[0])     # P = 1: the plain decode
print('RCCL_CONTROL_PREFLIGHT_OK', flush=True)
m.sp_force = False
m = fsdp.shard_model(m, device_id=0)
got = m([lat], t=t, context=[ctx], seq_len=32)[0]
assert torch.equal(got, ref)
print('RCCL_FSDP_OK', flush=True)
os.environ.pop('MOVIIGEN_SP_TRANSPORT')
dc.destroy()
dist.destroy_process_group()
