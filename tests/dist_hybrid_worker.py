"""worker for the BASELINE configs[3] composition ("FSDP shard + SP=4 on 8 GPUs", read as SURVEY §8(e) reads it):
cond / uncond halves of the ranks (CFG-parallel) x Ulysses inside each half x DiT block weights sharded over ALL
ranks and all-gathered one block ahead.  Two transports:

  MOVIIGEN_TEST_BACKEND=gloo (default): WORLD_SIZE ranks share cuda:0, collectives staged through the host
      (test plumbing; runs on a 1-GPU box — 4 ranks = 2 halves x Ulysses 2 x 4-way block shards);
  MOVIIGEN_TEST_BACKEND=nccl: one rank per GPU over RCCL (needs WORLD_SIZE visible GPUs) — the production path,
      including the pipelined packed exchange on the communication stream.

Every rank must end up with the (cond, uncond) pair of the plain single-process forwards, bit for bit (every op
is row-local except attention, whose key order does not change), for two consecutive steps (prompt cache warm and
cold), with the pipeline depth of the exchange forced to 1, 2 and the default."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')]

import weights as W  # noqa: E402
import wan  # noqa: E402
from wan.distributed.cfg_parallel import enable_cfg_parallel  # noqa: E402
from wan.distributed.fsdp import shard_model  # noqa: E402
from wan.distributed.xdit_context_parallel import enable_sequence_parallel  # noqa: E402

backend = os.environ.get('MOVIIGEN_TEST_BACKEND', 'gloo')
local = int(os.environ.get('LOCAL_RANK', '0')) if backend == 'nccl' else 0
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
torch.cuda.set_device(local)
dev = torch.device(f'cuda:{local}')
if backend == 'nccl':
    dist.init_process_group('nccl', device_id=dev)
else:
    dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
mode = os.environ.get('MOVIIGEN_TEST_LAYOUT', 'cfg_sp_fsdp')

if os.environ.get('MOVIIGEN_TEST_MODEL') == 'width40':
    # the 14B model's REAL width and head count (dim 5120, 40 heads x 128, ffn 13824, 512 text keys), 2 layers, 256 tokens:
    # what the 8-rank layouts of BASELINE configs[2] / [3] really look like — Ulysses 8 = 5 heads per rank = five ONE-head
    # pipeline groups; cond / uncond halves x Ulysses 4 = 10 heads per rank = five 2-head groups; 8-way block shards
    heads = 40
    cfg = dict(model_type='t2v', patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=5120, ffn_dim=13824, freq_dim=256,
               text_dim=4096, out_dim=16, num_heads=heads, num_layers=2, eps=1e-6)
    lat_shape = (16, 2, 16, 32)                               # grid (2, 8, 16) = 256 tokens
else:
    heads = 8                                                 # divisible by every Ulysses degree up to 8
    cfg = dict(model_type='t2v', patch_size=(1, 2, 2), text_len=64, in_dim=16, dim=heads * 128, ffn_dim=1536, freq_dim=64,
               text_dim=128, out_dim=16, num_heads=heads, num_layers=3, eps=1e-6)
    lat_shape = (16, 2, 16, 16)                               # grid (2, 8, 8) = 128 tokens
m = wan.modules.WanModel(**cfg, device=dev)
m.load_state_dict(W.make_dit_params(cfg, 0))
lat = W.randn(lat_shape, 20).to(dev)
ctx, ctx_null = W.randn((33, cfg['text_dim']), 30).to(dev), W.randn((7, cfg['text_dim']), 31).to(dev)
L = lat_shape[1] * (lat_shape[2] // 2) * (lat_shape[3] // 2)
ts = [torch.tensor([650], device=dev), torch.tensor([333], device=dev)]
refs = [(m([lat], t=t, context=[ctx], seq_len=L)[0].clone(), m([lat], t=t, context=[ctx_null], seq_len=L)[0].clone())
        for t in ts]

if mode == 'cfg_sp_fsdp':
    cp = enable_cfg_parallel(m)
    assert cp is not None and cp.sp_size == world // 2 and m.sp_size == max(1, world // 2)
else:                                                         # 'sp_fsdp': Ulysses over all ranks (BASELINE configs[2] + shards)
    cp = None
    enable_sequence_parallel(m)
    assert m.sp_size == world
shard_model(m, device_id=local)
assert m.blocks[1].ffn['0'].weight.numel() == 0              # full copies released: 1/world of the weights per rank

for depth in ('1', '2', '5', None):      # forced pipeline depths, then the shape-aware default (MOVIIGEN_SP_GROUPS=auto)
    if depth is None:
        os.environ.pop('MOVIIGEN_SP_GROUPS', None)
    else:
        os.environ['MOVIIGEN_SP_GROUPS'] = depth
    m._ws = {}                                               # rebuild the exchange buffers with this pipeline depth
    for rep in range(2):
        for t, (ref_c, ref_u) in zip(ts, refs):
            if cp is not None:
                mine = m([lat], t=t, context=[ctx_null if cp.branch else ctx], seq_len=L)[0]
                c, u = cp.exchange(mine)
            else:
                c = m([lat], t=t, context=[ctx], seq_len=L)[0].clone()
                u = m([lat], t=t, context=[ctx_null], seq_len=L)[0].clone()
            assert torch.equal(c, ref_c) and torch.equal(u, ref_u), \
                (depth, rep, (c - ref_c).abs().max().item(), (u - ref_u).abs().max().item())
    if m.sp_size > 1:
        x = m._ws[next(iter(m._ws))]['xchg']
        if depth is None:
            from wan.distributed.ulysses import choose_groups
            want = choose_groups(heads // m.sp_size, L, m.sp_size, dim=m.dim)[0]
        else:
            want = min(heads // m.sp_size, int(depth))
        assert len(x.groups) == want, (x.groups, want)
if os.environ.get('MOVIIGEN_SP_TRANSPORT') == 'peer_copy' and cp is None and m.sp_size > 1:
    # hard fallback AFTER the windows are open: ONE rank's runtime refuses a peer copy in the middle of a forward.
    # Nobody may hang in a rendezvous; every rank must learn of it, drop its windows, repeat the forward on the all-to-all collective
    # and return the right bits.
    x = m._ws[next(iter(m._ws))]['xchg']
    assert x.peer is not None

    class Refused:
        def view(self, *a):
            raise RuntimeError('injected fault: peer copy refused')
    if rank == world - 1:
        x.peer.views[0][0] = Refused()                       # rank 0's first receive buffer as mapped HERE
    c = m([lat], t=ts[0], context=[ctx], seq_len=L)[0].clone()
    assert torch.equal(c, refs[0][0]), (c - refs[0][0]).abs().max().item()
    assert x.peer is None and not m._peer_transport_failed()
    u = m([lat], t=ts[0], context=[ctx_null], seq_len=L)[0]
    assert torch.equal(u, refs[0][1])
    # ... and BEFORE the first exchange: a copy refused inside the self-check on one rank, and a mapping that reaches the wrong memory on
    # another — the vote says no on every rank, nobody waits in a rendezvous the failing rank skipped
    from wan.distributed import peer_copy
    bufs = [torch.zeros(world, 64, 48, dtype=torch.bfloat16, device=dev) for _ in range(2)]
    win = peer_copy.PeerWindows(None, bufs)
    assert peer_copy.self_check(win, dist.group.WORLD) is win and all(int(b.count_nonzero()) == 0 for b in bufs)      # intact: passes, buffers restored
    if rank == world - 1:
        win.views[1][0] = Refused()
    assert peer_copy.self_check(win, dist.group.WORLD) is None
    win2 = peer_copy.PeerWindows(None, bufs)
    if rank == 0 and world > 1:
        win2.views[0][1] = torch.zeros_like(bufs[0])             # "rank 1's buffer" as mapped on rank 0 is some other memory
    assert peer_copy.self_check(win2, dist.group.WORLD) is None
    print(f'PEER_FALLBACK_OK rank{rank}/{world}', flush=True)
torch.cuda.synchronize()
print(f'HYBRID_OK {mode} {backend} rank{rank}/{world}', flush=True)
dist.barrier()
dist.destroy_process_group()
