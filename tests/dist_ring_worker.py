"""worker for test_ring_attention_one_gpu: WORLD_SIZE processes share cuda:0 (gloo, staged via host
memory — test plumbing; production is RCCL isend/irecv).  Ring attention is not bit-identical to one
long softmax (the per-block results are merged in fp32 from bf16 partials): bf16 tolerance."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')]

import weights as W  # noqa: E402
import wan  # noqa: E402
from wan.backend import ops  # noqa: E402
from wan.distributed.ring import enable_ring_attention, ring_attention  # noqa: E402

dist.init_process_group('gloo')
rank, P = dist.get_rank(), dist.get_world_size()
dev = torch.device('cuda:0')

# ---- operator level: 3 heads (not divisible by 2), ragged block length, vs one full softmax ----------------
heads, Lloc = 3, 100
L = Lloc * P
q = (W.randn((L, heads * 128), 61) * 1.5).bfloat16().to(dev)
k = (W.randn((L, heads * 128), 62) * 1.5).bfloat16().to(dev)
v = W.randn((L, heads * 128), 63).bfloat16().to(dev)
k[17] = (q[150 % L] * 3).clone()                      # a key that dominates one row from another rank's block
scale = 128 ** -0.5
kp, vp = torch.empty(ops.packed_kv_numel(L, heads), dtype=torch.bfloat16, device=dev), torch.empty(
    ops.packed_kv_numel(L, heads), dtype=torch.bfloat16, device=dev)
ops.pack_kv(k, v, heads, kp, vp)
full = torch.empty(L, heads * 128, dtype=torch.bfloat16, device=dev)
ops.attention_hd128(q, kp, vp, full, L, heads, scale)
n1 = ops.packed_kv_numel(Lloc, heads)
e = lambda *s, dt=torch.bfloat16: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
ws = dict(kp0=e(n1), vp0=e(n1), kp1=e(n1), vp1=e(n1), part=e(Lloc, heads * 128), acc=e(Lloc, heads * 128, dt=torch.float32),
          lse=e(heads, Lloc, dt=torch.float32), lse_acc=e(heads, Lloc, dt=torch.float32))
sl = slice(rank * Lloc, (rank + 1) * Lloc)
out = e(Lloc, heads * 128)
ring_attention(q[sl], k[sl], v[sl], out, ws, dist.group.WORLD, P, rank, heads, scale)
err = (out.float() - full[sl].float()).abs().max().item()
assert err < 2e-2, err
# lse of the merged result == lse of the full softmax
s_full = (q[sl].float().view(Lloc, heads, 128).permute(1, 0, 2) @ k.float().view(L, heads, 128).permute(1, 2, 0)) * scale
assert (ws['lse_acc'] - torch.logsumexp(s_full, -1)).abs().max().item() < 2e-3
print(f'RING_OP_OK rank{rank}/{P} err {err:.2e}', flush=True)

# ---- model level: whole forward, token-sharded with ring attention (2 heads: not divisible by 3) ------------
cfg = dict(W.SMALL_DIT_HD128, num_layers=2)
m = wan.modules.WanModel(**cfg)
m.load_state_dict(W.make_dit_params(cfg, 0))
m.to(dev)
lat, ctx = W.randn((16, 3, 8, 16), 20).to(dev), W.randn((33, cfg['text_dim']), 30).to(dev)
t = torch.tensor([650], device=dev)
Lm = 3 * 4 * 8
single = m([lat], t=t, context=[ctx], seq_len=Lm)[0].clone()
enable_ring_attention(m)
assert m.sp_size == P and m.ring
got = m([lat], t=t, context=[ctx], seq_len=Lm)[0]
rel = ((got - single).norm() / single.norm()).item()
assert rel < 5e-3, rel
print(f'RING_MODEL_OK rank{rank}/{P} rel {rel:.2e}', flush=True)

# ---- hybrid: Ulysses 2 x ring P/2 (the reference CLI's --ulysses_size 2 --ring_size P/2) ---------------------
if P % 2 == 0 and P >= 4:
    from wan.distributed.ring import enable_hybrid_sp
    enable_hybrid_sp(m, 2, P // 2)
    assert m._sp_layout() == (2, P // 2) and m.sp_size == P
    got = m([lat], t=t, context=[ctx], seq_len=Lm)[0]
    rel = ((got - single).norm() / single.norm()).item()
    assert rel < 5e-3, rel
    print(f'HYBRID_OK rank{rank}/{P} rel {rel:.2e}', flush=True)
dist.barrier()
dist.destroy_process_group()
