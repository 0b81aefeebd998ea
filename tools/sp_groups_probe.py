"""What the Ulysses pipeline depth costs in attention time on ONE GPU: after the seq -> head exchange a rank attends ALL L queries for its
n_loc heads, cut into G launches (wan/distributed/ulysses.py: HeadExchange).  Times the G launches of every split G = 1..5 at the BASELINE
shapes and prints them beside attention_rounds() and the split choose_groups() picks.
    python tools/sp_groups_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'moviigen1.1_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from wan.backend import ops  # noqa: E402
from wan.distributed.ulysses import attention_rounds, choose_groups, split_heads  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for name, (n_loc, L, P) in {'configs[2] 1920x832 ulysses 8': (5, 131040, 8), 'configs[3] 1920x1056 cfg2 x ulysses 4': (10, 166320, 4),
                            '1280x720 ulysses 8 (published-reference case)': (5, 75600, 8), '1920x832 cfg2 x ulysses 4 (bench --gpus 8)': (10, 131040, 4)}.items():
    q = torch.randn(L, n_loc * 128, device=dev, generator=g).bfloat16()
    k = torch.randn(L, n_loc * 128, device=dev, generator=g).bfloat16()
    v = torch.randn(L, n_loc * 128, device=dev, generator=g).bfloat16()
    o = torch.empty_like(q)
    pick = choose_groups(n_loc, L, P)
    print(f'{name}: L = {L}, {n_loc} local heads; choose_groups -> {pick[0]} groups, {pick[1]} rounds')
    for G in range(1, min(5, n_loc) + 1):
        groups = split_heads(n_loc, G)
        bufs = []
        for h0, n in groups:
            kp = torch.empty(ops.packed_kv_numel(L, n), dtype=torch.bfloat16, device=dev)
            vp = torch.empty_like(kp)
            ops.pack_kv(k[:, h0 * 128:(h0 + n) * 128], v[:, h0 * 128:(h0 + n) * 128], n, kp, vp)
            bufs.append((q[:, h0 * 128:(h0 + n) * 128], kp, vp, o[:, h0 * 128:(h0 + n) * 128], n))

        def layer():
            for qq, kp, vp, oo, n in bufs:
                ops.attention_hd128(qq, kp, vp, oo, L, n, 1.0, prescaled=True)
        layer()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            layer()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        rounds = sum(attention_rounds(n, L) for _, n in groups)
        print(f'    G = {G}: heads per group {[n for _, n in groups]}  rounds {rounds:3d}  attention {ms:8.2f} ms per layer'
              f'  = {4.0 * L * L * 128 * n_loc / ms / 1e9:7.1f} TFLOP/s' + ('   <- chosen' if G == pick[0] else ''))
        del bufs
    del q, k, v, o
