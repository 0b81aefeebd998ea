"""HBM-bound row-wise kernels of a DiT block at the metric's size (L = 131 040 tokens x 5120): microseconds and GB/s of algorithmic traffic
(bytes each kernel must move), against the 6.29 TB/s a copy kernel reaches on this part (MI355X_MICROARCH.md).
    python tools/bench_rowwise.py [L]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'moviigen1.1_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from wan.backend import ops  # noqa: E402
from wan.modules.model import rope_cos_sin  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 131040
d, heads, grid = 5120, 40, (21, 52, 120)
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(L, 3 * d, device=dev, generator=g).bfloat16()
x32 = torch.randn(L, d, device=dev, generator=g)
w = 1 + 0.1 * torch.randn(d, device=dev, generator=g)
sc, sh = torch.randn(d, device=dev, generator=g), torch.randn(d, device=dev, generator=g)
tab = rope_cos_sin(128, grid).to(dev)
q = torch.empty(L, d, dtype=torch.bfloat16, device=dev)
k = torch.empty(L, d, dtype=torch.bfloat16, device=dev)
h = torch.empty(L, d, dtype=torch.bfloat16, device=dev)
kp = torch.empty(ops.packed_kv_numel(L, heads), dtype=torch.bfloat16, device=dev)
vp = torch.empty_like(kp)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


n = L * d
rows = [
    ('rmsnorm_rope q (strided q|k|v slice -> row-major, out_scale)', lambda: ops.rmsnorm_rope(qkv[:, :d], w, 1e-6, 128, q, tab, grid, 0, out_scale=0.1275), 4 * n),
    ('rmsnorm_rope k (-> row-major)', lambda: ops.rmsnorm_rope(qkv[:, d:2 * d], w, 1e-6, 128, k, tab, grid, 0), 4 * n),
    ('rmsnorm_rope_pack_k (-> packed K tiles)', lambda: ops.rmsnorm_rope_pack_k(qkv[:, d:2 * d], w, 1e-6, kp, tab, grid, 0), 4 * n),
    ('pack_kv k + v (round 5: both re-layouts)', lambda: ops.pack_kv(k, qkv[:, 2 * d:], heads, kp, vp), 8 * n),
    ('pack_kv v only (round 6)', lambda: ops.pack_kv(None, qkv[:, 2 * d:], heads, kp, vp), 4 * n),
    ('ln_modulate fp32 -> bf16', lambda: ops.ln_modulate(x32, sc, sh, True, 1e-6, h), 6 * n),
]
for name, fn, nbytes in rows:
    us = timed(fn)
    print(json.dumps({'kernel': name, 'us': round(us, 1), 'algorithmic_GB': round(nbytes / 1e9, 3), 'TB_per_s': round(nbytes / us / 1e6, 3),
                      'frac_of_6.29': round(nbytes / us / 1e6 / 6.29, 3)}))
