"""Time the umT5-XXL encoder (random-init, real architecture) on one MI355X: `python tools/bench_t5.py [tokens]`."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'moviigen1.1_amd'))
from wan.modules.t5 import umt5_xxl  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda:0')
m = umt5_xxl(device=dev)
g = torch.Generator(device=dev).manual_seed(0)
for name, p in m.named_parameters():
    if 'norm' in name:
        p.data.fill_(1.0)
    else:
        p.data.copy_(torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32).mul_(p.shape[-1] ** -0.5 * 0.5))
ids = torch.randint(1, 256384, (1, 512), generator=torch.Generator().manual_seed(1))
mask = torch.zeros(1, 512, dtype=torch.long)
mask[:, :n] = 1
for _ in range(2):
    out = m(ids.to(dev), mask.to(dev))
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    out = m(ids.to(dev), mask.to(dev))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
flops = 24 * (2 * n * 4096 * (4 * 4096 + 3 * 10240) + 4 * n * n * 4096)
print(f'umT5-XXL encode {n} tokens: {dt * 1e3:.2f} ms  ({flops / dt / 1e12:.1f} TFLOP/s), finite={bool(torch.isfinite(out.float()).all())}')
