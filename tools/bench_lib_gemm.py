"""Library GEMM (torch.nn.functional.linear on ROCm = hipBLASLt / rocBLAS) against mg_gemm_bf16 on the same operands, same
process, alternating rounds: the four weight shapes of a DiT block at M tokens, bf16 in / bf16 out, fp32 accumulate, + bias.
    python tools/bench_lib_gemm.py [M]
A reference point for DESIGN.md 3.2 — the product never calls torch arithmetic."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'moviigen1.1_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from wan.backend import ops  # noqa: E402

from wan.backend import lib as _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 131040
if os.environ.get('MG_GEMM_VARIANT'):       # measurement only: 0 = the library's own choice by shape
    assert _lib.ab_library().__enter__().mg_gemm_set_variant(int(os.environ['MG_GEMM_VARIANT'])) == 0      # the A/B library, for the whole process
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


SHAPES = ((15360, 5120, 'q|k|v'), (5120, 5120, 'cross q'), (13824, 5120, 'ffn.0 (bias only)'), (5120, 13824, 'ffn.2 (bias only)'))
if os.environ.get('MG_LIB_GEMM_SHAPES'):       # "N,K;N,K": other shapes (e.g. 5120,40960: 640 k-tiles per output tile, the k-loop alone)
    SHAPES = tuple((int(a.split(',')[0]), int(a.split(',')[1]), a) for a in os.environ['MG_LIB_GEMM_SHAPES'].split(';'))
for (N, K, name) in SHAPES:
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    Wt = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    bias_f = torch.randn(N, device=dev, generator=g)
    bias_b = bias_f.bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    res = {'shape': name, 'M': M, 'N': N, 'K': K}
    for r in range(2):
        t_lib = timed(lambda: torch.nn.functional.linear(A, Wt, bias_b))
        t_mg = timed(lambda: ops.gemm(A, Wt, bias_f, ops.BIAS_BF16, out))
        res[f'round{r}'] = {'torch_linear_tflops': round(fl / t_lib / 1e9, 1), 'mg_gemm_bf16_tflops': round(fl / t_mg / 1e9, 1)}
    ref = torch.nn.functional.linear(A[:256], Wt, bias_b).float()
    res['max_abs_diff_first_256_rows'] = (out[:256].float() - ref).abs().max().item()
    print(json.dumps(res), flush=True)
    del A, Wt, out
