"""Energy per launch of the two MFMA kernels and of everything that was tried against them ("attack energy per FLOP ... report J/launch
next to TFLOP/s for every variant ... or a table showing W and cycles per tile for each attempt that pins the floor").

Both kernels run against the package-power limit (PPT residency 0.6-0.7, ~1.9 GHz of 2.4): time per launch = energy per launch / 1.4 kW, so a variant is
faster exactly when it spends fewer joules on the same FLOPs.  For every row: ONE kernel back to back for `--secs` seconds on fixed operands, in-process
telemetry (tools/gpu_telemetry.py: power, clock, PPT residency, the energy accumulator) over exactly that loop, and

    TFLOP/s | ms per launch | W | MHz | J per launch = W x ms | pJ per FLOP | shader cycles per launch = ms x MHz | PPT residency

Rows: the shipped attention kernel (filler placement 1, items by ticket) and its A/B partners (placements 0 / 2 / 3 / 4 / 5, the static partition, the
round-2 kernel w64); the SAME shipped binary on operands that do not toggle (all zero / one constant): the instruction stream is identical, only the
switching activity differs — what the power limit itself costs; GEMM variants 12 (shipped) / 11 / 8 on the o-projection and ffn.0 shapes, random and zero A.

    python tools/energy_table.py [--secs 6] [--heads 8] [--only attn|gemm]          (A/B library: measurement only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from gpu_telemetry import GpuTelemetry  # noqa: E402
from wan.backend import lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--secs', type=float, default=6.0)
ap.add_argument('--heads', type=int, default=8)
ap.add_argument('--only', default=None, choices=['attn', 'gemm'])
ap.add_argument('--L', type=int, default=131040)
args = ap.parse_args()
dev = torch.device('cuda:0')
L = args.L
rows = []


def loop(label, fn, flops, group=4):
    fn()
    torch.cuda.synchronize()
    tel = GpuTelemetry(device_index=0, period_s=0.2).start()
    ms, n = 0.0, 0
    t0 = time.perf_counter()
    while ms < args.secs * 1e3:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(group):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms += a.elapsed_time(b)
        n += group
    wall = time.perf_counter() - t0
    t = tel.stop()
    per = ms / n
    w, mhz = t.get('power_w_mean'), t.get('sclk_mhz_mean')
    row = {'label': label, 'tflops': flops / (per * 1e-3) / 1e12, 'ms_per_launch': per, 'launches': n, 'power_w_mean': w, 'sclk_mhz_mean': mhz,
           'j_per_launch': (w * per * 1e-3) if w else None, 'pj_per_flop': (w * per * 1e-3 / flops * 1e12) if w else None,
           'mcycles_per_launch': (per * 1e-3 * mhz) if mhz else None, 'busy_frac_of_wall': ms / 1e3 / wall,
           'ppt_residency': (t.get('residency') or {}).get('ppt', (t.get('residency') or {}).get('viol_ppt_pwr')), 'vgfx_mv_mean': t.get('vgfx_mv_mean'),
           'energy_j_accumulator': t.get('energy_j'), 'temp_c_max': t.get('temp_c_max')}
    rows.append(row)
    print(json.dumps(row), flush=True)
    time.sleep(1.0)          # the same pause between rows: every row starts from a comparable die temperature
    return row


with lib.ab_library() as h:
    if args.only in (None, 'attn'):
        heads = args.heads
        g = torch.Generator(device=dev).manual_seed(1234)
        q = torch.randn(L, heads * 128, device=dev, generator=g).bfloat16()
        k = torch.randn(L, heads * 128, device=dev, generator=g).bfloat16()
        v = torch.randn(L, heads * 128, device=dev, generator=g).bfloat16()
        kp = torch.empty(ops.packed_kv_numel(L, heads), dtype=torch.bfloat16, device=dev)
        vp = torch.empty_like(kp)
        o = torch.empty_like(q)
        fl = 4.0 * L * L * 128 * heads

        def attn():
            ops.attention_hd128(q, kp, vp, o, L, heads, 1.0, prescaled=True)
        ops.pack_kv(k, v, heads, kp, vp)
        loop('attention m16, shipped (placement 1's gaps, MFMAs in quads, tickets), random operands', attn, fl)
        for name, dbg in (('placement 0 (round-4 order)', 12), ('placement 2', 4), ('placement 3', 6), ('placement 4', 8), ('placement 5', 10),
                          ('shipped placement, STATIC per-XCD partition', 16)):
            h.mg_attn_w64_debug(dbg)
            loop(f'attention m16, {name}, random operands', attn, fl)
        h.mg_attn_w64_debug(0)
        loop('attention m16, shipped, random operands (again: drift of the box over the table)', attn, fl)
        assert h.mg_attn_set_variant(3) == 0
        ops.pack_kv(k, v, heads, kp, vp)                       # the round-2 kernel reads K rows in natural order
        loop('attention w64 (round-2 kernel, 32x32x16), random operands', attn, fl)
        assert h.mg_attn_set_variant(0) == 0
        for name, fill in (('all-zero operands', 0.0), ('constant operands (0x3c3c)', None)):
            for t_ in (q, kp, vp):
                if fill is None:
                    t_.view(torch.int16).fill_(0x3c3c)
                else:
                    t_.zero_()
            loop(f'attention m16, shipped, {name}: the same instruction stream without switching', attn, fl)
        del q, k, v, kp, vp, o
        torch.cuda.empty_cache()
    if args.only in (None, 'gemm'):
        g = torch.Generator(device=dev).manual_seed(7)
        for name, N, K, epi in (('o-projection 5120 x 5120 + gated residual', 5120, 5120, ops.GATE_RESID_F32),
                                ('ffn.0 13824 x 5120 + GELU', 13824, 5120, ops.BIAS_GELU_BF16)):
            a_ = torch.randn(L, K, device=dev, generator=g).bfloat16()
            w_ = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
            b_ = torch.randn(N, device=dev, generator=g)
            gate = torch.randn(N, device=dev, generator=g) * 0.1
            out = torch.zeros(L, N, dtype=torch.float32 if epi == ops.GATE_RESID_F32 else torch.bfloat16, device=dev)
            fl = 2.0 * L * N * K

            def gemm():
                ops.gemm(a_, w_, b_, epi, out, gate=gate if epi == ops.GATE_RESID_F32 else None)
            for var in (12, 11, 8):
                assert h.mg_gemm_set_variant(var) == 0
                loop(f'GEMM variant {var}{" (shipped)" if var == 12 else ""}, {name}, random operands', gemm, fl, group=20)
            assert h.mg_gemm_set_variant(12) == 0
            a_.zero_()
            loop(f'GEMM variant 12 (shipped), {name}, A all zero: the same instruction stream without switching', gemm, fl, group=20)
            del a_, w_, b_, out
            torch.cuda.empty_cache()

print(f'\n{"kernel / variant":112s} {"TFLOP/s":>8s} {"ms":>8s} {"W":>6s} {"MHz":>6s} {"J/launch":>9s} {"pJ/FLOP":>8s} {"Mcyc":>7s} {"PPT":>5s}')
for r in rows:
    f = lambda v, fmt: (fmt % v) if v is not None else '-'  # noqa: E731
    print(f'{r["label"][:112]:112s} {r["tflops"]:8.1f} {r["ms_per_launch"]:8.3f} {f(r["power_w_mean"], "%6.0f"):>6s} {f(r["sclk_mhz_mean"], "%6.0f"):>6s} '
          f'{f(r["j_per_launch"], "%9.2f"):>9s} {f(r["pj_per_flop"], "%8.3f"):>8s} {f(r["mcycles_per_launch"], "%7.1f"):>7s} {f(r["ppt_residency"], "%5.2f"):>5s}')
