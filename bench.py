"""bench.py — denoise-steps/sec of the MoviiGen1.1 14B T2V hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 1080p|720p|1056p|tiny]

One "step" = what one iteration of the reference loop does (wan/text2video.py:233-254):
two WanModel forwards (cond / uncond), the CFG combine and one UniPC scheduler step, on
synthetic data of the configuration BASELINE.json's `metric` is quoted on: 14B T2V,
1920x832x81f (latent [16,21,104,240], L = 131 040 tokens) — it fits one MI355X, so N=1 runs
it unsharded (`--workload 720p` = BASELINE configs[1], 1280x720x81f, L = 75 600).
N>1 (one rank per GPU): the SAME video sharded over the ranks -> strong scaling.  The PRIMARY layout (`value`) is Ulysses sequence
parallelism over all N ranks on RCCL — the reference's layout and what BASELINE configs[2] names ("Ulysses SP=8"); for even N the same run
then measures the second layout — cond / uncond halves x Ulysses N/2 — with its own warm-up and K steps and reports it as `other_layout`
(`--cfg-parallel` swaps the two, `--single-layout` skips the second; `--dit-fsdp` = BASELINE configs[3]: cfg2 x ulysses_sp(N/2) x fsdpN only).
Before the warm-up a time-boxed PREFLIGHT (wan/distributed/preflight.py) records what the node can do: RCCL rank count, peer access per
rank pair, a 64 MiB all-to-all on the exchange's default transport (`link_gbps_measured`) and — only when the copy-engine transport is asked
for — IPC windows, self-check and the same exchange as peer copies.  Launched either by torch.distributed.run
(the driver's form; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or PLAINLY — `python bench.py
--gpus N` without WORLD_SIZE re-executes itself under torch.distributed.run on 127.0.0.1 (reference launch contract:
scripts/inference/generate.py:190-229, one process per GPU + init_process_group("nccl")).  The N > 1 line additionally
reports what RCCL saw: `rccl_ranks`, the device of every rank, the transport and the measured overlap (fraction of the
exchange time the compute stream did NOT wait for, from events on both streams).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (self-attention, 82 % of
the FLOPs at this size): algorithmic FLOPs per launch / mean launch duration, measured live with
events on the launch stream during the timed region.  After the timed region the same process
measures what is left of `sec/video` (reference wan/text2video.py:228-261): the WanVAE decode of a
latent of the workload's size and (reported separately, as SURVEY.md §8(d) prescribes) the two
umT5-XXL prompt encodes; `sec_per_video` = 50 x the measured step + the measured decode.
`cpu_baseline` times the ORACLE (oracle/dit.py, oracle/vae.py: the CPU restatements) on bounded
slices on this box's host cores and extrapolates by the FLOP formulas (rule stated in the line).
"""
import argparse
import json
import os
import sys
import time

# HIP maps streams onto 4 hardware queues by default; the compute stream, the exchange stream and RCCL's own stream then
# alias and the exchange of one head group serialises with the attention of the next (profiles/r02d_sp_overlap.txt:
# 0 % overlap with 4 queues, 38-44 % of the RCCL kernel time under attention with 8).  Must be set before HIP initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {  # name -> (W, H, frames, description)
    '720p': (1280, 720, 81, '14B T2V 1280x720x81f bf16 (BASELINE configs[1])'),
    '1080p': (1920, 832, 81, "14B T2V 1920x832x81f bf16 (the metric's configuration; BASELINE configs[2] shape)"),
    '1056p': (1920, 1056, 81, '14B T2V 1920x1056x81f bf16 (BASELINE configs[3] shape)'),
    'tiny': (128, 96, 9, 'plumbing check: 14B width, 2 layers, 128x96x9f'),
}
PEAK_BF16 = 2.5e15   # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_MFMA = 157.3e12   # exact-f32 MFMA peak (same table)
MODEL_14B = dict(dim=5120, ffn_dim=13824, freq_dim=256, num_heads=40, num_layers=40, text_len=512, text_dim=4096,
                 in_dim=16, out_dim=16, eps=1e-6)


def _telemetry(device_index, period_s=0.5):
    """tools/gpu_telemetry.py: an in-process thread reading libamd_smi (no child process, nothing on the GPU)"""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from gpu_telemetry import GpuTelemetry
    return GpuTelemetry(device_index=device_index, period_s=period_s)


def box_calibration(dev, device_index, secs_attn=6.0, secs_gemm=3.0):
    """What THIS box sustains on the two hot kernels alone, on fixed random operands, before anything else runs — so that a step time
    measured on one box can be compared with one measured on another (the same attention kernel ran 228-243 ms per launch on different
    boxes of one round): self-attention at L = 131 040 x 8 heads (1/5 of a launch of the workload) back to back for `secs_attn` seconds,
    the ffn.0 GEMM shape (131 040 x 13 824 x 5 120, bias + GELU) for `secs_gemm`, each with clock / power / temperature / firmware
    violation residencies sampled meanwhile."""
    from wan.backend import ops
    out = {}
    g = torch.Generator(device=dev).manual_seed(1234)
    L, heads = 131040, 8
    q = torch.randn(L, heads * 128, device=dev, generator=g).bfloat16()
    k = torch.randn(L, heads * 128, device=dev, generator=g).bfloat16()
    v = torch.randn(L, heads * 128, device=dev, generator=g).bfloat16()
    kp = torch.empty(ops.packed_kv_numel(L, heads), dtype=torch.bfloat16, device=dev)
    vp = torch.empty_like(kp)
    o = torch.empty_like(q)
    ops.pack_kv(k, v, heads, kp, vp)

    def loop(fn, secs, flops):
        fn()
        torch.cuda.synchronize()
        tel = _telemetry(device_index, 0.25).start()
        ms, n = 0.0, 0
        while ms < secs * 1e3:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4):
                fn()
            b.record()
            torch.cuda.synchronize()
            ms += a.elapsed_time(b)
            n += 4
        t = tel.stop()
        return {'tflops': flops * n / (ms * 1e-3) / 1e12, 'ms_per_launch': ms / n, 'launches': n,
                **{k_: t[k_] for k_ in ('sclk_mhz_mean', 'power_w_mean', 'temp_c_max', 'residency') if k_ in t}}
    out['attn'] = loop(lambda: ops.attention_hd128(q, kp, vp, o, L, heads, 1.0, prescaled=True), secs_attn, 4.0 * L * L * 128 * heads)
    del q, k, v, kp, vp, o
    N, K = 13824, 5120
    a_ = torch.randn(L, K, device=dev, generator=g).bfloat16()
    w_ = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    b_ = torch.randn(N, device=dev, generator=g)
    u_ = torch.empty(L, N, dtype=torch.bfloat16, device=dev)
    out['gemm_ffn0'] = loop(lambda: ops.gemm(a_, w_, b_, ops.BIAS_GELU_BF16, u_), secs_gemm, 2.0 * L * N * K)
    del a_, w_, b_, u_
    torch.cuda.empty_cache()
    out['how'] = ('before the warm-up, this process, random operands: mg_attn_fwd_bf16_hd128_prescaled at L = 131040 x 8 heads and mg_gemm_bf16 '
                  'at 131040 x 13824 x 5120 (bias + GELU) back to back, HIP events around groups of 4 launches; telemetry = tools/gpu_telemetry.py')
    return out


def flops_per_forward(L, cfg):
    """SURVEY.md §8(d) closed form (== FlopCounterMode on the reference)."""
    d, f, n = cfg['dim'], cfg['ffn_dim'], cfg['num_layers']
    return (n * (12 * L * d * d + 4 * 512 * d * d + 4 * L * d * f + 4 * L * L * d + 4 * L * 512 * d)
            + 2 * L * 64 * d * 2 + 2 * 512 * (4096 * d + d * d))


def _usable_cores():
    """host cores this process may really use: affinity, capped by the cgroup CPU quota and at 64
    (torch's CPU GEMMs stop scaling — and on an over-subscribed container collapse — beyond that)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def vae_decode_flops(T, h, w, up_taps=9):
    """(total, attention) FLOPs of WanVAE.decode for a latent [16,T,h,w] (reference vae.py:367-472 layout,
    SURVEY.md Appendix A): 1116.5 TF at (21,104,240), 639.2 TF at (21,90,160) — SURVEY §8(d).  up_taps = 4: what this
    engine EXECUTES for the three convs behind a nearest-2x upsample (four 2x2 phase convs instead of one 3x3)."""
    px, F0, F1, F2 = h * w, T, 1 + 2 * (T - 1), 1 + 4 * (T - 1)
    res = lambda cin, cout: 27 * cin * cout + 27 * cout * cout + (cin * cout if cin != cout else 0)  # noqa: E731
    attn = 2 * px * px * 384 * F0
    mac = (16 * 16 + 27 * 16 * 384 + 2 * res(384, 384) + 384 * 1152 + 384 * 384 + 3 * res(384, 384)) * px * F0 + attn
    mac += 3 * 384 * 768 * px * (F0 - 1) + (up_taps * 384 * 192 + res(192, 384) + 2 * res(384, 384)) * 4 * px * F1
    mac += 3 * 384 * 768 * 4 * px * (F1 - 1) + (up_taps * 384 * 192 + 3 * res(192, 192)) * 16 * px * F2
    mac += (up_taps * 192 * 96 + 3 * res(96, 96) + 27 * 96 * 3) * 64 * px * F2
    return 2 * mac, 2 * attn


def cpu_baseline(L_step, lat_shape):
    """The oracle (CPU restatement, fp32, all usable host cores) on the bounded slices SURVEY.md §8(d) names, each leg
    timed as the BEST of several runs (a single run swings +-50 % from box to box):
      * ONE 14B-width WanAttentionBlock (d=5120, 40 heads, ffn 13824, 512 text keys) at L = 4 096 (grid 4x32x32),
        and its self-attention alone on the same shapes, so the step is extrapolated in two parts:
        t_step = F_linear(step) / r_linear + F_attention(step) / r_attention  (FLOP formula of SURVEY §8(d));
      * the full 40-layer model at L = 1 024 (grid 4x16x16; the 40 blocks share one set of weights: same arithmetic,
        0.7 instead of 28 GB of host memory) — a whole-forward cross-check of the extrapolation rule;
      * BASELINE configs[0] in full (2-layer dim-128 DiT, latent [16,1,8,8], 2 UniPC steps with CFG);
      * the VAE decode of z[16,3,32,32] -> [3,9,256,256], extrapolated to the workload's latent by FLOPs."""
    import weights as W
    from oracle import dit, schedulers as osch, vae as ovae
    torch.set_num_threads(_usable_cores())
    d, f, N = 5120, 13824, 40
    cfg = dict(W.TINY_DIT, dim=d, ffn_dim=f, num_heads=N, num_layers=1, text_len=512)
    g = torch.Generator().manual_seed(0)
    shapes = {k: v for k, v in W.dit_param_shapes(cfg).items() if k.startswith('blocks.0.')}
    P = {k: (torch.randn(s, generator=g) * 0.02) for k, s in shapes.items()}
    L, grid = 4096, (4, 32, 32)
    x = torch.randn(L, d, generator=g)
    e0 = torch.randn(6, d, generator=g) * 0.1
    ctx = torch.randn(512, d, generator=g)
    tabs = dit.rope_table(128)

    def best(fn, reps, warm=True):
        if warm:
            fn()
        ts_ = []
        for _ in range(reps):
            t0 = time.time()
            fn()
            ts_.append(time.time() - t0)
        return min(ts_)
    t_blk = best(lambda: dit.block(P, 'blocks.0.', x, e0, L, grid, tabs, ctx, N, 1e-6, False, False), 2)
    q = torch.randn(L, N, 128, generator=g)
    t_att = best(lambda: dit.attention(q, q, q, L, False), 3)
    F_att = 4 * L * L * d
    F_lin = 12 * L * d * d + 4 * 512 * d * d + 4 * L * d * f + 4 * L * 512 * d
    r_att, r_lin = F_att / t_att, F_lin / max(t_blk - t_att, 1e-9)
    fl_fwd = flops_per_forward(L_step, MODEL_14B)
    F_att_step = 2 * MODEL_14B['num_layers'] * 4 * L_step * L_step * d
    F_lin_step = 2 * fl_fwd - F_att_step
    t_step = F_lin_step / r_lin + F_att_step / r_att

    # the full depth at L = 1024: 40 blocks over one block's weights
    class Shared(dict):
        def __missing__(self, key):
            if key.startswith('blocks.'):
                return self['blocks.0.' + key.split('.', 2)[2]]
            raise KeyError(key)
    cfg40 = dict(cfg, num_layers=40, text_dim=4096)
    P40 = Shared(P)
    for k, s_ in W.dit_param_shapes(dict(cfg40, num_layers=0)).items():
        if k not in P40:
            P40[k] = torch.randn(s_, generator=g) * 0.02
    lat40 = torch.randn(16, 4, 32, 32, generator=g)
    ctx40 = torch.randn(512, 4096, generator=g)
    t_l1024 = best(lambda: dit.dit_forward(P40, cfg40, lat40, torch.tensor([500]), ctx40, 1024), 1, warm=False)
    fl_l1024 = flops_per_forward(1024, MODEL_14B)
    pred_l1024 = (fl_l1024 - 40 * 4 * 1024 * 1024 * d) / r_lin + 40 * 4 * 1024 * 1024 * d / r_att

    # BASELINE configs[0], in full
    c0 = W.TINY_DIT
    P0 = W.make_dit_params(c0, 0)
    lat0, ctx0, ctxn0 = W.randn((16, 1, 8, 8), 1), W.randn((11, c0['text_dim']), 2), W.randn((5, c0['text_dim']), 3)
    t_cfg0 = best(lambda: osch.sample_loop(lambda lat, t, c: dit.dit_forward(P0, c0, lat, t, c, 16), lat0, ctx0, ctxn0, 2, 5.0,
                                           5.0, 'unipc'), 3)

    # VAE slice
    Pv = W.make_vae_params(96, 1)
    z = torch.randn(16, 3, 32, 32, generator=g)
    t_vae = best(lambda: ovae.vae_decode(Pv, z), 2, warm=False)
    fv_slice = vae_decode_flops(3, 32, 32)[0]
    fv_full = vae_decode_flops(*lat_shape[1:])[0]
    return {'value': 1.0 / t_step, 'unit': 'steps/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'block_L4096_s': round(t_blk, 3), 'attention_L4096_s': round(t_att, 3),
            'gflops_linear': round(r_lin / 1e9, 1), 'gflops_attention': round(r_att / 1e9, 1),
            'sec_per_step_extrapolated': round(t_step, 1),
            'model40_L1024_forward_s': round(t_l1024, 2), 'model40_L1024_predicted_by_rule_s': round(pred_l1024, 2),
            'model40_L1024_gflops': round(fl_l1024 / t_l1024 / 1e9, 1),
            'config0_2steps_s': round(t_cfg0, 4),
            'vae': {'slice_s': round(t_vae, 2), 'gflops': round(fv_slice / t_vae / 1e9, 1),
                    'decode_s_extrapolated': round(fv_full / (fv_slice / t_vae), 1),
                    'sample': 'oracle/vae.py vae_decode of z[16,3,32,32] -> [3,9,256,256] (dim-96 decoder), best of 2 runs, '
                              'extrapolated to the workload latent by the decoder FLOP count'},
            'sample': f'oracle/dit.py block (d=5120, 40 heads, ffn=13824, 512 text keys) at L=4096: {t_blk:.2f}s (best of 2 after a warm-up), '
                      f'its self-attention alone {t_att:.2f}s (best of 3); rule: t_step = F_linear/r_linear + F_attention/r_attention '
                      f'with the step FLOPs of SURVEY 8(d) ({2 * fl_fwd / 1e15:.2f} PFLOP at L={L_step}); the full 40-layer model at '
                      f'L=1024 (shared block weights): {t_l1024:.1f}s measured vs {pred_l1024:.1f}s by the rule; '
                      f'configs[0] (2-layer dim-128 DiT, 2 UniPC steps, CFG) in full: {t_cfg0:.3f}s (best of 3)'}


def measure_attention_traffic(L, heads, timeout=240):
    """HBM-side traffic of ONE launch of the dominant kernel, measured NOW on this box: two separate `rocprofv3 --pmc`
    passes (FETCH_SIZE, then WRITE_SIZE — they do not fit one pass; --kernel-trace only, as the guide's HBM section
    prescribes) over `mg_selftest attnpmc L heads` = two launches of mg_attn_fwd_bf16_hd128_prescaled at the workload's
    launch shape.  FETCH_SIZE is doubled (gfx950 reports half the bytes of a wide coalesced read stream,
    MI355X_MICROARCH.md), WRITE_SIZE taken as is; both are KiB.  -> (dict, None) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, 'moviigen1.1_amd', 'lib', 'mg_selftest')
    prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe) or not os.path.exists(prof):
        return None, 'mg_selftest or rocprofv3 not found'
    kib = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='mg_pmc_', dir='/tmp')
        try:
            r = subprocess.run([prof, '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p', '--', exe, 'attnpmc',
                                str(L), str(heads)], cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), capture_output=True, text=True,
                               timeout=timeout)
            vals = []
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for row in csv.DictReader(open(f)):
                    if 'attn_hd128_m16_kernel' in row['Kernel_Name'] and row.get('Counter_Name', ctr) == ctr:
                        vals.append(float(row['Counter_Value']))
            if r.returncode != 0 or not vals:
                return None, f'{ctr} pass failed (rc {r.returncode}, {len(vals)} dispatches): {(r.stdout + r.stderr)[-300:]}'
            kib[ctr] = (sum(vals) / len(vals), len(vals))
        except (subprocess.TimeoutExpired, OSError) as e:
            return None, f'{ctr} pass: {type(e).__name__}'
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {'traffic_bytes_per_launch': 2 * kib['FETCH_SIZE'][0] * 1024 + kib['WRITE_SIZE'][0] * 1024,
            'fetch_size_kib_raw': kib['FETCH_SIZE'][0], 'write_size_kib_raw': kib['WRITE_SIZE'][0],
            'dispatches': [kib['FETCH_SIZE'][1], kib['WRITE_SIZE'][1]]}, None


def launch_command(n, argv, port=None):
    """the torch.distributed.run command line `python bench.py --gpus n ...` re-executes itself under when it is
    started without WORLD_SIZE (one rank per GPU of this node, rendezvous on 127.0.0.1)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='1080p', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-video-tail', action='store_true',
                    help='skip the VAE decode / T5 legs measured after the timed region (profiling passes)')
    ap.add_argument('--cfg-parallel', action='store_true',
                    help='N > 1, even: make cond / uncond halves x Ulysses N/2 the PRIMARY layout (`value`); by default the primary is Ulysses '
                         "over all N ranks — the reference's layout and BASELINE configs[2] — and this one is measured second (`other_layout`)")
    ap.add_argument('--no-cfg-parallel', action='store_true',
                    help='N > 1: Ulysses over all N ranks as the primary layout (the default since round 6; kept for the round-5 command lines)')
    ap.add_argument('--single-layout', action='store_true', help='N > 1: measure the primary layout only (no `other_layout`)')
    ap.add_argument('--preflight', default='auto', choices=['auto', 'full'],
                    help="N > 1: the first-contact check before the warm-up (wan/distributed/preflight.py).  auto: ranks, peer access, a 64 MiB "
                         "all-to-all on the default transport; the IPC / peer-copy probe only when that transport is asked for.  full: always")
    ap.add_argument('--no-preflight', action='store_true')
    ap.add_argument('--preflight-budget', type=float, default=60.0, help='seconds the preflight may take before it skips its remaining stages')
    ap.add_argument('--dit-fsdp', action='store_true',
                    help="N > 1: DiT block weights sharded over ALL N ranks and all-gathered one block ahead (the reference's "
                         '--dit_fsdp, wan/text2video.py:107-108 -> shard_model); with the default layout on 8 GPUs this is BASELINE '
                         'configs[3]: cfg2 x ulysses_sp4 x fsdp8')
    ap.add_argument('--vae-parallel', nargs='?', const='spatial', default=None, choices=['spatial', 'pipeline'],
                    help='N > 1: the VAE decode of the sec/video tail over all ranks instead of rank 0 alone as in the reference '
                         '(text2video.py:260-261): spatial (the default of the flag) = every rank decodes its band of image columns '
                         '(WanVAE.decode_spatial: halo columns before each 3x3 conv, k|v all-gather in the attention block), pipeline = the layer '
                         'pipeline (WanVAE.decode_pipelined)')
    ap.add_argument('--transport', default=None, choices=['auto', 'torch', 'rccl_direct', 'peer_copy'],
                    help='N > 1: transport of the Ulysses exchange — torch.distributed nccl (default), the C-ABI collectives on the '
                         "library's own RCCL communicator, or one-sided peer copies on the copy engines")
    ap.add_argument('--no-shared-prefix-leg', action='store_true',
                    help='skip the short leg after the timed region that measures the same step through WanModel.forward_pair (`shared_prefix`)')
    ap.add_argument('--no-pmc', action='store_true',
                    help="N = 1: skip the two rocprofv3 --pmc passes that measure the dominant kernel's HBM traffic after the timed "
                         'region (roofline.traffic is then read from the newest committed summary and labelled so)')
    ap.add_argument('--emulate-rank', type=int, default=0, metavar='P',
                    help='no measurement of this box: ONE rank of a P-GPU run emulated on one GPU (tools/emulate_rank.py: the real engine on '
                         'loop-back process groups, per-rank shapes, real message sizes); prints lines marked `invalid: emulation`')
    ap.add_argument('--no-calibration', action='store_true',
                    help='skip the ~10 s box calibration (attention / GEMM alone on random operands before the warm-up: box_attn_tflops)')
    ap.add_argument('--layers', type=int, default=None, help='debug only: fewer layers (marks the line invalid)')
    ap.add_argument('--gemm-variant', type=int, default=0,
                    help='A/B only: force a tile schedule of mg_gemm_bf16 (same bits; 0 = the library default by shape, 8 = the round-3 '
                         'default); recorded in config.gemm_variant')
    args = ap.parse_args()
    if args.emulate_rank:
        cmd = [sys.executable, os.path.join(ROOT, 'tools', 'emulate_rank.py'), '--workload', args.workload if args.workload in ('720p', '1080p', '1056p') else '1080p',
               '--ranks', str(args.emulate_rank), '--steps', str(args.steps), '--fsdp-at'] + ([str(args.emulate_rank)] if args.dit_fsdp else [])
        os.execv(sys.executable, cmd)
    if args.transport is not None:        # read by wan.distributed at exchange-construction time; inherited by self-launched ranks
        os.environ['MOVIIGEN_SP_TRANSPORT'] = args.transport

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (the driver's torchrun form skips this)
        import subprocess
        cmd = launch_command(args.gpus, sys.argv[1:])
        if os.environ.get('MOVIIGEN_BENCH_DRYRUN'):
            print(json.dumps({'launch': cmd}), flush=True)
            return
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        sys.exit(subprocess.call(cmd, env=env))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and os.environ.get('MOVIIGEN_BENCH_BACKEND') == 'gloo':
        # test plumbing only (tests/test_gpu_parity.py::test_bench_multirank_code_path): N ranks share cuda:0 and
        # talk through gloo, so the N > 1 branches of this file run on a 1-GPU box.  Never a measurement.
        local = 0
        dist.init_process_group('gloo')
    elif world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if torch.cuda.device_count() < world:
            raise SystemExit(f'bench.py --gpus {world}: one rank per GPU over RCCL needs {world} visible GPUs, this node shows '
                             f'{torch.cuda.device_count()} (MOVIIGEN_BENCH_BACKEND=gloo runs the code path on one GPU: not a measurement)')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local}'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)'
    dev = torch.device(f'cuda:{local}')

    import wan
    from wan.backend import ops
    from wan.utils import FlowUniPCMultistepScheduler

    cfg = dict(MODEL_14B)
    if args.workload == 'tiny':
        cfg['num_layers'] = 2
    if args.layers:
        cfg['num_layers'] = args.layers
    Wd, Hd, frames, desc = WORKLOADS[args.workload]
    lat_shape = (16, (frames - 1) // 4 + 1, Hd // 8, Wd // 8)
    L = lat_shape[1] * (lat_shape[2] // 2) * (lat_shape[3] // 2)

    if args.gemm_variant:
        from wan.backend import lib as _lib
        _lib.ab_library().__enter__().mg_gemm_set_variant(args.gemm_variant)      # measurement only: the whole process runs on the A/B library
    calib = None
    if world == 1 and not args.no_calibration and args.workload != 'tiny' and not args.layers:
        calib = box_calibration(dev, local)
    # ---- N > 1: what this node / process group can do, measured before anything else (wan/distributed/preflight.py) -------------------
    pre = None
    if world > 1 and not args.no_preflight:
        from wan.distributed import preflight
        want_peer = (args.transport or os.environ.get('MOVIIGEN_SP_TRANSPORT') or 'torch') in ('auto', 'peer_copy') or args.preflight == 'full'
        try:
            pre = preflight.run(None, dev, probe_peer_copy=want_peer, budget_s=args.preflight_budget)
        except Exception as e:      # noqa: BLE001 — the preflight must never be the reason a bench line is missing
            pre = {'errors': [f'preflight: {type(e).__name__}: {e}'], 'rccl_ranks': 0 if os.environ.get('MOVIIGEN_BENCH_BACKEND') == 'gloo' else world}
        if rank == 0:
            print('preflight: ' + json.dumps(pre), file=sys.stderr, flush=True)

    # ---- layouts of the N > 1 forward ---------------------------------------------------------------------------------------------------
    #   'ulysses': Ulysses sequence parallelism over all N ranks — the reference's layout (scripts/inference/generate.py:216-229) and what
    #              BASELINE configs[2] names ("Ulysses SP=8"): the PRIMARY line, `value`
    #   'cfg'    : cond / uncond halves x Ulysses N/2 (wan/distributed/cfg_parallel.py: same math, half the exchange partners) — measured in
    #              the same run and reported as `other_layout` (even N); primary only with --cfg-parallel, or with --dit-fsdp (BASELINE
    #              configs[3]: "FSDP shard + SP=4 on 8 GPUs" = cfg2 x ulysses_sp4 x fsdp8)
    even = world > 1 and world % 2 == 0
    primary = 'single' if world == 1 else 'cfg' if even and (args.cfg_parallel or args.dit_fsdp) and not args.no_cfg_parallel else 'ulysses'
    secondary = None
    if even and not args.single_layout and not args.dit_fsdp:
        secondary = 'ulysses' if primary == 'cfg' else 'cfg'

    g = torch.Generator(device=dev).manual_seed(42)
    latent0 = torch.randn(*lat_shape, dtype=torch.float32, device=dev, generator=g)
    ctx = torch.randn(512, 4096, device=dev, generator=g).bfloat16()
    ctx_null = torch.randn(130, 4096, device=dev, generator=g).bfloat16()
    total = args.warmup + args.steps

    # live timing of the dominant kernel (self-attention) on the launch stream
    attn_events = []
    orig_attn = ops.attention_hd128
    recording = {'on': False}

    def timed_attn(q, k, vt, out, lk, heads, scale, **kw):
        if recording['on'] and lk > 1024:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = orig_attn(q, k, vt, out, lk, heads, scale, **kw)
            b.record()
            attn_events.append((a, b, heads))
            return r
        return orig_attn(q, k, vt, out, lk, heads, scale, **kw)
    ops.attention_hd128 = timed_attn

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    from wan.distributed.fsdp import BlockShards
    from wan.distributed.ulysses import HeadExchange

    def measure(layout, is_primary):
        """W warm-up steps + K timed steps of the loop body of wan/text2video.py:233-254 under one layout, its own model (same seed, same
        weights), scheduler and latent.  -> dict"""
        model = wan.modules.WanModel(**cfg, device=dev)
        model.init_weights(seed=0)
        model.eval().requires_grad_(False)
        cfgp = None
        if world > 1:
            if layout == 'cfg':
                from wan.distributed.cfg_parallel import enable_cfg_parallel
                cfgp = enable_cfg_parallel(model)
            else:
                from wan.distributed.xdit_context_parallel import enable_sequence_parallel
                enable_sequence_parallel(model)
            if args.dit_fsdp:
                from wan.distributed.fsdp import shard_model
                shard_model(model, device_id=local)       # 1/N of every block's GEMM weights per rank, gathered one block ahead
        sch = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        sch.set_timesteps(50, device=dev, shift=5.0)
        ts = sch.timesteps
        ts_host = ts.tolist()
        st = {'latent': latent0.clone()}
        noise_pred = torch.empty_like(latent0)

        def step(i):
            latent = st['latent']
            t = ts[i:i + 1]
            if cfgp is None:
                cond = model([latent], t=t, context=[ctx], seq_len=L)[0]
                uncond = model([latent], t=t, context=[ctx_null], seq_len=L)[0]
            else:
                mine = model([latent], t=t, context=[ctx_null if cfgp.branch else ctx], seq_len=L)[0]
                cond, uncond = cfgp.exchange(mine)
            ops.cfg_combine(noise_pred, uncond, cond, 5.0)
            st['latent'] = sch.step(noise_pred.unsqueeze(0), ts_host[i], latent.unsqueeze(0), return_dict=False)[0].squeeze(0)

        for i in range(args.warmup):
            step(i)
        fence()
        if world > 1:
            HeadExchange.trace = []          # events around every collective / every wait of the compute stream on one
            if args.dit_fsdp:
                BlockShards.trace = []
        recording['on'] = is_primary
        tel = _telemetry(local).start() if rank == 0 else None        # a thread reading sysfs through libamd_smi twice a second: nothing in the GPU's way
        t0 = time.perf_counter()
        for i in range(args.warmup, total):
            step(i)
        fence()
        elapsed = time.perf_counter() - t0
        R = {'layout': layout, 'model': model, 'cfgp': cfgp, 'sp': model.sp_size, 'telemetry': tel.stop() if tel is not None else None,
             'overlap': None, 'fsdp_trace': None, 'sp_groups': None, 'peer_used': False}
        recording['on'] = False
        if world > 1:
            tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = tt.item()
            R['overlap'] = HeadExchange.overlap_summary()
            HeadExchange.trace = None
            R['peer_used'] = any('xchg' in w_ and w_['xchg'].peer is not None for w_ in model._ws.values())
            xchgs = [w_['xchg'] for w_ in model._ws.values() if 'xchg' in w_]
            R['sp_groups'] = {'heads_per_group': [n for _, n in xchgs[0].groups], 'attention_rounds_per_layer': xchgs[0].rounds,
                              'chosen_by': os.environ.get('MOVIIGEN_SP_GROUPS', '') or 'auto'} if xchgs else None
            if args.dit_fsdp:
                from wan.distributed.collectives import trace_summary
                R['fsdp_trace'] = trace_summary(BlockShards.trace)
                BlockShards.trace = None
        R['elapsed'] = elapsed
        R['latent'] = st['latent']
        # ---- the same step through WanModel.forward_pair (what WanT2V.generate calls): everything in front of block 0's cross-attention is
        # computed once for the two guidance branches.  Its own short leg AFTER the timed region — `value` stays the two plain forwards.
        R['shared_prefix'] = None
        k2 = min(args.steps, 2, 50 - total)
        if cfgp is None and is_primary and k2 > 0 and not args.no_shared_prefix_leg:
            def step_pair(i):
                latent = st['latent']
                t = ts[i:i + 1]
                cond, uncond = model.forward_pair([latent], t, [ctx], [ctx_null], L)
                ops.cfg_combine(noise_pred, uncond[0], cond[0], 5.0)
                st['latent'] = sch.step(noise_pred.unsqueeze(0), ts_host[i], latent.unsqueeze(0), return_dict=False)[0].squeeze(0)
            fence()
            t1 = time.perf_counter()
            for i in range(total, total + k2):
                step_pair(i)
            fence()
            e2 = time.perf_counter() - t1
            t = ts[total + k2 - 1:total + k2]
            pc, pu = model.forward_pair([st['latent']], t, [ctx], [ctx_null], L)
            same = torch.equal(pu[0], model([st['latent']], t=t, context=[ctx_null], seq_len=L)[0])
            if world > 1:
                tt = torch.tensor([e2, 0.0 if same else 1.0], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                e2, same = tt[0].item(), tt[1].item() == 0.0
            R['shared_prefix'] = {'value': k2 / e2, 'ms_per_step': e2 / k2 * 1e3, 'steps': k2, 'second_branch_equals_plain_forward': same,
                                  'what': 'the same step through WanModel.forward_pair (the call WanT2V.generate makes): patch / time embedding and block 0 up to '
                                          'its self-attention residual read the latent and t only, so the second guidance branch starts from a copy of the '
                                          'first one\'s residual stream there — one self-attention launch and four GEMMs of a step\'s 80 are not repeated; '
                                          'bit-identical outputs (checked here on the last step, and by the GPU tests); NOT what `value` reports'}
            del pc, pu
        assert torch.isfinite(R['latent']).all().item(), 'non-finite latent'
        R['parallelism'] = ('single' if world == 1 else f'cfg2 x ulysses_sp{R["sp"]}' if cfgp is not None else f'ulysses_sp{R["sp"]}') + \
                           (f' x fsdp{world}' if args.dit_fsdp and world > 1 else '')
        return R

    R = measure(primary, True)
    other = None
    if secondary is not None:
        # the second layout of the same video on the same ranks, its own warm-up and K timed steps; the primary's model is released first
        lat_primary = R['latent']
        R['model']._ws = {}
        R['model'] = R['cfgp'] = None
        import gc
        gc.collect()                          # (module graphs hold reference cycles: without this the first model's 28.6 GB stay allocated)
        torch.cuda.empty_cache()
        R2 = measure(secondary, False)
        per = 1.0 / args.steps
        other = {'parallelism': R2['parallelism'], 'value': args.steps / R2['elapsed'], 'ms_per_step': R2['elapsed'] / args.steps * 1e3,
                 'latent_max_abs_diff_vs_primary': (R2['latent'] - lat_primary).abs().max().item(),
                 'overlap': ({'groups': R2['sp_groups'], 'exchange_ms_per_step': R2['overlap']['exchange_ms'] * per,
                              'exposed_ms_per_step': R2['overlap']['exposed_ms'] * per, 'hidden_frac': R2['overlap']['hidden_frac']}
                             if R2['overlap'] and R2['overlap']['collectives'] else None),
                 'note': 'same run, same ranks, same video, same K / W: the other layout of the N > 1 forward (both compute the same step; '
                         'latent_max_abs_diff_vs_primary is what the different summation order of the all-gathered rows leaves: 0)'}
        model, cfgp = R2['model'], R2['cfgp']
    else:
        model, cfgp = R['model'], R['cfgp']
    sp, elapsed, telemetry, latent = R['sp'], R['elapsed'], R['telemetry'], R['latent']
    overlap, fsdp_trace, sp_groups, peer_used = R['overlap'], R['fsdp_trace'], R['sp_groups'], R['peer_used']
    rank_devices = None
    if world > 1:
        mine = {'rank': rank, 'device': f'cuda:{local}', 'name': torch.cuda.get_device_name(dev),
                'uuid': str(getattr(torch.cuda.get_device_properties(dev), 'uuid', ''))}
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)

    # ---- the rest of sec/video (reference wan/text2video.py:228-261), measured in this same process after the timed
    # region: WanVAE.decode of a latent of this size on rank 0 (the reference decodes on rank 0 only) and, reported
    # separately, the two umT5-XXL prompt encodes.  Random-init weights of the shipped architectures.
    vae_s = t5_s = None
    vae_pipe = args.vae_parallel if world > 1 else None       # None | 'spatial' | 'pipeline'
    if (rank == 0 or vae_pipe) and not args.no_video_tail:
        import weights as Wt
        z = latent.clone()
        model._ws = {}                       # the DiT activations are not needed any more
        torch.cuda.empty_cache()
        vae = wan.modules.WanVAE(state_dict=Wt.make_vae_params(96, 1), device=dev)
        vae.decode([z[:, :2, :16, :16].contiguous()])        # warm-up launch of every kernel (tiny latent)
        if vae_pipe:
            # --vae-parallel: the decode over all ranks (every rank passes the same latent: the scheduler state is replicated);
            # timed between two barriers, video on rank 0
            multi = vae.decode_spatial if vae_pipe == 'spatial' else vae.decode_pipelined
            multi([z[:, :2, :16, :16].contiguous()])
            fence()
            t1 = time.perf_counter()
            video = multi([z])[0]
            fence()
        else:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            video = vae.decode([z])[0]
            torch.cuda.synchronize()
        vae_s = time.perf_counter() - t1
        if rank == 0:
            assert video.shape == (3, frames, Hd, Wd) and torch.isfinite(video).all().item()
        del video, vae
        torch.cuda.empty_cache()
    if rank == 0 and not args.no_video_tail:
        if not args.layers and args.workload != 'tiny':
            from wan.modules.t5 import umt5_xxl
            enc = umt5_xxl(device=dev)
            gt = torch.Generator(device=dev).manual_seed(0)
            for name, p in enc.named_parameters():
                if 'norm' in name:
                    p.data.fill_(1.0)
                else:
                    p.data.copy_(torch.randn(p.shape, generator=gt, device=dev, dtype=torch.float32)
                                 .mul_(p.shape[-1] ** -0.5 * 0.5))
            ids = torch.randint(1, 256384, (1, 512), generator=torch.Generator().manual_seed(1)).to(dev)
            masks = []
            for n in (512, 130):                             # prompt and negative prompt (SURVEY 8(d) lengths)
                mk = torch.zeros(1, 512, dtype=torch.long, device=dev)
                mk[:, :n] = 1
                masks.append(mk)
            for mk in masks:
                enc(ids, mk)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for mk in masks:
                enc(ids, mk)
            torch.cuda.synchronize()
            t5_s = time.perf_counter() - t1
            del enc
            torch.cuda.empty_cache()

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        fl_fwd = flops_per_forward(L, cfg)
        attn_ms = sum(a.elapsed_time(b) for a, b, _ in attn_events) / max(1, len(attn_events))
        heads_launch = attn_events[0][2] if attn_events else cfg['num_heads'] // sp      # heads of ONE timed launch (a head group when sharded)
        attn_flops = 4.0 * L * L * 128 * heads_launch
        ach = attn_flops / (attn_ms * 1e-3) / 1e12 if attn_events else None
        # the cross-attention K/V projections and the text embedding are per-prompt work cached outside the timed
        # region (WanModel._context): they are NOT counted in the executed-FLOP rate
        fl_cached = cfg['num_layers'] * 4 * 512 * cfg['dim'] ** 2 + 2 * 512 * (4096 * cfg['dim'] + cfg['dim'] ** 2)
        fl_step = 2 * (fl_fwd - fl_cached)
        line = {
            'metric': 'denoise-steps/sec', 'value': args.steps / elapsed, 'unit': 'steps/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_step, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': desc, 'latent': list(lat_shape), 'tokens': L, 'layers': cfg['num_layers'],
                       'parallelism': R['parallelism'], 'solver': 'unipc',
                       'guide_scale': 5.0, 'weights': 'random N(0,0.02) bf16, seed 0',
                       **({'gemm_variant': args.gemm_variant} if args.gemm_variant else {})},
            'sec_per_video': (ms_step * 50 / 1e3 + vae_s) if vae_s is not None else None,
            'shared_prefix': R['shared_prefix'],
            'sec_per_video_shared_prefix': (R['shared_prefix']['ms_per_step'] * 50 / 1e3 + vae_s) if vae_s is not None and R['shared_prefix'] else None,
            'sec_per_video_parts': {'denoise_50_steps_s': ms_step * 50 / 1e3, 'vae_decode_s': vae_s,
                                    't5_encode_2_prompts_s_not_included': t5_s,
                                    'note': '50 x the measured step + the measured WanVAE.decode of this latent size, same '
                                            'process; T5 reported separately (SURVEY 8(d))'},
            'model_tflops_per_gpu': fl_step / (elapsed / args.steps) / world / 1e12,
            'mfma_frac_whole_step': fl_step / (elapsed / args.steps) / world / PEAK_BF16,
            'roofline': {'kernel': 'attn_hd128_m16_kernel (self-attention, mg_attn_fwd_bf16_hd128_prescaled)', 'bound': 'mfma',
                         'achieved': ach, 'peak': PEAK_BF16 / 1e12, 'unit': 'TFLOP/s',
                         'frac': (ach * 1e12 / PEAK_BF16) if ach else None, 'traffic': None,
                         'launches_timed': len(attn_events), 'ms_per_launch': attn_ms,
                         'algorithmic_flops_per_launch': attn_flops},
        }
        # clock / power / temperature of the timed region and the firmware's own account of what held the clock (PPT = package power
        # tracking, thermal, VR, HBM, PROCHOT residencies: fraction of the interval each limiter was active), + what this box sustains on
        # the two hot kernels alone: without them a slow box and a slow build are indistinguishable
        line['telemetry'] = telemetry
        if calib is not None:
            line['box_attn_tflops'] = calib['attn']['tflops']
            line['box_gemm_tflops'] = calib['gemm_ffn0']['tflops']
            line['box_calibration'] = calib
        if vae_s is not None:
            fv = vae_decode_flops(*lat_shape[1:])[0]                    # the reference's arithmetic
            fx = vae_decode_flops(*lat_shape[1:], up_taps=4)[0]         # what the MFMAs execute (phase-decomposed up-convs)
            line['vae_decode'] = {'seconds': vae_s, 'latent': list(lat_shape), 'tflops_fp32': fx / vae_s / 1e12,
                                  'fp32_mfma_peak_tflops': PEAK_F32_MFMA / 1e12, 'frac': fx / vae_s / PEAK_F32_MFMA,
                                  'algorithmic_tflop': fv / 1e12, 'executed_tflop': fx / 1e12}
        # HBM-side traffic of the dominant kernel: measured HERE, after the timed region, by two rocprofv3 --pmc passes
        # over the same launch shape (measure_attention_traffic); only when that is impossible (--no-pmc, no rocprofv3)
        # the newest committed summary for this workload is reported instead, and labelled as such — never a guess
        live = None
        if world == 1 and not args.layers and not args.no_pmc and args.workload != 'tiny':
            live, why = measure_attention_traffic(L, cfg['num_heads'])
            if live is not None:
                line['roofline']['traffic'] = live['traffic_bytes_per_launch']
                line['roofline']['traffic_unit'] = ('bytes/launch, measured in this run: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + --pmc WRITE_SIZE, '
                                                    f'two passes over mg_selftest attnpmc {L} {cfg["num_heads"]} ({live["dispatches"]} dispatches)')
                line['roofline']['algorithmic_bytes_per_launch'] = 4 * L * cfg['dim'] * 2
            else:
                line['roofline']['traffic_live_failed'] = why
        if world == 1 and not args.layers and live is None:
            import glob
            found = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'*pmc_traffic_{args.workload}.json')))
            if not found and args.workload == '720p':
                found = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*pmc_traffic.json')))
            if found:
                with open(found[-1]) as f:
                    pmc = json.load(f)
                line['roofline']['traffic'] = pmc.get('traffic_bytes_per_launch')
                line['roofline']['traffic_unit'] = 'bytes/launch (FETCH_SIZE x2 + WRITE_SIZE, ' + os.path.basename(found[-1]) + ')'
        if world > 1:
            from wan.distributed import rccl_direct
            gloo = os.environ.get('MOVIIGEN_BENCH_BACKEND') == 'gloo'
            line['rccl_ranks'] = 0 if gloo else world
            if other is not None:
                line['other_layout'] = other
            if pre is not None:
                from wan.distributed import preflight
                line['preflight'] = {**preflight.parse(pre), 'peer_access': pre.get('peer_access'), 'stages_skipped': pre.get('stages_skipped'),
                                     'how': 'wan/distributed/preflight.py before the warm-up: backend / ranks, hipDeviceCanAccessPeer per peer, a '
                                            f'{pre.get("probe_bytes", 0) >> 20} MiB all-to-all on the exchange\'s default transport (GB/s a rank sends to the others, slowest '
                                            'rank), and — only when the copy-engine transport is asked for — IPC windows + self-check + the same exchange as peer copies'}
                line['link_gbps_measured'] = line['preflight']['link_gbps_measured']
            line['rank_devices'] = rank_devices
            # what the exchange objects REALLY use (a peer-copy request falls back to the collective when the IPC mapping fails)
            line['transport'] = {
                'requested': args.transport or os.environ.get('MOVIIGEN_SP_TRANSPORT') or 'torch',
                'used': ('gloo through host memory (test plumbing)' if gloo else
                         'one-sided peer copies (hipMemcpyAsync D2D into IPC-mapped receive buffers) between two flag all-reduces' if peer_used else
                         "C-ABI collectives on the library's RCCL communicator (mg_sp_all_to_all: grouped ncclSend/ncclRecv)"
                         if rccl_direct.enabled() else 'torch.distributed backend nccl (= RCCL): all_to_all_single on a comm stream')}
            if fsdp_trace is not None:
                per = 1.0 / args.steps
                shard_b = sum(sh.numel() for sh in model._shards.shards) * 2
                line['fsdp'] = {'ranks': world, 'gather_ms_per_step': fsdp_trace['comm_ms'] * per, 'exposed_ms_per_step': fsdp_trace['exposed_ms'] * per,
                                'hidden_frac': fsdp_trace['hidden_frac'], 'gathers_per_step': fsdp_trace['collectives'] * per,
                                'resident_weight_bytes_per_rank': shard_b, 'gathered_bytes_per_block': shard_b * world // cfg['num_layers'],
                                'how': 'rank 0: timing events around every block all-gather on its comm stream (gather) and around every '
                                       'wait of the compute stream for a gathered block (exposed); hidden = 1 - exposed / gather'}
            if vae_s is not None:
                line['vae_decode_layout'] = (f'W bands over {world} ranks (WanVAE.decode_spatial)' if vae_pipe == 'spatial' else
                                             f'layer pipeline over {world} ranks (WanVAE.decode_pipelined)' if vae_pipe else
                                             'rank 0 alone (reference text2video.py:260-261)')
            if overlap and overlap['collectives']:
                per = 1.0 / args.steps
                line['overlap'] = {'groups': sp_groups, 'exchange_ms_per_step': overlap['exchange_ms'] * per, 'exposed_ms_per_step': overlap['exposed_ms'] * per,
                                   'hidden_frac': overlap['hidden_frac'], 'collectives_per_step': overlap['collectives'] * per,
                                   'how': 'rank 0: timing events around every all-to-all on the comm stream (exchange) and around every '
                                          'wait of the compute stream on the comm stream (exposed); hidden = 1 - exposed / exchange'}
            else:
                line['overlap'] = None      # no per-layer exchange in this layout (e.g. cfg2 on 2 ranks: one forward per rank)
        if os.environ.get('MOVIIGEN_BENCH_BACKEND') == 'gloo':
            line['invalid'] = 'gloo test transport on a shared GPU (code-path check, not a measurement)'
        if args.layers or args.workload == 'tiny':
            line['invalid'] = 'debug configuration (not the BASELINE model)'
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(L, lat_shape)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
