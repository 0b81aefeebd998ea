"""bench.py — denoise-steps/sec of the MoviiGen1.1 14B T2V hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 720p|1080p|1056p|tiny]

One "step" = what one iteration of the reference loop does (wan/text2video.py:233-254):
two WanModel forwards (cond / uncond), the CFG combine and one UniPC scheduler step, on
synthetic data of BASELINE.json configs[1] (N=1: 14B, 1280x720x81f, L = 75 600 tokens).
N>1 (launched by torch.distributed.run, one rank per GPU): the SAME video with Ulysses
sequence parallelism over RCCL -> strong scaling.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (self-attention, 72 % of
the FLOPs): algorithmic FLOPs per launch / mean launch duration, measured live with events on the
launch stream during the timed region.  `cpu_baseline` times the ORACLE (oracle/dit.py, the CPU
restatement) on a bounded slice on this box's host cores and extrapolates by the FLOP formula.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {  # name -> (W, H, frames, description)
    '720p': (1280, 720, 81, '14B T2V 1280x720x81f bf16 (BASELINE configs[1])'),
    '1080p': (1920, 832, 81, '14B T2V 1920x832x81f bf16 (BASELINE configs[2] shape)'),
    '1056p': (1920, 1056, 81, '14B T2V 1920x1056x81f bf16 (BASELINE configs[3] shape)'),
    'tiny': (128, 96, 9, 'plumbing check: 14B width, 2 layers, 128x96x9f'),
}
PEAK_BF16 = 2.5e15   # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
MODEL_14B = dict(dim=5120, ffn_dim=13824, freq_dim=256, num_heads=40, num_layers=40, text_len=512, text_dim=4096,
                 in_dim=16, out_dim=16, eps=1e-6)


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


def cpu_baseline(budget_s=12.0):
    """oracle (CPU restatement, fp32) on a bounded slice: ONE 14B-width block at L=512
    (grid 2x16x16), all host cores; extrapolated to steps/s of the 720p workload by FLOPs."""
    import weights as W
    from oracle import dit
    torch.set_num_threads(_usable_cores())
    cfg = dict(W.TINY_DIT, dim=5120, ffn_dim=13824, num_heads=40, num_layers=1, text_len=512)
    g = torch.Generator().manual_seed(0)
    shapes = {k: v for k, v in W.dit_param_shapes(cfg).items() if k.startswith('blocks.0.')}
    P = {k: (torch.randn(s, generator=g) * 0.02) for k, s in shapes.items()}
    L = 512
    x = torch.randn(L, 5120, generator=g)
    e0 = torch.randn(6, 5120, generator=g) * 0.1
    ctx = torch.randn(512, 5120, generator=g)
    tabs = dit.rope_table(128)
    run = lambda: dit.block(P, 'blocks.0.', x, e0, L, (2, 16, 16), tabs, ctx, 40, 1e-6, False, False)  # noqa: E731
    run()
    t0, n = time.time(), 0
    while n < 1 or (time.time() - t0 < budget_s and n < 20):
        run()
        n += 1
    dt = (time.time() - t0) / n
    d, f = 5120, 13824
    blk = 12 * L * d * d + 4 * 512 * d * d + 4 * L * d * f + 4 * L * L * d + 4 * L * 512 * d
    gflops = blk / dt / 1e9
    step_flops = 2 * flops_per_forward(75600, MODEL_14B)
    return {'value': gflops * 1e9 / step_flops, 'unit': 'steps/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'gflops': round(gflops, 1),
            'sample': f'oracle/dit.py block (d=5120, 40 heads, ffn=13824) at L=512, {n} runs of {dt:.2f}s, '
                      f'extrapolated to the 720p step (13.05 PFLOP) by the FLOP formula'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='720p', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cfg-parallel', action='store_true',
                    help='N > 1: Ulysses over all N ranks (the reference layout) instead of cond/uncond halves x Ulysses N/2')
    ap.add_argument('--layers', type=int, default=None, help='debug only: fewer layers (marks the line invalid)')
    args = ap.parse_args()

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

    model = wan.modules.WanModel(**cfg, device=dev)
    model.init_weights(seed=0)
    model.eval().requires_grad_(False)
    cfgp = None
    if world > 1:
        # even N: cond / uncond halves, Ulysses inside each half (same math, less traffic: see
        # wan/distributed/cfg_parallel.py); odd N or --no-cfg-parallel: Ulysses over all ranks
        if world % 2 == 0 and not args.no_cfg_parallel:
            from wan.distributed.cfg_parallel import enable_cfg_parallel
            cfgp = enable_cfg_parallel(model)
        else:
            from wan.distributed.xdit_context_parallel import enable_sequence_parallel
            enable_sequence_parallel(model)
    sp = model.sp_size
    g = torch.Generator(device=dev).manual_seed(42)
    latent = torch.randn(*lat_shape, dtype=torch.float32, device=dev, generator=g)
    ctx = torch.randn(512, 4096, device=dev, generator=g).bfloat16()
    ctx_null = torch.randn(130, 4096, device=dev, generator=g).bfloat16()
    total = args.warmup + args.steps
    sch = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    sch.set_timesteps(50, device=dev, shift=5.0)
    ts = sch.timesteps
    ts_host = ts.tolist()

    # live timing of the dominant kernel (self-attention) on the launch stream
    attn_events = []
    orig_attn = ops.attention_hd128
    recording = {'on': False}

    def timed_attn(q, k, vt, out, lk, heads, scale):
        if recording['on'] and lk > 1024:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = orig_attn(q, k, vt, out, lk, heads, scale)
            b.record()
            attn_events.append((a, b))
            return r
        return orig_attn(q, k, vt, out, lk, heads, scale)
    ops.attention_hd128 = timed_attn

    noise_pred = torch.empty_like(latent)

    def step(i):
        nonlocal latent
        t = ts[i:i + 1]
        if cfgp is None:
            cond = model([latent], t=t, context=[ctx], seq_len=L)[0]
            uncond = model([latent], t=t, context=[ctx_null], seq_len=L)[0]
        else:
            mine = model([latent], t=t, context=[ctx_null if cfgp.branch else ctx], seq_len=L)[0]
            cond, uncond = cfgp.exchange(mine)
        ops.cfg_combine(noise_pred, uncond, cond, 5.0)
        latent = sch.step(noise_pred.unsqueeze(0), ts_host[i], latent.unsqueeze(0), return_dict=False)[0].squeeze(0)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    recording['on'] = True
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    recording['on'] = False
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    assert torch.isfinite(latent).all().item(), 'non-finite latent'

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        fl_fwd = flops_per_forward(L, cfg)
        attn_ms = sum(a.elapsed_time(b) for a, b in attn_events) / max(1, len(attn_events))
        heads_loc = cfg['num_heads'] // sp
        attn_flops = 4.0 * L * L * 128 * heads_loc
        ach = attn_flops / (attn_ms * 1e-3) / 1e12 if attn_events else None
        line = {
            'metric': 'denoise-steps/sec', 'value': args.steps / elapsed, 'unit': 'steps/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_step, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': desc, 'latent': list(lat_shape), 'tokens': L, 'layers': cfg['num_layers'],
                       'parallelism': ('single' if world == 1 else f'cfg2 x ulysses_sp{sp}' if cfgp is not None else f'ulysses_sp{sp}'), 'solver': 'unipc',
                       'guide_scale': 5.0, 'weights': 'random N(0,0.02) bf16, seed 0'},
            'sec_per_video_50steps_dit_only': ms_step * 50 / 1e3,
            'model_tflops_per_gpu': 2 * fl_fwd / (elapsed / args.steps) / world / 1e12,
            'mfma_frac_whole_step': 2 * fl_fwd / (elapsed / args.steps) / world / PEAK_BF16,
            'roofline': {'kernel': 'attn_hd128_w64_kernel (self-attention, mg_attn_fwd_bf16_hd128)', 'bound': 'mfma',
                         'achieved': ach, 'peak': PEAK_BF16 / 1e12, 'unit': 'TFLOP/s',
                         'frac': (ach * 1e12 / PEAK_BF16) if ach else None, 'traffic': None,
                         'launches_timed': len(attn_events), 'ms_per_launch': attn_ms,
                         'algorithmic_flops_per_launch': attn_flops},
        }
        # HBM-side traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this same
        # command (tools/round_end_gpu.sh); the newest committed summary is reported, never a guess
        if args.workload == '720p' and world == 1 and not args.layers:
            import glob
            found = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*pmc_traffic.json')))
            if found:
                with open(found[-1]) as f:
                    pmc = json.load(f)
                line['roofline']['traffic'] = pmc.get('traffic_bytes_per_launch')
                line['roofline']['traffic_unit'] = 'bytes/launch (FETCH_SIZE x2 + WRITE_SIZE, ' + os.path.basename(found[-1]) + ')'
        if os.environ.get('MOVIIGEN_BENCH_BACKEND') == 'gloo':
            line['invalid'] = 'gloo test transport on a shared GPU (code-path check, not a measurement)'
        if args.layers or args.workload == 'tiny':
            line['invalid'] = 'debug configuration (not the BASELINE model)'
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
