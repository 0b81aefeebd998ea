"""ONE rank of a multi-GPU run, emulated on ONE GPU — a prediction to check the first real scaling curve against (no node
with more than one GPU has been available to this repository, SCALE_r*.json are skip records).

The real engine runs — WanModel with sequence parallelism / CFG halves / block shards configured for P ranks — on process groups that are
`wan.distributed._test_transport.EmulatedGroup` objects: every collective is a LOOP-BACK device copy of the real message size on the
calling stream (the peers' data is this rank's own, repeated), so all the per-rank shapes are the real ones: GEMM M = L / P (or L / (P/2) in
the CFG-parallel layout, one forward per step), attention over all L tokens x heads / P heads in the pipeline groups choose_groups() picks, the
packed q|k|v exchange buffers, the 703 MB block gathers.  Printed per configuration (one JSON line each, marked `invalid: emulation`):
  compute_s_per_step      this rank's step with the loop-back copies in place of the transfers
  link_bytes_per_step     what the rank would send over ONE of its P - 1 xGMI links (all links carry the same amount, concurrently)
  exchange_ms_per_step    those bytes at the ASSUMED per-link rate (--link-gbps, default 50 GB/s per direction: 7 links x ~153 GB/s
                          bidirectional per GPU, task statement) — fully exposed, and hidden except pipeline fill / drain (1 / groups)
  predicted_s_per_step    compute + exchange (both bounds) -> implied strong-scaling efficiency against the 1-GPU step of the same build
    python tools/emulate_rank.py [--workload 1080p] [--ranks 2 4 8] [--layouts ulysses cfg2] [--fsdp-at 8] [--t1-s S]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import wan  # noqa: E402
from wan.backend import ops  # noqa: E402
from wan.distributed._test_transport import EmulatedGroup  # noqa: E402
from wan.distributed.cfg_parallel import CfgParallel  # noqa: E402
from wan.distributed.fsdp import BlockShards  # noqa: E402
from wan.distributed.ulysses import HeadExchange  # noqa: E402
from wan.distributed.xdit_context_parallel import enable_sequence_parallel  # noqa: E402
from wan.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler  # noqa: E402

WORKLOADS = {'720p': (1280, 720, 81), '1080p': (1920, 832, 81), '1056p': (1920, 1056, 81)}


def patch_dist():
    """torch.distributed's bookkeeping calls, taught about EmulatedGroup (this process only; nothing is initialised)"""
    orig = {n: getattr(dist, n) for n in ('get_world_size', 'get_rank', 'is_initialized', 'get_backend', 'get_global_rank')}
    em = lambda g: isinstance(g, EmulatedGroup)  # noqa: E731
    dist.get_world_size = lambda group=None: group.size if em(group) else orig['get_world_size'](group)
    dist.get_rank = lambda group=None: group.rank if em(group) else orig['get_rank'](group)
    dist.is_initialized = lambda: True
    dist.get_backend = lambda group=None: 'emulated' if em(group) else orig['get_backend'](group)
    dist.get_global_rank = lambda group, r: r if em(group) else orig['get_global_rank'](group, r)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='1080p', choices=sorted(WORKLOADS))
    ap.add_argument('--ranks', type=int, nargs='+', default=[2, 4, 8])
    ap.add_argument('--layouts', nargs='+', default=['ulysses', 'cfg2'], choices=['ulysses', 'cfg2'])
    ap.add_argument('--fsdp-at', type=int, nargs='*', default=[8], help='rank counts at which the block-sharded variant is run too (last: it releases the full weights)')
    ap.add_argument('--steps', type=int, default=1)
    ap.add_argument('--link-gbps', type=float, default=50.0)
    ap.add_argument('--t1-s', type=float, default=None, help='1-GPU s/step of this build on this box (measured here when omitted)')
    ap.add_argument('--layers', type=int, default=None, help='debug')
    args = ap.parse_args()
    patch_dist()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    cfg = dict(model_type='t2v', patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=5120, ffn_dim=13824, freq_dim=256, text_dim=4096,
               out_dim=16, num_heads=40, num_layers=args.layers or 40, eps=1e-6)
    Wd, Hd, frames = WORKLOADS[args.workload]
    lat_shape = (16, (frames - 1) // 4 + 1, Hd // 8, Wd // 8)
    L = lat_shape[1] * (lat_shape[2] // 2) * (lat_shape[3] // 2)
    model = wan.modules.WanModel(**cfg, device=dev)
    model.init_weights(seed=0)
    model.eval().requires_grad_(False)
    g = torch.Generator(device=dev).manual_seed(0)
    ctx = torch.randn(512, 4096, device=dev, generator=g).bfloat16()
    ctx_null = torch.randn(130, 4096, device=dev, generator=g).bfloat16()
    sch = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)

    def run(cfgp, steps):
        """warm-up step + `steps` timed steps of bench.py's loop body -> s/step"""
        latent = torch.randn(*lat_shape, device=dev, generator=g)
        sch.set_timesteps(50, device=dev, shift=5.0)
        ts, ts_host = sch.timesteps, sch.timesteps.tolist()
        noise_pred = torch.empty_like(latent)

        def step(i):
            nonlocal latent
            t = ts[i:i + 1]
            if cfgp is None:
                cond = model([latent], t=t, context=[ctx], seq_len=L)[0]
                uncond = model([latent], t=t, context=[ctx_null], seq_len=L)[0]
            else:
                mine = model([latent], t=t, context=[ctx], seq_len=L)[0]
                cond, uncond = cfgp.exchange(mine)
            ops.cfg_combine(noise_pred, uncond, cond, 5.0)
            latent = sch.step(noise_pred.unsqueeze(0), ts_host[i], latent.unsqueeze(0), return_dict=False)[0].squeeze(0)
        step(0)
        torch.cuda.synchronize()
        HeadExchange.trace, BlockShards.trace = [], []
        t0 = time.perf_counter()
        for i in range(1, 1 + steps):
            step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ov = HeadExchange.overlap_summary()
        HeadExchange.trace = BlockShards.trace = None
        assert torch.isfinite(latent).all().item()
        return dt, ov

    t1 = args.t1_s
    if t1 is None:
        t1, _ = run(None, args.steps)
    print(json.dumps({'workload': args.workload, 'tokens': L, 'single_gpu_s_per_step': t1, 'how': 'this process, same loop, no groups'}), flush=True)

    def configure(P, layout, fsdp):
        groups = {}
        if layout == 'cfg2':
            half = P // 2
            groups['pair'] = EmulatedGroup(2, 0, 'cfg pair')
            groups['sp'] = EmulatedGroup(half, 0, f'ulysses {half}') if half > 1 else None
            if half > 1:
                enable_sequence_parallel(model, group=groups['sp'])
            else:
                model.sp_size, model.sp_rank, model.sp_group, model._ws = 1, 0, None, {}
            cfgp = CfgParallel(0, groups['pair'], groups['sp'], half)
        else:
            groups['sp'] = EmulatedGroup(P, 0, f'ulysses {P}')
            enable_sequence_parallel(model, group=groups['sp'])
            cfgp = None
        if fsdp:
            groups['shard'] = EmulatedGroup(P, 0, f'shards {P}')
            BlockShards(model, group=groups['shard'], sync_module_states=False)
        return cfgp, groups

    todo = [(P, lay, False) for P in args.ranks for lay in args.layouts if not (lay == 'cfg2' and P % 2)]
    todo += [(P, lay, True) for P in (args.fsdp_at or []) for lay in args.layouts[-1:]]
    sharded = False
    for P, layout, fsdp in todo:
        if sharded:
            break                       # block shards released the full weights: one sharded configuration per process
        cfgp, groups = configure(P, layout, fsdp)
        sharded = fsdp
        dt, ov = run(cfgp, args.steps)
        sp = groups.get('sp')
        sp_size = sp.size if sp is not None else 1
        link = {k: (sp.link_bytes[k] if sp is not None else 0) for k in ('all_to_all', 'all_gather')}
        per_step = 1.0 / (args.steps + 1)          # the counters include the warm-up step
        a2a = link['all_to_all'] * per_step + link['all_gather'] * per_step
        pair = groups['pair'].link_bytes['all_gather'] * per_step if 'pair' in groups else 0
        shard = groups['shard'].link_bytes['all_gather'] * per_step if 'shard' in groups else 0
        xch = [w['xchg'] for w in model._ws.values() if 'xchg' in w]
        G = len(xch[0].groups) if xch else 1
        rate = args.link_gbps * 1e9
        ex_full = (a2a + pair) / rate * 1e3
        ex_hidden = (a2a / G + pair) / rate * 1e3
        sh_ms = shard / rate * 1e3
        line = {'invalid': 'emulation: one rank on one GPU, loop-back copies in place of the transfers', 'workload': args.workload, 'ranks': P,
                'layout': ('cfg2 x ' if layout == 'cfg2' else '') + f'ulysses_sp{sp_size}' + (f' x fsdp{P}' if fsdp else ''),
                'compute_s_per_step': dt, 'pipeline_groups': [n for _, n in xch[0].groups] if xch else None,
                'loopback_exchange_ms_per_step': ov['exchange_ms'] / (args.steps), 'loopback_exposed_ms_per_step': ov['exposed_ms'] / (args.steps),
                'link_bytes_per_step': {'ulysses_exchange': a2a, 'cfg_pair': pair, 'block_gathers': shard},
                'assumed_link_gbps': args.link_gbps,
                'exchange_ms_per_step': {'fully_exposed': ex_full, 'hidden_except_fill_drain': ex_hidden},
                'block_gather_ms_per_step_if_exposed': sh_ms,
                'predicted_s_per_step': {'worst': dt + (ex_full + sh_ms) / 1e3, 'best': dt + ex_hidden / 1e3},
                'single_gpu_s_per_step': t1,
                'implied_strong_scaling_efficiency': {'worst': t1 / (P * (dt + (ex_full + sh_ms) / 1e3)), 'best': t1 / (P * (dt + ex_hidden / 1e3))}}
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
