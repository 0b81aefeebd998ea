"""Evidence for the comm/compute overlap of the pipelined Ulysses exchange (wan/distributed/ulysses.py: HeadExchange).

    worker  :  python tools/sp_overlap_trace.py run          (under rocprofv3 --kernel-trace --output-format csv)
    analyse :  python tools/sp_overlap_trace.py analyse <dir with *kernel_trace.csv> [out.txt]

The worker runs the PRODUCTION transport — backend "nccl" (= RCCL) — with the sequence-parallel branch forced on a
1-rank group (a 1-GPU box cannot hold two RCCL ranks; with one rank the all-to-all is RCCL's own device copy kernel on
RCCL's stream, so the stream / event structure is exactly the multi-GPU one): a DiT of width 1024 (8 heads x 128),
16 384 tokens, 2 layers, exchange pipelined over MOVIIGEN_SP_GROUPS head groups (default 4; 8 = one head per group =
16 384 x 384 x 2 B = 12.6 MB per exchange, the per-peer message of BASELINE configs[2] at P = 8: 16 380 rows x 1 head x
(q|k|v) x 128 x 2 B).  MOVIIGEN_SP_TRANSPORT selects the transport as in production.  The analysis intersects the time intervals of the
RCCL kernels with those of the attention kernels of the same process: a non-zero overlap = the exchange of group g+1 /
the return of group g-1 really run under the attention of group g."""
import csv
import glob
import os
import sys


def run():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, 'moviigen1.1_amd'), os.path.join(root, 'tests', 'golden')]
    import torch
    import torch.distributed as dist
    import weights as W
    import wan
    from wan.distributed.xdit_context_parallel import enable_sequence_parallel
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29688')
    os.environ.setdefault('MOVIIGEN_SP_GROUPS', '4')
    torch.cuda.set_device(0)
    dev = torch.device('cuda:0')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    shape = os.environ.get('SP_TRACE_SHAPE', 'small')
    if shape == 'cfg2':
        # the per-rank group size of BASELINE configs[2] (1920x832x81f, Ulysses 8): L = 131 040 tokens x 5 local heads, ONE
        # head per pipeline group — 5.8 ms of attention per group; the loop-back exchange moves what 8 ranks together put
        # into this rank's receive buffer (131 040 x 384 x 2 B = 100 MB; per peer link it is 12.6 MB)
        os.environ['MOVIIGEN_SP_GROUPS'] = '5'
        cfg = dict(model_type='t2v', patch_size=(1, 2, 2), text_len=64, in_dim=16, dim=640, ffn_dim=1280, freq_dim=64,
                   text_dim=128, out_dim=16, num_heads=5, num_layers=2, eps=1e-6)
        lat_shape, L = (16, 21, 104, 240), 131040
    elif shape == 'fsdp':
        # block-shard prefetch: the 703 MB all-gather of a 14B-width block (loop-back on one rank) under the compute of
        # the previous block at the per-rank token count of configs[3] (L/4 = 41 580 rows)
        cfg = dict(model_type='t2v', patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=5120, ffn_dim=13824, freq_dim=256,
                   text_dim=4096, out_dim=16, num_heads=40, num_layers=4, eps=1e-6)
        lat_shape, L = (16, 21, 66, 120), 41580
    else:
        cfg = dict(model_type='t2v', patch_size=(1, 2, 2), text_len=64, in_dim=16, dim=1024, ffn_dim=2048, freq_dim=64,
                   text_dim=128, out_dim=16, num_heads=8, num_layers=2, eps=1e-6)
        lat_shape, L = (16, 16, 64, 64), 16384             # grid (16, 32, 32) = 16 384 tokens
    m = wan.modules.WanModel(**cfg, device=dev).init_weights(0)
    lat = W.randn(lat_shape, 3).to(dev)
    ctx = W.randn((20, cfg['text_dim']), 4).to(dev)
    t = torch.tensor([500.0], device=dev)
    ref = m([lat], t=t, context=[ctx], seq_len=L)[0].clone()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    m([lat], t=t, context=[ctx], seq_len=L)
    b.record()
    torch.cuda.synchronize()
    plain_ms = a.elapsed_time(b)
    if shape == 'fsdp':
        from wan.distributed.collectives import trace_summary
        from wan.distributed.fsdp import BlockShards, shard_model
        shard_model(m, device_id=0)
        for _ in range(2):
            got = m([lat], t=t, context=[ctx], seq_len=L)[0]
        BlockShards.trace = []
        a.record()
        got = m([lat], t=t, context=[ctx], seq_len=L)[0]
        b.record()
        torch.cuda.synchronize()
        tr = trace_summary(BlockShards.trace)
        BlockShards.trace = None
        assert torch.equal(got, ref)
        print(f'FSDP_OVERLAP_RUN_OK forward_ms unsharded {plain_ms:.2f} sharded(prefetch) {a.elapsed_time(b):.2f} gathers {tr["collectives"]} '
              f'gather_ms {tr["comm_ms"]:.2f} exposed_ms {tr["exposed_ms"]:.2f} hidden_frac {tr["hidden_frac"]}', flush=True)
        dist.destroy_process_group()
        return
    enable_sequence_parallel(m)
    m.sp_force = True
    from wan.distributed.ulysses import HeadExchange
    for _ in range(2):
        got = m([lat], t=t, context=[ctx], seq_len=L)[0]
    HeadExchange.trace = []
    a.record()
    got = m([lat], t=t, context=[ctx], seq_len=L)[0]
    b.record()
    torch.cuda.synchronize()
    ov = HeadExchange.overlap_summary()
    HeadExchange.trace = None
    assert torch.equal(got, ref)
    x = m._ws[next(iter(m._ws))]['xchg']
    print(f'SP_OVERLAP_RUN_OK shape {shape} groups {x.groups} reserve_cus {x.reserve_cus} forward_ms plain {plain_ms:.2f} exchanged {a.elapsed_time(b):.2f} '
          f'exchange_ms {ov["exchange_ms"]:.2f} exposed_ms {ov["exposed_ms"]:.2f} hidden_frac {ov["hidden_frac"]}', flush=True)
    dist.destroy_process_group()


def analyse(d, out=None):
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp']),
                         r.get('Stream_Id', r.get('Queue_Id', '?'))))
    attn = [(s, e) for n, s, e, _ in rows if 'attn_hd128' in n]
    comm = [(s, e, n) for n, s, e, _ in rows if 'nccl' in n.lower() or 'rccl' in n.lower() or 'copyBuffer' in n]     # RCCL kernels; blit copies of the peer_copy transport (same-device loop-back: no SDMA)
    pack = [(s, e) for n, s, e, _ in rows if 'sp_copy_blocks' in n]
    attn.sort()

    def overlap(iv):
        tot = 0
        for s, e in iv:
            for a, b in attn:
                if b <= s:
                    continue
                if a >= e:
                    break
                tot += min(e, b) - max(s, a)
        return tot
    c_tot = sum(e - s for s, e, _ in comm)
    c_ov = overlap([(s, e) for s, e, _ in comm])
    label = os.environ.get('SP_TRACE_LABEL', 'default')
    lines = [f'# tools/sp_overlap_trace.py [{label}]: rocprofv3 --kernel-trace of the pipelined Ulysses exchange, backend nccl (RCCL), 1 rank, shape ' + os.environ.get('SP_TRACE_SHAPE', 'small') + ', transport ' + (os.environ.get('MOVIIGEN_SP_TRANSPORT') or 'torch') + ', reserve_cus ' + os.environ.get('MOVIIGEN_SP_RESERVE_CUS', 'default') + ',',
             f'attention kernels          : n={len(attn)} total_us={sum(b - a for a, b in attn) / 1e3:.1f} longest_us={max([b - a for a, b in attn] or [0]) / 1e3:.1f}',
             f'RCCL kernels (all-to-all)  : n={len(comm)} total_us={c_tot / 1e3:.1f}  names={sorted({n.split("(")[0][:50] for _, _, n in comm})}',
             f'  of which UNDER attention : {c_ov / 1e3:.1f} us = {100.0 * c_ov / max(c_tot, 1):.1f} % of the RCCL kernel time',
             f'pack / unpack kernels      : n={len(pack)} total_us={sum(b - a for a, b in pack) / 1e3:.1f}',
             f'streams seen               : {sorted({st for *_, st in rows})}']
    # a slice of the timeline (one layer of the last forward) for the record
    rows.sort(key=lambda r: r[1])
    last_attn = [i for i, r in enumerate(rows) if 'attn_hd128' in r[0]]
    if last_attn:
        i0 = max(0, last_attn[-9] - 6) if len(last_attn) >= 9 else 0
        t0 = rows[i0][1]
        lines.append('timeline slice (us from the first row; stream; duration us; kernel):')
        for n, s_, e_, st in rows[i0:i0 + 34]:
            lines.append(f'  {(s_ - t0) / 1e3:9.1f}  st={st:>3s}  {(e_ - s_) / 1e3:7.1f}  {n.split("(")[0][:60]}')
    text = '\n'.join(lines) + '\n'
    if out:
        open(out, 'a').write(text)
    print(text)


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run()
    else:
        analyse(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
