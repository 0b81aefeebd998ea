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
    cfg = dict(model_type='t2v', patch_size=(1, 2, 2), text_len=64, in_dim=16, dim=1024, ffn_dim=2048, freq_dim=64,
               text_dim=128, out_dim=16, num_heads=8, num_layers=2, eps=1e-6)
    m = wan.modules.WanModel(**cfg, device=dev).init_weights(0)
    lat = W.randn((16, 16, 64, 64), 3).to(dev)             # grid (16, 32, 32) = 16 384 tokens
    ctx = W.randn((20, 128), 4).to(dev)
    t = torch.tensor([500.0], device=dev)
    ref = m([lat], t=t, context=[ctx], seq_len=16384)[0].clone()
    enable_sequence_parallel(m)
    m.sp_force = True
    for _ in range(3):
        got = m([lat], t=t, context=[ctx], seq_len=16384)[0]
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    print('SP_OVERLAP_RUN_OK groups', m._ws[next(iter(m._ws))]['xchg'].groups, flush=True)
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
    lines = [f'# tools/sp_overlap_trace.py [{label}]: rocprofv3 --kernel-trace of the pipelined Ulysses exchange, backend nccl (RCCL), 1 rank, ' + os.environ.get('MOVIIGEN_SP_GROUPS', '4') + ' head groups, transport ' + (os.environ.get('MOVIIGEN_SP_TRANSPORT') or 'torch') + ',',
             f'attention kernels          : n={len(attn)} total_us={sum(b - a for a, b in attn) / 1e3:.1f}',
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
