"""Time WanVAE decode (BASELINE.json configs[4]: 1920x832x81f latent -> pixels) on one MI355X.
    python tools/bench_vae.py [--size 1920x832] [--frames 81] [--chunk N]
Synthetic weights of the shipped decoder shape (dim 96), z = randn seed 7."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import weights as W  # noqa: E402
import wan  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', default='1920x832')
ap.add_argument('--frames', type=int, default=81)
ap.add_argument('--chunk', type=int, default=1, help='latent frames per decoder chunk after the first')
ap.add_argument('--mode', default='exact', choices=('exact', 'bf16x3'), help='bf16x3: the opt-in split-bf16 convolutions (not the reference arithmetic)')
ap.add_argument('--tile', default='auto', help="voxels per workgroup of the wide exact convolutions: auto (by shape), 128, 256")
ap.add_argument('--stages', action='store_true', help='also time every decoder stage (first chunk / steady chunk) and model the layer pipeline of decode_pipelined for 2 / 4 / 8 ranks')
ap.add_argument('--bands', type=int, nargs='*', default=[], metavar='P',
                help='EMULATION on this one GPU of ONE rank of the W-band decode over P ranks (WanVAE.decode_spatial on a loop-back process group: the real '
                     "band width, halo columns, gather copies and kernel launches of an interior rank; the neighbours' data is its own): measured seconds of the "
                     'rank + the bytes it would put on a link, priced at --link-gbps.  Lines are marked `invalid: emulation`')
ap.add_argument('--link-gbps', type=float, default=45.0)
ap.add_argument('--upconv', default='phases', choices=('phases', 'gather'), help="the convs behind a 2x upsample: four 2x2 phase convs / one 3x3 through the upsample")
args = ap.parse_args()
Wd, Hd = (int(v) for v in args.size.split('x'))
T = (args.frames - 1) // 4 + 1
dev = torch.device('cuda:0')
vae = wan.modules.WanVAE(state_dict=W.make_vae_params(96, 1), device=dev, upconv=args.upconv, mode=args.mode,
                         tile=args.tile if args.tile == 'auto' else int(args.tile))
z = torch.randn(16, T, Hd // 8, Wd // 8, generator=torch.Generator().manual_seed(7)).to(dev)
chunks = [1] + [args.chunk] * ((T - 1) // args.chunk) + ([(T - 1) % args.chunk] if (T - 1) % args.chunk else [])
torch.cuda.synchronize()
t0 = time.perf_counter()
video = vae.model.decode(z, chunks=chunks)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
del video
t1 = time.perf_counter()
video = vae.model.decode(z, chunks=chunks)           # second decode: the caching allocator already holds every block
torch.cuda.synchronize()
dt_warm = time.perf_counter() - t1
# conv FLOPs of the decoder at this size (1116.5 TF at 1920x832x81, scales with voxels)
# (what the MFMAs EXECUTE: the phase-decomposed up-convs do 4/9 of their taps — 1065.8 instead of 1116.5 TF, bench.py vae_decode_flops)
flops = (1065.8e12 if args.upconv == 'phases' else 1116.5e12) * (Wd * Hd * args.frames) / (1920 * 832 * 81)
print(json.dumps({'metric': 'vae_decode_sec', 'value': dt, 'second_decode_sec': dt_warm, 'upconv': args.upconv, 'mode': args.mode, 'tile': args.tile, 'size': args.size, 'frames': args.frames, 'chunks': chunks[:3],
                  'tflops_fp32': flops / dt / 1e12, 'fp32_mfma_peak_tflops': 157.3, 'frac': flops / dt / 157.3e12,
                  'finite': bool(torch.isfinite(video).all().item()), 'peak_mem_gb': torch.cuda.max_memory_allocated() / 2**30}))

if args.bands:
    # one rank of P, emulated: what the W split costs a rank in kernels and copies (measured) and in link time (bytes / rate, not overlapped)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from emulate_rank import patch_dist
    from wan.distributed._test_transport import EmulatedGroup
    patch_dist()
    for P in args.bands:
        r = 1 if P > 2 else 0                                    # an interior rank (two neighbours) whenever there is one
        grp = EmulatedGroup(P, r, f'vae bands {P}')
        vae.model.decode_spatial(z[:, :2], group=grp)            # warm-up of the band-shaped launches
        torch.cuda.synchronize()
        grp.link_bytes = {k: 0 for k in grp.link_bytes}
        t1 = time.perf_counter()
        vae.model.decode_spatial(z, group=grp)
        torch.cuda.synchronize()
        sec = time.perf_counter() - t1
        halo_b, gather_b = grp.link_bytes['p2p'], grp.link_bytes['all_gather']
        band_video = 3 * args.frames * Hd * (Wd // P) * 4
        # halos: per direction on its own link; the k|v all-gather: 1/P of the buffer to each peer, all links concurrently (EmulatedGroup counts per link);
        # the band of the video: ranks 1..P-1 -> rank 0 on P - 1 links concurrently
        link_s = (halo_b - (band_video if r else 0) + gather_b + band_video) / (args.link_gbps * 1e9)
        print(json.dumps({'metric': 'vae_decode_band_rank_sec', 'invalid': 'emulation: one rank of P on a loop-back group, link time modelled',
                          'ranks': P, 'emulated_rank': r, 'band_columns_latent': (Wd // 8) // P, 'rank_seconds_measured': sec,
                          'link_bytes_per_link': {'halo_columns_and_video_band': halo_b, 'kv_all_gather': gather_b}, 'link_seconds_modelled_not_overlapped': link_s,
                          'modelled_makespan_s': sec + link_s, 'single_gpu_s': dt_warm, 'modelled_efficiency': dt_warm / (P * (sec + link_s)), 'size': args.size, 'frames': args.frames}))

if args.stages:
    # per-stage milliseconds (HIP events around every _run_stage call of one more decode), the cost per MAC of each kernel
    # class relative to the wide 3x3x3 convolutions (-> wan/modules/vae.py REL_MS_PER_MAC), and the modelled makespan of
    # the layer pipeline cut by these times (WanVAE_.decode_pipelined)
    from wan.modules.vae import partition_costs, pipeline_makespan
    m = vae.model
    ev, orig = [], m._run_stage

    def timed(stage, x, cache, idx):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        y = orig(stage, x, cache, idx)
        b.record()
        ev.append((a, b))
        return y
    m._run_stage = timed
    m.decode(z, chunks=chunks)
    torch.cuda.synchronize()
    m._run_stage = orig
    stages = m._stages()
    S = len(stages)
    ms = [[a.elapsed_time(b) for a, b in ev[c * S:(c + 1) * S]] for c in range(len(chunks))]
    first, steady = ms[0], [sum(ms[c][i] for c in range(1, len(chunks)) if chunks[c] == chunks[1]) / max(1, sum(1 for c in range(1, len(chunks)) if chunks[c] == chunks[1])) for i in range(S)]
    macs = m.stage_costs(Hd // 8, Wd // 8)
    cls = {}
    base = m.P['decoder.head.0.gamma'].numel()
    for (kind, pre, _), c, t in zip(stages, macs, steady):
        k = kind
        if kind in ('conv1', 'res'):
            k = 'narrow' if m.P[pre + ('.weight' if kind == 'conv1' else 'residual.6.weight')].shape[0] == base else 'wide'
        a = cls.setdefault(k, [0.0, 0.0])
        a[0] += t
        a[1] += c * chunks[1]
    rel = {k: (v[0] / v[1]) / (cls['wide'][0] / cls['wide'][1]) for k, v in cls.items()}
    print('stage                                kind   first_ms  steady_ms  GMAC/frame')
    for (kind, pre, _), a, b, c in zip(stages, first, steady, macs):
        print(f'{pre:36s} {kind:6s} {a:9.2f} {b:10.2f} {c / 1e9:11.1f}')
    print('REL_MS_PER_MAC =', {k: round(v, 3) for k, v in rel.items()})
    h, w = Hd // 8, Wd // 8
    for P in (2, 4, 8):
        for name, weights in (('measured ms', steady), ('MACs', macs)):
            cuts = partition_costs(weights, P)
            seg_f = [sum(first[a:b]) for a, b in zip(cuts, cuts[1:])]
            seg_s = [sum(steady[a:b]) for a, b in zip(cuts, cuts[1:])]
            # bytes over a cut per steady chunk at ~45 GB/s effective per xGMI link (one direction, one peer)
            xfer = []
            for cpos in cuts[1:-1]:
                f, hh, ww, cc = m.stage_out_shape(cpos, chunks[1], False, h, w)
                xfer.append(f * hh * ww * cc * 4 / 45e9 * 1e3)
            mk, eff = pipeline_makespan(seg_f, seg_s, len(chunks), xfer)
            print(f'P={P} cut by {name:11s}: cuts {cuts}  steady segment ms {[round(v) for v in seg_s]}  transfer ms/chunk {[round(v, 1) for v in xfer]}  '
                  f'modelled makespan {mk / 1e3:.2f} s  (single GPU {(sum(first) + (len(chunks) - 1) * sum(steady)) / 1e3:.2f} s, efficiency {eff:.2f})')
