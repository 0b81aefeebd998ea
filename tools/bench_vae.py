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
ap.add_argument('--upconv', default='phases', choices=('phases', 'gather'), help="the convs behind a 2x upsample: four 2x2 phase convs / one 3x3 through the upsample")
args = ap.parse_args()
Wd, Hd = (int(v) for v in args.size.split('x'))
T = (args.frames - 1) // 4 + 1
dev = torch.device('cuda:0')
vae = wan.modules.WanVAE(state_dict=W.make_vae_params(96, 1), device=dev, upconv=args.upconv, mode=args.mode)
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
print(json.dumps({'metric': 'vae_decode_sec', 'value': dt, 'second_decode_sec': dt_warm, 'upconv': args.upconv, 'mode': args.mode, 'size': args.size, 'frames': args.frames, 'chunks': chunks[:3],
                  'tflops_fp32': flops / dt / 1e12, 'fp32_mfma_peak_tflops': 157.3, 'frac': flops / dt / 157.3e12,
                  'finite': bool(torch.isfinite(video).all().item()), 'peak_mem_gb': torch.cuda.max_memory_allocated() / 2**30}))
