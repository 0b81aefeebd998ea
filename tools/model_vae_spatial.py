"""MODEL (no GPU, nothing measured here): WanVAE decode split over P ranks along the W axis, from the measured per-stage times of the single-GPU
decode (`tools/bench_vae.py --stages` -> profiles/r04b_vae_stages.txt; the decoder has not changed since).

Round 4 modelled the LAYER pipeline (`decode_pipelined`, what is built): 3.6-3.8 s on 8 ranks, efficiency 0.30 — six chunks through eight stages is
mostly fill and drain, and the cuts behind the 96-channel stage carry 9.8 GB per chunk.  The review asked for the other axis: every
rank owns W / P columns of every activation; a 3x3 convolution needs ONE halo column from each neighbour (sent before the convolution: T x H x C_in x 4
bytes per side); the per-frame attention block gathers its frame (all-gather of the 104 x 240 x 384 input); rank 0 collects the video.  Per stage:
    time = measured ms / P x (rounds the rank's tiles take on 512 workgroup slots / the same for its share of the full launch)     [grid quantisation]
         + halo bytes / link rate per convolution of the stage (counted as exposed)
Not built — the numbers say what building it would buy.
    python tools/model_vae_spatial.py [profiles/r04b_vae_stages.txt] [--link-gbps 45]"""
import math
import re
import sys

path = next((a for a in sys.argv[1:] if not a.startswith('--')), 'profiles/r04b_vae_stages.txt')
link = float(sys.argv[sys.argv.index('--link-gbps') + 1]) if '--link-gbps' in sys.argv else 45.0
rows = []
for ln in open(path):
    m = re.match(r'(decoder\.\S+)\s+(\w+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)', ln)
    if m:
        rows.append((m.group(1), m.group(2), float(m.group(3)), float(m.group(4)), float(m.group(5))))
assert len(rows) == 20, len(rows)
# geometry of the 1920x832 decode: (H, W, channels in) per stage, latent frames per steady chunk = 4 -> frames T at that depth
H0, W0 = 104, 240
geo = {}
lvl = {'conv1': (1, 16), 'middle': (1, 384), 'up0': (1, 384), 'up1': (2, 384), 'up2': (4, 192), 'up3': (8, 96)}
for name, kind, *_ in rows:
    idx = int(re.search(r'upsamples\.(\d+)\.', name).group(1)) if 'upsamples' in name else None
    if 'conv1' in name:
        s, c, t = 1, 16, 4
    elif 'middle' in name or (idx is not None and idx <= 3):
        s, c, t = 1, 384, 4
    elif idx is not None and idx <= 7:
        s, c, t = 2, 384 if idx == 4 else 192 * 2 if idx < 7 else 192 * 2, 8          # 208 x 480, temporal 2x done
    elif idx is not None and idx <= 11:
        s, c, t = 4, 192, 16
    else:
        s, c, t = 8, 96, 16
    geo[name] = (H0 * s, W0 * s, c, t)
convs = {'conv1': 1, 'res': 2, 'attn': 0, 'up': 1, 'head': 1}
SLOTS = 512


def rounds(voxels, cout):
    tiles = math.ceil(voxels / 128) * math.ceil(cout / 96)
    return math.ceil(tiles / SLOTS), tiles


print(f'stage table: {path}; link rate {link:.0f} GB/s per direction; steady chunk = 4 latent frames, 5 steady chunks + the first')
single = sum(r[2] for r in rows) + 5 * sum(r[3] for r in rows)
print(f'single GPU: {single / 1e3:.2f} s (sum of the measured stage times)')
for P in (2, 4, 8):
    tot_first = tot_steady = halo_ms = quant_loss = 0.0
    for name, kind, first, steady, _ in rows:
        H, W, C, T = geo[name]
        full_r, _ = rounds(T * H * W, C)
        mine_r, _ = rounds(T * H * (W // P), C)
        q = (mine_r * P) / full_r if full_r else 1.0          # > 1: the rank's share takes more rounds than 1 / P of the launch's
        q = max(q, 1.0)
        t_s = steady / P * q
        t_f = first / P * max((rounds(1 * H * (W // P), C)[0] * P) / max(rounds(1 * H * W, C)[0], 1), 1.0)
        hb = convs[kind] * 2 * T * H * C * 4                    # both sides, per chunk
        if kind == 'attn':                                      # gather the frame's input (T frames x H x W x C), (P - 1) / P of it arrives over P - 1 links
            hb = T * H * W * C * 4 / P
        h_ms = hb / (link * 1e9) * 1e3
        tot_steady += t_s + h_ms
        tot_first += t_f + h_ms / 4
        halo_ms += h_ms
        quant_loss += t_s - steady / P
    video_ms = 3 * 81 * 832 * 1920 * 4 / P / (link * 1e9) * 1e3      # each rank's share to rank 0, the P - 1 links concurrently
    make = tot_first + 5 * tot_steady + video_ms
    print(f'P = {P}: modelled makespan {make / 1e3:.2f} s  (efficiency {single / (P * make):.2f}); per steady chunk: compute {tot_steady - halo_ms:.0f} ms '
          f'(of it grid quantisation {quant_loss:.0f}), halo + gather {halo_ms:.1f} ms; video to rank 0 {video_ms:.1f} ms')
