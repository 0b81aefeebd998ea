"""WanVAE decode on MI355X — drop-in for the decode side of reference wan/modules/vae.py.

Public surface kept: `WanVAE(z_dim=16, vae_pth=..., dtype=torch.float, device=...)`,
`.model.z_dim`, `.decode(list[Tensor[16,T,h,w]]) -> list[Tensor[3,4T-3,8h,8w]]` fp32 in [-1,1]
(reference vae.py:619-663), reading the reference checkpoint `Wan2.1_VAE.pth` (a plain
state_dict; encoder tensors are ignored: VAE *encode* is training-side preprocessing, out of the
denoising hot path).

Execution: activations are channels-last [T][H][W][C] fp32 on the device; every convolution
(3x3x3 causal, 3x1x1 time_conv, 3x3 after nearest-2x, 1x1) is ONE implicit-GEMM launch on the
exact-f32 MFMA (mg_vae_conv_f32) with the causal cache, the zero padding, the 2x upsample, the
bias and the residual add folded in; RMS_norm+SiLU and the per-frame d=384 attention are HIP
kernels as well.  The chunked decode with its 2-frame feature cache follows the reference
protocol (vae.py:101-141,202-220,423-472,544-568) slot for slot.
"""
import logging
import os

import torch

from ..backend import ops

__all__ = ['WanVAE']

CACHE_T = 2

_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
         0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
        3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


class _Band:
    """One rank's band of image columns in the W-split multi-GPU decode (`WanVAE_.decode_spatial`): rank r of P owns the latent columns
    [starts[r], starts[r] + widths[r]) — widths differ by at most one — and the same band, scaled, of every later activation."""

    def __init__(self, group, P, rank, W):
        base, extra = divmod(W, P)
        if base < 1:
            raise ValueError(f'decode_spatial: {P} ranks need at least {P} latent columns, the latent has {W}')
        self.widths = [base + (1 if r < extra else 0) for r in range(P)]
        self.starts = [sum(self.widths[:r]) for r in range(P)]
        self.group, self.P, self.rank = group, P, rank
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < P - 1 else None
        self.scale = 1                     # 2x per spatial up-sampler passed
        self.halo_bytes = 0                # what this rank sent as halos (reported by tools/bench_vae.py)

    def halo(self, x):
        """x [T,H,Wl,C] -> (x with the neighbours' border columns attached: [T,H,nl+Wl+nr,C], nl).  One halo column per side is what a
        3x3 convolution reads across the cut; an edge rank has no neighbour on that side — there the convolution's own
        zero padding is the reference's."""
        from ..distributed import collectives
        T, H, Wl, C = x.shape
        nl, nr = int(self.left is not None), int(self.right is not None)
        xh = x.new_empty(T, H, nl + Wl + nr, C)
        xh[:, :, nl:nl + Wl].copy_(x)
        sends, recvs = [], []
        lbuf = rbuf = None
        if nl:
            sends.append((x[:, :, :1].contiguous(), self.left))
            lbuf = x.new_empty(T, H, 1, C)
            recvs.append((lbuf, self.left))
        if nr:
            sends.append((x[:, :, -1:].contiguous(), self.right))
            rbuf = x.new_empty(T, H, 1, C)
            recvs.append((rbuf, self.right))
        collectives.neighbor_exchange(sends, recvs, self.group)
        self.halo_bytes += sum(t.numel() * 4 for t, _ in sends)
        if nl:
            xh[:, :, :1].copy_(lbuf)
        if nr:
            xh[:, :, -1:].copy_(rbuf)
        return xh, nl

    def gather_cols(self, t):
        """t [T,H,Wl,K] (this band) -> [T,H,W,K] of all bands in image order, on every rank (the attention block's keys / values)."""
        from ..distributed import collectives
        T, H, Wl, K = t.shape
        ws = [w * self.scale for w in self.widths]
        wmax = max(ws)
        mine = t.new_zeros(T, H, wmax, K)
        mine[:, :, :Wl].copy_(t)
        parts = t.new_empty(self.P, T, H, wmax, K)
        collectives.all_gather(parts, mine, self.group)
        full = t.new_empty(T, H, sum(ws), K)
        c = 0
        for r in range(self.P):
            full[:, :, c:c + ws[r]].copy_(parts[r][:, :, :ws[r]])
            c += ws[r]
        return full


class WanVAE_:
    """decoder-side container: parameters keyed by the reference state_dict names, repacked for
    the channels-last kernels ([Cout,Cin,kt,kh,kw] -> [Cout,kt,kh,kw,Cin])."""

    def __init__(self, state_dict, z_dim=16, device='cuda', upconv='phases', mode='exact', tile='auto'):
        """upconv: how the 3x3 conv behind a nearest-2x upsample runs — 'phases' = four 2x2 convs of the image with
        pre-summed taps (4/9 of the multiply-adds; default), 'gather' = the 3x3 conv reading through the upsample (the two
        agree to fp32 rounding of the weight sums; kept as the cross-check)."""
        if upconv not in ('phases', 'gather'):
            raise ValueError(f"upconv must be 'phases' or 'gather', got {upconv!r}")
        if mode not in ('exact', 'bf16x3'):
            raise ValueError(f"mode must be 'exact' (the reference's fp32 arithmetic) or 'bf16x3', got {mode!r}")
        self.upconv = upconv
        self.mode = mode          # 'bf16x3': opt-in split-bf16 convolutions (~1e-5 relative per conv)
        if tile not in ('auto', 128, 256):
            raise ValueError(f"tile must be 'auto' (= 128), 128 or 256 (voxels per workgroup of the wide exact convolutions; 256 is the slower "
                             f"measurement variant), got {tile!r}")
        # passed with every conv call (ABI 7): the arithmetic mode, and in bits 8-9 the measurement override of the voxel tile
        self._conv_mode = (ops.VAE_BF16X3 if mode == 'bf16x3' else ops.VAE_EXACT) | ({'auto': 0, 128: 1, 256: 2}[tile] << 8)
        self.z_dim = z_dim
        self.device = torch.device(device)
        self.P = {}
        for k, v in state_dict.items():
            if not (k.startswith('decoder.') or k.startswith('conv2.')):
                continue
            v = v.to(torch.float32)
            if k.endswith('weight') and v.dim() == 5:
                v = v.permute(0, 2, 3, 4, 1)
            elif k.endswith('weight') and v.dim() == 4:      # Conv2d -> kt = 1
                v = v.permute(0, 2, 3, 1).unsqueeze(1)
            elif k.endswith('gamma'):
                v = v.reshape(-1)
            self.P[k] = v.contiguous().to(self.device)
        n_up = 1 + max(int(k.split('.')[2]) for k in self.P if k.startswith('decoder.upsamples.'))
        self.layout = []
        for i in range(n_up):
            pre = f'decoder.upsamples.{i}.'
            self.layout.append(('up' if (pre + 'resample.1.weight') in self.P else 'res', pre))
        self.n_slots = sum(1 for k, v in self.P.items()
                           if k.startswith('decoder.') and k.endswith('weight') and v.dim() == 5 and
                           'resample' not in k and 'to_qkv' not in k and 'proj' not in k)
        self.mean = torch.tensor(_MEAN[:z_dim], dtype=torch.float32, device=self.device)
        self.inv_std = (1.0 / torch.tensor(_STD[:z_dim], dtype=torch.float32)).to(self.device)
        self._band = None                  # set for the duration of a decode_spatial call: this rank's column band

    # ---- building blocks -----------------------------------------------------------------------
    def _new(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def _conv(self, name, x, cache=None, up2=False, residual=None):
        w = self.P[name + '.weight']
        kt, kh, kw = w.shape[1:4]
        T, H, W, _ = x.shape
        out = self._new(T, 2 * H if up2 else H, 2 * W if up2 else W, w.shape[0])
        return ops.vae_conv(x, w, self.P[name + '.bias'], out, kt, kh, kw, cache=cache, up2=up2, residual=residual,
                            mode=self._conv_mode)

    def _cached_conv(self, name, x, cache, idx, residual=None):
        """the feat_cache protocol of every 3x3x3 conv (reference vae.py:205-217)."""
        i = idx[0]
        prev = cache[i]
        col0 = None
        if self._band is not None:
            # W-band decode: the convolution reads one column across each cut — fetch the neighbours' border columns first; the causal cache
            # keeps the frames WITH their halos (so nothing is exchanged twice)
            cols = x.shape[2]
            x, col0 = self._band.halo(x)
        if x.shape[0] >= CACHE_T:
            cx = x[-CACHE_T:].clone()
        elif prev is not None:
            cx = torch.cat([prev[-1:], x[-1:]], dim=0)
        else:
            cx = x[-1:].clone()
        if col0 is None:
            out = self._conv(name, x, cache=prev, residual=residual)
        else:
            w = self.P[name + '.weight']
            kt, kh, kw = w.shape[1:4]
            out = ops.vae_conv_cols(x, w, self.P[name + '.bias'], self._new(x.shape[0], x.shape[1], cols, w.shape[0]), kt, kh, kw, col0,
                                    cache=prev, residual=residual, mode=self._conv_mode)
        cache[i] = cx
        idx[0] += 1
        return out

    def _norm_silu(self, x, gamma, silu=True):
        return ops.vae_rmsnorm_silu(x, self.P[gamma], self._new(*x.shape), silu)

    def _res(self, pre, x, cache, idx):
        h = x
        if (pre + 'shortcut.weight') in self.P:
            h = self._conv(pre + 'shortcut', x)
        y = self._norm_silu(x, pre + 'residual.0.gamma')
        y = self._cached_conv(pre + 'residual.2', y, cache, idx)
        y = self._norm_silu(y, pre + 'residual.3.gamma')
        return self._cached_conv(pre + 'residual.6', y, cache, idx, residual=h)

    def _attn(self, pre, x):
        T, H, W, C = x.shape
        L = H * W
        y = self._norm_silu(x, pre + 'norm.gamma', silu=False)
        qkv = self._conv(pre + 'to_qkv', y)
        a = self._new(T, H, W, C)
        if self._band is not None:
            # W-band decode: queries = this band's pixels, keys / values = every band's (one all-gather of k|v per chunk); a row's result does
            # not depend on which rows are computed with it (mg_vae_attn_rows_f32)
            kv = self._band.gather_cols(qkv[..., C:])
            Lk = kv.shape[1] * kv.shape[2]
            ws = self._new(ops.vae_attn_workspace_floats(Lk, C))
            ops.vae_attn_rows(qkv.view(T, L, 3 * C)[..., :C], kv.view(T, Lk, 2 * C), a.view(T, L, C), ws)
            return self._conv(pre + 'proj', a, residual=x)
        ws = self._new(ops.vae_attn_workspace_floats(L, C))
        ops.vae_attn(qkv.view(T, L, 3 * C), a.view(T, L, C), ws)
        return self._conv(pre + 'proj', a, residual=x)

    def _up(self, pre, x, cache, idx):
        if (pre + 'time_conv.weight') in self.P:       # upsample3d (reference vae.py:103-137)
            i = idx[0]
            if cache[i] is None:
                cache[i] = 'Rep'
                idx[0] += 1
            else:
                rep = isinstance(cache[i], str)
                if x.shape[0] >= CACHE_T:
                    cx = x[-CACHE_T:].clone()
                elif rep:
                    cx = torch.cat([torch.zeros_like(x[-1:]), x[-1:]], dim=0)
                else:
                    cx = torch.cat([cache[i][-1:], x[-1:]], dim=0)
                y = self._conv(pre + 'time_conv', x, cache=None if rep else cache[i])
                cache[i] = cx
                idx[0] += 1
                T, H, W, C2 = y.shape
                x = ops.vae_time_interleave(y, self._new(2 * T, H, W, C2 // 2))
        if self.upconv == 'gather':
            if self._band is not None:
                raise NotImplementedError("decode_spatial runs the up-sampling convolutions as phase convolutions (upconv='phases')")
            return self._conv(pre + 'resample.1', x, up2=True)
        name = pre + 'resample.1'
        wp = self.P.get(name + '.phases')
        if wp is None:                                   # folded once per checkpoint
            wp = self.P[name + '.phases'] = ops.vae_upconv_fold_weights(self.P[name + '.weight'])
        T, H, W, _ = x.shape
        if self._band is not None:
            xh, col0 = self._band.halo(x)
            self._band.scale *= 2
            return ops.vae_upconv_phases_cols(xh, wp, self.P[name + '.bias'], self._new(T, 2 * H, 2 * W, wp.shape[1]), col0, mode=self._conv_mode)
        return ops.vae_upconv_phases(x, wp, self.P[name + '.bias'], self._new(T, 2 * H, 2 * W, wp.shape[1]), mode=self._conv_mode)

    # ---- the decoder as a list of stages (each owns a contiguous range of feat_cache slots) ---------
    def _stages(self):
        """[(kind, prefix, cache slots used)] in execution order (reference Decoder3d.forward, vae.py:423-472)."""
        st = [('conv1', 'decoder.conv1', 1), ('res', 'decoder.middle.0.', 2), ('attn', 'decoder.middle.1.', 0),
              ('res', 'decoder.middle.2.', 2)]
        for kind, pre in self.layout:
            st.append((kind, pre, 2 if kind == 'res' else (1 if (pre + 'time_conv.weight') in self.P else 0)))
        st.append(('head', 'decoder.head.', 1))
        return st

    def _run_stage(self, stage, x, cache, idx):
        kind, pre, _ = stage
        if kind == 'conv1':
            return self._cached_conv(pre, x, cache, idx)
        if kind == 'res':
            return self._res(pre, x, cache, idx)
        if kind == 'attn':
            return self._attn(pre, x)
        if kind == 'up':
            return self._up(pre, x, cache, idx)
        x = self._norm_silu(x, 'decoder.head.0.gamma')
        return self._cached_conv('decoder.head.2', x, cache, idx)

    def _decoder_chunk(self, x, cache, first=0, last=None):
        """stages [first, last) on one chunk; `cache` is the full feat_cache list (a rank only ever
        touches the slots of its own stages)."""
        stages = self._stages()
        idx = [sum(s[2] for s in stages[:first])]
        for stage in stages[first:len(stages) if last is None else last]:
            x = self._run_stage(stage, x, cache, idx)
        return x

    def stage_costs(self, h, w):
        """MACs per steady-state chunk (one latent frame) of every stage at latent size h x w — the
        weights of the layer-pipelined multi-GPU decode."""
        costs, frames, px = [], 1, h * w
        for kind, pre, _ in self._stages():
            names = [k for k in self.P if k.startswith(pre) and k.endswith('weight') and self.P[k].dim() == 5]
            c = 0
            for k in names:
                scale = 1
                if 'resample' in k:                                   # the conv runs at the upsampled resolution: 9 taps
                    scale = 4 if self.upconv == 'gather' else 16 / 9  # per output, or 4 as phase convs
                f = 2 * frames if ('resample' in k and (pre + 'time_conv.weight') in self.P) else frames
                c += self.P[k].numel() * px * scale * f
            if kind == 'attn':
                c += 2 * px * px * self.P[pre + 'proj.weight'].shape[0]
            costs.append(c)
            if kind == 'up':
                px *= 4
                if (pre + 'time_conv.weight') in self.P:
                    frames *= 2
        return costs

    @torch.no_grad()
    def decode(self, z, chunks=None):
        """z [16,T,h,w] -> [3, 1+4(T-1), 8h, 8w] fp32 clamped to [-1,1]."""
        return self._decode(z, chunks)

    def _decode(self, z, chunks=None):
        z = z.to(self.device, torch.float32).contiguous()
        C, T, H, W = z.shape
        x = ops.vae_latent_in(z, self.mean, self.inv_std, self._new(T, H, W, C))
        x = self._conv('conv2', x)
        # The reference decodes ONE latent frame per decoder call (vae.py:555-566) to bound its peak memory (2.45 GB per
        # activation at 832x1920); the cache protocol makes any chunking of frames 1.. give the same values (SURVEY
        # Appendix A; tests: chunked == unchunked).  With 288 GB of HBM the engine decodes 4 latent frames per call: the
        # low-resolution stages then launch enough tiles to fill 256 CUs (9.77 vs 10.16 s at 1920x832, peak 45 GB).
        # MOVIIGEN_VAE_CHUNK=1 restores the reference's chunking.
        chunks = self._chunks(T, chunks)
        cache = [None] * (self.n_slots + 8)
        video = self._new(3, 1 + 4 * (T - 1), 8 * H, 8 * W)
        t0, f0 = 0, 0
        for n in chunks:
            y = self._decoder_chunk(x[t0:t0 + n], cache)
            ops.vae_video_out(y, video, f0)
            t0 += n
            f0 += y.shape[0]
        assert f0 == video.shape[1]
        return video

    def _chunks(self, T, chunks=None):
        """the latent-frame chunking of a decode: frame 0 alone (the 'Rep' protocol of the temporal up-samplers needs it),
        then MOVIIGEN_VAE_CHUNK (default 4) frames per decoder call — see _decode."""
        if chunks is None:
            n = max(1, int(os.environ.get('MOVIIGEN_VAE_CHUNK', '4')))
            chunks = [1] + [n] * ((T - 1) // n) + ([(T - 1) % n] if (T - 1) % n else [])
        assert sum(chunks) == T and chunks[0] == 1
        return chunks

    def stage_out_shape(self, last, n, first_chunk, H, W):
        """[frames, H', W', C] of the activation that leaves stage `last - 1` for a chunk of n latent frames at latent size
        H x W — what crosses a pipeline cut (the receiver allocates from this; no shape header travels)."""
        f, C = n, self.z_dim
        for kind, pre, _ in self._stages()[:last]:
            if kind == 'conv1':
                C = self.P[pre + '.weight'].shape[0]
            elif kind == 'res':
                C = self.P[pre + 'residual.6.weight'].shape[0]
            elif kind == 'up':
                if (pre + 'time_conv.weight') in self.P and not first_chunk:
                    f *= 2                              # the first chunk skips time_conv ('Rep', reference vae.py:106-108)
                H, W, C = 2 * H, 2 * W, self.P[pre + 'resample.1.weight'].shape[0]
            elif kind == 'head':
                C = self.P['decoder.head.2.weight'].shape[0]
        return (f, H, W, C)

    def stage_weights(self, h, w, stage_ms=None):
        """what the pipelined decode balances: measured milliseconds per stage and steady-state chunk when given
        (tools/bench_vae.py --stages prints them for a size), else MACs x the measured cost per MAC of the stage's kernel
        class (REL_MS_PER_MAC: the 96-channel stage runs at a different efficiency than the 384-channel ones, the Cout = 3
        head re-reads its input 27 times for 3 output channels, the attention block is three launches per 2048 query rows)."""
        if stage_ms is not None:
            assert len(stage_ms) == len(self._stages())
            return list(stage_ms)
        out, base = [], self.P['decoder.head.0.gamma'].numel()         # the decoder's `dim` (96): its narrowest stage
        for (kind, pre, _), c in zip(self._stages(), self.stage_costs(h, w)):
            k = kind
            if kind in ('conv1', 'res'):
                cout = self.P[pre + ('.weight' if kind == 'conv1' else 'residual.6.weight')].shape[0]
                k = 'narrow' if cout == base else 'wide'
            out.append(c * REL_MS_PER_MAC[k])
        return out

    @torch.no_grad()
    def decode_spatial(self, z, group=None, chunks=None):
        """Multi-GPU decode split along W (the reference decodes on rank 0 alone, wan/text2video.py:260-261 ->
        vae.py:544-568): rank r owns a band of image columns of EVERY activation and runs the whole decoder on it, chunk by chunk with the
        single-GPU chunk list and its own causal caches.  What crosses ranks: one halo column per side in front of every 3x3 convolution
        (T x H x C floats per side, `_Band.halo`), the attention block's keys / values (one all-gather per chunk at latent resolution), and at
        the end each band of the video to rank 0.  All P ranks compute all the time — unlike the layer pipeline (`decode_pipelined`), whose
        six chunks through eight segments are mostly fill and drain.  Every output voxel is the same dot product in the same order as in the
        one-GPU launch: the video on rank 0 is bit-identical to `decode` (tests/dist_vae_worker.py, 2 / 4 / 8 ranks, uneven bands).
        Every rank passes the same latent; returns the video on rank 0, None elsewhere."""
        import torch.distributed as dist
        from ..distributed import collectives
        P, rank = dist.get_world_size(group), dist.get_rank(group)
        if P == 1:
            return self.decode(z, chunks)
        z = z.to(self.device, torch.float32)
        C, T, H, W = z.shape
        band = _Band(group, P, rank, W)
        c0, wl = band.starts[rank], band.widths[rank]
        chunks = self._chunks(T, chunks)
        self._band = band
        try:
            zb = z[:, :, :, c0:c0 + wl].contiguous()
            x = ops.vae_latent_in(zb, self.mean, self.inv_std, self._new(T, H, wl, C))
            x = self._conv('conv2', x)                                  # 1x1x1: no halo
            cache = [None] * (self.n_slots + 8)
            mine = self._new(3, 1 + 4 * (T - 1), 8 * H, 8 * wl)         # this band of the video
            t0, f0 = 0, 0
            for n in chunks:
                band.scale = 1
                y = self._decoder_chunk(x[t0:t0 + n], cache)
                ops.vae_video_out(y, mine, f0)
                t0 += n
                f0 += y.shape[0]
            assert f0 == mine.shape[1] and band.scale == 8
        finally:
            self._band = None
        self.last_halo_bytes = band.halo_bytes
        if rank != 0:
            collectives.send(mine, 0 if group is None else dist.get_global_rank(group, 0), group)
            return None
        video = self._new(3, mine.shape[1], 8 * H, 8 * W)
        video[:, :, :, :8 * wl].copy_(mine)
        for r in range(1, P):
            part = collectives.recv((3, mine.shape[1], 8 * H, 8 * band.widths[r]), r if group is None else dist.get_global_rank(group, r),
                                    self.device, group)
            video[:, :, :, 8 * band.starts[r]:8 * (band.starts[r] + band.widths[r])].copy_(part)
        return video

    @torch.no_grad()
    def decode_pipelined(self, z, group=None, chunks=None, stage_ms=None):
        """Multi-GPU decode: the stage list is cut into world_size contiguous segments of about
        equal TIME (stage_weights); segment s runs on rank P-1-s, so rank 0 owns the head and assembles the video.  The
        chunks of the single-GPU decode ([1, 4, 4, ...] latent frames) flow through the ranks as a pipeline — the causal
        feat_cache of a conv stays on the rank that owns the conv, only the activation of the cut crosses ranks: sent
        from a side stream (the next chunk's kernels do not wait for it) and received one chunk ahead on another.
        Exactly the single-GPU arithmetic.  Every rank passes the same latent; returns the video on rank 0, None elsewhere."""
        import torch.distributed as dist
        P, rank = dist.get_world_size(group), dist.get_rank(group)
        if P == 1:
            return self.decode(z, chunks)
        return self._decode_pipelined(z, group, P, rank, chunks, stage_ms)

    def _decode_pipelined(self, z, group, P, rank, chunks=None, stage_ms=None):
        import torch.distributed as dist
        from ..distributed import collectives
        z = z.to(self.device, torch.float32).contiguous()
        C, T, H, W = z.shape
        chunks = self._chunks(T, chunks)
        stages = self._stages()
        cuts = partition_costs(self.stage_weights(H, W, stage_ms), min(P, len(stages)))
        nseg = len(cuts) - 1
        if rank >= nseg:                                   # more ranks than stages: the rest idle
            return None
        seg = nseg - 1 - rank                              # rank 0 owns the last segment (head + video)
        first, last = cuts[seg], cuts[seg + 1]
        src = dist.get_global_rank(group, rank + 1) if group is not None else rank + 1      # upstream: segment seg-1
        dst = dist.get_global_rank(group, rank - 1) if group is not None else rank - 1      # downstream: segment seg+1
        cache = [None] * (self.n_slots + 8)
        video = self._new(3, 1 + 4 * (T - 1), 8 * H, 8 * W) if seg == nseg - 1 else None
        cuda = self.device.type == 'cuda'
        cur = torch.cuda.current_stream(self.device) if cuda else None
        s_send = torch.cuda.Stream(device=self.device) if cuda and video is None else None
        s_recv = torch.cuda.Stream(device=self.device) if cuda and first != 0 else None

        def post_recv(ci):
            """the cut activation of chunk ci, received on the side stream; -> (tensor, event the compute stream waits on)"""
            shape = self.stage_out_shape(first, chunks[ci], ci == 0, H, W)
            if s_recv is None:
                return collectives.recv(shape, src, self.device, group), None
            with torch.cuda.stream(s_recv):
                x = collectives.recv(shape, src, self.device, group)
                ev = torch.cuda.Event()
                ev.record(s_recv)
            return x, ev
        if first == 0:
            x_all = self._conv('conv2', ops.vae_latent_in(z, self.mean, self.inv_std, self._new(T, H, W, C)))
        else:
            nxt = post_recv(0)
        t0, f0 = 0, 0
        for ci, n in enumerate(chunks):
            if first == 0:
                x = x_all[t0:t0 + n]
            else:
                x, ev = nxt
                if ev is not None:
                    cur.wait_event(ev)
                    x.record_stream(cur)
                if ci + 1 < len(chunks):
                    nxt = post_recv(ci + 1)                # in flight while this chunk computes
            y = self._decoder_chunk(x, cache, first, last)
            if video is not None:
                ops.vae_video_out(y, video, f0)
                f0 += y.shape[0]
            elif s_send is None:
                collectives.send(y, dst, group)
            else:
                done = torch.cuda.Event()
                done.record(cur)
                with torch.cuda.stream(s_send):
                    s_send.wait_event(done)
                    collectives.send(y, dst, group)
                    y.record_stream(s_send)
            t0 += n
        if s_send is not None:
            cur.wait_stream(s_send)
        return video


# measured cost per multiply-add of the decoder's kernel classes, relative to the 384- / 192-channel 3x3x3 convolutions
# (vae_conv_kernel<4>), 1920x832 latent, 4-frame chunks — profiles/r04*_vae_stages.txt, tools/bench_vae.py --stages:
#   narrow = the 96-channel stage (vae_conv_kernel<3>: three of four MFMA column blocks of a 128-wide tile),
#   up   = Resample stages (time_conv 3x1x1 + the four 2x2 phase convs + interleave),
#   attn = the per-frame attention block (qkv / proj 1x1 convs + score blocks of 2048 rows + row softmax),
#   head = RMS_norm + SiLU + the 96 -> 3 convolution (vae_conv_kernel<0> on v_mfma_f32_4x4x1: bound by its 27-tap input re-read).
REL_MS_PER_MAC = {'wide': 1.0, 'narrow': 1.06, 'up': 1.08, 'attn': 1.5, 'head': 11.7}


def pipeline_makespan(seg_ms_first, seg_ms_steady, n_chunks, xfer_ms=None):
    """modelled wall time of the layer pipeline: segment s takes seg_ms_first[s] for chunk 0 (one latent frame, the
    temporal up-samplers skipped) and seg_ms_steady[s] for every later chunk; a chunk enters segment s when it has left
    segment s-1 (+ xfer_ms[s-1] on the link, overlapped with compute on both sides) and segment s is done with the chunk
    before.  -> (makespan, efficiency = sum of all work / (segments x makespan))."""
    S = len(seg_ms_steady)
    xfer_ms = xfer_ms or [0.0] * (S - 1)
    done = [[0.0] * n_chunks for _ in range(S)]
    for c in range(n_chunks):
        for s_ in range(S):
            t = seg_ms_first[s_] if c == 0 else seg_ms_steady[s_]
            ready = done[s_ - 1][c] + xfer_ms[s_ - 1] if s_ else 0.0
            free = done[s_][c - 1] if c else 0.0
            done[s_][c] = max(ready, free) + t
    total = sum(seg_ms_first) + (n_chunks - 1) * sum(seg_ms_steady)
    return done[-1][-1], total / (S * done[-1][-1])


def partition_costs(costs, parts):
    """cut points [0, ..., len(costs)] of `parts` contiguous non-empty segments minimising the largest
    segment sum (exact DP; a dozen stages)."""
    n = len(costs)
    parts = max(1, min(parts, n))
    pre = [0]
    for c in costs:
        pre.append(pre[-1] + c)
    INF = float('inf')
    best = [[INF] * (n + 1) for _ in range(parts + 1)]
    arg = [[0] * (n + 1) for _ in range(parts + 1)]
    best[0][0] = 0
    for p in range(1, parts + 1):
        for j in range(p, n + 1):
            for k in range(p - 1, j):
                v = max(best[p - 1][k], pre[j] - pre[k])
                if v < best[p][j]:
                    best[p][j], arg[p][j] = v, k
    cuts, j = [n], n
    for p in range(parts, 0, -1):
        j = arg[p][j]
        cuts.append(j)
    return cuts[::-1]


class WanVAE:

    def __init__(self, z_dim=16, vae_pth='cache/vae_step_411000.pth', dtype=torch.float, device='cuda',
                 state_dict=None, upconv='phases', mode='exact', tile='auto'):
        if dtype not in (torch.float, torch.float32):
            raise NotImplementedError('the reference decodes in fp32 (vae.py:623,658); so does this engine')
        self.dtype = dtype
        self.device = device
        if state_dict is None:
            logging.info(f'loading {vae_pth}')
            state_dict = torch.load(vae_pth, map_location='cpu', weights_only=True)
        self.model = WanVAE_(state_dict, z_dim=z_dim, device=device, upconv=upconv, mode=mode, tile=tile)
        self.mean, self.std = torch.tensor(_MEAN[:z_dim]), torch.tensor(_STD[:z_dim])
        self.scale = [self.mean, 1.0 / self.std]

    def encode(self, videos):
        raise NotImplementedError('VAE encode is training-side preprocessing (reference '
                                  'scripts/data_preprocess), outside the denoising hot path')

    def decode(self, zs):
        return [self.model.decode(u) for u in zs]

    def decode_peak_bytes(self, latent_shape):
        """upper bound of the device memory a decode of a [16, T, h, w] latent needs at its peak: the video itself
        plus, per decoder call of 4 latent frames, a handful of live fp32 activations of the widest stage (96 channels
        at 8h x 8w, 16 output frames; measured peak at 1920x832x81f: 45 GB) — used by WanT2V.generate to decide whether
        `offload_model` has to move anything."""
        _, T, h, w = latent_shape
        frames = 1 + 4 * (T - 1)
        video = 3 * frames * 64 * h * w * 4
        act = 16 * 64 * h * w * 96 * 4          # one [16 frames, 8h, 8w, 96] fp32 activation
        return video + 8 * act

    def decode_pipelined(self, zs, group=None):
        """multi-GPU decode, layer pipeline: every rank calls it with the same latents; videos on rank 0, None elsewhere."""
        return [self.model.decode_pipelined(u, group) for u in zs]

    def decode_spatial(self, zs, group=None):
        """multi-GPU decode, W bands (every rank computes its columns of every layer): same call contract as decode_pipelined."""
        return [self.model.decode_spatial(u, group) for u in zs]
