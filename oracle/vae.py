"""ORACLE — CPU restatement of the WanVAE decode path (test infrastructure, NOT product code).

Restates wan/modules/vae.py of the reference on a flat {state_dict-name: tensor} dict:

  vae.py:17-36    CausalConv3d (temporal left pad 2, cache shrinks the pad)   -> causal_conv3d()
  vae.py:39-54    RMS_norm (F.normalize over channels * sqrt(C) * gamma)      -> rms_norm()
  vae.py:57-141   Upsample / Resample upsample2d, upsample3d + 'Rep' protocol -> resample()
  vae.py:186-220  ResidualBlock with the 2-frame feat_cache protocol           -> residual_block()
  vae.py:223-262  AttentionBlock (single head, per frame)                      -> attention_block()
  vae.py:423-472  Decoder3d.forward                                            -> decoder_chunk()
  vae.py:544-568  WanVAE_.decode (one latent frame per chunk, cache across chunks)
  vae.py:619-663  WanVAE: per-channel mean/std, clamp(-1, 1)                   -> vae_decode()

The chunking and the cache list (one slot per CausalConv3d in module order, including the never-
used slot of the 1x1x1 shortcut) follow the reference exactly, so intermediate caches can be
compared slot by slot.  `chunks=` lets tests drive other chunkings (any
chunking that isolates latent frame 0 gives the same video up to fp32 summation order).
"""
import torch
import torch.nn.functional as F

CACHE_T = 2

VAE_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
            0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
VAE_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
           3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


def causal_conv3d(x, w, b, cache=None):
    """x [1,C,T,H,W]; pads (kw//2, kh//2) spatially and 2*(kt//2) frames on the LEFT of time."""
    kt, kh, kw = w.shape[2:]
    pad_t = 2 * (kt // 2)
    if cache is not None and pad_t > 0:
        x = torch.cat([cache, x], dim=2)
        pad_t -= cache.shape[2]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, pad_t, 0))
    return F.conv3d(x, w, b)


def rms_norm(x, gamma):
    """channel-first L2 normalise * sqrt(C) * gamma (gamma broadcast over the trailing dims)."""
    c = x.shape[1]
    g = gamma.reshape(1, c, *([1] * (x.dim() - 2)))
    return F.normalize(x, dim=1) * (c ** 0.5) * g


def _next_cache(x, prev):
    """the feat_cache update shared by every 3x3x3 conv (vae.py:205-214)."""
    cx = x[:, :, -CACHE_T:].clone()
    if cx.shape[2] < 2 and prev is not None:
        cx = torch.cat([prev[:, :, -1:], cx], dim=2)
    return cx


def _cached_conv(P, name, x, cache, idx):
    i = idx[0]
    cx = _next_cache(x, cache[i])
    y = causal_conv3d(x, P[name + '.weight'], P[name + '.bias'], cache[i])
    cache[i] = cx
    idx[0] += 1
    return y


def residual_block(P, pre, x, cache, idx):
    h = x
    if (pre + 'shortcut.weight') in P:
        h = causal_conv3d(x, P[pre + 'shortcut.weight'], P[pre + 'shortcut.bias'])
    y = F.silu(rms_norm(x, P[pre + 'residual.0.gamma']))
    y = _cached_conv(P, pre + 'residual.2', y, cache, idx)
    y = F.silu(rms_norm(y, P[pre + 'residual.3.gamma']))
    y = _cached_conv(P, pre + 'residual.6', y, cache, idx)
    return y + h


def attention_block(P, pre, x):
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = rms_norm(y, P[pre + 'norm.gamma'])
    qkv = F.conv2d(y, P[pre + 'to_qkv.weight'], P[pre + 'to_qkv.bias'])
    q, k, v = qkv.reshape(b * t, 1, 3 * c, h * w).permute(0, 1, 3, 2).chunk(3, dim=-1)
    a = F.scaled_dot_product_attention(q, k, v)
    a = a.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
    a = F.conv2d(a, P[pre + 'proj.weight'], P[pre + 'proj.bias'])
    a = a.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)
    return a + x


def resample(P, pre, x, cache, idx):
    b, c, t, h, w = x.shape
    if (pre + 'time_conv.weight') in P:  # upsample3d
        i = idx[0]
        if cache[i] is None:
            cache[i] = 'Rep'
            idx[0] += 1
        else:
            cx = x[:, :, -CACHE_T:].clone()
            if cx.shape[2] < 2:
                if isinstance(cache[i], str):
                    cx = torch.cat([torch.zeros_like(cx), cx], dim=2)
                else:
                    cx = torch.cat([cache[i][:, :, -1:], cx], dim=2)
            prev = None if isinstance(cache[i], str) else cache[i]
            y = causal_conv3d(x, P[pre + 'time_conv.weight'], P[pre + 'time_conv.bias'], prev)
            cache[i] = cx
            idx[0] += 1
            y = y.reshape(b, 2, c, t, h, w)
            x = torch.stack((y[:, 0], y[:, 1]), dim=3).reshape(b, c, t * 2, h, w)
            t = t * 2
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.interpolate(y, scale_factor=(2.0, 2.0), mode='nearest-exact')
    y = F.conv2d(y, P[pre + 'resample.1.weight'], P[pre + 'resample.1.bias'], padding=1)
    return y.reshape(b, t, y.shape[1], 2 * h, 2 * w).permute(0, 2, 1, 3, 4)


def decoder_layout(P):
    """module order of decoder.upsamples from the key names: list of ('res'|'up', prefix)."""
    n = 1 + max(int(k.split('.')[2]) for k in P if k.startswith('decoder.upsamples.'))
    out = []
    for i in range(n):
        pre = f'decoder.upsamples.{i}.'
        out.append(('up' if (pre + 'resample.1.weight') in P else 'res', pre))
    return out


def count_cache_slots(P):
    """count_conv3d(decoder) (vae.py:475-480): every CausalConv3d incl. shortcuts and time_convs."""
    n = 0
    for k, v in P.items():
        if k.startswith('decoder.') and k.endswith('.weight') and v.dim() == 5:
            n += 1
    return n


def decoder_chunk(P, x, cache):
    """Decoder3d.forward on one chunk (vae.py:423-472)."""
    idx = [0]
    x = _cached_conv(P, 'decoder.conv1', x, cache, idx)
    x = residual_block(P, 'decoder.middle.0.', x, cache, idx)
    x = attention_block(P, 'decoder.middle.1.', x)
    x = residual_block(P, 'decoder.middle.2.', x, cache, idx)
    for kind, pre in decoder_layout(P):
        if kind == 'res':
            x = residual_block(P, pre, x, cache, idx)
            # the 1x1x1 shortcut owns a cache slot it never touches; the reference's feat_idx only
            # advances on the two 3x3x3 convs, so slots at the tail of the list stay None.
        else:
            x = resample(P, pre, x, cache, idx)
    x = F.silu(rms_norm(x, P['decoder.head.0.gamma']))
    x = _cached_conv(P, 'decoder.head.2', x, cache, idx)
    return x


def vae_decode(P, z, chunks=None, return_cache=False):
    """WanVAE.decode for one latent z [16,T,h,w] -> video [3, 1+4(T-1), 8h, 8w] in [-1,1]
    (vae.py:544-568, 657-663).  chunks: list of latent-frame counts (default: all ones)."""
    zc = z.shape[0]
    mean = torch.tensor(VAE_MEAN[:zc]).view(1, zc, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(VAE_STD[:zc])).view(1, zc, 1, 1, 1)
    x = z[None].to(torch.float32) / inv_std + mean
    x = causal_conv3d(x, P['conv2.weight'], P['conv2.bias'])
    T = x.shape[2]
    if chunks is None:
        chunks = [1] * T
    assert sum(chunks) == T and chunks[0] == 1
    cache = [None] * count_cache_slots(P)
    outs, t0 = [], 0
    for n in chunks:
        outs.append(decoder_chunk(P, x[:, :, t0:t0 + n], cache))
        t0 += n
    video = torch.cat(outs, dim=2).to(torch.float32).clamp_(-1, 1)[0]
    return (video, cache) if return_cache else video
