"""ORACLE — CPU restatement of the WanModel DiT forward (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (moviigen1.1_amd/) never does.

Restates, in plain torch CPU ops on a flat {state_dict-name: tensor} parameter dict, the
algorithm of the reference files (paths relative to the reference tree):

  wan/modules/model.py:15-25    sinusoidal_embedding_1d        -> sinusoid()
  wan/modules/model.py:28-67    rope_params / rope_apply       -> rope_table(), rope()
  wan/modules/model.py:70-86    WanRMSNorm                     -> rmsnorm()
  wan/modules/model.py:89-99    WanLayerNorm                   -> layernorm()
  wan/modules/attention.py:24-130 flash_attention (varlen, non-causal, k_lens mask) -> attention()
  wan/modules/model.py:102-181  WanSelfAttention / WanT2VCrossAttention
  wan/modules/model.py:228-313  WanAttentionBlock              -> block()
  wan/modules/model.py:316-343  Head                           -> head()
  wan/modules/model.py:486-609  WanModel.forward / unpatchify  -> dit_forward(), unpatchify()
  wan/distributed/xdit_context_parallel.py:23-62,65-152,155-198 (Ulysses SP)  -> dit_forward_sp_sim()
  scripts/train/model/model_seq.py:37-76,197-294,621-790 (training-side SP forward) -> dit_forward_train_sp_sim()

Two numeric modes:
  emulate_bf16=False : everything fp32 (RoPE fp64) — what the reference itself computes on CPU,
                       where every `autocast("cuda")` region is inert.  Pinned against the imported
                       reference by tests/golden (see tests/golden/make_golden.py).
  emulate_bf16=True  : inserts the bf16 roundings the reference performs on a GPU under
                       `torch.autocast(bf16)`: GEMM inputs/outputs bf16,
                       fp32 accumulate, fp32 residual stream, fp32 norms/modulation, fp32 head.
                       Pinned against the imported reference run under torch.autocast("cpu", bf16).
"""
import math

import torch
import torch.nn.functional as F


def _bf(x, on):
    """Round to bf16 and come back to fp32 (a no-op in fp32 mode)."""
    return x.to(torch.bfloat16).to(torch.float32) if on else x


def linear(x, w, b, bf):
    """nn.Linear under autocast: bf16 operands, fp32 accumulate, bf16 result (model.py:120-123)."""
    y = F.linear(_bf(x, bf), _bf(w, bf), None)
    if b is not None:
        y = y + (_bf(b, bf))
    return _bf(y, bf)


def sinusoid(dim, t):
    """model.py:15-25 — fp64 evaluation; caller casts to fp32."""
    half = dim // 2
    pos = t.to(torch.float64)
    freqs = torch.pow(10000.0, -torch.arange(half, dtype=torch.float64) / half)
    ang = torch.outer(pos, freqs)
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=1)


def rope_table(head_dim, max_len=1024, theta=10000.0):
    """model.py:28-36 + :473-478 — three frequency groups (temporal, height, width) as fp64 angles.

    Returns angle tables [max_len, c0], [max_len, c1], [max_len, c1] with c = head_dim/2,
    c1 = c//3, c0 = c - 2*c1 (dims d-4*(d//6), 2*(d//6), 2*(d//6) of the reference)."""
    d = head_dim

    def ang(dim):
        inv = 1.0 / torch.pow(theta, torch.arange(0, dim, 2, dtype=torch.float64) / dim)
        return torch.outer(torch.arange(max_len, dtype=torch.float64), inv)

    return ang(d - 4 * (d // 6)), ang(2 * (d // 6)), ang(2 * (d // 6))


def rope(x, grid, tables, pos0=0):
    """model.py:39-67 — rotate adjacent pairs of x [L, N, D] by the angle of token (f, h, w).

    Tokens beyond f*h*w (padding) are left untouched.  pos0 shifts the token index (sequence-
    parallel rank slice, xdit_context_parallel.py:43-57)."""
    f, h, w = grid
    L = x.shape[0]
    n_tok = f * h * w
    ta, th, tw = tables
    full = torch.cat([
        ta[:f].view(f, 1, 1, -1).expand(f, h, w, -1),
        th[:h].view(1, h, 1, -1).expand(f, h, w, -1),
        tw[:w].view(1, 1, w, -1).expand(f, h, w, -1),
    ], dim=-1).reshape(n_tok, -1)
    idx = torch.arange(L) + pos0
    valid = idx < n_tok
    ang = torch.zeros(L, full.shape[1], dtype=torch.float64)
    ang[valid] = full[idx[valid]]
    xd = x.to(torch.float64).reshape(L, x.shape[1], -1, 2)
    c, s = torch.cos(ang)[:, None, :], torch.sin(ang)[:, None, :]
    a, b = xd[..., 0], xd[..., 1]
    out = torch.stack([a * c - b * s, a * s + b * c], dim=-1).flatten(2)
    return out.to(torch.float32)


def rmsnorm(x, weight, eps, bf):
    """model.py:70-86 — fp32 normalise, cast back to x's dtype (bf16 in autocast mode), * weight."""
    xf = x.to(torch.float32)
    y = xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    return _bf(y, bf) * weight


def layernorm(x, eps, weight=None, bias=None):
    """model.py:89-99 — fp32 LayerNorm over the last dim."""
    return F.layer_norm(x.to(torch.float32), (x.shape[-1],), weight, bias, eps)


def attention(q, k, v, k_len, bf):
    """attention.py:24-130 semantics: softmax(q k^T / sqrt(D)) v per head, keys >= k_len masked;
    inputs cast to bf16 in autocast mode, fp32 accumulate, result in bf16 then back to q's dtype.
    q [Lq,N,D], k/v [Lk,N,D] -> [Lq,N,D]."""
    qh = _bf(q, bf).permute(1, 0, 2)
    kh = _bf(k, bf).permute(1, 0, 2)[:, :k_len]
    vh = _bf(v, bf).permute(1, 0, 2)[:, :k_len]
    s = torch.matmul(qh, kh.transpose(1, 2)) / math.sqrt(q.shape[-1])
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, vh).permute(1, 0, 2)
    return _bf(o, bf)


def block(P, pre, x, e0, seq_len_valid, grid, tables, ctx, num_heads, eps, bf, first_block,
          pos0=0, attn_fn=None):
    """model.py:274-313.  x [L, dim]; e0 [6, dim] fp32; ctx [Lc, dim]."""
    L, dim = x.shape
    hd = dim // num_heads
    e = (P[pre + 'modulation'][0] + e0).to(torch.float32)  # [6, dim]

    def norm_mod(xx, shift, scale):
        y = layernorm(xx, eps)
        if first_block and bf:
            y = _bf(y, True)  # model.py:99 .type_as(x): x is bf16 until the first residual add
        return y * (1 + scale) + shift

    # --- self attention (model.py:127-156)
    h = norm_mod(x, e[0], e[1])
    sa = pre + 'self_attn.'
    q = rmsnorm(linear(h, P[sa + 'q.weight'], P[sa + 'q.bias'], bf), P[sa + 'norm_q.weight'], eps, bf)
    k = rmsnorm(linear(h, P[sa + 'k.weight'], P[sa + 'k.bias'], bf), P[sa + 'norm_k.weight'], eps, bf)
    v = linear(h, P[sa + 'v.weight'], P[sa + 'v.bias'], bf)
    q = rope(q.view(L, num_heads, hd), grid, tables, pos0)
    k = rope(k.view(L, num_heads, hd), grid, tables, pos0)
    v = v.view(L, num_heads, hd)
    if attn_fn is None:
        a = attention(q, k, v, seq_len_valid, bf)
    else:
        a = attn_fn(q, k, v)
    y = linear(a.reshape(L, dim), P[sa + 'o.weight'], P[sa + 'o.bias'], bf)
    x = x.to(torch.float32) + y * e[2]

    # --- cross attention (model.py:159-181, 306)
    ca = pre + 'cross_attn.'
    h = layernorm(x, eps, P[pre + 'norm3.weight'], P[pre + 'norm3.bias'])
    q = rmsnorm(linear(h, P[ca + 'q.weight'], P[ca + 'q.bias'], bf), P[ca + 'norm_q.weight'], eps, bf)
    kc = rmsnorm(linear(ctx, P[ca + 'k.weight'], P[ca + 'k.bias'], bf), P[ca + 'norm_k.weight'], eps, bf)
    vc = linear(ctx, P[ca + 'v.weight'], P[ca + 'v.bias'], bf)
    Lc = ctx.shape[0]
    a = attention(q.view(L, num_heads, hd), kc.view(Lc, num_heads, hd), vc.view(Lc, num_heads, hd), Lc, bf)
    x = x + linear(a.reshape(L, dim), P[ca + 'o.weight'], P[ca + 'o.bias'], bf)

    # --- ffn (model.py:267-269, 307-309)
    h = layernorm(x, eps) * (1 + e[4]) + e[3]
    u = linear(h, P[pre + 'ffn.0.weight'], P[pre + 'ffn.0.bias'], bf)
    u = _bf(F.gelu(u, approximate='tanh'), bf)
    y = linear(u, P[pre + 'ffn.2.weight'], P[pre + 'ffn.2.bias'], bf)
    return x + y * e[5]


def time_embed(P, t, freq_dim):
    """model.py:541-545 — fp32 even under autocast.  Returns e [B, dim], e0 [B, 6, dim]."""
    s = sinusoid(freq_dim, t).to(torch.float32)
    e = F.linear(F.silu(F.linear(s, P['time_embedding.0.weight'], P['time_embedding.0.bias'])),
                 P['time_embedding.2.weight'], P['time_embedding.2.bias'])
    e0 = F.linear(F.silu(e), P['time_projection.1.weight'], P['time_projection.1.bias'])
    return e, e0.unflatten(1, (6, -1))


def text_embed(P, ctx, text_len, bf):
    """model.py:548-554 — zero-pad the prompt to text_len rows, Linear-GELU(tanh)-Linear."""
    pad = torch.cat([ctx.to(torch.float32), torch.zeros(text_len - ctx.shape[0], ctx.shape[1])])
    u = linear(pad, P['text_embedding.0.weight'], P['text_embedding.0.bias'], bf)
    u = _bf(F.gelu(u, approximate='tanh'), bf)
    return linear(u, P['text_embedding.2.weight'], P['text_embedding.2.bias'], bf)


def patch_embed(P, lat, patch, bf):
    """model.py:445-450,529-531 — k=s=patch Conv3d as a GEMM over (c,pt,ph,pw); tokens (f,h,w)."""
    C, Fr, H, W = lat.shape
    pt, ph, pw = patch
    g = (Fr // pt, H // ph, W // pw)
    cols = lat.reshape(C, g[0], pt, g[1], ph, g[2], pw).permute(1, 3, 5, 0, 2, 4, 6).reshape(
        g[0] * g[1] * g[2], C * pt * ph * pw)
    w = P['patch_embedding.weight'].reshape(P['patch_embedding.weight'].shape[0], -1)
    return linear(cols, w, P['patch_embedding.bias'], bf), g


def head(P, x, e, eps):
    """model.py:333-343 — all fp32."""
    m = (P['head.modulation'][0] + e[None, :]).to(torch.float32)  # [2, dim]
    y = layernorm(x, eps) * (1 + m[1]) + m[0]
    return F.linear(y, P['head.head.weight'], P['head.head.bias'])


def unpatchify(tok, grid, patch, out_dim):
    """model.py:581-609 — [L, pt*ph*pw*c] (c fastest) -> [c, F*pt, H*ph, W*pw]."""
    f, h, w = grid
    pt, ph, pw = patch
    u = tok[:f * h * w].view(f, h, w, pt, ph, pw, out_dim)
    u = torch.einsum('fhwpqrc->cfphqwr', u)
    return u.reshape(out_dim, f * pt, h * ph, w * pw)


def dit_forward(P, cfg, lat, t, ctx, seq_len, emulate_bf16=False, return_tokens=False):
    """WanModel.forward for ONE sample (model.py:486-579).

    P: state-dict-named fp32 tensors; cfg: dict(dim, ffn_dim, freq_dim, num_heads, num_layers,
    text_len, patch_size, out_dim, eps); lat [C,F,H,W] fp32; t: 0-d / [1] tensor; ctx [Lc, text_dim].
    Returns the predicted latent [out_dim, F, H, W] fp32."""
    bf = emulate_bf16
    x, grid = patch_embed(P, lat.to(torch.float32), cfg['patch_size'], bf)
    L = x.shape[0]
    assert L <= seq_len
    x = torch.cat([x, torch.zeros(seq_len - L, x.shape[1])])
    e, e0 = time_embed(P, t.reshape(1), cfg['freq_dim'])
    c = text_embed(P, ctx, cfg['text_len'], bf)
    tables = rope_table(cfg['dim'] // cfg['num_heads'])
    for i in range(cfg['num_layers']):
        x = block(P, f'blocks.{i}.', x, e0[0], L, grid, tables, c, cfg['num_heads'], cfg['eps'], bf,
                  first_block=(i == 0))
    y = head(P, x, e[0], cfg['eps'])
    if return_tokens:
        return y
    return unpatchify(y, grid, cfg['patch_size'], cfg['out_dim']).to(torch.float32)


# ------------------------------------------------------------------------------------------------
# Ulysses sequence parallelism, simulated in ONE process (xdit_context_parallel.py)
# ------------------------------------------------------------------------------------------------
def all_to_all_seq_to_head(parts):
    """forward a2a (scatter heads, gather sequence): P tensors [L/P, N, D] -> P tensors [L, N/P, D]."""
    Pn = len(parts)
    n = parts[0].shape[1] // Pn
    return [torch.cat([parts[src][:, r * n:(r + 1) * n] for src in range(Pn)], dim=0) for r in range(Pn)]


def all_to_all_head_to_seq(parts):
    """inverse a2a: P tensors [L, N/P, D] -> P tensors [L/P, N, D]."""
    Pn = len(parts)
    l = parts[0].shape[0] // Pn
    return [torch.cat([parts[src][r * l:(r + 1) * l] for src in range(Pn)], dim=1) for r in range(Pn)]


def dit_forward_sp_sim(P, cfg, lat, t, ctx, seq_len, sp, emulate_bf16=False):
    """usp_dit_forward + usp_attn_forward with the collectives replaced by list shuffles.
    Must equal dit_forward() when seq_len % sp == 0 and seq_len == L (no padding: the SP path does
    not mask padded keys, xdit_context_parallel.py:178-193)."""
    bf = emulate_bf16
    x, grid = patch_embed(P, lat.to(torch.float32), cfg['patch_size'], bf)
    L = x.shape[0]
    assert L == seq_len and seq_len % sp == 0 and cfg['num_heads'] % sp == 0
    e, e0 = time_embed(P, t.reshape(1), cfg['freq_dim'])
    c = text_embed(P, ctx, cfg['text_len'], bf)
    tables = rope_table(cfg['dim'] // cfg['num_heads'])
    lr = seq_len // sp
    xs = [x[r * lr:(r + 1) * lr] for r in range(sp)]
    for i in range(cfg['num_layers']):
        # run every rank up to the exchange point, exchange, finish: done by capturing q,k,v
        caps = []

        def grab(q, k, v):
            caps.append((q, k, v))
            return torch.zeros_like(q)

        for r in range(sp):
            block(P, f'blocks.{i}.', xs[r], e0[0], lr, grid, tables, c, cfg['num_heads'], cfg['eps'], bf,
                  first_block=(i == 0), pos0=r * lr, attn_fn=grab)
        qh = all_to_all_seq_to_head([cp[0] for cp in caps])
        kh = all_to_all_seq_to_head([cp[1] for cp in caps])
        vh = all_to_all_seq_to_head([cp[2] for cp in caps])
        oh = [attention(qh[r], kh[r], vh[r], seq_len, bf) for r in range(sp)]
        os_ = all_to_all_head_to_seq(oh)
        xs = [block(P, f'blocks.{i}.', xs[r], e0[0], lr, grid, tables, c, cfg['num_heads'], cfg['eps'], bf,
                    first_block=(i == 0), pos0=r * lr, attn_fn=(lambda q, k, v, rr=r: os_[rr]))
              for r in range(sp)]
    ys = [head(P, xs[r], e[0], cfg['eps']) for r in range(sp)]
    y = torch.cat(ys, dim=0)  # all_gather(dim=1) of the reference
    return unpatchify(y, grid, cfg['patch_size'], cfg['out_dim']).to(torch.float32)


# ------------------------------------------------------------------------------------------------
# training-side sequence-parallel forward (scripts/train/model/model_seq.py), simulated in ONE process
# ------------------------------------------------------------------------------------------------
def dit_forward_train_sp_sim(P, cfg, lat, t, batch_context, seq_len, sp, emulate_bf16=False):
    """scripts/train/model/model_seq.py:621-790 with its collectives replaced by list shuffles:

      * the token sequence is zero-padded to seq_len AFTER the patch embedding (:704-706) and chunked over the sp
        ranks (:757) — a rank may hold padded rows;
      * rope_apply_dist (:37-76): the rank's slice of the position table, padded with identity rotations;
      * self-attention (:197-256): all_to_all_4D to [all tokens, heads/sp], flash_attention with k_lens = the video's
        token count (padded keys masked), all_to_all_4D back;
      * cross-attention (:271-294): q goes through the same all-to-all, K/V keep this rank's heads (shrink_head),
        attention over all tokens x local heads, all-to-all back;
      * all_gather of x along tokens (:780), head and unpatchify on the gathered sequence.

    batch_context [text_len, text_dim]: the prompt embedding already padded (the `batch_context` argument, :748)."""
    bf = emulate_bf16
    N, eps = cfg['num_heads'], cfg['eps']
    x, grid = patch_embed(P, lat.to(torch.float32), cfg['patch_size'], bf)
    L, dim = x.shape
    hd = dim // N
    assert L <= seq_len and seq_len % sp == 0 and N % sp == 0
    x = torch.cat([x, torch.zeros(seq_len - L, dim)])
    e, e0 = time_embed(P, t.reshape(1), cfg['freq_dim'])
    assert batch_context.shape[0] == cfg['text_len']
    c = text_embed(P, batch_context, cfg['text_len'], bf)
    Lc = c.shape[0]
    tables = rope_table(hd)
    lr = seq_len // sp
    nl = N // sp
    xs = [x[r * lr:(r + 1) * lr] for r in range(sp)]
    for i in range(cfg['num_layers']):
        pre = f'blocks.{i}.'
        em = (P[pre + 'modulation'][0] + e0[0]).to(torch.float32)
        sa, ca = pre + 'self_attn.', pre + 'cross_attn.'
        qs, ks, vs = [], [], []
        for r in range(sp):
            y = layernorm(xs[r], eps)
            if i == 0 and bf:
                y = _bf(y, True)
            h = y * (1 + em[1]) + em[0]
            q = rmsnorm(linear(h, P[sa + 'q.weight'], P[sa + 'q.bias'], bf), P[sa + 'norm_q.weight'], eps, bf)
            k = rmsnorm(linear(h, P[sa + 'k.weight'], P[sa + 'k.bias'], bf), P[sa + 'norm_k.weight'], eps, bf)
            v = linear(h, P[sa + 'v.weight'], P[sa + 'v.bias'], bf)
            qs.append(rope(q.view(lr, N, hd), grid, tables, r * lr))
            ks.append(rope(k.view(lr, N, hd), grid, tables, r * lr))
            vs.append(v.view(lr, N, hd))
        qh, kh, vh = all_to_all_seq_to_head(qs), all_to_all_seq_to_head(ks), all_to_all_seq_to_head(vs)
        os_ = all_to_all_head_to_seq([attention(qh[r], kh[r], vh[r], L, bf) for r in range(sp)])
        for r in range(sp):
            y = linear(os_[r].reshape(lr, dim), P[sa + 'o.weight'], P[sa + 'o.bias'], bf)
            xs[r] = xs[r].to(torch.float32) + y * em[2]
        # cross-attention, head-sharded
        kc = rmsnorm(linear(c, P[ca + 'k.weight'], P[ca + 'k.bias'], bf), P[ca + 'norm_k.weight'], eps, bf).view(Lc, N, hd)
        vc = linear(c, P[ca + 'v.weight'], P[ca + 'v.bias'], bf).view(Lc, N, hd)
        qs = []
        for r in range(sp):
            h = layernorm(xs[r], eps, P[pre + 'norm3.weight'], P[pre + 'norm3.bias'])
            qs.append(rmsnorm(linear(h, P[ca + 'q.weight'], P[ca + 'q.bias'], bf), P[ca + 'norm_q.weight'], eps,
                              bf).view(lr, N, hd))
        qh = all_to_all_seq_to_head(qs)
        os_ = all_to_all_head_to_seq([attention(qh[r], kc[:, r * nl:(r + 1) * nl], vc[:, r * nl:(r + 1) * nl], Lc, bf)
                                      for r in range(sp)])
        for r in range(sp):
            xs[r] = xs[r] + linear(os_[r].reshape(lr, dim), P[ca + 'o.weight'], P[ca + 'o.bias'], bf)
            h = layernorm(xs[r], eps) * (1 + em[4]) + em[3]
            u = linear(h, P[pre + 'ffn.0.weight'], P[pre + 'ffn.0.bias'], bf)
            u = _bf(F.gelu(u, approximate='tanh'), bf)
            xs[r] = xs[r] + linear(u, P[pre + 'ffn.2.weight'], P[pre + 'ffn.2.bias'], bf) * em[5]
    xg = torch.cat(xs, dim=0)                       # all_gather(x, dim=1)
    y = head(P, xg, e[0], eps)
    return unpatchify(y, grid, cfg['patch_size'], cfg['out_dim']).to(torch.float32)
