"""ORACLE — CPU restatement of the umT5 encoder (test infrastructure, NOT product code).

Restates reference wan/modules/t5.py (encoder-only path of T5EncoderModel) on a flat
{state_dict-name: tensor} dict:

  t5.py:47-59     T5LayerNorm (RMS, fp32 statistics, weight)             -> t5_norm()
  t5.py:40-44     GELU (tanh form) ; :117-141 T5FeedForward gate*fc1 -> fc2  -> t5_ffn()
  t5.py:62-113    T5Attention: q,k,v,o without bias, NO 1/sqrt(d) scaling, additive position bias,
                  key mask filled with finfo.min, softmax in fp32          -> t5_attention()
  t5.py:222-263   T5RelativeEmbedding (bidirectional log buckets)        -> rel_buckets(), pos_bias()
  t5.py:144-167   T5SelfAttention block (pre-norm residual)               -> t5_block()
  t5.py:266-312   T5Encoder.forward (per-layer pos embedding: shared_pos=False for umT5)
  t5.py:498-518   T5EncoderModel.__call__: encode, then cut each row at its mask length

`emulate_bf16=True` models the reference's deployment dtype (`t5_dtype = bfloat16`: every parameter
and activation tensor is bf16, each op accumulates in fp32 and rounds its result)."""
import math

import torch
import torch.nn.functional as F


def _bf(x, on):
    return x.to(torch.bfloat16).to(torch.float32) if on else x


def rel_buckets(lq, lk, num_buckets=32, max_dist=128):
    """T5RelativeEmbedding._relative_position_bucket, bidirectional (t5.py:242-263): [lq, lk] int64."""
    rel = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = torch.abs(rel)
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_dist / max_exact) *
                         (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(rel < max_exact, rel, large)


def pos_bias(emb, lq, lk, num_buckets):
    """[heads, lq, lk] additive bias from the [num_buckets, heads] embedding table."""
    return emb[rel_buckets(lq, lk, num_buckets)].permute(2, 0, 1)


def t5_norm(x, w, bf, eps=1e-6):
    y = x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)
    return _bf(w * _bf(y, bf), bf)


def _lin(x, w, bf):
    return _bf(F.linear(_bf(x, bf), _bf(w, bf)), bf)


def t5_attention(P, pre, x, bias, klen, heads, bf):
    L, _ = x.shape
    q = _lin(x, P[pre + 'q.weight'], bf).view(L, heads, -1).permute(1, 0, 2)
    k = _lin(x, P[pre + 'k.weight'], bf).view(L, heads, -1).permute(1, 0, 2)
    v = _lin(x, P[pre + 'v.weight'], bf).view(L, heads, -1).permute(1, 0, 2)
    s = _bf(_bf(torch.matmul(q, k.transpose(1, 2)), bf) + _bf(bias, bf), bf)
    s = s[:, :, :klen]                              # masked keys get finfo.min -> weight 0
    p = _bf(torch.softmax(s.float(), dim=-1), bf)
    o = _bf(torch.matmul(p, v[:, :klen]), bf).permute(1, 0, 2).reshape(L, -1)
    return _lin(o, P[pre + 'o.weight'], bf)


def t5_ffn(P, pre, x, bf):
    g = _lin(x, P[pre + 'gate.0.weight'], bf)
    # GELU (t5.py:46-50) is a chain of elementwise ops: in the bf16 deployment dtype EVERY one of them rounds
    r = lambda v: _bf(v, bf)  # noqa: E731
    inner = r(math.sqrt(2.0 / math.pi) * r(g + r(0.044715 * r(torch.pow(g, 3.0)))))
    g = r(r(0.5 * g) * r(1.0 + r(torch.tanh(inner))))
    h = _bf(_lin(x, P[pre + 'fc1.weight'], bf) * g, bf)
    return _lin(h, P[pre + 'fc2.weight'], bf)


def t5_encode(P, cfg, ids, klen, emulate_bf16=False):
    """T5Encoder.forward for ONE padded sequence ids [L] with klen valid tokens -> [klen, dim]."""
    bf = emulate_bf16
    L = ids.shape[0]
    x = _bf(P['token_embedding.weight'][ids], bf)
    for i in range(cfg['num_layers']):
        pre = f'blocks.{i}.'
        bias = pos_bias(P[pre + 'pos_embedding.embedding.weight'], L, L, cfg['num_buckets'])
        x = _bf(x + t5_attention(P, pre + 'attn.', t5_norm(x, P[pre + 'norm1.weight'], bf), bias, klen,
                                 cfg['num_heads'], bf), bf)
        x = _bf(x + t5_ffn(P, pre + 'ffn.', t5_norm(x, P[pre + 'norm2.weight'], bf), bf), bf)
    return t5_norm(x, P['norm.weight'], bf)[:klen]
