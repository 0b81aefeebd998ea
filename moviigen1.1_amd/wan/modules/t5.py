"""umT5-XXL text encoder on the MI355X kernels — drop-in for the encoder side of reference
wan/modules/t5.py (`T5EncoderModel`, `umt5_xxl(encoder_only=True)`), SURVEY.md §8(f) rank 1.

Same checkpoint (`models_t5_umt5-xxl-enc-bf16.pth`: token_embedding / blocks.N.{norm1,attn.{q,k,v,o},
norm2,ffn.{gate.0,fc1,fc2},pos_embedding.embedding} / norm) and the same call contract:
`T5EncoderModel(text_len, dtype, device, checkpoint_path, tokenizer_path)(texts, device)` ->
list of `[len_i, 4096]` bf16 tensors (each prompt cut at its attention-mask length, t5.py:504-518).

Execution: per prompt only the `len_i` valid rows are pushed through the 24 blocks (padded keys are
masked and padded query rows discarded by the reference, so this is exact): T5LayerNorm =
mg_rmsnorm_rope_bf16 (no RoPE), q|k|v as one fused GEMM, mg_t5_attn_bf16 (relative-position bias,
no scaling), gate / fc1 / fc2 GEMMs with mg_ew_bf16 for `fc1(x) * gelu(gate(x))` and the bf16
residual adds.  ~5 TFLOP per prompt: a few ms, once per video."""
import logging
import math

import torch
import torch.nn as nn

from ..backend import ops

__all__ = ['T5Encoder', 'T5EncoderModel', 'umt5_xxl']


class _W(nn.Module):
    def __init__(self, shape, dtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape, dtype=dtype, device=device), requires_grad=False)


class _Attn(nn.Module):
    def __init__(self, dim, dim_attn, device):
        super().__init__()
        bf = torch.bfloat16
        self.q, self.k, self.v = _W((dim_attn, dim), bf, device), _W((dim_attn, dim), bf, device), _W((dim_attn, dim), bf, device)
        self.o = _W((dim, dim_attn), bf, device)


class _FFN(nn.Module):
    def __init__(self, dim, dim_ffn, device):
        super().__init__()
        bf = torch.bfloat16
        self.gate = nn.ModuleDict({'0': _W((dim_ffn, dim), bf, device)})
        self.fc1, self.fc2 = _W((dim_ffn, dim), bf, device), _W((dim, dim_ffn), bf, device)


class _PosEmb(nn.Module):
    def __init__(self, num_buckets, num_heads, device):
        super().__init__()
        self.embedding = _W((num_buckets, num_heads), torch.bfloat16, device)


class _Block(nn.Module):
    def __init__(self, dim, dim_attn, dim_ffn, num_heads, num_buckets, device):
        super().__init__()
        self.norm1, self.norm2 = _W((dim,), torch.float32, device), _W((dim,), torch.float32, device)
        self.attn = _Attn(dim, dim_attn, device)
        self.ffn = _FFN(dim, dim_ffn, device)
        self.pos_embedding = _PosEmb(num_buckets, num_heads, device)


def relative_buckets(n, num_buckets=32, max_dist=128):
    """bucket of every relative position j - i in [-(n-1), n-1] (reference t5.py:242-263, same fp32
    torch expression), as the int32 table mg_t5_attn_bf16 indexes with (j - i) + n - 1."""
    rel = torch.arange(-(n - 1), n)
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = torch.abs(rel)
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return (out + torch.where(rel < max_exact, rel, large)).to(torch.int32)


class T5Encoder(nn.Module):

    def __init__(self, vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, shared_pos=False,
                 dropout=0.1, device=None):
        super().__init__()
        if shared_pos:
            raise NotImplementedError('umT5 uses per-layer position embeddings (shared_pos=False, t5.py:466)')
        self.dim, self.dim_attn, self.dim_ffn = dim, dim_attn, dim_ffn
        self.num_heads, self.num_layers, self.num_buckets = num_heads, num_layers, num_buckets
        self.token_embedding = _W((vocab, dim), torch.bfloat16, device)
        self.blocks = nn.ModuleList([_Block(dim, dim_attn, dim_ffn, num_heads, num_buckets, device)
                                     for _ in range(num_layers)])
        self.norm = _W((dim,), torch.float32, device)
        self._wqkv = None
        self._buckets = {}

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        for name, p in self.named_parameters():
            want = torch.float32 if 'norm' in name else torch.bfloat16
            if p.dtype != want:
                p.data = p.data.to(want)
        self._wqkv, self._buckets = None, {}
        return out

    def load_state_dict(self, state_dict, strict=True, assign=False):
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        self._wqkv = None
        return out

    def _fused(self):
        if self._wqkv is None:
            self._wqkv = []
            for b in self.blocks:
                w = torch.cat([b.attn.q.weight, b.attn.k.weight, b.attn.v.weight], 0).contiguous()
                da = self.dim_attn
                b.attn.q.weight.data, b.attn.k.weight.data, b.attn.v.weight.data = w[:da], w[da:2 * da], w[2 * da:]
                self._wqkv.append(w)
        return self._wqkv

    @torch.no_grad()
    def _encode_one(self, ids, klen):
        dev = self.token_embedding.weight.device
        if dev.type != 'cuda':
            raise RuntimeError('T5Encoder needs the model on a HIP device; there is no CPU path (oracle/t5.py is '
                               'the CPU reference)')
        bf = torch.bfloat16
        d, da, f, N = self.dim, self.dim_attn, self.dim_ffn, self.num_heads
        hd = da // N
        n = int(klen)
        e = lambda *s: torch.empty(*s, dtype=bf, device=dev)  # noqa: E731
        x, h, qkv, a, y = e(n, d), e(n, d), e(n, 3 * da), e(n, da), e(n, d)
        g, f1 = e(n, f), e(n, f)
        ops.embed_rows(self.token_embedding.weight, ids[:n].to(dev, torch.int64).contiguous(), x)
        if n not in self._buckets:
            self._buckets = {n: relative_buckets(n, self.num_buckets).to(dev)}
        rb = self._buckets[n]
        for blk, wqkv in zip(self.blocks, self._fused()):
            ops.rmsnorm_rope(x, blk.norm1.weight, 1e-6, d, h)
            ops.gemm(h, wqkv, None, ops.BIAS_BF16, qkv)
            ops.t5_attention(qkv[:, :da], qkv[:, da:2 * da], qkv[:, 2 * da:], blk.pos_embedding.embedding.weight, rb, a,
                             n, N, hd)
            ops.gemm(a, blk.attn.o.weight, None, ops.BIAS_BF16, y)
            ops.ew_bf16(x, y, x, 0)
            ops.rmsnorm_rope(x, blk.norm2.weight, 1e-6, d, h)
            ops.gemm(h, blk.ffn.gate['0'].weight, None, ops.BIAS_BF16, g)
            ops.gemm(h, blk.ffn.fc1.weight, None, ops.BIAS_BF16, f1)
            ops.ew_bf16(f1, g, f1, 1)
            ops.gemm(f1, blk.ffn.fc2.weight, None, ops.BIAS_BF16, y)
            ops.ew_bf16(x, y, x, 0)
        out = e(n, d)
        ops.rmsnorm_rope(x, self.norm.weight, 1e-6, d, out)
        return out

    def forward(self, ids, mask=None):
        """ids [B, L] int64, mask [B, L] (1 = token) -> [B, L, dim] bf16, zero beyond each mask length."""
        B, L = ids.shape
        dev = self.token_embedding.weight.device
        out = torch.zeros(B, L, self.dim, dtype=torch.bfloat16, device=dev)
        lens = [L] * B if mask is None else mask.gt(0).sum(dim=1).tolist()
        for b in range(B):
            if lens[b]:
                out[b, :lens[b]] = self._encode_one(ids[b], lens[b])
        return out


def umt5_xxl(encoder_only=True, return_tokenizer=False, dtype=torch.bfloat16, device='cpu', **kwargs):
    if not encoder_only or return_tokenizer:
        raise NotImplementedError('only the encoder of umT5-XXL is on the T2V path (reference t5.py:482-487)')
    if dtype != torch.bfloat16:
        raise NotImplementedError('the reference runs the text encoder in bf16 (config t5_dtype)')
    cfg = dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32)
    cfg.update(**kwargs)
    return T5Encoder(shared_pos=False, device=device, **cfg)


class T5EncoderModel:

    def __init__(self, text_len, dtype=torch.bfloat16, device='cuda', checkpoint_path=None, tokenizer_path=None,
                 shard_fn=None, model=None, tokenizer=None):
        self.text_len, self.dtype, self.device = text_len, dtype, device
        self.checkpoint_path, self.tokenizer_path = checkpoint_path, tokenizer_path
        if shard_fn is not None:
            raise NotImplementedError('t5_fsdp: the 9.4 GB bf16 encoder fits one MI355X many times over')
        if model is None:
            model = umt5_xxl(encoder_only=True, dtype=dtype, device='cpu').eval().requires_grad_(False)
            logging.info(f'loading {checkpoint_path}')
            model.load_state_dict(torch.load(checkpoint_path, map_location='cpu', weights_only=True))
        self.model = model.to(device)
        if tokenizer is None:
            from .tokenizers import HuggingfaceTokenizer
            tokenizer = HuggingfaceTokenizer(name=tokenizer_path, seq_len=text_len, clean='whitespace')
        self.tokenizer = tokenizer

    def __call__(self, texts, device):
        ids, mask = self.tokenizer(texts, return_mask=True, add_special_tokens=True)
        seq_lens = mask.gt(0).sum(dim=1).long().tolist()
        self.model.to(device)
        context = self.model(ids.to(device), mask.to(device))
        return [u[:v] for u, v in zip(context, seq_lens)]
