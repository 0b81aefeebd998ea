"""flash_attention — the reference's single operator seam (wan/modules/attention.py:24-130),
served by the MI355X attention kernels instead of flash_attn's CUDA kernels.

Same signature and dtype contract: q [B,Lq,N,C], k/v [B,Lk,N,C]; inputs that are not fp16/bf16 are
cast to `dtype`; the result comes back in q's original dtype.  Supported subset = what the DiT
uses: non-causal, no dropout, no window, N_q == N_k.  WanModel itself does not go through this
wrapper (it calls the kernels on packed buffers).  The reference imports it BY NAME into
wan/modules/model.py (:10) and calls that module-level name (:146-151, :176); the same binding
exists here (`wan.modules.model.flash_attention`) and a caller who re-binds it is honoured: WanModel's
layer loop then calls the bound function with the reference's arguments for every self- and
cross-attention instead of the fused kernels (tests/test_gpu_parity.py::test_operator_seam_flash_attention)."""
import math

import torch

from ..backend import ops

__all__ = ['flash_attention', 'attention']


def flash_attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None,
                    causal=False, window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, version=None):
    assert dtype == torch.bfloat16, 'the MI355X kernels are bf16'
    assert q.is_cuda and q.size(-1) <= 256
    if causal or dropout_p != 0. or tuple(window_size) != (-1, -1):
        raise NotImplementedError('causal / dropout / window attention are never used by the reference DiT')
    b, lq, n, c = q.shape
    lk = k.size(1)
    assert k.size(2) == n and v.size(2) == n, 'grouped-query attention is not used by the reference DiT'
    out_dtype = q.dtype
    if q_scale is not None:
        q = q * q_scale
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(c)
    out = torch.zeros(b, lq, n, c, dtype=torch.bfloat16, device=q.device)
    for i in range(b):
        ql = lq if q_lens is None else int(q_lens[i])
        kl = lk if k_lens is None else int(k_lens[i])
        qi = q[i, :ql].to(torch.bfloat16).reshape(ql, n * c).contiguous()
        ki = k[i, :kl].to(torch.bfloat16).reshape(kl, n * c).contiguous()
        vi = v[i, :kl].to(torch.bfloat16).reshape(kl, n * c).contiguous()
        oi = torch.empty(ql, n * c, dtype=torch.bfloat16, device=q.device)
        if c == 128:
            n_pk = ops.packed_kv_numel(kl, n)
            kp = torch.empty(n_pk, dtype=torch.bfloat16, device=q.device)
            vp = torch.empty(n_pk, dtype=torch.bfloat16, device=q.device)
            ops.pack_kv(ki, vi, n, kp, vp)
            ops.attention_hd128(qi, kp, vp, oi, kl, n, scale)
        else:
            ops.attention_generic(qi, ki, vi, oi, kl, n, c, scale)
        out[i, :ql] = oi.view(ql, n, c)
    return out.type(out_dtype)


def attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None, causal=False,
              window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, fa_version=None):
    return flash_attention(q=q, k=k, v=v, q_lens=q_lens, k_lens=k_lens, dropout_p=dropout_p,
                           softmax_scale=softmax_scale, q_scale=q_scale, causal=causal, window_size=window_size,
                           deterministic=deterministic, dtype=dtype, version=fa_version)
