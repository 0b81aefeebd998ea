"""WanModel — the MoviiGen1.1 / Wan2.1 DiT, executed by hand-written HIP kernels on MI355X.

Drop-in for the reference class of the same name (reference wan/modules/model.py:361-633):
same constructor arguments, same state_dict key names and shapes (so `from_pretrained` reads the
reference checkpoint layout: config.json + diffusion_pytorch_model*.safetensors), same
`forward(x, t, context, seq_len, clip_fea=None, y=None) -> List[Tensor[C,F,H,W] fp32]`.

What differs is HOW a forward runs.  The reference composes ~40 torch ops per block under
autocast; here a block is 14 kernel launches from libmoviigen_hip.so:

    ln_modulate -> gemm(QKV fused, N=3*dim) -> rmsnorm_rope(q, x softmax_scale*log2e) / rmsnorm_rope(k) / pack_kv
    -> attention -> gemm(o) with `x += y*gate` fused  -> ln_modulate(norm3 affine) -> gemm(q)
    -> rmsnorm -> attention(512 cached text keys) -> gemm(o) with `x += y` fused
    -> ln_modulate -> gemm(ffn.0)+GELU fused -> gemm(ffn.2) with `x += y*gate` fused

The rounding points of the reference's autocast(bf16) execution are reproduced (SURVEY.md
Appendix B): bf16 GEMM operands/results, fp32 accumulate, fp32 residual stream, fp32 norms and
modulation, fp32 time embedding and head.  GEMM weights are stored in bf16 (what autocast feeds
the GEMM anyway); norm weights, modulation tables, time embedding and head stay fp32.
Per-prompt work that does not depend on the timestep (text_embedding, the 40 cross-attention
K/V projections) is computed once per prompt and cached — identical values, 100x fewer times.

There is no torch fallback: without the HIP library (or a GPU) forward() raises.
"""
import json
import math
import os

import numpy as np
import torch
import torch.nn as nn

from ..backend import ops
from .attention import flash_attention      # operator seam (1): the reference binds it here by name (model.py:10)

__all__ = ['WanModel']

_ENGINE_FLASH_ATTENTION = flash_attention


def _rebound_flash_attention():
    """the function a caller bound as `wan.modules.model.flash_attention` (the reference calls that module-level
    name at model.py:146-151 and :176, so assigning it replaces the attention of every block), or None while the name
    still is the engine's own operator — then the fused kernels on packed operands run instead of the wrapper."""
    fn = globals()['flash_attention']
    return None if fn is _ENGINE_FLASH_ATTENTION else fn

_BF16_SUFFIXES = ('.q.weight', '.k.weight', '.v.weight', '.o.weight', 'ffn.0.weight', 'ffn.2.weight',
                  'text_embedding.0.weight', 'text_embedding.2.weight', 'patch_embedding.weight')


class _Lin(nn.Module):
    """parameter holder with nn.Linear's names (weight [out,in], bias [out]); no forward."""

    def __init__(self, out_f, in_f, wdtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_f, in_f, dtype=wdtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_f, dtype=torch.float32, device=device), requires_grad=False)


class _Vec(nn.Module):
    def __init__(self, dim, device, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, dtype=torch.float32, device=device), requires_grad=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(dim, dtype=torch.float32, device=device), requires_grad=False)


class _Attn(nn.Module):
    """parameters of WanSelfAttention / WanT2VCrossAttention (reference model.py:102-181).  Inside WanModel.forward
    attention runs fused in the engine's layer loop; `forward` below is the same operator stand-alone, with the
    reference's call shape — the operator seam of text2video.py:97-100 (`types.MethodType(fn, block.self_attn)`)."""

    def __init__(self, dim, device, num_heads=None, eps=1e-6):
        super().__init__()
        for n in 'qkvo':
            setattr(self, n, _Lin(dim, dim, torch.bfloat16, device))
        self.norm_q = _Vec(dim, device)
        self.norm_k = _Vec(dim, device)
        self.dim, self.num_heads, self.eps = dim, num_heads, eps
        self._xchg = {}     # (group, Lloc, device) -> HeadExchange of the stand-alone sequence-parallel operator

    def forward(self, x, seq_lens, grid_sizes, freqs=None, sp=None):
        """WanSelfAttention.forward (model.py:127-156): x [B, L, C] -> [B, L, C] bf16.  `freqs` (the reference's
        complex table) is accepted and ignored: the rotation angles are rebuilt from grid_sizes (same formula).
        sp = (group, size, rank in the group, rank of the token shard): x is the rank's token shard, Ulysses exchange
        around the attention (usp_attn_forward, xdit_context_parallel.py:155-198)."""
        if sp == 'ring':
            raise NotImplementedError('the stand-alone self-attention operator does not rotate K/V blocks: under the ring / '
                                      'hybrid layouts call WanModel.forward (wan/distributed/ring.py)')
        d, N = self.dim, self.num_heads
        hd = d // N
        scale = 1.0 / math.sqrt(hd)
        bf = torch.bfloat16
        outs = []
        for b in range(x.shape[0]):
            h = x[b].to(bf).contiguous()
            L, dev = h.shape[0], h.device
            grid = tuple(int(v) for v in grid_sizes[b].tolist())
            rope = rope_cos_sin(hd, grid).to(dev)
            q, k, v = (torch.empty(L, d, dtype=bf, device=dev) for _ in range(3))
            for lin, dst in ((self.q, q), (self.k, k), (self.v, v)):
                ops.gemm(h, lin.weight, lin.bias, ops.BIAS_BF16, dst)
            P = sp[1] if sp else 1
            pos_rank = (sp[3] if len(sp) > 3 else sp[2]) if sp else 0
            qn, kn = torch.empty_like(q), torch.empty_like(k)
            # head_dim 128: softmax_scale * log2(e) folded into q before its one rounding to bf16, exactly as the fused
            # layer loop does (WanModel._q_scale): the stand-alone operator and the fused one give the same bits
            pre = hd == 128
            ops.rmsnorm_rope(q, self.norm_q.weight, self.eps, hd, qn, rope, grid, pos_rank * L,
                             out_scale=scale * ops.ATTN_LOG2E if pre else 1.0)
            ops.rmsnorm_rope(k, self.norm_k.weight, self.eps, hd, kn, rope, grid, pos_rank * L)
            a = torch.empty(L, d, dtype=bf, device=dev)

            def attend(qg, kg, vg, ag, heads, klen):
                if hd == 128:
                    n_pk = ops.packed_kv_numel(klen, heads)
                    kp, vp = torch.empty(n_pk, dtype=bf, device=dev), torch.empty(n_pk, dtype=bf, device=dev)
                    ops.pack_kv(kg[:klen], vg[:klen], heads, kp, vp)
                    ops.attention_hd128(qg, kp, vp, ag, klen, heads, scale, prescaled=pre)
                else:
                    ops.attention_generic(qg, kg, vg, ag, klen, heads, hd, scale)
            if P > 1:
                from ..distributed.ulysses import HeadExchange
                key = (sp[0], L, str(dev))
                ex = self._xchg.get(key)
                if ex is None:          # one live shape: the buffers and the stream are persistent, not per call
                    self._xchg = {key: HeadExchange(sp[0], P, N, hd, L, dev)}
                    ex = self._xchg[key]
                ex.run(qn, kn, v, a, lambda qg, kg, vg, ag, n: attend(qg, kg, vg, ag, n, kg.shape[0]))
            else:
                klen = min(L, int(seq_lens[b])) if seq_lens is not None else L
                attend(qn, kn, v, a, N, klen)
            o = torch.empty(L, d, dtype=bf, device=dev)
            ops.gemm(a, self.o.weight, self.o.bias, ops.BIAS_BF16, o)
            outs.append(o)
        return torch.stack(outs)


class _Block(nn.Module):
    def __init__(self, dim, ffn_dim, device, num_heads=None, eps=1e-6):
        super().__init__()
        self.self_attn = _Attn(dim, device, num_heads, eps)
        self.norm3 = _Vec(dim, device, bias=True)
        self.cross_attn = _Attn(dim, device, num_heads, eps)
        self.ffn = nn.ModuleDict({'0': _Lin(ffn_dim, dim, torch.bfloat16, device),
                                  '2': _Lin(dim, ffn_dim, torch.bfloat16, device)})
        self.modulation = nn.Parameter(torch.empty(1, 6, dim, dtype=torch.float32, device=device),
                                       requires_grad=False)


class _Head(nn.Module):
    def __init__(self, dim, out_f, device):
        super().__init__()
        self.head = _Lin(out_f, dim, torch.float32, device)
        self.modulation = nn.Parameter(torch.empty(1, 2, dim, dtype=torch.float32, device=device),
                                       requires_grad=False)


class _PatchEmbed(nn.Module):
    def __init__(self, dim, in_dim, patch, device):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(dim, in_dim, *patch, dtype=torch.bfloat16, device=device),
                                   requires_grad=False)
        self.bias = nn.Parameter(torch.empty(dim, dtype=torch.float32, device=device), requires_grad=False)


def _replaced_forward(attn):
    """the instance-level `forward` a caller installed on a self-attention module (types.MethodType), unless it is
    one of this engine's own entry points (usp_attn_forward only re-states what the fused loop does anyway)."""
    fn = attn.__dict__.get('forward')
    if fn is None:
        return None
    from ..distributed.xdit_context_parallel import usp_attn_forward
    if getattr(fn, '__func__', fn) in (usp_attn_forward, _Attn.forward):
        return None
    return fn


def rope_cos_sin(head_dim, grid, theta=10000.0):
    """(cos, sin) tables of reference model.py:28-36,473-478 for positions < (F, H, W), fp64 ->
    fp32, laid out [F][c0] ++ [H][c1] ++ [W][c1] as mg_rmsnorm_rope_bf16 expects."""
    c = head_dim // 2
    c1 = c // 3
    c0 = c - 2 * c1
    parts = []
    for n, cnt in zip(grid, (c0, c1, c1)):
        inv = 1.0 / np.power(theta, np.arange(0, 2 * cnt, 2, dtype=np.float64) / (2 * cnt))
        ang = np.outer(np.arange(n, dtype=np.float64), inv)
        parts.append(np.stack([np.cos(ang), np.sin(ang)], axis=-1).reshape(-1, 2))
    return torch.from_numpy(np.concatenate(parts).astype(np.float32))


class WanModel(nn.Module):
    ignore_for_config = ['patch_size', 'cross_attn_norm', 'qk_norm', 'text_dim', 'window_size']
    _no_split_modules = ['WanAttentionBlock']

    def __init__(self, model_type='t2v', patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192,
                 freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32, window_size=(-1, -1),
                 qk_norm=True, cross_attn_norm=True, eps=1e-6, device=None):
        super().__init__()
        if model_type != 't2v':
            raise NotImplementedError('only the t2v path of MoviiGen1.1 is implemented (reference configs '
                                      'register t2v-14B / t2i-14B only)')
        if tuple(window_size) != (-1, -1) or not qk_norm or not cross_attn_norm:
            raise NotImplementedError('window attention / qk_norm=False / cross_attn_norm=False are never used '
                                      'by the reference configs')
        assert dim % num_heads == 0 and (dim // num_heads) % 2 == 0
        self.model_type, self.patch_size, self.text_len = model_type, tuple(patch_size), text_len
        self.in_dim, self.dim, self.ffn_dim, self.freq_dim, self.text_dim = in_dim, dim, ffn_dim, freq_dim, text_dim
        self.out_dim, self.num_heads, self.num_layers, self.eps = out_dim, num_heads, num_layers, eps
        self.window_size, self.qk_norm, self.cross_attn_norm = window_size, qk_norm, cross_attn_norm
        self.config = dict(model_type=model_type, text_len=text_len, in_dim=in_dim, dim=dim, ffn_dim=ffn_dim,
                           freq_dim=freq_dim, out_dim=out_dim, num_heads=num_heads, num_layers=num_layers, eps=eps)
        dv = device
        self.patch_embedding = _PatchEmbed(dim, in_dim, self.patch_size, dv)
        self.text_embedding = nn.ModuleDict({'0': _Lin(dim, text_dim, torch.bfloat16, dv),
                                             '2': _Lin(dim, dim, torch.bfloat16, dv)})
        self.time_embedding = nn.ModuleDict({'0': _Lin(dim, freq_dim, torch.float32, dv),
                                             '2': _Lin(dim, dim, torch.float32, dv)})
        self.time_projection = nn.ModuleDict({'1': _Lin(6 * dim, dim, torch.float32, dv)})
        self.blocks = nn.ModuleList([_Block(dim, ffn_dim, dv, num_heads, eps) for _ in range(num_layers)])
        self.head = _Head(dim, math.prod(self.patch_size) * out_dim, dv)
        # sequence-parallel placement, installed by wan.distributed (Ulysses); 1 = single GPU
        self.sp_size, self.sp_rank, self.sp_group = 1, 0, None
        self.sp_force = False   # run the Ulysses collectives even on a 1-rank group (RCCL smoke test)
        # training-side SP forward (scripts/train/model/model_seq.py): the sequence is zero-padded to seq_len before it
        # is chunked over the ranks and padded keys are masked (k_lens); cross-attention is head-sharded
        self.sp_mask_padded_keys = False
        self.cross_attn_head_sharded = False
        self.ring = False       # sequence parallelism by ring attention instead of Ulysses (wan/distributed/ring.py)
        # hybrid layout (wan/distributed/ring.py: enable_hybrid_sp): Ulysses inside groups of `uly_size` consecutive
        # ranks, ring attention across the `ring_size` groups; None = derive from sp_size / ring
        self.uly_group, self.uly_size, self.ring_group, self.ring_size, self.ring_rank = None, None, None, None, 0
        self._packed = None
        self._ws = {}
        self._rope = {}
        self._ctx_cache = {}

    @property
    def freqs(self):
        """the reference's complex RoPE table [1024, head_dim/2] (model.py:473-478), built on demand for callers of
        the operator seam; the engine's kernels use rope_cos_sin() instead."""
        if getattr(self, '_freqs', None) is None:
            d = self.dim // self.num_heads

            def params(dim):
                ang = torch.outer(torch.arange(1024, dtype=torch.float64),
                                  1.0 / torch.pow(10000.0, torch.arange(0, dim, 2, dtype=torch.float64) / dim))
                return torch.polar(torch.ones_like(ang), ang)
            self._freqs = torch.cat([params(d - 4 * (d // 6)), params(2 * (d // 6)), params(2 * (d // 6))], dim=1)
        return self._freqs

    # ------------------------------------------------------------------------------------------
    # checkpoint layout (reference text2video.py:87)
    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, checkpoint_dir, device=None):
        from safetensors import safe_open
        with open(os.path.join(checkpoint_dir, 'config.json')) as f:
            cfg = json.load(f)
        keys = ('model_type', 'text_len', 'in_dim', 'dim', 'ffn_dim', 'freq_dim', 'text_dim', 'out_dim', 'num_heads',
                'num_layers', 'eps')
        kw = {k: cfg[k] for k in keys if k in cfg}
        if 'patch_size' in cfg:
            kw['patch_size'] = tuple(cfg['patch_size'])
        model = cls(**kw, device=device)
        idx = os.path.join(checkpoint_dir, 'diffusion_pytorch_model.safetensors.index.json')
        if os.path.exists(idx):
            with open(idx) as f:
                files = sorted(set(json.load(f)['weight_map'].values()))
        else:
            files = sorted(f for f in os.listdir(checkpoint_dir)
                           if f.startswith('diffusion_pytorch_model') and f.endswith('.safetensors'))
        if not files:
            raise FileNotFoundError(f'no diffusion_pytorch_model*.safetensors under {checkpoint_dir}')
        own = dict(model.named_parameters())
        seen = set()
        for fn in files:
            with safe_open(os.path.join(checkpoint_dir, fn), framework='pt', device='cpu') as sf:
                for k in sf.keys():
                    if k not in own:
                        raise KeyError(f'unexpected checkpoint tensor {k}')
                    own[k].data.copy_(sf.get_tensor(k).reshape(own[k].shape))
                    seen.add(k)
        missing = set(own) - seen
        if missing:
            raise KeyError(f'checkpoint is missing {sorted(missing)[:5]} ...')
        return model

    def load_state_dict(self, state_dict, strict=True, assign=False):
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        self._invalidate()
        return out

    def init_weights(self, seed=0, std=0.02):
        """seeded synthetic weights: N(0, std) GEMM weights, N(0,1)/sqrt(dim)
        modulation, unit norm weights; generated directly on the parameter's device."""
        g = torch.Generator(device=self.patch_embedding.weight.device)
        g.manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith('modulation'):
                p.data.copy_(torch.randn(p.shape, generator=g, device=p.device) / math.sqrt(self.dim))
            elif 'norm' in name and name.endswith('weight'):
                p.data.fill_(1.0)
            elif name.endswith('bias'):
                p.data.copy_(torch.randn(p.shape, generator=g, device=p.device) * std)
            else:
                p.data.copy_((torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32) * std))
        self._invalidate()
        return self

    def _invalidate(self):
        self._packed = None
        self._ctx_cache = {}

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        # keep the storage dtypes fixed: `.to(torch.float32)`-style casts must not widen bf16 weights
        for name, p in self.named_parameters():
            want = torch.bfloat16 if name.endswith(_BF16_SUFFIXES) else torch.float32
            if p.dtype != want:
                p.data = p.data.to(want)
        self._invalidate()
        self._ws, self._rope = {}, {}
        return out

    # ------------------------------------------------------------------------------------------
    # engine-side packing: fused QKV / cross-KV weights, stacked modulation
    # ------------------------------------------------------------------------------------------
    def _pack(self):
        if self._packed is not None:
            return self._packed
        dev = self.patch_embedding.weight.device
        if dev.type != 'cuda':
            raise RuntimeError('WanModel.forward needs the model on a HIP device (model.to("cuda")): the hot '
                               'path has no CPU implementation — use oracle/ for CPU reference numbers')
        pk = {'layers': []}
        sharded = getattr(self, '_shards', None) is not None
        for b in self.blocks:
            sa, ca = b.self_attn, b.cross_attn
            lw = dict(bqkv=torch.cat([sa.q.bias, sa.k.bias, sa.v.bias]).contiguous(),
                      bkv_c=torch.cat([ca.k.bias, ca.v.bias]).contiguous())
            if not sharded:
                wqkv = torch.cat([sa.q.weight, sa.k.weight, sa.v.weight], 0).contiguous()
                # re-point the three parameters at the fused storage: no second copy of the weights
                d = self.dim
                sa.q.weight.data, sa.k.weight.data, sa.v.weight.data = wqkv[:d], wqkv[d:2 * d], wqkv[2 * d:]
                wkv = torch.cat([ca.k.weight, ca.v.weight], 0).contiguous()
                ca.k.weight.data, ca.v.weight.data = wkv[:d], wkv[d:]
                lw.update({'wqkv': wqkv, 'wkv_c': wkv, 'self_attn.o': sa.o.weight, 'cross_attn.q': ca.q.weight,
                           'cross_attn.o': ca.o.weight, 'ffn.0': b.ffn['0'].weight, 'ffn.2': b.ffn['2'].weight})
            pk['layers'].append(lw)
        pk['modulation'] = torch.cat([b.modulation.data.reshape(6, self.dim) for b in self.blocks], 0).contiguous()
        pk['patch_w'] = self.patch_embedding.weight.data.reshape(self.dim, -1)
        self._packed = pk
        return pk

    def _layer(self, i):
        """GEMM operands of block i: resident, or (block-sharded mode, wan.distributed.fsdp) views of
        the all-gather buffer, with block i+1's gather already in flight on the comm stream."""
        lw = self._pack()['layers'][i]
        sh = getattr(self, '_shards', None)
        if sh is None:
            return lw
        return {**lw, **sh.fetch(i)}

    def _sp_layout(self):
        """(U, R): Ulysses degree inside a group, ring degree across groups; U * R == sp_size."""
        if self.uly_size is not None:
            return self.uly_size, self.ring_size
        return (1, self.sp_size) if self.ring else (self.sp_size, 1)

    def _workspace(self, L, dev):
        key = (L, str(dev), self.sp_force, self.sp_size, self._sp_layout())
        ws = self._ws.get(key)
        if ws is None:
            d, f = self.dim, self.ffn_dim
            bf, f32 = torch.bfloat16, torch.float32
            hd = d // self.num_heads
            Ltot = L * self.sp_size
            e = lambda *s, dt=bf: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
            ws = dict(x=e(L, d, dt=f32), h=e(L, d), qkv=e(L, 3 * d), q=e(L, d), k=e(L, d), a=e(L, d),
                      u=e(L, f), tok=e(L, self.in_dim * math.prod(self.patch_size)),
                      hf=e(L, d, dt=f32), y=e(L, math.prod(self.patch_size) * self.out_dim, dt=f32),
                      sin=e(1, self.freq_dim, dt=f32), e1=e(d, dt=f32), e=e(d, dt=f32), e0=e(6, d, dt=f32),
                      mod=e(6 * self.num_layers, d, dt=f32), hmod=e(2, d, dt=f32))
            U, R = self._sp_layout()
            n_loc, Lg = self.num_heads // U, L * U          # heads and tokens a rank attends after the Ulysses exchange
            n_att = n_loc
            if U > 1 or self.sp_force:     # pipelined packed exchange (wan/distributed/ulysses.py): buffers live there
                from ..distributed.ulysses import HeadExchange
                ug = self.uly_group if self.uly_group is not None else self.sp_group
                ws['xchg'] = HeadExchange(ug, U, self.num_heads, hd, L, dev, max_groups=1 if R > 1 else None)
                n_att = max(n for _, n in ws['xchg'].groups)
            if hd == 128 and R == 1:   # K / V packed into 64-key tiles (operand layout of the MFMA attention kernel)
                n_pk = ops.packed_kv_numel(Lg, n_att)
                ws['kp'], ws['vp'] = e(n_pk), e(n_pk)
            if R > 1:                  # ring attention: two packed K/V blocks in flight, fp32 running result
                n1 = ops.packed_kv_numel(Lg, n_loc)
                ws['kp0'], ws['vp0'], ws['kp1'], ws['vp1'] = e(n1), e(n1), e(n1), e(n1)
                ws['part'], ws['acc'] = e(Lg, n_loc * hd), e(Lg, n_loc * hd, dt=f32)
                ws['lse'], ws['lse_acc'] = e(n_loc, Lg, dt=f32), e(n_loc, Lg, dt=f32)
            self._ws = {key: ws}  # one live shape at a time (activations are GBs at 14B/720p)
        return ws

    def _rope_tab(self, grid, dev):
        key = (tuple(grid), str(dev))
        if key not in self._rope:
            self._rope = {key: rope_cos_sin(self.dim // self.num_heads, grid).to(dev)}
        return self._rope[key]

    # ------------------------------------------------------------------------------------------
    # prompt-only work: text_embedding + per-layer cross-attention K/V (reference model.py:548-554,
    # 168-170) — independent of t, so done once per prompt tensor and cached
    # ------------------------------------------------------------------------------------------
    def _context(self, ctx):
        """-> (ctx, emb [text_len, dim] bf16, layers): `layers[i]` is block i's cross-attention (K, V) — filled by
        _cross_kv the first time block i runs for this prompt, i.e. while that block's weights are at hand anyway
        (block-sharded mode: no extra all-gather sweep over the 28 GB of weights per new prompt)."""
        key = (ctx.data_ptr(), tuple(ctx.shape), ctx._version)
        hit = self._ctx_cache.get(key)
        if hit is not None:
            return hit
        dev, d = ctx.device, self.dim
        Lc = self.text_len
        bf = torch.bfloat16
        pad = torch.zeros(Lc, self.text_dim, dtype=bf, device=dev)
        pad[:ctx.shape[0]].copy_(ctx)
        t0 = torch.empty(Lc, d, dtype=bf, device=dev)
        emb = torch.empty(Lc, d, dtype=bf, device=dev)
        te = self.text_embedding
        ops.gemm(pad, te['0'].weight, te['0'].bias, ops.BIAS_GELU_BF16, t0)
        ops.gemm(t0, te['2'].weight, te['2'].bias, ops.BIAS_BF16, emb)
        if len(self._ctx_cache) >= 4:
            self._ctx_cache.pop(next(iter(self._ctx_cache)))
        self._ctx_cache[key] = (ctx, emb, [None] * self.num_layers)  # keep ctx alive so data_ptr stays unique
        return self._ctx_cache[key]

    def _cross_kv(self, i, lw, emb, row_major=False):
        """cross-attention K (RMS-normed) and V of block i for one prompt (reference model.py:168-170): packed tile
        buffers for the head_dim 128 kernel, row-major [text_len, dim] otherwise (and for a rebound flash_attention)."""
        dev, d, hd = emb.device, self.dim, self.dim // self.num_heads
        Lc, bf = self.text_len, torch.bfloat16
        kv = torch.empty(Lc, 2 * d, dtype=bf, device=dev)
        ops.gemm(emb, lw['wkv_c'], lw['bkv_c'], ops.BIAS_BF16, kv)
        kc = torch.empty(Lc, d, dtype=bf, device=dev)
        ops.rmsnorm_rope(kv[:, :d], self.blocks[i].cross_attn.norm_k.weight, self.eps, hd, kc)
        if hd == 128 and not row_major:
            n_pk = ops.packed_kv_numel(Lc, self.num_heads)
            kcp, vcp = torch.empty(n_pk, dtype=bf, device=dev), torch.empty(n_pk, dtype=bf, device=dev)
            ops.pack_kv(kc, kv[:, d:], self.num_heads, kcp, vcp)
            return kcp, vcp
        return kc, kv[:, d:].clone()

    # ------------------------------------------------------------------------------------------
    def _q_scale(self):
        """factor the fused forward folds into q when it is RMS-normed (ops.rmsnorm_rope out_scale): the attention's
        softmax_scale * log2(e) for the head_dim 128 kernel — q is rounded to bf16 once either way, and the kernel then
        needs no per-score multiply (flash_attn applies softmax_scale to the fp32 scores: same product) — else 1."""
        hd = self.dim // self.num_heads
        return ops.ATTN_LOG2E / math.sqrt(hd) if hd == 128 else 1.0

    def _attention(self, q, k, v, out, lk, heads):
        """head_dim 128: q PRE-SCALED (_q_scale), k, v PACKED tile buffers (ops.pack_kv); otherwise row-major."""
        hd = self.dim // self.num_heads
        scale = 1.0 / math.sqrt(hd)
        if hd == 128:
            ops.attention_hd128(q, k, v, out, lk, heads, scale, prescaled=True)
        else:
            ops.attention_generic(q, k, v, out, lk, heads, hd, scale)

    def _self_attention(self, ws, blk, grid, rope, L, pos0):
        """model.py:127-156 (and the Ulysses variant xdit_context_parallel.py:155-198)."""
        d, hd, N = self.dim, self.dim // self.num_heads, self.num_heads
        qkv = ws['qkv']
        sa = blk.self_attn
        fa = _rebound_flash_attention() if self.sp_size == 1 and not self.sp_force else None
        ops.rmsnorm_rope(qkv[:, :d], sa.norm_q.weight, self.eps, hd, ws['q'], rope, grid, pos0,
                         out_scale=self._q_scale() if fa is None else 1.0)
        ops.rmsnorm_rope(qkv[:, d:2 * d], sa.norm_k.weight, self.eps, hd, ws['k'], rope, grid, pos0)
        if fa is not None:
            # operator seam (1): the caller's function gets the reference's call (model.py:146-151): q, k roped
            # [1, L, N, hd] (UNSCALED: softmax_scale is the callee's business), v, k_lens, window_size
            y = fa(q=ws['q'].view(1, L, N, hd), k=ws['k'].view(1, L, N, hd), v=qkv[:, 2 * d:].reshape(1, L, N, hd),
                   k_lens=torch.tensor([self._kv_valid]), window_size=self.window_size)
            ws['a'].copy_(y.reshape(L, d))
            return
        if self.sp_size == 1 and not self.sp_force:
            if hd == 128:
                kv = self._kv_valid        # rows past the video's tokens are padding: never packed, never attended
                ops.pack_kv(ws['k'][:kv], qkv[:kv, 2 * d:], N, ws['kp'], ws['vp'])
                self._attention(ws['q'], ws['kp'], ws['vp'], ws['a'], kv, N)
            else:
                self._attention(ws['q'], ws['k'], qkv[:, 2 * d:], ws['a'], self._kv_valid, N)
            return
        U, R = self._sp_layout()
        rg = self.ring_group if self.ring_group is not None else self.sp_group
        rr = self.ring_rank if self.uly_size is not None else self.sp_rank
        scale = 1.0 / math.sqrt(hd)
        q, k, v, a = ws['q'], ws['k'], qkv[:, 2 * d:], ws['a']

        def attend(qg, kg, vg, ag, heads):      # operands may be strided column views; ag contiguous
            if R > 1:                           # ring attention across the groups (hd 128 only)
                from ..distributed.ring import ring_attention
                ring_attention(qg, kg, vg, ag, ws, rg, R, rr, heads, scale, prescaled=True)
            elif hd == 128:
                kv = min(kg.shape[0], self._kv_valid)       # keys past the video's tokens are padding (k_lens)
                ops.pack_kv(kg[:kv], vg[:kv], heads, ws['kp'], ws['vp'])
                # while an exchange is in flight the persistent attention grid leaves `reserve_cus` CUs to RCCL's kernels
                ops.attention_hd128(qg, ws['kp'], ws['vp'], ag, kv, heads, scale, prescaled=True,
                                    reserve_cus=ws['xchg'].reserve_cus if 'xchg' in ws else 0)
            else:
                ops.attention_generic(qg, kg, vg, ag, min(kg.shape[0], self._kv_valid), heads, hd, scale)

        if U > 1 or self.sp_force:
            # Ulysses: packed q|k|v exchange per head group on the comm stream, pipelined against the attention
            # launches (xdit_context_parallel.py:185-190 / model_seq.py:232-256)
            ws['xchg'].run(q, k, v, a, attend)
        else:                                   # pure ring: every rank keeps all heads
            attend(q, k, v, a, N)

    def _cross_attention_head_sharded(self, ws, kc, vc):
        """model_seq.py:271-294: q through the seq->head all-to-all, K/V narrowed to this rank's heads (shrink_head),
        attention over ALL tokens x local heads, head->seq all-to-all back.  Same values as the token-local form."""
        from ..distributed import ulysses
        U = self._sp_layout()[0]
        ug = self.uly_group if self.uly_group is not None else self.sp_group
        ur = self.sp_rank % U
        N, hd = self.num_heads, self.dim // self.num_heads
        nl, L = N // U, ws['k'].shape[0]
        qg = torch.empty(L * U, nl * hd, dtype=torch.bfloat16, device=ws['k'].device)
        ulysses.seq_to_head(ws['k'], qg, ug, U, N, hd)
        ag = torch.empty_like(qg)
        if hd == 128:      # packed K/V are head-major: this rank's heads are one contiguous range of tiles
            per_head = kc.numel() // N
            ks, vs = kc[ur * nl * per_head:(ur + 1) * nl * per_head], vc[ur * nl * per_head:(ur + 1) * nl * per_head]
        else:
            ks, vs = kc[:, ur * nl * hd:(ur + 1) * nl * hd], vc[:, ur * nl * hd:(ur + 1) * nl * hd]
        self._attention(qg, ks, vs, ag, self.text_len, nl)
        ulysses.head_to_seq(ag, ws['a'], ug, U, N, hd)

    @torch.no_grad()
    def _forward_one(self, lat, t, ctx, seq_len, share=None):
        """share: None, or the two halves of forward_pair — 'save' keeps the residual stream as it stands behind block 0's self-attention
        (ws['x0']), 'reuse' starts from that copy instead of recomputing it (same latent, same t: nothing before that point reads `ctx`)."""
        pk = self._pack()
        dev = lat.device
        lat = lat.to(torch.float32).contiguous()
        C, F, H, W = lat.shape
        pt, ph, pw = self.patch_size
        assert pt == 1, 'temporal patch size 1 (reference config)'
        grid = (F, H // ph, W // pw)
        Lfull = grid[0] * grid[1] * grid[2]
        assert Lfull <= seq_len, 'seq_lens.max() <= seq_len (reference model.py:534)'
        P = self.sp_size
        Ltot = Lfull
        if P > 1 and self.sp_mask_padded_keys:
            # model_seq.py:704-706,757: pad to seq_len, chunk; padded keys masked through k_lens (:247-252)
            assert seq_len % P == 0 and self.num_heads % self._sp_layout()[0] == 0 and self._sp_layout()[1] == 1, \
                'training-side sequence parallel needs seq_len % sp == 0, heads % sp == 0 (Ulysses only)'
            Ltot = seq_len
        elif P > 1:
            # reference SP path does not mask padded keys (xdit_context_parallel.py:178-193): it is
            # only correct without padding, which is what every supported size gives
            assert seq_len == Lfull and Lfull % P == 0 and self.num_heads % self._sp_layout()[0] == 0, \
                'sequence parallel needs L % sp == 0, heads % sp == 0 and no padding'
        L = Ltot // P
        pos0 = self.sp_rank * L
        n_valid = min(max(Lfull - pos0, 0), L)          # video tokens among this rank's rows (all of them unless padded)
        self._kv_valid = Lfull
        d, eps = self.dim, self.eps
        ws = self._workspace(L, dev)
        x = ws['x']

        reuse = share == 'reuse'
        if share == 'save' and 'x0' not in ws:
            ws['x0'] = torch.empty_like(x)
        assert not reuse or 'x0' in ws, "share='reuse' follows a share='save' forward of the same latent and t"

        # patch embedding (model.py:529-531): bf16 result, residual stream kept in fp32 storage
        if reuse:
            pass                                         # x comes back whole behind block 0's self-attention
        elif P == 1:
            ops.patchify(lat, ph, pw, ws['tok'])
        else:
            full = torch.empty(Lfull, ws['tok'].shape[1], dtype=torch.bfloat16, device=dev)
            ops.patchify(lat, ph, pw, full)
            ws['tok'][:n_valid].copy_(full[pos0:pos0 + n_valid])  # torch.chunk(x, P, dim=1)[rank]
        if n_valid and not reuse:
            ops.gemm(ws['tok'][:n_valid], pk['patch_w'], self.patch_embedding.bias, ops.BIAS_F32, x[:n_valid])
        if n_valid < L and not reuse:
            x[n_valid:].zero_()                          # rows padded AFTER the patch embedding (no bias), :704-706

        # time embedding (model.py:541-545), fp32
        tt = t.reshape(1).to(dev)
        if tt.dtype not in (torch.int64, torch.float32, torch.float64):
            tt = tt.to(torch.float32)
        ops.sinusoid_embed(tt, self.freq_dim, ws['sin'])
        te, tp = self.time_embedding, self.time_projection['1']
        ops.gemv(te['0'].weight, te['0'].bias, ws['sin'], ws['e1'])
        ops.gemv(te['2'].weight, te['2'].bias, ws['e1'], ws['e'], silu_in=True)
        ops.gemv(tp.weight, tp.bias, ws['e'], ws['e0'], silu_in=True)
        ops.add_rows(pk['modulation'], ws['e0'], ws['mod'], 6)                       # model.py:292-295
        ops.add_rows(self.head.modulation.data.reshape(2, d), ws['e'].reshape(1, d), ws['hmod'], 1)

        _, ctx_emb, ctx_layers = self._context(ctx)
        rope = self._rope_tab(grid, dev)
        mod = ws['mod']

        for i, blk in enumerate(self.blocks):
            lw = self._layer(i)
            m = mod[6 * i:6 * i + 6]
            # self attention
            if i == 0 and reuse:
                x.copy_(ws['x0'])                        # block 0 up to here saw the latent and t only: the 'save' forward's stream, bit for bit
            else:
                ops.ln_modulate(x, m[1], m[0], True, eps, ws['h'], round_norm_bf16=(i == 0))
                custom = _replaced_forward(blk.self_attn)
                if custom is None:
                    ops.gemm(ws['h'], lw['wqkv'], lw['bqkv'], ops.BIAS_BF16, ws['qkv'])
                    self._self_attention(ws, blk, grid, rope, L, pos0)
                    ops.gemm(ws['a'], lw['self_attn.o'], blk.self_attn.o.bias, ops.GATE_RESID_F32, x, gate=m[2])
                else:
                    # operator seam (2) of the reference (text2video.py:97-100): the caller replaced
                    # block.self_attn.forward — call it with the reference's arguments and keep the fused rest
                    y = custom(ws['h'][None], torch.tensor([Lfull]), torch.tensor([list(grid)]), self.freqs)
                    ops.gate_residual(x, y[0].to(torch.bfloat16).contiguous(), m[2])
                if i == 0 and share == 'save':
                    ws['x0'].copy_(x)
            # cross attention (text keys/values cached per prompt)
            ca = blk.cross_attn
            fa = _rebound_flash_attention()
            if ctx_layers[i] is None or (len(ctx_layers[i]) == 3) != (fa is not None):  # first forward with this prompt
                ctx_layers[i] = self._cross_kv(i, lw, ctx_emb) if fa is None else \
                    self._cross_kv(i, lw, ctx_emb, row_major=True) + ('row-major',)
            kc, vc = ctx_layers[i][:2]
            ops.ln_modulate(x, blk.norm3.weight, blk.norm3.bias, False, eps, ws['h'])
            ops.gemm(ws['h'], lw['cross_attn.q'], ca.q.bias, ops.BIAS_BF16, ws['q'])
            ops.rmsnorm_rope(ws['q'], ca.norm_q.weight, eps, d // self.num_heads, ws['k'],
                             out_scale=self._q_scale() if fa is None else 1.0)
            if fa is not None:
                # operator seam (1), reference model.py:176: flash_attention(q, k, v, k_lens=context_lens) with
                # context_lens = None for T2V
                N, hd = self.num_heads, d // self.num_heads
                y = fa(ws['k'].view(1, L, N, hd), kc.view(1, -1, N, hd), vc.reshape(1, -1, N, hd), k_lens=None)
                ws['a'].copy_(y.reshape(L, d))
            elif self.cross_attn_head_sharded and P > 1:
                self._cross_attention_head_sharded(ws, kc, vc)
            else:
                self._attention(ws['k'], kc, vc, ws['a'], self.text_len, self.num_heads)
            ops.gemm(ws['a'], lw['cross_attn.o'], ca.o.bias, ops.GATE_RESID_F32, x, gate=None)
            # ffn
            ops.ln_modulate(x, m[4], m[3], True, eps, ws['h'])
            ops.gemm(ws['h'], lw['ffn.0'], blk.ffn['0'].bias, ops.BIAS_GELU_BF16, ws['u'])
            ops.gemm(ws['u'], lw['ffn.2'], blk.ffn['2'].bias, ops.GATE_RESID_F32, x, gate=m[5])

        # head (model.py:333-343): fp32 end to end
        ops.ln_modulate(x, ws['hmod'][1], ws['hmod'][0], True, eps, ws['hf'])
        ops.head_gemm(ws['hf'], self.head.head.weight, self.head.head.bias, ws['y'])
        y = ws['y']
        if P > 1:
            from ..distributed import ulysses
            y = ulysses.all_gather_seq(ws['y'], self.sp_group, P)                   # get_sp_group().all_gather
        out = torch.empty(self.out_dim, F, H, W, dtype=torch.float32, device=dev)
        ops.unpatchify(y, self.out_dim, grid[0], grid[1], grid[2], ph, pw, out)
        return out

    def forward(self, x, t, context, seq_len, clip_fea=None, y=None):
        if clip_fea is not None or y is not None:
            raise NotImplementedError('image conditioning (i2v) is not part of MoviiGen1.1 T2V')
        t = t.reshape(-1)
        outs = [self._forward_one(u, t[i if t.numel() > 1 else 0], c, seq_len)
                for i, (u, c) in enumerate(zip(x, context))]
        if self._peer_transport_failed():
            # a copy of the copy-engine transport was refused on some rank during this forward (every rank reads the same answer): the
            # whole group goes back to the all-to-all collective, for good, and the forward is repeated on it
            outs = [self._forward_one(u, t[i if t.numel() > 1 else 0], c, seq_len)
                    for i, (u, c) in enumerate(zip(x, context))]
        return outs

    def forward_pair(self, x, t, context, context_null, seq_len):
        """The two guidance branches of ONE denoising step (reference text2video.py:237-240: two calls of the model on the same latent and t,
        with the prompt's and the negative prompt's embeddings) -> (cond, uncond), each what forward() returns for its context, bit for bit.
        Nothing in front of block 0's cross-attention reads the context — patch embedding, time embedding, and block 0's modulated LayerNorm,
        q|k|v projection, RMS-norm + RoPE, self-attention over all L tokens and its gated residual see the latent and t only — so the second
        branch starts from a copy of the first one's residual stream at that point (2.7 GB at 1920x832x81f) instead of computing it again:
        one self-attention launch and four GEMMs of 80 per step.  A replaced self-attention forward or a rebound flash_attention (the operator
        seams) may be anything, also non-deterministic: then, and with MOVIIGEN_CFG_SHARED_PREFIX=0, the branches run as two plain forwards."""
        plain = os.environ.get('MOVIIGEN_CFG_SHARED_PREFIX', '1') == '0' or _rebound_flash_attention() is not None or \
            _replaced_forward(self.blocks[0].self_attn) is not None or len(x) != 1 or len(context) != 1 or len(context_null) != 1
        if plain:
            return self.forward(x, t, context, seq_len), self.forward(x, t, context_null, seq_len)
        t = t.reshape(-1)
        for _ in range(2):
            cond = self._forward_one(x[0], t[0], context[0], seq_len, share='save')
            uncond = self._forward_one(x[0], t[0], context_null[0], seq_len, share='reuse')
            if not self._peer_transport_failed():       # (see forward: a refused peer copy sends the whole group back to the collective, once)
                break
        return [cond], [uncond]

    def _peer_transport_failed(self):
        """once per forward (one 4-byte read per open window set — nothing at all on the default collective transport): did a peer copy
        fail?  If so every HeadExchange of this model drops its windows (wan/distributed/peer_copy.py)."""
        xs = [w['xchg'] for w in self._ws.values() if 'xchg' in w]
        xs = [x for x in xs if x.peer is not None]
        if not xs or not any(x.peer_failed() for x in xs):
            return False
        for x in xs:
            x.drop_peer()
        return True

    def unpatchify(self, x, grid_sizes):
        outs = []
        for u, v in zip(x, grid_sizes.tolist()):
            f, h, w = v
            out = torch.empty(self.out_dim, f * self.patch_size[0], h * self.patch_size[1], w * self.patch_size[2],
                              dtype=torch.float32, device=u.device)
            ops.unpatchify(u.to(torch.float32).contiguous(), self.out_dim, f, h, w, self.patch_size[1],
                           self.patch_size[2], out)
            outs.append(out)
        return outs
