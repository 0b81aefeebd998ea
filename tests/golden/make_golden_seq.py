"""Generate tests/golden/g8_train_seq.npz by IMPORTING the reference's training-side sequence-parallel
DiT (scripts/train/model/model_seq.py) — SURVEY.md §8(f) rank 4, second half.

    python tests/golden/make_golden_seq.py

The module imports un-vendored libraries (fastvideo, xfuser) that only move data.  They are replaced by
an in-process simulation: the P sequence-parallel ranks run as P Python threads in lock step, and
`all_to_all_4D` / `all_gather` exchange their operands through a barrier-protected slot list with the
semantics of SURVEY.md Appendix C (all_to_all_4D(scatter_dim=2, gather_dim=1): [B,L/P,N,D] ->
[B,L,N/P,D]; (scatter_dim=1, gather_dim=2) the inverse; all_gather(dim=1): rank-order concat).
`nccl_info.sp_size / rank_within_group` and `dist.get_rank()` are per-thread.  flash_attention and the
autocast mapping are the same stand-ins as in make_golden.py.  A fixture is data: inputs + the reference's
outputs; weights are regenerated from seeds by tests/golden/weights.py."""
import importlib.util
import os
import sys
import threading
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
import weights as W  # noqa: E402

REF = '/root/reference'


class Sim:
    """P ranks = P threads; exchange() hands every rank the list of all ranks' operands."""

    def __init__(self, P):
        self.P, self.tls = P, threading.local()
        self.bar = threading.Barrier(P)
        self.slots = [None] * P
        self.enabled = True

    @property
    def rank(self):
        return getattr(self.tls, 'rank', 0)

    def exchange(self, x):
        self.slots[self.rank] = x
        self.bar.wait()
        out = list(self.slots)
        self.bar.wait()
        return out


SIM = Sim(1)


def all_to_all_4D(x, scatter_dim=2, gather_dim=1):
    parts, r, P = SIM.exchange(x), SIM.rank, SIM.P
    if scatter_dim == 2 and gather_dim == 1:          # [B, L/P, N, D] -> [B, L, N/P, D]
        nl = x.shape[2] // P
        return torch.cat([p[:, :, r * nl:(r + 1) * nl] for p in parts], dim=1).contiguous()
    if scatter_dim == 1 and gather_dim == 2:          # [B, L, N/P, D] -> [B, L/P, N, D]
        ll = x.shape[1] // P
        return torch.cat([p[:, r * ll:(r + 1) * ll] for p in parts], dim=2).contiguous()
    raise NotImplementedError


def all_gather(x, dim=1):
    return torch.cat(SIM.exchange(x), dim=dim)


class _NcclInfo:
    @property
    def sp_size(self):
        return SIM.P

    @property
    def rank_within_group(self):
        return SIM.rank


class _Dist:
    @staticmethod
    def get_rank():
        return SIM.rank

    @staticmethod
    def barrier():
        return None


def load_model_seq():
    G.install_shims()

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod('fastvideo')
    mod('fastvideo.utils')
    mod('fastvideo.utils.communications', all_gather=all_gather, all_to_all_4D=all_to_all_4D)
    mod('fastvideo.utils.parallel_states', get_sequence_parallel_state=lambda: SIM.enabled and SIM.P > 1,
        nccl_info=_NcclInfo())
    mod('xfuser')
    mod('xfuser.core')
    mod('xfuser.core.distributed', get_sequence_parallel_rank=lambda: SIM.rank,
        get_sequence_parallel_world_size=lambda: SIM.P, get_sp_group=lambda: None)
    mod('xfuser.core.long_ctx_attention', xFuserLongContextAttention=object)
    spec = importlib.util.spec_from_file_location('ref_model_seq', os.path.join(REF, 'scripts/train/model/model_seq.py'))
    ms = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ms)
    ms.flash_attention = G.sdpa_flash_attention
    ms.dist = _Dist
    return ms


def run_ranks(P, fn):
    """fn(rank) on P lock-step threads; returns the per-rank results."""
    global SIM
    SIM.P, SIM.bar, SIM.slots = P, threading.Barrier(P), [None] * P
    out, err = [None] * P, []

    def work(r):
        SIM.tls.rank = r
        try:
            with torch.no_grad():
                out[r] = fn(r)
        except Exception as e:  # noqa: BLE001
            err.append(e)
            SIM.bar.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if err:
        raise err[0]
    return out


@torch.no_grad()
def main():
    ms = load_model_seq()
    cfg = W.SMALL_DIT_HD128                        # 2 heads x 128: one head per rank at P = 2
    P = W.make_dit_params(cfg, 0)
    m = G.build_ref_dit(ms, cfg, P)
    lat = W.randn((16, 2, 8, 12), 20)              # grid (2, 4, 6) = 48 tokens
    ctx = torch.zeros(1, cfg['text_len'], cfg['text_dim'])
    ctx[0, :33] = W.randn((33, cfg['text_dim']), 30)        # batch_context: already padded to text_len
    t = torch.tensor([650])
    arrs = dict(lat=lat, batch_context=ctx, t=t)
    amp_real = importlib.import_module('torch.amp')
    for tag, seq_len in (('nopad', 48), ('pad', 56)):      # 56: rank 1 holds 20 video tokens + 8 padded rows
        def fwd32(rank):
            return m([lat], t=t, context=None, seq_len=seq_len, batch_context=ctx)[0]

        def fwdbf(rank):
            with torch.autocast('cpu', dtype=torch.bfloat16):
                return m([lat], t=t, context=None, seq_len=seq_len, batch_context=ctx)[0]
        ms.amp = amp_real
        single = run_ranks(1, fwd32)[0]
        sp32 = run_ranks(2, fwd32)
        assert torch.equal(sp32[0], sp32[1])
        assert (sp32[0] - single).abs().max().item() < 1e-4, (sp32[0] - single).abs().max().item()
        ms.amp = G.AmpCpu
        spbf = run_ranks(2, fwdbf)
        assert torch.equal(spbf[0], spbf[1])
        ms.amp = amp_real
        arrs.update({f'seq_len_{tag}': seq_len, f'single_fp32_{tag}': single, f'sp2_fp32_{tag}': sp32[0],
                     f'sp2_bf16_{tag}': spbf[0].float()})
        print(tag, 'sp2 vs single fp32 max-abs', (sp32[0] - single).abs().max().item(),
              ' bf16 vs fp32 rel-L2', ((spbf[0].float() - single).norm() / single.norm()).item())
    # the pieces the SP forward adds: rope_apply_dist on a rank slice (with padded rows), head-sharded cross-attention
    xq = W.randn((1, 28, 2, 128), 81)
    grid = torch.tensor([[2, 4, 6]])

    def rope_rank(rank):
        return ms.rope_apply_dist(xq, grid, m.freqs)
    rr = run_ranks(2, rope_rank)
    arrs.update(rope_dist_x=xq, rope_dist_grid=grid, rope_dist_rank0=rr[0], rope_dist_rank1=rr[1])
    G.save('g8_train_seq', **arrs)


if __name__ == '__main__':
    main()
