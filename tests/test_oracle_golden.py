"""The oracle (oracle/*.py, our CPU restatement) against golden vectors produced by the imported
reference (tests/golden/make_golden.py).  CPU only.  Tolerances: fp32 paths <= 1e-5 relative to
the tensor scale; bf16-emulated paths 2e-2 rel-L2 (the stated bf16 tolerance of the north star)."""
import numpy as np
import pytest
import torch

import weights as W
from oracle import dit, schedulers, vae


def T(a):
    return torch.from_numpy(np.asarray(a))


def maxerr(a, b):
    a, b = T(a).double(), T(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def rel_l2(a, b):
    a, b = T(a).double(), T(b).double()
    return ((a - b).norm() / b.norm()).item()


def test_primitives(golden):
    g = golden('g1_primitives')
    x = T(g['x'])[0]
    assert maxerr(dit.rmsnorm(x, T(g['rms_w']), 1e-6, False), g['rmsnorm'][0]) < 1e-6
    assert maxerr(dit.layernorm(x, 1e-6), g['layernorm'][0]) < 1e-6
    assert maxerr(dit.sinusoid(64, T(g['sinus_t'])), g['sinus']) < 1e-12
    tabs = dit.rope_table(32)
    out = dit.rope(T(g['rope_x'])[0], tuple(int(v) for v in g['rope_grid'][0]), tabs)
    assert maxerr(out, g['rope'][0]) < 1e-6
    f, h, w = (int(v) for v in g['rope_grid'][0])
    up = dit.unpatchify(T(g['unpatch_in'])[0], (f, h, w), (1, 2, 2), 16)
    assert maxerr(up, g['unpatch']) == 0.0


@pytest.mark.parametrize('tag,cfg', [('tiny', W.TINY_DIT), ('tiny_pad', W.TINY_DIT), ('hd128', W.SMALL_DIT_HD128)])
def test_dit_forward(golden, tag, cfg):
    g = golden(f'g3_dit_{tag}')
    P = W.make_dit_params(cfg, 0)
    j = 0
    while f'ctx{j}' in g:
        args = (P, cfg, T(g['lat']), T(g[f't{j}']), T(g[f'ctx{j}']), int(g['seq_len']))
        out32 = dit.dit_forward(*args, emulate_bf16=False)
        assert maxerr(out32, g[f'out_fp32_{j}']) < 1e-5, tag
        outbf = dit.dit_forward(*args, emulate_bf16=True)
        # reference under torch.autocast(cpu, bf16) vs our bf16 rounding model
        assert rel_l2(outbf, g[f'out_bf16_{j}']) < 1e-2, tag
        # and the size of the bf16 effect itself stays inside the stated tolerance
        assert rel_l2(g[f'out_bf16_{j}'], g[f'out_fp32_{j}']) < 2e-2
        j += 1
    assert j >= 1


def test_block(golden):
    g = golden('g2_block')
    cfg = W.TINY_DIT
    P = W.make_dit_params(cfg, 0)
    tabs = dit.rope_table(cfg['dim'] // cfg['num_heads'])
    out = dit.block(P, 'blocks.1.', T(g['x'])[0], T(g['e0'])[0], 16, (1, 4, 4), tabs, T(g['ctx'])[0],
                    cfg['num_heads'], cfg['eps'], False, first_block=False)
    assert maxerr(out, g['out'][0]) < 1e-5


def test_sp_simulation_equals_single_rank():
    cfg = W.SMALL_DIT_HD128
    P = W.make_dit_params(cfg, 0)
    lat, ctx = W.randn((16, 2, 8, 8), 3), W.randn((17, cfg['text_dim']), 4)
    t = torch.tensor([500])
    ref = dit.dit_forward(P, cfg, lat, t, ctx, 32)
    sp = dit.dit_forward_sp_sim(P, cfg, lat, t, ctx, 32, sp=2)
    assert maxerr(sp, ref) < 1e-5


def test_scheduler_tables(golden):
    g = golden('g4_schedulers')
    for n, shift in ((50, 5.0), (2, 5.0), (6, 3.0)):
        s = schedulers.UniPCOracle(shift=1.0)
        ts = s.set_timesteps(n, shift=shift)
        assert np.array_equal(ts.numpy(), g[f'unipc_t_{n}'])
        assert np.array_equal(s.sigmas.numpy(), g[f'unipc_sigma_{n}'])
        d = schedulers.DPMppOracle()
        ts = d.set_timesteps(n, shift)
        assert np.array_equal(ts.numpy(), g[f'dpm_t_{n}'])
        assert np.array_equal(d.sigmas.numpy(), g[f'dpm_sigma_{n}'])
    assert list(g['unipc_t_50'][:6]) == [999, 995, 991, 987, 982, 978]   # SURVEY §8 a17 [probe]
    assert list(g['unipc_t_2']) == [999, 833]
    assert list(g['dpm_t_50'][:3]) == [1000, 995, 991]


@pytest.mark.parametrize('name,n,shift', [('unipc', 6, 3.0), ('unipc', 2, 5.0), ('dpm', 6, 3.0), ('dpm', 2, 5.0),
                                          ('unipc', 50, 5.0), ('dpm', 50, 5.0)])
def test_scheduler_trajectories(golden, name, n, shift):
    """(50, 5.0) is the production setting (text2video.py:114-124): every one of the 50 steps of the reference's
    trajectory is in g9_sampling50 (order ramp-up at step 1, lower_order_final at the end)."""
    g = golden('g9_sampling50' if n == 50 else 'g4_schedulers')
    if name == 'unipc':
        s = schedulers.UniPCOracle(shift=1.0)
        ts = s.set_timesteps(n, shift=shift)
    else:
        s = schedulers.DPMppOracle()
        ts = s.set_timesteps(n, shift)
    traj = g[f'traj_{name}'] if n == 50 else g[f'traj_{name}_{n}']
    if n == 50:
        assert np.array_equal(ts.numpy(), g[f'{name}_t']) and len(traj) == 50
    lat = T(g['traj_x0']).clone()
    for i, t in enumerate(ts):
        v = 0.5 * torch.tanh(lat) + 0.1 * torch.sin(t.float() / 100.0)
        lat = s.step(v, lat)
        assert maxerr(lat, traj[i]) < 2e-6, (name, i)


def test_vae_pieces(golden):
    g = golden('g5_vae_d8_t3')
    P = W.make_vae_params(8, 1)
    w, b = P['decoder.middle.0.residual.2.weight'], P['decoder.middle.0.residual.2.bias']
    x, c = T(g['conv_x']), T(g['conv_cache'])
    assert maxerr(vae.causal_conv3d(x, w, b), g['conv_nocache']) < 1e-5
    assert maxerr(vae.causal_conv3d(x, w, b, c), g['conv_cache2']) < 1e-5
    assert maxerr(vae.causal_conv3d(x, w, b, c[:, :, -1:]), g['conv_cache1']) < 1e-5
    assert maxerr(vae.attention_block(P, 'decoder.middle.1.', x), g['attn']) < 1e-5
    cache, idx = [None, None], [0]
    o1 = vae.residual_block(P, 'decoder.middle.0.', x[:, :, :1], cache, idx)
    idx = [0]
    o2 = vae.residual_block(P, 'decoder.middle.0.', x[:, :, 1:], cache, idx)
    assert maxerr(torch.cat([o1, o2], 2), g['res_chunked']) < 1e-5
    xu = T(g['up_x'])
    cache = [None]
    for i in range(3):
        o = vae.resample(P, 'decoder.upsamples.3.', xu[:, :, i:i + 1], cache, [0])
        assert maxerr(o, g[f'up_c{i}']) < 1e-5, i


@pytest.mark.parametrize('dim,t', [(8, 3), (8, 5), (32, 2)])
def test_vae_decode(golden, dim, t):
    g = golden(f'g5_vae_d{dim}_t{t}')
    P = W.make_vae_params(dim, 1)
    out = vae.vae_decode(P, T(g['z']))
    assert out.shape == g['video'].shape
    assert maxerr(out, g['video']) < 2e-5
    if t >= 3:  # chunking invariance: 1 + rest in one chunk
        out2 = vae.vae_decode(P, T(g['z']), chunks=[1, t - 1])
        assert maxerr(out2, g['video']) < 5e-5


def test_pipeline_cfg1(golden):
    """BASELINE.json configs[0]: 2-layer DiT, [16,1,8,8] latent, 2 steps, CPU fp32."""
    g = golden('g6_pipeline_cfg1')
    cfg = W.TINY_DIT
    P = W.make_dit_params(cfg, 0)

    def model_fn(lat, t, ctx):
        return dit.dit_forward(P, cfg, lat, t, ctx, 16)

    for solver in ('unipc', 'dpm++'):
        x0, _ = schedulers.sample_loop(model_fn, T(g['noise']), T(g['ctx']), T(g['ctx_null']), 2, 5.0, 5.0, solver)
        assert maxerr(x0, g[f'x0_{solver}']) < 2e-5, solver
    x0, _ = schedulers.sample_loop(model_fn, T(g['noise']), T(g['ctx']), T(g['ctx_null']), 2, 5.0, 5.0, 'unipc')
    video = vae.vae_decode(W.make_vae_params(8, 1), x0)
    assert maxerr(video, g['video_unipc']) < 5e-5


# ---- umT5 encoder ---------------------------------------------------------
def test_t5_oracle_vs_reference(golden):
    """oracle/t5.py against the imported reference T5Encoder (g7): fp32 exact to 1e-5; the bf16
    emulation must sit as close to the fp32 truth as the reference's own bf16 run does."""
    from oracle import t5 as ot5
    g = golden('g7_t5')
    P = W.make_t5_params(W.TINY_T5, 2)
    L = g['ids'].shape[1]
    assert (ot5.rel_buckets(L, L).numpy() == g['buckets']).all()
    for b in range(g['ids'].shape[0]):
        kl = int(g['mask'][b].sum())
        ids = T(g['ids'][b])
        truth = g['out_fp32'][b, :kl]
        assert maxerr(ot5.t5_encode(P, W.TINY_T5, ids, kl, False), truth) < 1e-5
        ob = ot5.t5_encode(P, W.TINY_T5, ids, kl, True)
        ref_bf = g['out_bf16'][b, :kl].astype(np.float32)
        assert rel_l2(ob, ref_bf) < 3e-2
        assert rel_l2(ob, truth) < 1.25 * rel_l2(ref_bf, truth) + 1e-3


def test_t5_product_bucket_table():
    """the host table the product hands mg_t5_attn_bf16 equals the reference bucket matrix."""
    from oracle import t5 as ot5
    from wan.modules.t5 import relative_buckets
    for n in (1, 2, 17, 24, 200, 512):
        full, tab = ot5.rel_buckets(n, n), relative_buckets(n)
        i, j = torch.meshgrid(torch.arange(n), torch.arange(n), indexing='ij')
        assert tab.dtype == torch.int32 and tab.numel() == 2 * n - 1
        assert (tab[(j - i) + n - 1].long() == full).all()


def test_t5_state_dict_names():
    from wan.modules.t5 import T5Encoder, umt5_xxl
    cfg = {('vocab' if k == 'vocab_size' else k): v for k, v in W.TINY_T5.items()}
    m = T5Encoder(**cfg)
    assert sorted(m.state_dict()) == sorted(W.t5_param_shapes(W.TINY_T5))
    m.load_state_dict(W.make_t5_params(W.TINY_T5, 2))
    assert m.blocks[0].norm1.weight.dtype == torch.float32 and m.blocks[0].attn.q.weight.dtype == torch.bfloat16
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, dtype=torch.long))          # no CPU path
    with pytest.raises(NotImplementedError):
        umt5_xxl(encoder_only=False)


@pytest.mark.parametrize('tag', ['nopad', 'pad'])
def test_train_side_sp_forward(golden, tag):
    """g8: the reference's training-side sequence-parallel DiT (scripts/train/model/model_seq.py) run as 2 lock-step
    ranks (tests/golden/make_golden_seq.py) vs the oracle's list-shuffle simulation of the same forward; 'pad' has
    seq_len 56 > 48 video tokens (rank 1 holds padded rows, padded keys are masked)."""
    g = golden('g8_train_seq')
    cfg = W.SMALL_DIT_HD128
    P = W.make_dit_params(cfg, 0)
    args = (P, cfg, T(g['lat']), T(g['t']), T(g['batch_context'])[0], int(g[f'seq_len_{tag}']), 2)
    out32 = dit.dit_forward_train_sp_sim(*args, emulate_bf16=False)
    assert maxerr(out32, g[f'sp2_fp32_{tag}']) < 1e-5
    assert maxerr(out32, g[f'single_fp32_{tag}']) < 1e-5          # the reference's own SP == its single-rank forward
    outbf = dit.dit_forward_train_sp_sim(*args, emulate_bf16=True)
    assert rel_l2(outbf, g[f'sp2_bf16_{tag}']) < 1e-2
    assert rel_l2(g[f'sp2_bf16_{tag}'], g[f'sp2_fp32_{tag}']) < 2e-2
    # one-rank simulation == the plain inference forward of the oracle (same algorithm, different plumbing)
    plain = dit.dit_forward(P, cfg, T(g['lat']), T(g['t']), T(g['batch_context'])[0, :33], int(g[f'seq_len_{tag}']))
    assert maxerr(plain, out32) < 1e-5


def test_rope_apply_dist(golden):
    """model_seq.py:37-76: the rank's slice of the position table, identity rotation on the padded rows."""
    g = golden('g8_train_seq')
    grid = tuple(int(v) for v in g['rope_dist_grid'][0])
    tabs = dit.rope_table(128)
    x = T(g['rope_dist_x'])[0]
    for r in range(2):
        assert maxerr(dit.rope(x, grid, tabs, pos0=r * x.shape[0]), g[f'rope_dist_rank{r}'][0]) < 1e-6


@pytest.mark.parametrize('solver', ['unipc', 'dpm++'])
def test_generate_loop_50_steps(golden, solver):
    """The oracle at the PRODUCTION sampling setting: the loop body of text2video.py:233-254 (two DiT forwards, CFG 5.0,
    scheduler step) for N = 50, shift 5.0 on the small head-dim-128 DiT, fp32, against the imported reference's latents
    after steps 1, 10, 20, 30, 40, 50 (tests/golden/make_golden_sampling50.py)."""
    g = golden('g9_sampling50')
    cfg = W.SMALL_DIT_HD128
    P = W.make_dit_params(cfg, 0)
    if solver == 'unipc':
        s = schedulers.UniPCOracle(shift=1.0)
        ts = s.set_timesteps(50, shift=5.0)
    else:
        s = schedulers.DPMppOracle()
        ts = s.set_timesteps(50, 5.0)
    lat, ctx, ctxn = T(g['noise']), T(g['ctx']), T(g['ctx_null'])
    keep = list(g['keep'])
    for i, t in enumerate(ts):
        tt = t.reshape(1)
        c = dit.dit_forward(P, cfg, lat, tt, ctx, 48)
        u = dit.dit_forward(P, cfg, lat, tt, ctxn, 48)
        lat = s.step((u + 5.0 * (c - u))[None], lat[None])[0]
        if i + 1 in keep:
            ref = T(g[f'lat_{solver}_fp32'][keep.index(i + 1)])
            assert ((lat - ref).norm() / ref.norm()).item() < 2e-5, (solver, i + 1)
