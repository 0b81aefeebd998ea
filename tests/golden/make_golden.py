"""Generate the golden vectors under tests/golden/*.npz by IMPORTING the reference.

Runs only in the build container (needs /root/reference); the GPU box and the tests never see
the reference, only the .npz files written here.  A fixture is data: inputs and the reference's
outputs.  Weights are regenerated from seeds by tests/golden/weights.py.

    python tests/golden/make_golden.py            # rewrites every fixture

Import shims — the reference hot path is pure Python + torch but
  (1) wan/__init__.py and wan/modules/__init__.py pull easydict / T5 (torch.cuda at import):
      synthetic parent packages with __path__ only are registered instead;
  (2) diffusers is absent: ConfigMixin/ModelMixin/SchedulerMixin/... stubs (no arithmetic);
  (3) flash_attention asserts CUDA: rebound to an fp32 SDPA with the k_lens mask — the
      third-party flash_attn kernel itself is "parity unpinned" by the reference;
  (4) for the bf16 fixtures `wan.modules.model.amp.autocast("cuda", ...)` is mapped to
      torch.autocast("cpu", ...) so the nested fp32 regions behave as they do on a GPU.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import weights as W  # noqa: E402

REF = '/root/reference'


def install_shims():
    for name, sub in (('wan', 'wan'), ('wan.modules', 'wan/modules'), ('wan.utils', 'wan/utils'),
                      ('wan.distributed', 'wan/distributed')):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = m

    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    d = mod('diffusers')
    cu = mod('diffusers.configuration_utils')

    class ConfigMixin:
        config_name = 'config.json'

        def register_to_config(self, **kw):
            self.config.__dict__.update(kw)

    def register_to_config(init):
        import functools
        import inspect

        @functools.wraps(init)
        def wrapper(self, *a, **kw):
            sig = inspect.signature(init)
            ba = sig.bind(self, *a, **kw)
            ba.apply_defaults()
            cfg = {k: v for k, v in ba.arguments.items() if k != 'self'}
            self.config = types.SimpleNamespace(**cfg)
            init(self, *a, **kw)
        return wrapper

    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    mm = mod('diffusers.models')
    mu = mod('diffusers.models.modeling_utils')
    mu.ModelMixin = torch.nn.Module
    ss = mod('diffusers.schedulers')
    su = mod('diffusers.schedulers.scheduling_utils')

    class SchedulerMixin:
        pass

    class SchedulerOutput:
        def __init__(self, prev_sample):
            self.prev_sample = prev_sample

    su.SchedulerMixin, su.SchedulerOutput, su.KarrasDiffusionSchedulers = SchedulerMixin, SchedulerOutput, []
    ut = mod('diffusers.utils')
    ut.deprecate = lambda *a, **k: None
    ut.is_scipy_available = lambda: True
    tu = mod('diffusers.utils.torch_utils')
    tu.randn_tensor = lambda shape, generator=None, device=None, dtype=None: torch.randn(
        shape, generator=generator, device=device, dtype=dtype)
    d.configuration_utils, d.models, d.schedulers, d.utils = cu, mm, ss, ut


def sdpa_flash_attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None,
                         causal=False, window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16,
                         version=None):
    """stand-in for flash_attn_varlen_func with the wrapper's dtype contract (attention.py:56-130)."""
    out_dtype = q.dtype
    half = (torch.float16, torch.bfloat16)
    emul = torch.is_autocast_enabled('cpu')
    if emul:
        q, k, v = [u if u.dtype in half else u.to(dtype) for u in (q, k, v)]
    b, lq, lk = q.size(0), q.size(1), k.size(1)
    outs = []
    for i in range(b):
        kl = lk if k_lens is None else int(k_lens[i])
        qi = q[i].float().transpose(0, 1)
        ki = k[i, :kl].float().transpose(0, 1)
        vi = v[i, :kl].float().transpose(0, 1)
        with torch.autocast('cpu', enabled=False):
            o = torch.nn.functional.scaled_dot_product_attention(qi, ki, vi)
        o = o.transpose(0, 1)
        if emul:
            o = o.to(dtype)
        outs.append(o)
    return torch.stack(outs).type(out_dtype)


class AmpCpu:
    """amp shim: autocast("cuda", ...) -> torch.autocast("cpu", ...)."""

    @staticmethod
    def autocast(device_type='cuda', dtype=None, enabled=True, **kw):
        if dtype is None:
            return torch.autocast('cpu', enabled=enabled)
        return torch.autocast('cpu', dtype=dtype, enabled=enabled)


def load_ref():
    install_shims()
    model = importlib.import_module('wan.modules.model')
    model.flash_attention = sdpa_flash_attention
    vae = importlib.import_module('wan.modules.vae')
    # t5.py evaluates torch.cuda.current_device() in a default argument and imports ftfy via tokenizers
    torch.cuda.current_device = lambda: 0
    sys.modules.setdefault('ftfy', types.ModuleType('ftfy'))
    global t5ref
    t5ref = importlib.import_module('wan.modules.t5')
    unipc = importlib.import_module('wan.utils.fm_solvers_unipc')
    dpm = importlib.import_module('wan.utils.fm_solvers')
    return model, vae, unipc, dpm


def build_ref_dit(model, cfg, P):
    kw = {k: cfg[k] for k in ('model_type', 'patch_size', 'text_len', 'in_dim', 'dim', 'ffn_dim', 'freq_dim',
                              'text_dim', 'out_dim', 'num_heads', 'num_layers', 'eps')}
    m = model.WanModel(**kw)
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return m.eval().requires_grad_(False)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB')


@torch.no_grad()
def main():
    import contextlib
    import io
    torch.manual_seed(0)
    model, vae, unipc, dpm = load_ref()

    # ---- G1 primitives ---------------------------------------------------------------------
    x = W.randn((1, 24, 128), 10)
    wq = 1 + 0.1 * W.randn((128,), 11)
    rn = model.WanRMSNorm(128, eps=1e-6)
    rn.weight.data.copy_(wq)
    ln = model.WanLayerNorm(128, 1e-6)
    grid = torch.tensor([[2, 3, 4]])
    freqs = torch.cat([model.rope_params(1024, 32 - 4 * (32 // 6)), model.rope_params(1024, 2 * (32 // 6)),
                       model.rope_params(1024, 2 * (32 // 6))], dim=1)
    xr = W.randn((1, 30, 4, 32), 12)  # 24 grid tokens + 6 padding tokens
    tok = W.randn((1, 24, 64), 13)
    save('g1_primitives', x=x, rms_w=wq, rmsnorm=rn(x), layernorm=ln(x),
         sinus_t=torch.tensor([999., 500., 3.]), sinus=model.sinusoidal_embedding_1d(64, torch.tensor([999., 500., 3.])),
         rope_x=xr, rope_grid=grid, rope=model.rope_apply(xr, grid, freqs),
         unpatch_in=tok, unpatch=build_ref_dit(model, W.TINY_DIT, W.make_dit_params(W.TINY_DIT, 0)).unpatchify(
             tok, grid)[0])

    # ---- G2/G3 DiT forwards ------------------------------------------------------------------
    for tag, cfg, latshape, ctxlens, seq_pad in (('tiny', W.TINY_DIT, (16, 1, 8, 8), (11, 5), 0),
                                                 ('tiny_pad', W.TINY_DIT, (16, 2, 8, 8), (7,), 6),
                                                 ('hd128', W.SMALL_DIT_HD128, (16, 2, 8, 12), (33, 9), 0)):
        P = W.make_dit_params(cfg, 0)
        m = build_ref_dit(model, cfg, P)
        lat = W.randn(latshape, 20)
        L = latshape[1] * (latshape[2] // 2) * (latshape[3] // 2)
        seq_len = L + seq_pad
        arrs = dict(lat=lat, seq_len=seq_len)
        for j, cl in enumerate(ctxlens):
            ctx = W.randn((cl, cfg['text_dim']), 30 + j)
            t = torch.tensor([999 - 333 * j])
            model.amp = importlib.import_module('torch.amp')
            out32 = m([lat], t=t, context=[ctx], seq_len=seq_len)[0]
            model.amp = AmpCpu
            with torch.autocast('cpu', dtype=torch.bfloat16):
                outbf = m([lat], t=t, context=[ctx], seq_len=seq_len)[0]
            model.amp = importlib.import_module('torch.amp')
            arrs.update({f'ctx{j}': ctx, f't{j}': t, f'out_fp32_{j}': out32, f'out_bf16_{j}': outbf.float()})
        save(f'g3_dit_{tag}', **arrs)

    # one block in isolation (G2)
    cfg = W.TINY_DIT
    P = W.make_dit_params(cfg, 0)
    m = build_ref_dit(model, cfg, P)
    xb = W.randn((1, 20, 128), 40)
    e0 = 0.3 * W.randn((1, 6, 128), 41)
    ctxe = W.randn((1, 32, 128), 42)
    yb = m.blocks[1](xb, e=e0, seq_lens=torch.tensor([16]), grid_sizes=torch.tensor([[1, 4, 4]]), freqs=m.freqs,
                     context=ctxe, context_lens=None)
    save('g2_block', x=xb, e0=e0, ctx=ctxe, out=yb)

    # ---- G4 schedulers -----------------------------------------------------------------------------
    arrs = {}
    sink = io.StringIO()
    for n, shift in ((50, 5.0), (2, 5.0), (6, 3.0)):
        s = unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        s.set_timesteps(n, device='cpu', shift=shift)
        arrs[f'unipc_t_{n}'] = s.timesteps
        arrs[f'unipc_sigma_{n}'] = s.sigmas
        d = dpm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        ts, _ = dpm.retrieve_timesteps(d, device='cpu', sigmas=dpm.get_sampling_sigmas(n, shift))
        arrs[f'dpm_t_{n}'] = ts
        arrs[f'dpm_sigma_{n}'] = d.sigmas
    x0 = W.randn((1, 16, 2, 4, 4), 50)
    arrs['traj_x0'] = x0
    for name, n, shift in (('unipc', 6, 3.0), ('unipc', 2, 5.0), ('dpm', 6, 3.0), ('dpm', 2, 5.0)):
        if name == 'unipc':
            s = unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
            s.set_timesteps(n, device='cpu', shift=shift)
            ts = s.timesteps
        else:
            s = dpm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
            ts, _ = dpm.retrieve_timesteps(s, device='cpu', sigmas=dpm.get_sampling_sigmas(n, shift))
        lat = x0.clone()
        traj = []
        with contextlib.redirect_stdout(sink):
            for t in ts:
                v = 0.5 * torch.tanh(lat) + 0.1 * torch.sin(t.float() / 100.0)
                lat = s.step(v, t, lat, return_dict=False)[0]
                traj.append(lat.clone())
        arrs[f'traj_{name}_{n}'] = torch.stack(traj)
    save('g4_schedulers', **arrs)

    # ---- G5 VAE ---------------------------------------------------------------------------------------
    for dim, zshape in ((8, (16, 3, 8, 8)), (8, (16, 5, 4, 6)), (32, (16, 2, 4, 4))):
        P = W.make_vae_params(dim, 1)
        ref = vae.WanVAE_(dim=dim, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                          temperal_downsample=[False, True, True], dropout=0.0)
        sd = ref.state_dict()
        sd.update({k: v.reshape(sd[k].shape) for k, v in P.items()})
        ref.load_state_dict(sd)
        ref.eval().requires_grad_(False)
        z = W.randn(zshape, 60)
        mean = torch.tensor([-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
                             0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921])
        std = torch.tensor([2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
                            3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160])
        out = ref.decode(z[None], [mean, 1.0 / std]).float().clamp_(-1, 1)[0]
        arrs = dict(z=z, video=out)
        if dim == 8 and zshape[1] == 3:
            # module-level pieces: causal conv with/without cache, residual block, attention, resample
            xin = W.randn((1, 32, 2, 5, 6), 61)
            cache = W.randn((1, 32, 2, 5, 6), 62)
            rb = ref.decoder.middle[0]
            c3 = rb.residual[2]
            arrs.update(conv_x=xin, conv_cache=cache, conv_nocache=c3(xin), conv_cache2=c3(xin, cache),
                        conv_cache1=c3(xin, cache[:, :, -1:]))
            arrs['attn'] = ref.decoder.middle[1](xin)
            fc = [None] * 2
            o1 = rb(xin[:, :, :1], fc, [0])
            o2 = rb(xin[:, :, 1:], fc, [0])
            arrs['res_chunked'] = torch.cat([o1, o2], dim=2)
            up = ref.decoder.upsamples[3]
            xu = W.randn((1, 32, 3, 4, 5), 63)
            fc = [None]
            arrs['up_x'] = xu
            arrs['up_c0'] = up(xu[:, :, :1], fc, [0])
            arrs['up_c1'] = up(xu[:, :, 1:2], fc, [0])
            arrs['up_c2'] = up(xu[:, :, 2:3], fc, [0])
        save(f'g5_vae_d{dim}_t{zshape[1]}', **arrs)

    # ---- G6 end-to-end config 1: 2-layer DiT, [16,1,8,8] latent, 2 UniPC steps, tiny VAE ---------------
    cfg = W.TINY_DIT
    m = build_ref_dit(model, cfg, W.make_dit_params(cfg, 0))
    noise = W.randn((16, 1, 8, 8), 70)
    ctx, ctxn = W.randn((11, 64), 71), W.randn((5, 64), 72)
    arrs = dict(noise=noise, ctx=ctx, ctx_null=ctxn)
    for solver in ('unipc', 'dpm++'):
        if solver == 'unipc':
            s = unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
            s.set_timesteps(2, device='cpu', shift=5.0)
            ts = s.timesteps
        else:
            s = dpm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
            ts, _ = dpm.retrieve_timesteps(s, device='cpu', sigmas=dpm.get_sampling_sigmas(2, 5.0))
        lat = [noise]
        with contextlib.redirect_stdout(sink):
            for t in ts:  # the loop body of wan/text2video.py:233-254
                tt = torch.stack([t])
                c = m(lat, t=tt, context=[ctx], seq_len=16)[0]
                u = m(lat, t=tt, context=[ctxn], seq_len=16)[0]
                v = u + 5.0 * (c - u)
                lat = [s.step(v.unsqueeze(0), t, lat[0].unsqueeze(0), return_dict=False)[0].squeeze(0)]
        arrs[f'x0_{solver}'] = lat[0]
    Pv = W.make_vae_params(8, 1)
    ref = vae.WanVAE_(dim=8, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                      temperal_downsample=[False, True, True], dropout=0.0)
    sd = ref.state_dict()
    sd.update({k: v.reshape(sd[k].shape) for k, v in Pv.items()})
    ref.load_state_dict(sd)
    arrs['video_unipc'] = ref.eval().decode(arrs['x0_unipc'][None], [mean, 1.0 / std]).float().clamp_(-1, 1)[0]
    save('g6_pipeline_cfg1', **arrs)


@torch.no_grad()
def t5_golden():
    """G7: umT5 encoder (tiny config) in fp32 and in the deployment dtype bf16, two padded prompts."""
    cfg = W.TINY_T5
    P = W.make_t5_params(cfg, 2)
    enc = t5ref.T5Encoder(cfg['vocab_size'], cfg['dim'], cfg['dim_attn'], cfg['dim_ffn'], cfg['num_heads'],
                          cfg['num_layers'], cfg['num_buckets'], shared_pos=False, dropout=0.0).eval()
    missing, unexpected = enc.load_state_dict(P, strict=True)
    rs = np.random.RandomState(3)
    L = 24
    ids = torch.from_numpy(rs.randint(1, cfg['vocab_size'], size=(2, L))).long()
    lens = [17, 24]
    mask = torch.zeros(2, L, dtype=torch.long)
    for i, n in enumerate(lens):
        mask[i, :n] = 1
        ids[i, n:] = 0
    out32 = enc(ids, mask)
    encb = enc.to(torch.bfloat16)
    outbf = encb(ids, mask).float()
    save('g7_t5', ids=ids, mask=mask, out_fp32=out32, out_bf16=outbf,
         buckets=enc.blocks[0].pos_embedding._relative_position_bucket(
             torch.arange(L).unsqueeze(0) - torch.arange(L).unsqueeze(1)))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 't5':
        load_ref()
        t5_golden()
    else:
        main()
        t5_golden()
