"""Deterministic synthetic weights for the parity tests (no reference import, no checkpoint).

numpy's legacy RandomState stream is stable across numpy versions and platforms, so the fixture
files only need to hold inputs and expected outputs; both tests/golden/make_golden.py (which
loads these tensors into the imported reference modules) and the tests regenerate the same
parameters from the seed.  Key names are the reference's state_dict names."""
import numpy as np
import torch

# BASELINE.json configs[0]: the CPU-runnable plumbing case (text_dim 64 so K % 64 == 0 holds for
# the HIP GEMM; everything else as SURVEY §8(c) G3)
TINY_DIT = dict(model_type='t2v', patch_size=(1, 2, 2), text_len=32, in_dim=16, dim=128, ffn_dim=256,
                freq_dim=64, text_dim=64, out_dim=16, num_heads=4, num_layers=2, eps=1e-6)
# head_dim 128 variant (the only head size the MFMA attention kernel implements)
SMALL_DIT_HD128 = dict(model_type='t2v', patch_size=(1, 2, 2), text_len=64, in_dim=16, dim=256,
                       ffn_dim=512, freq_dim=64, text_dim=128, out_dim=16, num_heads=2, num_layers=2,
                       eps=1e-6)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def dit_param_shapes(cfg):
    d, f = cfg['dim'], cfg['ffn_dim']
    pd = int(np.prod(cfg['patch_size']))
    sh = {
        'patch_embedding.weight': (d, cfg['in_dim'], *cfg['patch_size']),
        'patch_embedding.bias': (d,),
        'text_embedding.0.weight': (d, cfg['text_dim']), 'text_embedding.0.bias': (d,),
        'text_embedding.2.weight': (d, d), 'text_embedding.2.bias': (d,),
        'time_embedding.0.weight': (d, cfg['freq_dim']), 'time_embedding.0.bias': (d,),
        'time_embedding.2.weight': (d, d), 'time_embedding.2.bias': (d,),
        'time_projection.1.weight': (6 * d, d), 'time_projection.1.bias': (6 * d,),
        'head.modulation': (1, 2, d),
        'head.head.weight': (pd * cfg['out_dim'], d), 'head.head.bias': (pd * cfg['out_dim'],),
    }
    for i in range(cfg['num_layers']):
        p = f'blocks.{i}.'
        sh[p + 'modulation'] = (1, 6, d)
        for a in ('self_attn.', 'cross_attn.'):
            for n in 'qkvo':
                sh[p + a + n + '.weight'] = (d, d)
                sh[p + a + n + '.bias'] = (d,)
            sh[p + a + 'norm_q.weight'] = (d,)
            sh[p + a + 'norm_k.weight'] = (d,)
        sh[p + 'norm3.weight'] = (d,)
        sh[p + 'norm3.bias'] = (d,)
        sh[p + 'ffn.0.weight'] = (f, d)
        sh[p + 'ffn.0.bias'] = (f,)
        sh[p + 'ffn.2.weight'] = (d, f)
        sh[p + 'ffn.2.bias'] = (d,)
    return sh


def make_dit_params(cfg, seed=0):
    rs = np.random.RandomState(seed)
    P = {}
    for name, shape in dit_param_shapes(cfg).items():
        if name.endswith('modulation'):
            a = rs.standard_normal(shape) / np.sqrt(cfg['dim'])
        elif 'norm' in name and name.endswith('weight'):
            a = 1.0 + 0.1 * rs.standard_normal(shape)
        elif name.endswith('bias'):
            a = 0.05 * rs.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = rs.standard_normal(shape) * (1.0 / np.sqrt(fan_in))
        P[name] = _t(a)
    return P


def vae_decoder_shapes(dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2,
                       temporal_upsample=(True, True, False)):
    """Shapes of conv2.* and decoder.* following Decoder3d.__init__ (reference vae.py:367-420)."""
    sh = {'conv2.weight': (z_dim, z_dim, 1, 1, 1), 'conv2.bias': (z_dim,)}
    dims = [dim * u for u in [dim_mult[-1]] + list(dim_mult[::-1])]

    def res(pre, cin, cout):
        sh[pre + 'residual.0.gamma'] = (cin, 1, 1, 1)
        sh[pre + 'residual.2.weight'] = (cout, cin, 3, 3, 3)
        sh[pre + 'residual.2.bias'] = (cout,)
        sh[pre + 'residual.3.gamma'] = (cout, 1, 1, 1)
        sh[pre + 'residual.6.weight'] = (cout, cout, 3, 3, 3)
        sh[pre + 'residual.6.bias'] = (cout,)
        if cin != cout:
            sh[pre + 'shortcut.weight'] = (cout, cin, 1, 1, 1)
            sh[pre + 'shortcut.bias'] = (cout,)

    sh['decoder.conv1.weight'] = (dims[0], z_dim, 3, 3, 3)
    sh['decoder.conv1.bias'] = (dims[0],)
    res('decoder.middle.0.', dims[0], dims[0])
    sh['decoder.middle.1.norm.gamma'] = (dims[0], 1, 1)
    sh['decoder.middle.1.to_qkv.weight'] = (3 * dims[0], dims[0], 1, 1)
    sh['decoder.middle.1.to_qkv.bias'] = (3 * dims[0],)
    sh['decoder.middle.1.proj.weight'] = (dims[0], dims[0], 1, 1)
    sh['decoder.middle.1.proj.bias'] = (dims[0],)
    res('decoder.middle.2.', dims[0], dims[0])
    idx = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(num_res_blocks + 1):
            res(f'decoder.upsamples.{idx}.', cin, cout)
            idx += 1
            cin = cout
        if i != len(dim_mult) - 1:
            pre = f'decoder.upsamples.{idx}.'
            sh[pre + 'resample.1.weight'] = (cout // 2, cout, 3, 3)
            sh[pre + 'resample.1.bias'] = (cout // 2,)
            if temporal_upsample[i]:
                sh[pre + 'time_conv.weight'] = (2 * cout, cout, 3, 1, 1)
                sh[pre + 'time_conv.bias'] = (2 * cout,)
            idx += 1
    sh['decoder.head.0.gamma'] = (dims[-1], 1, 1, 1)
    sh['decoder.head.2.weight'] = (3, dims[-1], 3, 3, 3)
    sh['decoder.head.2.bias'] = (3,)
    return sh


def make_vae_params(dim=8, seed=1, z_dim=16):
    rs = np.random.RandomState(seed)
    P = {}
    for name, shape in vae_decoder_shapes(dim, z_dim).items():
        if name.endswith('gamma'):
            a = 1.0 + 0.1 * rs.standard_normal(shape)
        elif name.endswith('bias'):
            a = 0.05 * rs.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = rs.standard_normal(shape) * (1.0 / np.sqrt(fan_in))
        P[name] = _t(a)
    return P


def randn(shape, seed):
    return _t(np.random.RandomState(seed).standard_normal(shape))


# ---- umT5 encoder ---------------------------------------------------------------
TINY_T5 = dict(vocab_size=100, dim=128, dim_attn=128, dim_ffn=256, num_heads=4, num_layers=2, num_buckets=32)


def t5_param_shapes(cfg):
    d, da, f = cfg['dim'], cfg['dim_attn'], cfg['dim_ffn']
    sh = {'token_embedding.weight': (cfg['vocab_size'], d), 'norm.weight': (d,)}
    for i in range(cfg['num_layers']):
        p = f'blocks.{i}.'
        sh[p + 'norm1.weight'] = (d,)
        sh[p + 'norm2.weight'] = (d,)
        for n in 'qkv':
            sh[p + f'attn.{n}.weight'] = (da, d)
        sh[p + 'attn.o.weight'] = (d, da)
        sh[p + 'ffn.gate.0.weight'] = (f, d)
        sh[p + 'ffn.fc1.weight'] = (f, d)
        sh[p + 'ffn.fc2.weight'] = (d, f)
        sh[p + 'pos_embedding.embedding.weight'] = (cfg['num_buckets'], cfg['num_heads'])
    return sh


def make_t5_params(cfg, seed=2):
    rs = np.random.RandomState(seed)
    P = {}
    for name, shape in t5_param_shapes(cfg).items():
        if 'norm' in name:
            a = 1.0 + 0.1 * rs.standard_normal(shape)
        elif name == 'token_embedding.weight':
            a = rs.standard_normal(shape)
        elif 'pos_embedding' in name:
            a = 0.5 * rs.standard_normal(shape)
        elif name.endswith('q.weight'):
            a = rs.standard_normal(shape) * (shape[1] ** -0.5) * 0.5   # T5 has no 1/sqrt(d): keep logits O(1)
        else:
            a = rs.standard_normal(shape) * (shape[1] ** -0.5)
        P[name] = _t(a)
    return P
