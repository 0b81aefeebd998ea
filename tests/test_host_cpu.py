"""CPU-only tests of the host side: C-ABI surface, configs, checkpoint layout, scheduler host
logic (coefficients) against the reference's golden trajectories, Ulysses data movement with
2 gloo processes against the oracle's single-process simulation."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    """both libraries load and export exactly what include/moviigen_hip.h declares for them (no compute): the PRODUCT library
    everything outside the header's MG_AB_BUILD section — in particular no kernel-selection switch and no profiling hook —, the A/B
    library all of it."""
    import subprocess
    from wan.backend import lib
    hdr = open(os.path.join(ROOT, 'include', 'moviigen_hip.h')).read()
    m = re.search(r'#ifdef MG_AB_BUILD(.*?)#endif /\* MG_AB_BUILD \*/', hdr, re.S)
    assert m
    decl = lambda txt: set(re.findall(r'^(?:int|void|const char\*|int64_t)\s+(mg_[a-z0-9_]+)\s*\(', txt, re.M))
    ab_only = decl(m.group(1))
    product = decl(hdr.replace(m.group(1), ''))
    assert len(product) >= 40 and ab_only == set(lib.SIGNATURES_AB), ab_only ^ set(lib.SIGNATURES_AB)
    assert product == set(lib.SIGNATURES), product ^ set(lib.SIGNATURES)
    assert not any('set_variant' in n or 'debug' in n or 'profile' in n for n in product)
    for path, want, handle in ((lib.LIB_PATH, product, lib.load()), (lib.LIB_AB_PATH, product | ab_only, lib.load_ab())):
        for name in want:
            assert hasattr(handle, name), (path, name)
        assert handle.mg_abi_version() == 9
        assert b'gfx950' in handle.mg_version()
        nm = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True, check=True).stdout
        exported = set(re.findall(r'\bT (mg_[a-z0-9_]+)$', nm, re.M))
        assert exported == want, (path, exported ^ want)      # nothing undeclared leaks out, nothing declared is missing


def test_ab_library_switches_refuse_unknown_numbers():
    """the A/B library's kernel-selection switches are host-side bookkeeping (no GPU needed): numbers that name a kernel or a measurement build are
    accepted, everything else is refused with MG_ERR_ARG instead of being aliased to some kernel; the scope resets them."""
    from wan.backend import lib
    with lib.ab_library() as h:
        for v in (0, 1, 2, 7, 8, 11, 12, 110, 173, 200, 208, 232, 296, 968, 1224, 4296):      # 0 = the product's rule by shape
            assert h.mg_gemm_set_variant(v) == 0, v
        for v in (-1, 3, 4, 5, 6, 9, 10, 13, 99, 174, 199, 264, 264 + 8, 200 + 8192, 1 << 20):      # 264 = variant 12, generated body 1: not compiled
            assert h.mg_gemm_set_variant(v) != 0, v
        assert h.mg_gemm_set_variant(lib.DEFAULT_GEMM_VARIANT) == 0
        for v in (0, 3):
            assert h.mg_attn_set_variant(v) == 0, v
        for v in (-1, 1, 2, 4, 16):
            assert h.mg_attn_set_variant(v) != 0, v


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'moviigen1.1_amd')
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), os.path.join(dp, fn)
                assert '/root/reference' not in src


def test_missing_gpu_fails_loudly():
    import wan
    m = wan.modules.WanModel(**{k: v for k, v in W.TINY_DIT.items()})
    m.load_state_dict(W.make_dit_params(W.TINY_DIT, 0))
    with pytest.raises(RuntimeError):
        m([torch.zeros(16, 1, 8, 8)], t=torch.tensor([999]), context=[torch.zeros(5, 64)], seq_len=16)


def test_configs_match_reference_values():
    from wan.configs import MAX_AREA_CONFIGS, SIZE_CONFIGS, SUPPORTED_SIZES, WAN_CONFIGS
    c = WAN_CONFIGS['t2v-14B']
    assert (c.dim, c.ffn_dim, c.num_heads, c.num_layers, c.freq_dim, c.text_len) == (5120, 13824, 40, 40, 256, 512)
    assert c.patch_size == (1, 2, 2) and c.vae_stride == (4, 8, 8) and c.eps == 1e-6
    assert c.num_train_timesteps == 1000 and c.sample_fps == 16 and c.param_dtype == torch.bfloat16
    assert c.t5_checkpoint == 'models_t5_umt5-xxl-enc-bf16.pth' and c.vae_checkpoint == 'Wan2.1_VAE.pth'
    assert SIZE_CONFIGS['1920*832'] == (1920, 832) and SIZE_CONFIGS['1280*720'] == (1280, 720)
    assert MAX_AREA_CONFIGS['1280*720'] == 921600 and '1920*1056' in SUPPORTED_SIZES['t2v-14B']
    assert set(WAN_CONFIGS) == {'t2v-14B', 't2i-14B'}


def test_state_dict_layout_and_from_pretrained(tmp_path):
    """checkpoint layout drop-in: same key names/shapes as the reference, loadable
    from config.json + diffusion_pytorch_model*.safetensors shards."""
    from safetensors.torch import save_file
    import wan
    cfg = W.TINY_DIT
    P = W.make_dit_params(cfg, 0)
    m = wan.modules.WanModel(**cfg)
    own = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert own == {k: tuple(s) for k, s in W.dit_param_shapes(cfg).items()}
    keys = sorted(P)
    save_file({k: P[k] for k in keys[:40]}, str(tmp_path / 'diffusion_pytorch_model-00001-of-00002.safetensors'))
    save_file({k: P[k] for k in keys[40:]}, str(tmp_path / 'diffusion_pytorch_model-00002-of-00002.safetensors'))
    json.dump({k: cfg[k] for k in ('model_type', 'text_len', 'in_dim', 'dim', 'ffn_dim', 'freq_dim', 'out_dim',
                                   'num_heads', 'num_layers', 'eps')} | {'text_dim': cfg['text_dim']},
              open(tmp_path / 'config.json', 'w'))
    # text_dim is not in the reference's config.json (14B uses the default 4096); from_pretrained honours it when present
    m2 = wan.modules.WanModel.from_pretrained(str(tmp_path))
    sd = m2.state_dict()
    for k in keys:
        ref = P[k].to(sd[k].dtype)
        assert torch.equal(sd[k], ref), k
    assert sd['blocks.0.self_attn.q.weight'].dtype == torch.bfloat16
    assert sd['blocks.0.self_attn.norm_q.weight'].dtype == torch.float32
    assert sd['head.head.weight'].dtype == torch.float32 and sd['time_embedding.0.weight'].dtype == torch.float32


def _np_lincomb(like, terms):
    acc = None
    for t, c in terms:
        v = t.to(torch.float32) * torch.tensor(c, dtype=torch.float32)
        acc = v if acc is None else acc + v
    return acc


@pytest.mark.parametrize('name,n,shift', [('unipc', 6, 3.0), ('unipc', 2, 5.0), ('dpm', 6, 3.0), ('dpm', 2, 5.0),
                                          ('unipc', 50, 5.0), ('dpm', 50, 5.0)])
def test_scheduler_host_logic_vs_reference(golden, name, n, shift):
    """product schedulers (coefficient algebra on the host) driven with a CPU lincomb injected by
    the test: must follow the reference's trajectories (the GPU run of the same thing is in
    test_gpu_parity.py)."""
    from wan.utils import (FlowDPMSolverMultistepScheduler, FlowUniPCMultistepScheduler, get_sampling_sigmas,
                           retrieve_timesteps)
    g = golden('g4_schedulers')
    if name == 'unipc':
        s = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False,
                                        lincomb=_np_lincomb)
        s.set_timesteps(n, device='cpu', shift=shift)
        ts = s.timesteps
        assert np.array_equal(s.sigmas.numpy(), g[f'unipc_sigma_{n}'])
    else:
        s = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False,
                                            lincomb=_np_lincomb)
        ts, _ = retrieve_timesteps(s, device='cpu', sigmas=get_sampling_sigmas(n, shift))
        assert np.array_equal(s.sigmas.numpy(), g[f'dpm_sigma_{n}'])
    assert np.array_equal(ts.numpy(), g[f'{name}_t_{n}'])
    # (50, 5.0) = the production setting (text2video.py:114-124): all 50 steps of the reference's trajectory
    g9 = golden('g9_sampling50')
    traj = g9[f'traj_{name}'] if n == 50 else g[f'traj_{name}_{n}']
    if n == 50:
        assert np.array_equal(ts.numpy(), g9[f'{name}_t']) and len(traj) == 50
    lat = torch.from_numpy(g['traj_x0']).clone()
    assert np.array_equal(g['traj_x0'], g9['traj_x0'])
    for i, t in enumerate(ts):
        v = 0.5 * torch.tanh(lat) + 0.1 * torch.sin(t.float() / 100.0)
        lat = s.step(v, t, lat, return_dict=False)[0]
        ref = torch.from_numpy(traj[i])
        assert ((lat - ref).abs().max() / ref.abs().max()).item() < 5e-6, (name, i)


def test_preflight_report_logic():
    """wan/distributed/preflight.py: what bench.py copies into its line from a report, and when the copy-engine transport is recommended —
    only if its windows opened, its self-check raised nothing and it moved the probe >= 5 % faster than the collective."""
    from wan.distributed import preflight
    base = {'rccl_ranks': 8, 'peer_access': [[True] * 8] * 8, 'elapsed_s': 3.2, 'errors': [], 'ipc_open': True,
            'link_gbps_measured': {'all_to_all': 40.0, 'peer_copy': 44.0}}
    assert preflight.recommend(base) == 'peer_copy'
    assert preflight.recommend({**base, 'link_gbps_measured': {'all_to_all': 40.0, 'peer_copy': 41.0}}) == 'torch'          # < 5 %
    assert preflight.recommend({**base, 'ipc_open': False}) == 'torch' and preflight.recommend({**base, 'ipc_open': None}) == 'torch'
    assert preflight.recommend({**base, 'errors': ['peer_copy: the probe pattern did not arrive intact']}) == 'torch'
    assert preflight.recommend({**base, 'link_gbps_measured': {'all_to_all': 40.0}}) == 'torch'
    assert preflight.recommend({'errors': ['preflight: RuntimeError: x']}) == 'torch'
    p = preflight.parse({**base, 'recommended': preflight.recommend(base)})
    assert p == {'rccl_ranks': 8, 'transport_recommended': 'peer_copy', 'link_gbps_measured': {'all_to_all': 40.0, 'peer_copy': 44.0},
                 'peer_access_all': True, 'ipc_open': True, 'preflight_s': 3.2, 'preflight_errors': []}
    row = [True, None, False] + [True] * 5            # None = a peer on another host / not visible: not counted; False = no access
    assert preflight.parse({**base, 'peer_access': [row] + [[True] * 8] * 7})['peer_access_all'] is False
    assert preflight.parse({})['transport_recommended'] == 'torch' and preflight.parse({})['peer_access_all'] is None


def test_preflight_gloo_world2():
    """the preflight end to end on 2 gloo ranks with host tensors (tests/dist_preflight_worker.py): stages, the time box decided by rank 0 for
    the whole group, the report; the control-plane reductions and the copy-engine transport's all-or-none vote."""
    script = os.path.join(ROOT, 'tests', 'dist_preflight_worker.py')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                        '--master-port', '29547', script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'PREFLIGHT_OK rank0/2' in r.stdout and 'PREFLIGHT_OK rank1/2' in r.stdout


def test_unsupported_solver_config_raises():
    from wan.utils import FlowUniPCMultistepScheduler
    with pytest.raises(NotImplementedError):
        FlowUniPCMultistepScheduler(solver_type='bh1')


def test_rope_table_matches_oracle():
    from oracle import dit
    from wan.modules.model import rope_cos_sin
    tab = rope_cos_sin(128, (3, 5, 7)).double()
    ta, th, tw = dit.rope_table(128)
    c0, c1 = 22, 21
    ref = torch.cat([torch.stack([torch.cos(a), torch.sin(a)], -1).reshape(-1, 2)
                     for a in (ta[:3], th[:5], tw[:7])])
    assert tab.shape == (3 * c0 + 5 * c1 + 7 * c1, 2)
    assert (tab - ref).abs().max() < 1e-7


def test_ulysses_gloo_world2():
    """2 processes, gloo, CPU tensors: seq_to_head / head_to_seq / all_gather_seq against the
    oracle's single-process all-to-all simulation."""
    script = os.path.join(ROOT, 'tests', 'dist_ulysses_worker.py')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                        '--master-addr', '127.0.0.1', '--master-port', '29541', script],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'ULYSSES_OK rank0' in r.stdout and 'ULYSSES_OK rank1' in r.stdout
    assert 'SHARDS_OK rank0' in r.stdout and 'SHARDS_OK rank1' in r.stdout
    assert 'CFGP_HOST_OK rank0' in r.stdout and 'CFGP_HOST_OK rank1' in r.stdout
    assert 'TRAIN_SP_STATE_OK rank0' in r.stdout and 'TRAIN_SP_STATE_OK rank1' in r.stdout


def test_vae_stage_partition():
    """cut points of the layer-pipelined VAE decode: contiguous, non-empty, minimal bottleneck."""
    from wan.modules.vae import partition_costs
    assert partition_costs([5, 1, 1, 1, 9, 2, 2, 8], 3) == [0, 4, 6, 8]
    assert partition_costs([1, 2, 3], 5) == [0, 1, 2, 3]            # more ranks than stages
    assert partition_costs([4, 4, 4, 4], 2) == [0, 2, 4]
    assert partition_costs([7], 1) == [0, 1]


def test_vae_pipeline_host_logic():
    """host side of the layer-pipelined decode (round 4): the chunk list is the single-GPU one, the activation shape a
    cut receives is computed locally (no shape header travels), the cut is weighed by time, and the makespan model is the
    pipeline recurrence."""
    from wan.modules.vae import REL_MS_PER_MAC, WanVAE_, partition_costs, pipeline_makespan
    v = WanVAE_(W.make_vae_params(8, 1), device='cpu')
    assert v._chunks(21) == [1, 4, 4, 4, 4, 4] and v._chunks(10) == [1, 4, 4, 1] and v._chunks(1) == [1]
    assert v._chunks(4, [1, 3]) == [1, 3]
    st = v._stages()
    n = len(st)
    # dim-8 decoder: 32 -> 32 -> 16 -> 8 channels, x8 in space; frames x4 after the two temporal up-samplers (not on the first chunk)
    assert v.stage_out_shape(1, 1, True, 6, 10) == (1, 6, 10, 32)
    assert v.stage_out_shape(n, 1, True, 6, 10) == (1, 48, 80, 3)
    assert v.stage_out_shape(n, 4, False, 6, 10) == (16, 48, 80, 3)
    assert v.stage_out_shape(n - 1, 4, False, 6, 10) == (16, 48, 80, 8)
    first_up = [i for i, s_ in enumerate(st) if s_[0] == 'up'][0]
    assert v.stage_out_shape(first_up + 1, 4, False, 6, 10) == (8, 12, 20, 16)
    assert v.stage_out_shape(first_up + 1, 1, True, 6, 10) == (1, 12, 20, 16)
    # weights: MACs x the class factor, or the measured milliseconds as given
    macs, wts = v.stage_costs(6, 10), v.stage_weights(6, 10)
    kinds = [k for k, _, _ in st]
    assert wts[kinds.index('attn')] == macs[kinds.index('attn')] * REL_MS_PER_MAC['attn']
    assert wts[-1] == macs[-1] * REL_MS_PER_MAC['head'] and wts[-2] == macs[-2] * REL_MS_PER_MAC['narrow']
    assert v.stage_weights(6, 10, stage_ms=list(range(1, n + 1))) == list(range(1, n + 1))
    assert REL_MS_PER_MAC['wide'] == 1.0 and REL_MS_PER_MAC['head'] > REL_MS_PER_MAC['attn'] > 1.0
    # the recurrence: one segment = the plain sum; balanced two-stage pipeline of 6 chunks = fill + 6 steady chunks
    mk, eff = pipeline_makespan([3.0], [10.0], 6)
    assert mk == 53.0 and abs(eff - 1.0) < 1e-12
    mk, eff = pipeline_makespan([1.0, 1.0], [10.0, 10.0], 6)
    assert mk == 61.0 and abs(eff - (2 + 100) / (2 * 61.0)) < 1e-12
    mk2, _ = pipeline_makespan([1.0, 1.0], [10.0, 10.0], 6, xfer_ms=[5.0])
    assert mk2 == mk + 5.0                                  # a transfer only adds latency while it is shorter than a segment
    assert partition_costs(wts, 3)[0] == 0 and partition_costs(wts, 3)[-1] == n


def test_collectives_have_one_test_transport_guard():
    """every RCCL call of the product lives in wan/distributed/collectives.py, and the host-staged gloo transport of the
    one-GPU multi-process tests is reachable through exactly one predicate."""
    import re
    dist_dir = os.path.join(ROOT, 'moviigen1.1_amd', 'wan', 'distributed')
    for fn in os.listdir(dist_dir):
        if not fn.endswith('.py') or fn in ('_test_transport.py', 'collectives.py'):
            continue
        src = open(os.path.join(dist_dir, fn)).read()
        code = '\n'.join(ln.split('#')[0] for ln in src.split('"""')[::2] for ln in ln.splitlines())
        assert "'gloo'" not in code and '"gloo"' not in code, fn       # (round 6: peer_copy's fallback vote went to collectives.control_reduce)
        assert '.cpu()' not in code, fn
    col = open(os.path.join(dist_dir, 'collectives.py')).read()
    assert len(re.findall(r'if _test_transport\.staged\(', col)) == 8          # all_to_all, all_gather, broadcast, send, recv, neighbor_exchange, ring_hop, rendezvous
    tt = open(os.path.join(dist_dir, '_test_transport.py')).read()
    assert 'def staged(t, group):' in tt and "t.is_cuda and dist.get_backend(group) == 'gloo'" in tt


def test_vae_upconv_option_host_side():
    """WanVAE_'s `upconv` keyword: validated at construction, and the layer-pipelined decode weighs the up-conv stages by
    what they execute (9 taps at 4x the pixels through the upsample, 16/9 as phase convs)."""
    from wan.modules.vae import WanVAE_
    P = W.make_vae_params(8, 1)
    with pytest.raises(ValueError):
        WanVAE_(P, device='cpu', upconv='bilinear')
    ph, ga = WanVAE_(P, device='cpu'), WanVAE_(P, device='cpu', upconv='gather')
    assert ph.upconv == 'phases' and ga.upconv == 'gather'
    cp, cg = ph.stage_costs(8, 8), ga.stage_costs(8, 8)
    kinds = [k for k, _, _ in ph._stages()]
    assert len(cp) == len(cg) == len(kinds)
    for k, a, b in zip(kinds, cp, cg):
        assert (a < b) if k == 'up' else (a == b)
    w_up = [(b - a) for k, a, b in zip(kinds, cp, cg) if k == 'up']
    assert all(d > 0 for d in w_up)


def _launcher():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('mg_generate', os.path.join(root, 'scripts', 'inference', 'generate.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_flags_and_defaults():
    """scripts/inference/generate.py keeps the reference CLI's flags, defaults and validation
    (reference generate.py:36-179) and its output naming (:300-306)."""
    import datetime
    g = _launcher()
    a = g._parse_args(['--ckpt_dir', '/x'])
    assert (a.task, a.size, a.frame_num, a.sample_steps, a.sample_shift, a.sample_guide_scale) == \
        ('t2v-14B', '1280*720', 81, 50, 5.0, 5.0)
    assert (a.ulysses_size, a.ring_size, a.sample_solver, a.offload_model) == (1, 1, 'unipc', None)
    assert not (a.t5_fsdp or a.t5_cpu or a.dit_fsdp or a.use_prompt_extend) and a.base_seed >= 0
    assert g._parse_args(['--ckpt_dir', '/x', '--task', 't2i-14B']).frame_num == 1
    b = g._parse_args(['--ckpt_dir', '/x', '--base_seed', '42', '--offload_model', 'False', '--sample_solver', 'dpm++',
                       '--ulysses_size', '4', '--dit_fsdp', '--t5_fsdp', '--prompt', 'A cat walks on the grass/now'])
    assert b.base_seed == 42 and b.offload_model is False and b.sample_solver == 'dpm++' and b.dit_fsdp and b.t5_fsdp
    with pytest.raises(AssertionError, match='Please specify the checkpoint directory'):
        g._parse_args([])
    with pytest.raises(AssertionError, match='Unsupport size'):
        g._parse_args(['--ckpt_dir', '/x', '--size', '1024*1024'])
    with pytest.raises(AssertionError, match='Unsupport frame_num'):
        g._parse_args(['--ckpt_dir', '/x', '--task', 't2i-14B', '--frame_num', '5'])
    with pytest.raises(SystemExit):
        g._parse_args(['--ckpt_dir', '/x', '--sample_solver', 'euler'])
    name = g.default_save_name(b, datetime.datetime(2026, 1, 2, 3, 4, 5))
    assert name == 't2v-14B_1280*720_4_1_A_cat_walks_on_the_grass_now_20260102_030405.mp4'
    # single process: the reference's assertions on distributed-only flags
    with pytest.raises(AssertionError, match='not supported in non-distributed'):
        g.generate(g._parse_args(['--ckpt_dir', '/x', '--dit_fsdp']))
    with pytest.raises(AssertionError, match='context parallel'):
        g.generate(g._parse_args(['--ckpt_dir', '/x', '--ulysses_size', '2']))


def test_tokenizer_cleaning_modes():
    """prompt cleaning of HuggingfaceTokenizer: (input, whitespace, lower, canonicalize) rows produced by the
    reference's basic_clean / whitespace_clean / canonicalize (wan/modules/tokenizers.py:13-47, ftfy = NFC)."""
    from wan.modules.tokenizers import HuggingfaceTokenizer
    rows = [('  A  cat &amp;amp; dog\n walks ', 'A cat & dog walks', 'a cat & dog walks', 'a cat dog walks'),
            ('Hello_World!!  It&#39;s  OK', "Hello_World!! It's OK", "hello_world!! it's ok", 'hello world its ok'),
            (' MiXed   Case ', 'MiXed Case', 'mixed case', 'mixed case'),
            ('tab\tsep  &lt;b&gt;', 'tab sep <b>', 'tab sep <b>', 'tab sep b'),
            ('ünï_cödé!!', 'ünï_cödé!!', 'ünï_cödé!!', 'ünï cödé'),
            ('a/b  c_d. E', 'a/b c_d. E', 'a/b c_d. e', 'ab c d e')]
    tok = HuggingfaceTokenizer.__new__(HuggingfaceTokenizer)
    for text, ws, lo, ca in rows:
        for mode, want in (('whitespace', ws), ('lower', lo), ('canonicalize', ca), (None, text)):
            tok.clean = mode
            assert tok._prepare(text) == want, (mode, text)
    with pytest.raises(AssertionError):
        HuggingfaceTokenizer('x', clean='bogus')


def test_tokenizer_ftfy_defaults_restated():
    """ftfy.fix_text's default fixes that touch well-formed prompts (reference tokenizers.py:13), restated because ftfy is
    not in this image: the default negative prompt's fullwidth commas become ',', as ftfy's fix_character_width does
    (so umT5 sees the ids the reference sees), quotes are uncurled, ligatures expanded, line breaks unified, control
    characters and terminal escapes removed.  Expected strings are ftfy 6's documented outputs for these inputs."""
    from wan.configs import WAN_CONFIGS
    from wan.modules import tokenizers as tk
    neg = WAN_CONFIGS['t2v-14B'].sample_neg_prompt
    assert '，' in neg
    got = tk.clean_text(neg, 'whitespace')
    assert got == neg.replace('，', ',') and '，' not in got
    rows = [('“quote” and ‘single’', '"quote" and \'single\''),
            ('ﬁne ﬂow', 'fine flow'),
            ('ＡＢＣ　１２！', 'ABC 12!'),
            ('a\r\nb\rc d', 'a b c d'),
            ('x\x07y﻿z', 'xyz'),
            ('\x1b[31mred\x1b[0m', 'red'),
            ('é', 'é')]
    for text, want in rows:
        assert tk.clean_text(text, 'whitespace') == want, ascii(text)


def test_bench_flop_formulas():
    """bench.py's closed forms reproduce SURVEY.md 8(d): DiT FLOPs per forward at the three video sizes and the VAE decode
    FLOPs at 1920x832x81 / 1280x720x81 (the numbers `roofline.achieved` and `vae_decode.tflops_fp32` are computed from)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ['bench.py']
    try:
        spec.loader.exec_module(b)
    finally:
        sys.argv = argv
    for L, pf in ((75600, 6.5234), (131040, 17.2571), (166320, 26.7095)):
        assert abs(b.flops_per_forward(L, b.MODEL_14B) / 1e15 - pf) < 1e-3
    tot, attn = b.vae_decode_flops(21, 104, 240)
    assert abs(tot / 1e12 - 1116.5) < 0.1 and abs(attn / 1e12 - 20.1) < 0.05
    assert abs(b.vae_decode_flops(21, 90, 160)[0] / 1e12 - 639.2) < 0.1
    # executed by this engine: the three convs behind a 2x upsample as four 2x2 phase convs = 4/9 of their taps
    ex = b.vae_decode_flops(21, 104, 240, up_taps=4)[0]
    up = 2 * (384 * 192 * 4 * 41 + 384 * 192 * 16 * 81 + 192 * 96 * 64 * 81) * 104 * 240      # FLOPs per tap of the three convs
    assert abs((tot - ex) - 5 * up) < 1e6 and abs(ex / 1e12 - 1065.8) < 0.1
    assert b.WORKLOADS['1080p'][:3] == (1920, 832, 81)


def test_bench_self_launch_command():
    """`python bench.py --gpus N` without WORLD_SIZE becomes the launcher of its own ranks (reference launch contract
    scripts/inference/generate.py:190-229: one process per GPU): the command it re-executes, checked without running it."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['MOVIIGEN_BENCH_DRYRUN'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])['launch']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '8', '--steps', '3', '--warmup', '1']
    # the command of BASELINE configs[3] ("FSDP shard + SP=4 on 8 GPUs"): every flag reaches the ranks, the transport
    # choice travels in their environment too
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--workload', '1056p', '--dit-fsdp', '--vae-parallel',
                         '--transport', 'peer_copy'], capture_output=True, text=True, timeout=300, env=env)
    assert r3.returncode == 0, r3.stderr[-2000:]
    cmd3 = json.loads([ln for ln in r3.stdout.splitlines() if ln.startswith('{')][0])['launch']
    assert cmd3[cmd3.index(os.path.join(ROOT, 'bench.py')) + 1:] == ['--gpus', '8', '--workload', '1056p', '--dit-fsdp', '--vae-parallel',
                                                                     '--transport', 'peer_copy']
    # under a launcher (WORLD_SIZE set) nothing is re-executed: the dry-run marker is not printed, the rank path runs
    # (and stops at the missing GPU here)
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '1', '--workload', 'tiny'],
                        capture_output=True, text=True, timeout=300, env=env2)
    assert '"launch"' not in r2.stdout


def test_hot_loops_static_audit():
    """the steady-state loops of the two MFMA kernels that carry 97 % of the step, audited in the gfx950 assembly hipcc
    emits for EVERY template instantiation: MFMA count per iteration, an upper bound on everything else, no scratch
    access and no select chains (tools/audit_hot_loops.py: a computed array index hipcc failed to fold made one
    epilogue instantiation of the GEMM 7x slower in round 3 while every numerical test stayed green)."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import audit_hot_loops
    problems = []
    for k in audit_hot_loops.KERNELS:
        rep, prob = audit_hot_loops.audit(*k)
        assert rep, k
        problems += prob
    assert not problems, problems


def test_no_packed_write_behind_a_wide_store():
    """every kernel of the library: no store of more than 64 bits directly followed by a packed-fp32 VALU instruction that overwrites
    its data registers — hipcc leaves no wait state there for buffer stores with a scalar offset register, and on gfx950 the store
    then picks up part of the new value in four lanes of every 16 (csrc/gemm_bf16_v11.hip)."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import audit_hot_loops
    hits = []
    for src in sorted(os.listdir(audit_hot_loops.CSRC)):
        if src.endswith('.hip'):
            hits += audit_hot_loops.store_hazards(src)
    assert not hits, hits


def test_gemm_v11_schedule_is_the_generators_output(tmp_path):
    """the k-tile of GEMM variant 11 is generated code: the committed .inc files are what tools/gen_gemm_v11_schedule.py writes."""
    env = dict(os.environ, MG_V11_GEN_DIR=str(tmp_path))
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen_gemm_v11_schedule.py')], check=True, env=env, capture_output=True)
    names = sorted(os.listdir(tmp_path))
    assert names == ['gemm_bf16_v11_ktile_s4.inc', 'gemm_bf16_v11_ktile_s6.inc', 'gemm_bf16_v11_tail.inc']
    for n in names:
        assert open(os.path.join(tmp_path, n)).read() == open(os.path.join(ROOT, 'moviigen1.1_amd', 'csrc', n)).read(), n


def test_sp_group_chooser_on_the_baseline_shapes():
    """the Ulysses pipeline depth follows the shape: rounds of the persistent attention grid per layer, then exposed
    exchange; the four BASELINE multi-GPU shapes + the published-reference case (720p x 8 GPUs)."""
    from wan.distributed.ulysses import attention_rounds, choose_groups, split_heads
    rounds = lambda n, L, G: sum(attention_rounds(m, L) for _, m in split_heads(n, G))
    # configs[2]: 1920x832, Ulysses 8 -> 5 local heads, 512 query blocks: every split takes 10 rounds, so the deepest pipeline
    assert choose_groups(5, 131040, 8)[:2] == (5, 10)
    # configs[3]: 1920x1056, cfg2 x Ulysses 4 -> 10 local heads, 650 query blocks: 5 x 2 heads = 6 rounds per launch for 5.08 of work
    assert rounds(10, 166320, 5) == 30 and rounds(10, 166320, 2) == 26 and rounds(10, 166320, 1) == 26
    assert choose_groups(10, 166320, 4)[:2] == (2, 26)
    # what `bench.py --gpus 8` runs by default (1080p, cfg2 x Ulysses 4): 1024 items per 2-head launch = 4 full rounds
    assert choose_groups(10, 131040, 4)[:2] == (5, 20)
    # the published-reference case, 1280x720 on 8 GPUs: a 1-head launch has 296 items = 2 rounds for 1.16 of work -> one launch of 5 heads
    assert rounds(5, 75600, 5) == 10 and choose_groups(5, 75600, 8)[:2] == (1, 6)
    # single rank: nothing to exchange, nothing to pipeline over
    assert attention_rounds(40, 131040) == 80 and attention_rounds(1, 300) == 1
    # the explicit setting still wins (HeadExchange reads MOVIIGEN_SP_GROUPS) and sizes differ by at most one
    assert [n for _, n in split_heads(10, 4)] == [3, 3, 2, 2]


def test_mp4_mjpeg_container(tmp_path):
    """wan/utils/mp4_mjpeg.py — what `cache_video` writes when imageio is absent (reference wan/utils/utils.py:23-60 hands the frames to
    imageio / libx264).  The file must be a well-formed ISO base media file: ftyp | mdat | moov in that order, every box size exact, the one chunk
    offset pointing at a JPEG start-of-image, the sample sizes summing to the mdat payload, mp4v + objectTypeIndication 0x6C, the frame rate in
    the media timescale — and the pictures must decode back to the frames (JPEG quality 95, 4:4:4)."""
    import struct
    from wan.utils import mp4_mjpeg as M
    yy, xx = np.mgrid[0:50, 0:86]
    frames = np.stack([np.stack([(xx * 3 + t * 10) % 256, (yy * 5 + t * 7) % 256, ((xx + yy) * 2 + t) % 256], -1) for t in range(5)]).astype(np.uint8)
    for fps, fr in ((16, frames), (23.976, frames[:1]), (30, frames[:, :17, :33])):        # odd sizes, a single frame, a fractional rate
        path = str(tmp_path / f'v{fps}.mp4')
        n = M.write_mp4_mjpeg(path, fr, fps=fps)
        buf = open(path, 'rb').read()
        assert n == len(buf)
        top = list(M._walk(buf, 0, len(buf)))
        assert [k for k, _, _ in top] == [b'ftyp', b'mdat', b'moov'] and top[-1][2] == len(buf)
        assert buf[8:12] == b'isom'
        (_, ma, mb) = top[1]
        sa, sb = M._find(buf, 0, len(buf), b'moov', b'trak', b'mdia', b'minf', b'stbl')
        oa, _ = M._find(buf, sa, sb, b'stco')
        count, off = struct.unpack_from('>II', buf, oa + 4)
        assert count == 1 and off == ma and buf[off:off + 2] == b'\xff\xd8'
        za, _ = M._find(buf, sa, sb, b'stsz')
        fixed, cnt = struct.unpack_from('>II', buf, za + 4)
        sizes = struct.unpack_from(f'>{cnt}I', buf, za + 12)
        assert fixed == 0 and cnt == fr.shape[0] and sum(sizes) == mb - ma
        pos = off
        for sz in sizes:                                        # every sample is one whole JPEG
            assert buf[pos:pos + 2] == b'\xff\xd8' and buf[pos + sz - 2:pos + sz] == b'\xff\xd9'
            pos += sz
        with pytest.raises(ValueError):
            M._find(buf, sa, sb, b'stss')                       # no sync-sample table: every sample is a sync sample
        back = M.read_mp4_mjpeg(path)
        assert (back['codec'], back['object_type'], back['width'], back['height']) == ('mp4v', 0x6C, fr.shape[2], fr.shape[1])
        assert abs(back['fps'] - fps) < 1e-3
        assert back['frames'].shape == fr.shape
        err = np.abs(back['frames'].astype(np.int32) - fr.astype(np.int32))
        assert err.mean() < 2.0 and err.max() < 48, (err.mean(), err.max())
        ha, _ = M._find(buf, 0, len(buf), b'moov', b'mvhd')
        timescale, duration = struct.unpack_from('>II', buf, ha + 12)
        assert timescale == 1000 and duration == round(fr.shape[0] * 1000.0 / fps)
    for bad in (frames.astype(np.float32), frames[0], frames[:0]):
        with pytest.raises(ValueError):
            M.write_mp4_mjpeg(str(tmp_path / 'bad.mp4'), bad)
    with pytest.raises(ValueError):
        M.write_mp4_mjpeg(str(tmp_path / 'bad.mp4'), frames, fps=0)
