"""worker for test_train_side_sp_forward_one_gpu: the training-side sequence-parallel forward
(scripts/train/model/model_seq.py on the engine) with 2 ranks sharing cuda:0 over gloo, against the golden
produced by the imported reference (g8_train_seq.npz, tests/golden/make_golden_seq.py) and against the engine's own
single-rank forward (bit-identical: every op is row-local except attention, whose key order does not change)."""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')]

import weights as W  # noqa: E402

spec = importlib.util.spec_from_file_location('model_seq', os.path.join(ROOT, 'scripts', 'train', 'model', 'model_seq.py'))
ms = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ms)

dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device('cuda:0')
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_train_seq.npz'))
cfg = W.SMALL_DIT_HD128
m = ms.WanModel(**cfg)
m.load_state_dict(W.make_dit_params(cfg, 0))
m.to(dev)
lat = torch.from_numpy(g['lat']).to(dev)
bctx = torch.from_numpy(g['batch_context']).to(dev)
t = torch.from_numpy(g['t']).to(dev)


def rel_l2(a, b):
    a, b = a.double().cpu(), torch.from_numpy(b).double()
    return ((a - b).norm() / b.norm()).item()


single = {}
for tag in ('nopad', 'pad'):
    single[tag] = m([lat], t=t, context=None, seq_len=int(g[f'seq_len_{tag}']), batch_context=bctx)[0].clone()
    assert rel_l2(single[tag], g[f'single_fp32_{tag}']) < 2e-2            # stated bf16 tolerance vs the fp32 reference
    assert rel_l2(single[tag], g[f'sp2_bf16_{tag}']) < 1.2e-2             # vs the reference under bf16 autocast
ms.initialize_sequence_parallel_state(world)
assert ms.get_sequence_parallel_state() and ms.nccl_info.sp_size == world
for rep in range(2):
    for tag in ('nopad', 'pad'):
        out = m([lat], t=t, context=None, seq_len=int(g[f'seq_len_{tag}']), batch_context=bctx)[0]
        assert m.sp_size == world and m.cross_attn_head_sharded and m.sp_mask_padded_keys
        assert torch.equal(out, single[tag]), (tag, (out - single[tag]).abs().max().item())
        assert rel_l2(out, g[f'sp2_bf16_{tag}']) < 1.2e-2 and rel_l2(out, g[f'sp2_fp32_{tag}']) < 2e-2
print(f'TRAIN_SEQ_OK rank{rank}/{world}', flush=True)
dist.barrier()
dist.destroy_process_group()
