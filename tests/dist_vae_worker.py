"""worker for test_vae_pipelined_one_gpu / test_vae_spatial_one_gpu: WORLD_SIZE processes share cuda:0 (gloo, staged via host
memory — test plumbing; production is RCCL send/recv).  The layer-pipelined decode and the W-band decode must each give rank 0
the single-GPU video bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'moviigen1.1_amd'), os.path.join(ROOT, 'tests', 'golden')]

import weights as W  # noqa: E402
import wan  # noqa: E402
from wan.modules.vae import partition_costs  # noqa: E402

dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
vae = wan.modules.WanVAE(state_dict=W.make_vae_params(8, 1), device='cuda:0')
z = W.randn((16, 10, 6, 10), 41)                        # 10 latent frames: chunks [1, 4, 4, 1] — the single-GPU chunk list
ref = vae.decode([z])[0]
assert vae.model._chunks(10) == [1, 4, 4, 1]
WHICH = os.environ.get('MOVIIGEN_VAE_TEST', 'both')      # 'pipeline' | 'spatial' | 'both': which multi-rank decode this run checks
out = vae.decode_pipelined([z])[0] if WHICH != 'spatial' else None
# cut by "measured" stage times instead of the cost model (any positive weights must give the same video), and the
# reference's one-frame chunks
n_st = len(vae.model._stages())
out2 = vae.model.decode_pipelined(z, stage_ms=[1.0 + (i % 3) for i in range(n_st)]) if WHICH != 'spatial' else None
out3 = vae.model.decode_pipelined(z, chunks=[1] * 10) if WHICH != 'spatial' else None
if WHICH == 'spatial':
    pass
elif rank == 0:
    assert out is not None and torch.equal(out, ref), (out - ref).abs().max().item()
    assert torch.equal(out2, ref) and torch.equal(out3, ref)
    costs = vae.model.stage_weights(6, 10)
    cuts = partition_costs(costs, world)
    assert cuts[0] == 0 and cuts[-1] == len(costs) and all(b > a for a, b in zip(cuts, cuts[1:]))
    # the shape the receiver of a cut allocates == what the upstream stages really produce
    m = vae.model
    x = torch.zeros(1, 6, 10, 16, device='cuda:0')
    for last in (1, 5, 8, 12, n_st):                    # first chunk: run the stages [0, last) and compare
        y = m._decoder_chunk(x, [None] * (m.n_slots + 8), 0, last)
        assert tuple(y.shape) == m.stage_out_shape(last, 1, True, 6, 10), (last, y.shape)
    assert m.stage_out_shape(n_st, 4, False, 6, 10) == (16, 48, 80, 3)
else:
    assert out is None and out2 is None and out3 is None
if WHICH != 'spatial':
    print(f'VAEPIPE_OK rank{rank}/{world}', flush=True)
if WHICH == 'pipeline':
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)

# ---- the W-band decode (WanVAE.decode_spatial): every rank runs the whole decoder on its band of image columns; 10 latent columns over
# 2 / 4 / 8 ranks = bands of 5+5, 3+3+2+2 (uneven), 2+2+1+1+1+1+1+1 (one-column bands: both halo columns of a band come from other ranks) --
from wan.modules.vae import _Band  # noqa: E402
b = _Band(None, world, rank, 10)
assert sum(b.widths) == 10 and max(b.widths) - min(b.widths) <= 1 and b.starts[rank] == sum(b.widths[:rank])
sp = vae.decode_spatial([z])[0]
sp1 = vae.model.decode_spatial(z, chunks=[1] * 10)              # the reference's one-frame chunks
z2 = W.randn((16, 3, 4, 9), 43)                                  # another geometry: 9 columns, odd everything
ref2 = vae.decode([z2])[0]
sp2 = vae.model.decode_spatial(z2) if world <= 9 else None
if rank == 0:
    assert sp is not None and tuple(sp.shape) == tuple(ref.shape) and torch.equal(sp, ref), (sp - ref).abs().max().item()
    assert torch.equal(sp1, ref) and torch.equal(sp2, ref2)
else:
    assert sp is None and sp1 is None and sp2 is None
assert vae.model._band is None and (vae.model.last_halo_bytes > 0) == (world > 1)
print(f'VAESPATIAL_OK rank{rank}/{world}', flush=True)
dist.barrier()
dist.destroy_process_group()
